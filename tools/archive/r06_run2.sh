#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06b; rm -rf $O; mkdir -p $O
(cd tools/ubench && /opt/rocm/bin/hipcc -O3 -Wno-unused-value host_costs.hip -o /tmp/host_costs 2>/dev/null && /tmp/host_costs) > $O/host_costs.txt 2>&1; cat $O/host_costs.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.txt
timeout 300 python tools/slab_host_cost.py 128 8 --still > $O/slab_128.txt 2>&1; echo "slab rc=$?"
TFL_SIDE_STREAM=0 timeout 300 python tools/slab_host_cost.py 128 8 --still > $O/slab_128_onestream.txt 2>&1
cat $O/slab_*.txt | grep -v amdgpu.ids
