#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06h; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_simulate.py tests/test_native_transport.py -m gpu -q -x -k "zslab or transport or graph" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for a in "128 8" "256 8"; do
  timeout 200 python tools/slab_host_cost.py $a --still --no-graph 2>&1 | grep "^res"
  timeout 200 python tools/slab_host_cost.py $a --still --check-reach 2>&1 | grep "^res" | sed 's/^/check_reach=1: /'
done | tee $O/slab_check_reach.txt
