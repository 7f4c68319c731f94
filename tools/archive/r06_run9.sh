#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06i; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_simulate.py tests/test_native_transport.py tests/test_hip_fullsize.py tests/test_abi.py -m gpu -q -x -k "zslab or transport or graph or c_multi" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.txt | tail -1
for a in "128 8" "256 8" "128 4" "128 2"; do
  timeout 200 python tools/slab_host_cost.py $a --still --no-graph --kernels 2>&1 | grep -E "^res|k_adv|k_vel|k_scal|sum"
  TFL_ADV_PAIR=0 timeout 200 python tools/slab_host_cost.py $a --still --no-graph 2>&1 | grep "^res" | sed 's/^/TFL_ADV_PAIR=0: /'
done | tee $O/slab_pair.txt
