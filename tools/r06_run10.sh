#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06j; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_simulate.py tests/test_hip_fullsize.py tests/test_example_sim3d.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.txt | tail -1; grep -E "^FAILED|^E " $O/pytest.txt | head
for r in 1 2; do for v in 0 1; do echo "-- TFL_ADV_PAIR=$v"; TFL_ADV_PAIR=$v python bench.py --no-cpu-baseline --no-config5 --steps 30 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|adv|scalar|vel_"; done; done 2>&1 | tee $O/ab_pair.txt
