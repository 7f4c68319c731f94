#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06e; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_default.txt 2>&1; echo "default flavour rc=$?"; grep -E "passed|failed" $O/pytest_default.txt | tail -2
TFL_LIBRARY=$REPO/fluidnet_amd/libtfluids_hip_exp.so timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_exp.txt 2>&1; echo "experiments flavour rc=$?"; grep -E "passed|failed" $O/pytest_exp.txt | tail -2
grep -E "^FAILED|Error" $O/pytest_default.txt $O/pytest_exp.txt | head -20
python bench.py --no-cpu-baseline --no-config5 --steps 50 2>/dev/null | python tools/bench_kernels.py | tee $O/bench_kernels.txt
