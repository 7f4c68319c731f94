#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06f; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_simulate.py tests/test_hip_fullsize.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
echo "== A/B: k_project fast division + split_pair in the first conv layer"; bash tools/ab_run.sh "project|conv3_in|bcs" base new 2>&1 | tee $O/ab.txt
