#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "advect or golden or oracle or ragged or sample_outside or zslab_decomposition_equals" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5
for round in 1 2; do
for tz in 2 12 22 14 1; do
  echo "== TZ $tz: $(TFL_SCAL3_TZ=$tz timeout 120 python tools/adv_abl.py 2>&1 | tail -2 | tr '\n' ' ')"
done
done
