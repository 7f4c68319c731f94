#!/bin/bash
# round 4: chunk length of the z-marched MFMA conv kernels (rounds of resident blocks) -- full GPU suite, then A/B
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r04convcz; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 | tee $O/pytest.txt
run() { # label, env...
  local label=$1; shift
  for res in 128 256; do
    echo "== $label res $res"
    env "$@" python bench.py --no-cpu-baseline --no-config5 --no-configs --blocks 3 --res $res --steps $((res == 128 ? 40 : 10)) 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|k_conv3"
  done
}
for round in 1 2; do
  run auto TFL_DEBUG=1
  run cz11 TFL_M16_CZ=11
  run cz8 TFL_M16_CZ=8
  run cz32 TFL_M16_CZ=32
done 2>&1 | tee $O/ab.txt
TFL_DEBUG=1 python bench.py --no-cpu-baseline --no-config5 --no-configs --blocks 1 --steps 2 2>&1 | grep "tfl\]" | sort | uniq | head
