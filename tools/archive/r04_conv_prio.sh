#!/bin/bash
# round 4: z-marched MFMA conv with the epilogue deferred by one plane -- scheduling groups / wave priority A/B
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r04convprio; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "model or conv or simulate_parity or smoke or witness" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 | tee $O/pytest.txt
for round in 1 2; do
  for n in base noprio nosched nosched_noprio; do
    for res in 128 256; do
      echo "== $n res $res"
      TFL_LIBRARY=$PWD/ab/$n.so python bench.py --no-cpu-baseline --no-config5 --no-configs --blocks 3 --res $res --steps $((res == 128 ? 40 : 10)) 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|k_conv3"
    done
  done
done 2>&1 | tee $O/ab.txt
