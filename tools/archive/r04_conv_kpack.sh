#!/bin/bash
# round 4: K-packed z-marched MFMA conv (k_conv3_m16p) -- parity tests, then A/B against k_conv3_m16z
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r04kpack; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "model or conv or parity_at_baseline or zslab or witness or range" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 | tee $O/pytest.txt
for round in 1 2; do
  for v in "kpack:1" "tile_in:2"; do
    for res in 128 256; do
      echo "== ${v%%:*} res $res"
      TFL_DEBUG=1 TFL_M16_KPACK=${v##*:} python bench.py --no-cpu-baseline --no-config5 --no-configs --blocks 3 --res $res --steps $((res == 128 ? 40 : 10)) 2>$O/err.txt | python tools/bench_kernels.py | grep -E "ms/step|k_conv3"
      grep "m16p" $O/err.txt | sort -u | head -3
    done
  done
done 2>&1 | tee $O/ab.txt
