#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
run() { local label=$1; shift
  for res in 128 256; do
    echo "== $label res $res"
    env TFL_LIBRARY=$PWD/ab/new.so "$@" timeout 120 python bench.py --no-cpu-baseline --no-config5 --no-configs --blocks 3 --res $res --steps $((res == 128 ? 40 : 10)) 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|k_vel|k_vort|k_curl|k_confine|k_add_buoy"
  done
}
for round in 1 2; do
  run fused TFL_VORT_FUSED=1
  run fused_cz12 TFL_VORT_FUSED=1 TFL_VORT_CZ=12
  run auto
  run unfused TFL_VORT_FUSED=0
done
