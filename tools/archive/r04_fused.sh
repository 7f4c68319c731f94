#!/bin/bash
# round 4: fused curl + confinement (k_vort_fused) and the per-size block depth of advect_vel3 -- tests, then A/B
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r04fused; rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -x -q -k "fused_vorticity or zslab or parity_at_baseline or native_simulate or golden or long_horizon or native_transport" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 | tee $O/pytest.txt
run() { # label, lib, env...
  local label=$1 lib=$2; shift; shift
  for res in 128 256; do
    echo "== $label res $res"
    env TFL_LIBRARY=$PWD/ab/$lib.so "$@" timeout 120 python bench.py --no-cpu-baseline --no-config5 --no-configs --blocks 3 --res $res --steps $((res == 128 ? 40 : 10)) 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|k_vel|k_vort|k_curl|k_confine|k_add_buoy"
  done
}
for round in 1 2; do
  run new new
  run unfused new TFL_VORT_FUSED=0
  run oldvel oldvel
done 2>&1 | tee $O/ab.txt
