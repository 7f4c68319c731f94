#!/bin/bash
# round 4: advect_scalar3.hip -- parity (whole GPU suite) + A/B of the block depth TZ against the round-2 gather kernels
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r04scal3; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 | tee $O/pytest.txt
run() { # label, env...
  local label=$1; shift
  for res in 128 256; do
    echo "== $label res $res"
    env "$@" python bench.py --no-cpu-baseline --no-config5 --no-configs --res $res --steps $((res == 128 ? 40 : 10)) 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|k_scalar|k_minmax|k_vel"
  done
}
for round in 1 2; do
  run gather TFL_SCALAR_GATHER=1
  run tz1 TFL_SCAL3_TZ=1
  run tz2 TFL_SCAL3_TZ=2
  run tz4 TFL_SCAL3_TZ=4
done 2>&1 | tee $O/ab.txt
