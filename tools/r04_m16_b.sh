#!/bin/bash
python tools/m16_check.py 2>&1 | tail -12
export TFL_CONV_PATH=mfma16
bash tools/r04_m16_abl.sh m16_base m16_lb4 m16_abl1 m16_abl4 m16_abl8 m16_abl15 m16_base
for nt in 1 2 8; do echo "== nt=$nt"; TFL_M16_NT=$nt TFL_LIBRARY=$PWD/ab/m16_base.so python bench.py --no-cpu-baseline --no-config5 --steps 30 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|conv3"; done
