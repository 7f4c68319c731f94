#!/bin/bash
python tools/m16_check.py 2>&1 | tail -12
export TFL_CONV_PATH=mfma16
run() { python bench.py --no-cpu-baseline --no-config5 --steps 30 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|conv3"; }
echo "== tiled"; TFL_M16_TILED=3 TFL_LIBRARY=$PWD/ab/z_lb2.so run
for cz in 8 16; do
  for l in z_lb2 z_abl4 z_abl8; do echo "== $l cz=$cz"; TFL_M16_CZ=$cz TFL_LIBRARY=$PWD/ab/$l.so run; done
done
