#!/bin/bash
# ablation timing of the fp16-MFMA conv kernels (ab/m16_abl<mask>.so; mask bits in conv_mfma16.hip)
export TFL_CONV_PATH=mfma16
for n in "$@"; do
  echo "== $n"
  TFL_LIBRARY=$PWD/ab/$n.so python bench.py --no-cpu-baseline --no-config5 --steps 30 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|conv3"
done
