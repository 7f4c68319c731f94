#!/bin/bash
# round 4, first GPU session of the fp16-MFMA conv path: numerics, then the step with both conv paths
mkdir -p gpurun_out
python tools/m16_check.py > gpurun_out/m16_check.txt 2>&1; tail -20 gpurun_out/m16_check.txt
for p in winograd mfma16; do
  echo "== $p"
  TFL_CONV_PATH=$p python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tee gpurun_out/bench_$p.json | python tools/bench_kernels.py | head -24
done
