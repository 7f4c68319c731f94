#!/bin/bash
# round 5, session 1: the fused conv layers + the MFMA tail -- parity tests, then kernel times with the switches on and off
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_hip_simulate.py -x -q -m gpu -k "conv_fused or conv_paths_agree_3d or fp16_range or forward_matches" 2>&1 | tail -15 | tee gpurun_out/r05a/tests.txt
for cfg in "" "TFL_M16_FUSE12=0" "TFL_M16_TAIL_MFMA=0" "TFL_M16_FUSE12=0 TFL_M16_TAIL_MFMA=0" "" ; do
  echo "== conv_abl [$cfg]" | tee -a gpurun_out/r05a/conv.txt
  env $cfg TFL_DEBUG=1 timeout 300 python tools/conv_abl.py 2>&1 | grep -v "^$" | tail -12 | tee -a gpurun_out/r05a/conv.txt
done
for cz in 8 12 16 22 32 43 64; do
  echo "== TFL_M16_CZ_F2=$cz" | tee -a gpurun_out/r05a/conv.txt
  TFL_M16_CZ_F2=$cz timeout 300 python tools/conv_abl.py 2>&1 | tail -2 | tee -a gpurun_out/r05a/conv.txt
done
