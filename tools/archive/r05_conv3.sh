#!/bin/bash
# round 5, session 3: ablations of k_conv3_m16p (ab/abl<mask>.so: 1 no DMA, 2 no MFMA, 4 no stores, 8 no LDS reads, 16 no barrier, 32 no epilogue)
mkdir -p gpurun_out/r05c
out=gpurun_out/r05c/abl.txt
for a in "" 1 2 4 8 16 32 3 11 36 47 63; do
  lib=$PWD/fluidnet_amd/libtfluids_hip.so; [ -n "$a" ] && lib=$PWD/ab/abl$a.so
  echo "== abl [$a]" | tee -a $out
  TFL_M16_FUSE12=0 TFL_LIBRARY=$lib timeout 300 python tools/conv_abl.py 2>&1 | grep "\^3" | tee -a $out
done
