#!/bin/bash
# round 5, session 2: de-phasing the resident blocks (TFL_M16_STAGGER*), deferred stores (ab/defer.so)
mkdir -p gpurun_out/r05b
out=gpurun_out/r05b/conv.txt
run() { echo "== [$*]" | tee -a $out; env "$@" timeout 300 python tools/conv_abl.py 2>&1 | grep "\^3" | tee -a $out; }
run TFL_M16_FUSE12=0
for s in 4 8 16 24 31 47; do run TFL_M16_FUSE12=0 TFL_M16_STAGGER=$s TFL_M16_STAGGER_IN=$s; done
run TFL_M16_FUSE12=1
for s in 8 16 31 47 63 94; do run TFL_M16_FUSE12=1 TFL_M16_STAGGER_F2=$s; done
run TFL_M16_FUSE12=0 TFL_LIBRARY=$PWD/ab/defer.so
run TFL_M16_FUSE12=0 TFL_LIBRARY=$PWD/ab/defer.so TFL_M16_STAGGER=16
run TFL_M16_FUSE12=0
