#!/bin/bash
# gpurun_out/r05final -> profiles/r05_* (run here after tools/r05_gpu_final.sh ran on the GPU box)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"
O=gpurun_out/r05final; C=$(git rev-parse --short HEAD)
cp $O/bench.json profiles/r05_bench.json
cp $O/bench_fast.json profiles/r05_bench_fast_advection.json
cp $O/128_kernel_stats.csv profiles/r05_kernel_stats.csv
cp $O/256_kernel_stats.csv profiles/r05_256_kernel_stats.csv
cp $O/slab_128.txt profiles/r05_slab_host_cost.txt
python tools/pmc_traffic.py $O/128_FETCH_SIZE.csv $O/128_WRITE_SIZE.csv r05 2097152 $C | head -24
python tools/pmc_traffic.py $O/256_FETCH_SIZE.csv $O/256_WRITE_SIZE.csv r05_256 16777216 $C | head -24
python tools/pmc_sq.py $O/128_sq.csv r05 $C json | head -24
