#!/bin/bash
# gpurun_out/r04final -> profiles/r04_* (run here after tools/r04_gpu_final.sh ran on the GPU box)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"
O=gpurun_out/r04final; C=$(git rev-parse --short HEAD)
cp $O/bench.json profiles/r04_bench.json
cp $O/bench_fast.json profiles/r04_bench_fast_advection.json
cp $O/128_kernel_stats.csv profiles/r04_kernel_stats.csv
cp $O/256_kernel_stats.csv profiles/r04_256_kernel_stats.csv
cp $O/slab_128.txt profiles/r04_slab_host_cost.txt
python tools/pmc_traffic.py $O/128_FETCH_SIZE.csv $O/128_WRITE_SIZE.csv r04 2097152 $C | head -20
python tools/pmc_traffic.py $O/256_FETCH_SIZE.csv $O/256_WRITE_SIZE.csv r04_256 16777216 $C | head -20
python tools/pmc_sq.py $O/128_sq.csv r04 $C | head -20
