#!/bin/bash
# gpurun_out/r06final -> profiles/r06_* (run here after tools/r06_gpu_final.sh ran on the GPU box)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"
O=gpurun_out/r06final; C=$(git rev-parse --short HEAD)
cp $O/bench.json profiles/r06_bench.json
cp $O/bench_fast.json profiles/r06_bench_fast_advection.json
cp $O/128_kernel_stats.csv profiles/r06_kernel_stats.csv
cp $O/256_kernel_stats.csv profiles/r06_256_kernel_stats.csv
cp $O/gloo2.json profiles/r06_multirank_controlflow_gloo2_selflaunch.json
{ echo "# tools/slab_host_cost.py (round 6: the library's native transport over tests/stub_rccl.cpp in STUB_RCCL_NULL mode), one MI355X, commit $C"; cat $O/slab.txt; echo; echo "# tools/ubench/host_costs.hip on the same box"; cat $O/host_costs.txt; } > profiles/r06_slab_host_cost.txt
python tools/pmc_traffic.py $O/128_FETCH_SIZE.csv $O/128_WRITE_SIZE.csv r06 2097152 $C | head -24
python tools/pmc_traffic.py $O/256_FETCH_SIZE.csv $O/256_WRITE_SIZE.csv r06_256 16777216 $C | head -24
python tools/pmc_sq.py $O/128_sq.csv r06 $C json | head -24
