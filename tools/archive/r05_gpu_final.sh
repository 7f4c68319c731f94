#!/bin/bash
# round-5 evidence set -> gpurun_out/r05final/ (tools/r05_collect.sh copies what is kept into profiles/)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r05final; rm -rf $O; mkdir -p $O
timeout 300 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench rc=$?"
TFL_ADVECT_MODE=fast timeout 120 python bench.py --no-cpu-baseline --no-config5 --no-configs 2>/dev/null | tail -1 > $O/bench_fast.json
timeout 120 python tools/slab_host_cost.py 128 8 --kernels --still > $O/slab_128.txt 2>&1; timeout 120 python tools/slab_host_cost.py 256 8 --kernels --still >> $O/slab_128.txt 2>&1
cd /tmp
for res in 128 256; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$res -o bench -- python $REPO/bench.py --no-cpu-baseline --no-config5 --no-configs --res $res --steps $((res==128?50:10)) --blocks 1 --preroll $((res==128?16:4)) > $O/stats$res.log 2>&1
  cp "$(find $O/stats$res -name '*kernel_stats.csv' | head -1)" $O/${res}_kernel_stats.csv; rm -rf $O/stats$res
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --no-configs --res $res --steps 4 --blocks 1 --warmup 1 --preroll 2 > $O/pmc_${res}_$c.log 2>&1
    cp "$(find $O/pmc -name '*counter_collection.csv' | head -1)" $O/${res}_$c.csv; rm -rf $O/pmc
  done
done
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --no-configs --steps 10 --blocks 1 --warmup 2 > $O/pmc_sq.log 2>&1
cp "$(find $O/pmc -name '*counter_collection.csv' | head -1)" $O/128_sq.csv; rm -rf $O/pmc
cd $REPO; ls -la $O | head -30; head -c 600 $O/bench.json; echo
