#!/bin/bash
# round-3 evidence set -> gpurun_out/r03final/ (copy what is wanted into profiles/)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r03final; rm -rf $O; mkdir -p $O
python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench rc=$?"
python tools/slab_host_cost.py 128 8 --kernels > $O/slab_128.txt 2>&1; python tools/slab_host_cost.py 256 8 >> $O/slab_128.txt 2>&1
cd /tmp
for res in 128 256; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$res -o bench -- python $REPO/bench.py --no-cpu-baseline --no-config5 --res $res --steps $((res==128?50:10)) --preroll $((res==128?16:4)) > $O/stats$res.log 2>&1
  cp "$(find $O/stats$res -name '*kernel_stats.csv' | head -1)" $O/${res}_kernel_stats.csv; rm -rf $O/stats$res
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --res $res --steps 4 --warmup 1 --preroll 2 > $O/pmc_${res}_$c.log 2>&1
    cp "$(find $O/pmc -name '*counter_collection.csv' | head -1)" $O/${res}_$c.csv; rm -rf $O/pmc
  done
done
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --steps 10 --warmup 2 > $O/pmc_sq.log 2>&1
cp "$(find $O/pmc -name '*counter_collection.csv' | head -1)" $O/128_sq.csv; rm -rf $O/pmc
cd $REPO; ls -la $O | head -30; head -c 400 $O/bench.json; echo
