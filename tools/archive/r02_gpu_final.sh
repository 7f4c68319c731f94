#!/bin/bash
# round-2 evidence set -> gpurun_out/r02final/ (copy what is wanted into profiles/)
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r02final; rm -rf $O; mkdir -p $O
python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; echo "bench rc=$?"
for n in 2 8; do
  TFL_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2960$n bench.py --gpus $n --steps 5 --warmup 1 --preroll 2 2> $O/gloo$n.err | tail -1 > $O/bench_gloo$n.json
done
python tools/pcg_bench.py 128 > $O/pcg.txt 2>&1
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
cd $REPO; ls -la $O | head -30; head -c 600 $O/bench.json; echo; cat $O/pcg.txt | tail -5
