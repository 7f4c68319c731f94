#!/bin/bash
# round-4 mid-round evidence: full GPU tests with the fp16-MFMA conv path as the default, kernel stats + PMC at 128^3
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r04mid; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt
python bench.py --no-cpu-baseline 2> $O/bench.err | tail -1 > $O/bench.json; python tools/bench_kernels.py $O/bench.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --steps 4 --warmup 1 --preroll 2 > $O/pmc_$c.log 2>&1
  cp "$(find $O/pmc -name '*counter_collection.csv' | head -1)" $O/128_$c.csv; rm -rf $O/pmc
done
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --steps 10 --warmup 2 > $O/pmc_sq.log 2>&1
cp "$(find $O/pmc -name '*counter_collection.csv' | head -1)" $O/128_sq.csv; rm -rf $O/pmc
cd $REPO
python tools/pmc_traffic.py $O/128_FETCH_SIZE.csv $O/128_WRITE_SIZE.csv r04mid 2097152 $(cat .git_head 2>/dev/null || echo wip) | head -24
python tools/pmc_sq.py $O/128_sq.csv r04mid wip | head -24
