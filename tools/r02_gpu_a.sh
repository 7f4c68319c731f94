#!/bin/bash
# round-2 GPU session A: test suite (incl. the full-size parity tests) + un-sharded 256^3 kernel profile + PMC traffic
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"; export TMPDIR=/tmp
O=gpurun_out/r02a; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --durations=12 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -25 $O/tests.log
bash tools/prof_bench.sh --res 256 --steps 10 --warmup 2 --preroll 4 > $O/prof256.log 2>&1; cp gpurun_out/prof/kernel_stats.csv $O/256_kernel_stats.csv; tail -22 $O/prof256.log
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/$O/pmc_$c -o run -- python $REPO/bench.py --no-cpu-baseline --res 256 --steps 4 --warmup 1 --preroll 2 > $REPO/$O/pmc_$c.log 2>&1
  cp "$(find $REPO/$O/pmc_$c -name '*counter_collection.csv' | head -1)" $REPO/$O/256_$c.csv; rm -rf $REPO/$O/pmc_$c
done
cd $REPO
python tools/pmc_traffic.py $O/256_FETCH_SIZE.csv $O/256_WRITE_SIZE.csv r02_256 16777216 > $O/traffic256.txt 2>&1; cp profiles/r02_256_pmc_traffic.txt $O/ 2>/dev/null
git checkout profiles/pmc_traffic.json 2>/dev/null
tail -20 $O/traffic256.txt
