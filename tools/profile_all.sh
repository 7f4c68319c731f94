#!/bin/bash
# The judged evidence set, produced on the GPU box into gpurun_out/final/ (copy what is wanted into profiles/):
#   bench.json            python bench.py (default flags, with cpu_baseline)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   fetch/ write/ sq/     PMC passes (counters only + kernel trace), one counter set per pass
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT="$REPO/gpurun_out/final"
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
python bench.py 2> "$OUT/bench.err" | tail -1 > "$OUT/bench.json"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python "$REPO/bench.py" --no-cpu-baseline > "$OUT/stats.log" 2>&1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/$name" -o run -- \
    python "$REPO/bench.py" --no-cpu-baseline --steps 10 --warmup 2 --preroll 16 > "$OUT/$name.log" 2>&1
  cp "$(find "$OUT/$name" -name '*counter_collection.csv' | head -1)" "$OUT/$name.csv"
  rm -rf "$OUT/$name"
done
rm -rf "$OUT/stats"
ls -la "$OUT"
head -c 400 "$OUT/bench.json"; echo
