#!/bin/bash
# Kernel-trace profile of the default bench.py run -> gpurun_out/prof/kernel_stats.csv (rocprofv3 --stats).
# usage (GPU box): bash tools/prof_bench.sh [extra bench.py args]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$REPO/gpurun_out/prof"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -o bench -- \
  python "$REPO/bench.py" --no-cpu-baseline "$@" > "$REPO/gpurun_out/prof/bench.log" 2>&1 || { tail -20 "$REPO/gpurun_out/prof/bench.log"; exit 1; }
f=$(find "$REPO/gpurun_out/prof" -name '*kernel_stats.csv' | head -1)
cp "$f" "$REPO/gpurun_out/prof/kernel_stats.csv"
python - "$REPO/gpurun_out/prof/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} pct {r['Percentage']}")
PY
tail -1 "$REPO/gpurun_out/prof/bench.log" | cut -c1-300
