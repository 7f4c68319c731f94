#!/bin/bash
# One rocprofv3 PMC pass over the default bench.py run (counters only + kernel trace, as the pool requires) and a
# per-kernel average table. usage (GPU box): bash tools/pmc_bench.sh "SQ_INSTS_VALU SQ_WAVES ..." [tag]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
ctrs=$1; tag=${2:-pmc}
out="$REPO/gpurun_out/$tag"
rm -rf "$out"; mkdir -p "$out"
cd /tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o run -- \
  python "$REPO/bench.py" --no-cpu-baseline --steps 10 --warmup 2 --preroll 4 > "$out/bench.log" 2>&1 || { tail -20 "$out/bench.log"; exit 1; }
f=$(find "$out" -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[k] += 1
names = sorted({c for v in acc.values() for c in v})
print("%-48s %6s " % ("kernel", "n") + " ".join("%16s" % c[:16] for c in names))
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVES", 0) if "tfl" in k else 1):
    if "tfl::" not in k:
        continue
    print("%-48s %6d " % (k, n[k]) + " ".join("%16.1f" % (acc[k][c] / n[k]) for c in names))
PY
