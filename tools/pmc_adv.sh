#!/bin/bash
# SQ / LDS counters of the advection kernels (two PMC passes, --kernel-trace only) -> gpurun_out/<tag>/pmc_adv.txt
# usage: tools/pmc_adv.sh <tag> [library.so]
REPO=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
tag=$1; lib=${2:-}
O=$REPO/gpurun_out/$tag; mkdir -p $O
[ -n "$lib" ] && export TFL_LIBRARY=$lib
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --steps 6 --warmup 2 > $O/pmc$i.log 2>&1
  cp "$(find $O/pmc$i -name '*counter_collection.csv' | head -1)" $O/pmc$i.csv; rm -rf $O/pmc$i
done
python - "$O" <<'PY'
import csv, collections, re, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in ("pmc1.csv", "pmc2.csv"):
    for r in csv.DictReader(open(O + "/" + f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
out = ["# per WAVE averages (counter / SQ_WAVES), bench.py 128^3; clocks = SQ_BUSY_CYCLES/32", "%-22s %8s " % ("kernel", "waves") + " ".join("%9s" % c.replace("SQ_", "")[:9] for c in cols) + "   clocks"]
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)):
    a = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
    w = a.get("SQ_WAVES", 1) or 1
    out.append("%-22s %8d " % (k[:22], w) + " ".join("%9.1f" % (a.get(c, 0) / w) for c in cols) + "  %7.0f" % (a.get("SQ_BUSY_CYCLES", 0) / 32))
open(O + "/pmc_adv.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
