#!/bin/bash
# SQ counters of the PCG sweep kernels on one sub-box (two PMC passes, --kernel-trace only) -> gpurun_out/<tag>/pmc_pcg.txt
# usage: tools/pmc_pcg.sh <tag> [Z,Y,X] [library.so]
REPO=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
tag=$1; dims=${2:-18,66,128}; lib=${3:-}
O=$REPO/gpurun_out/$tag; mkdir -p $O
[ -n "$lib" ] && export TFL_LIBRARY=$lib
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o run -- python $REPO/tools/pcg_wf_probe.py $dims > $O/pmc$i.log 2>&1
  cp "$(find $O/pmc$i -name '*counter_collection.csv' | head -1)" $O/pmc$i.csv; rm -rf $O/pmc$i
done
python - "$O" <<'PY'
import csv, collections, re, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in ("pmc1.csv", "pmc2.csv", "pmc3.csv"):
    try: rows = list(csv.DictReader(open(O + "/" + f)))
    except OSError: continue
    for r in rows:
        m = re.search(r"(k_wf_\w+?)(?:<|I|$)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1) + ("<-1>" if "-1" in r["Kernel_Name"] or "ILin1" in r["Kernel_Name"] else ""); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = []
for k in sorted(acc):
    a = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
    w = a.get("SQ_WAVES", 1) or 1
    out.append("%s: waves %d, clocks (BUSY_CYCLES/32... raw) %.0f" % (k, w, a.get("SQ_BUSY_CYCLES", 0)))
    out.append("   per wave: " + "  ".join("%s %.0f" % (c.replace("SQ_", ""), v / w) for c, v in sorted(a.items()) if c not in ("SQ_WAVES", "SQ_BUSY_CYCLES")))
open(O + "/pmc_pcg.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
