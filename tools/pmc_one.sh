#!/bin/bash
# SQ counters of the kernels of one command (two PMC passes, --kernel-trace only) -> gpurun_out/<tag>/pmc_one.txt
# usage: tools/pmc_one.sh <tag> <kernel regex> -- <command...>
REPO=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
tag=$1; pat=$2; shift; shift; shift
O=$REPO/gpurun_out/$tag; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -o run -- "$@" > $O/pmc$i.log 2>&1
  cp "$(find $O/pmc$i -name '*counter_collection.csv' | head -1)" $O/pmc$i.csv; rm -rf $O/pmc$i
done
python - "$O" "$pat" <<'PY'
import csv, collections, re, sys
O, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in ("pmc1.csv", "pmc2.csv"):
    for r in csv.DictReader(open(O + "/" + f)):
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        if not m or not re.search(pat, m.group(1)): continue
        k = m.group(1) + " grid " + r.get("Grid_Size", "?"); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = []
for k in acc:
    a = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
    w = a.get("SQ_WAVES", 1) or 1
    out.append(k)
    out.append("   waves %d  clocks %.0f  per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f" % (w, a.get("SQ_BUSY_CYCLES", 0) / 32, a.get("SQ_INSTS_VALU", 0) / w, a.get("SQ_INSTS_SALU", 0) / w, a.get("SQ_INSTS_LDS", 0) / w, a.get("SQ_INSTS_VMEM_RD", 0) / w, a.get("SQ_INSTS_VMEM_WR", 0) / w))
    wc = a.get("SQ_WAVE_CYCLES", 1) or 1
    out.append("   of wave cycles: wait_any %.2f wait_inst_any %.2f active_inst_any %.2f active_valu %.2f active_lds %.2f active_sca %.2f; lds bank conflict cycles/wave %.0f" % (
        a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_VALU", 0) / wc, a.get("SQ_ACTIVE_INST_LDS", 0) / wc, a.get("SQ_ACTIVE_INST_SCA", 0) / wc, a.get("SQ_LDS_BANK_CONFLICT", 0) / w))
open(O + "/pmc_one.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
