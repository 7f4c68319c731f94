"""Per-kernel summary of one rocprofv3 SQ counter pass (SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU) -> profiles/<tag>_pmc_sq.txt.
usage: python tools/pmc_sq.py <counter_collection.csv> <tag> <commit>"""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag, commit = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
order = []
for r in csv.DictReader(open(src)):
    m = re.search(r"::(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    if not m:
        continue
    k = m.group(1)
    if k not in acc:
        order.append(k)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
rows = []
for k in order:
    a = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    rows.append((a.get("SQ_BUSY_CYCLES", 0.0), k, cnt[k]["SQ_WAVES"], a.get("SQ_WAVES", 0), a.get("SQ_WAIT_ANY", 0) / wc,
                 a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                 a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (a.get("SQ_BUSY_CYCLES", 0) or 1.0) / 32.0,
                 a.get("SQ_INSTS_VALU", 0) / (a.get("SQ_WAVES", 0) or 1.0), a.get("SQ_BUSY_CYCLES", 0) / 32.0))
rows.sort(reverse=True)
out = ["# rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU",
       "# (own pass, --kernel-trace only), bench.py 128^3, commit %s. per-launch averages; wait_any/wait_inst/active = fractions of" % commit,
       "# SQ_WAVE_CYCLES; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the 1024 SIMDs) / (1024 x clocks): the busy fraction of",
       "# the matrix pipes; valu/wave = SQ_INSTS_VALU / SQ_WAVES; clocks = SQ_BUSY_CYCLES / 32 (the counter is summed over 32 shader engines)",
       "%-46s %4s %9s %8s %8s %8s %9s %10s %9s" % ("kernel", "n", "waves", "wait_any", "wait_ins", "active", "mfma_util", "valu/wave", "clocks")]
for _, k, n, w, wa, wi, ac, mf, vw, clk in rows:
    out.append("%-46s %4d %9d %8.2f %8.2f %8.2f %9.2f %10.0f %9.0f" % (k[:46], n, w, wa, wi, ac, mf, vw, clk))
open(os.path.join(ROOT, "profiles", tag + "_pmc_sq.txt"), "w").write("\n".join(out) + "\n")
# the file bench.py reads for roofline.valu_issue_frac / mfma_util: per PROFILER kernel name (tools/pmc_traffic.py short()),
# with the git blob hash of the kernel's source at measurement time (bench.py refuses the figures when it differs)
if len(sys.argv) > 4 and sys.argv[4] == "json":
    import json
    sys.path.insert(0, os.path.join(ROOT, "fluidnet_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _kernels
    saved_argv, sys.argv = sys.argv, sys.argv[:1]
    from pmc_traffic import short
    sys.argv = saved_argv
    j = {}
    for _, k, n, w, wa, wi, ac, mf, vw, clk in rows:
        s = short("tfl::" + k)
        if s and s not in j:
            j[s] = {"pmc_name": k, "waves": w, "valu_per_wave": vw, "clocks": clk, "mfma_util": mf, "wait_any": wa, "wait_inst": wi, "active": ac}
    conv_path = os.environ.get("TFL_CONV_PATH", "mfma16")
    j["_meta"] = {"commit": commit, "source": tag + "_pmc_sq.txt", "simds": 1024,
                  "source_sha": {k: list(_kernels.source_sha(k, conv_path)) for k in j}}
    json.dump(j, open(os.path.join(ROOT, "profiles", "pmc_sq.json"), "w"), indent=1)
print("\n".join(out[:12]))
