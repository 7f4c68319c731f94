#!/bin/bash
# FETCH_SIZE per launch of the halo-reading kernels under the hardware's block order (TFL_XCD_ORDER=0) and the XCD-contiguous
# one (=1), at 128^3 and 256^3 (own PMC pass, --kernel-trace only) -> gpurun_out/<tag>/xcd_traffic.txt
REPO=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
tag=${1:-xcd}; O=$REPO/gpurun_out/$tag; mkdir -p $O
for res in 128 256; do for o in ${ORDERS:-0 1}; do
  RES=$res TFL_XCD_ORDER=$o timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p -o run -- python $REPO/tools/halo_ops.py > $O/p.log 2>&1
  cp "$(find $O/p -name '*counter_collection.csv' | head -1)" $O/fetch_${res}_$o.csv; rm -rf $O/p
done; done
python - "$O" "${ORDERS:-0 1}" <<'PY'
import csv, collections, re, sys
O = sys.argv[1]; orders = sys.argv[2].split()
out = ["# 2 x FETCH_SIZE (KiB -> bytes, gfx950 correction) per cell and launch, tools/halo_ops.py; columns: block order " + " / ".join(orders)]
for res in (128, 256):
    tab = collections.OrderedDict()
    for o in orders:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open("%s/fetch_%d_%s.csv" % (O, res, o))):
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            if m and r["Counter_Name"] == "FETCH_SIZE": acc[m.group(1)].append(float(r["Counter_Value"]))
        for k, v in acc.items(): tab.setdefault(k, {})[o] = sum(v) / len(v) * 1024 * 2 / res ** 3
    for k, v in tab.items():
        if k.startswith(("k_vel3", "k_scal3", "k_curl", "k_confine", "k_vort")):
            out.append("%d^3 %-40s " % (res, k[:40]) + "  ".join("%6.1f" % v.get(o, float("nan")) for o in orders) + "  B/cell read")
open(O + "/xcd_traffic.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
