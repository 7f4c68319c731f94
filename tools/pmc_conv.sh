#!/bin/bash
# SQ counters of the conv kernels (two --pmc passes, kernel-trace only) -> gpurun_out/pmc_conv/*.csv + summary
# usage: tools/pmc_conv.sh [env assignments for the bench, e.g. TFL_CONV_PATH=mfma]
REPO=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
O=$REPO/gpurun_out/pmc_conv; mkdir -p $O
tag=${1:-default}; shift
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/tmp -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --steps 6 --warmup 2 --preroll 2 > $O/${tag}_$i.log 2>&1
  cp "$(find $O/tmp -name '*counter_collection.csv' | head -1)" $O/${tag}_$i.csv; rm -rf $O/tmp
done
python - "$O" "$tag" <<'P'
import csv, sys, collections
O, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for i in (1, 2, 3):
    try: rows = list(csv.DictReader(open("%s/%s_%d.csv" % (O, tag, i))))
    except Exception as e: print("pass", i, e); continue
    for r in rows:
        k = r["Kernel_Name"]
        if "conv3" not in k: continue
        k = k.split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    print("==", k)
    for c in sorted(acc[k]):
        print("  %-26s %14.0f per launch" % (c, acc[k][c] / n[k][c]))
P
