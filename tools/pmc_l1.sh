#!/bin/bash
# L1 / texture-addresser counters of the advection kernels (separate --pmc passes, kernel-trace only) -> gpurun_out/pmc_l1/
# Every pass runs under `timeout`: a counter set the hardware cannot collect makes rocprofv3 abort and then hang.
REPO=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
O=$REPO/gpurun_out/pmc_l1; mkdir -p $O
cd /tmp
i=0
for set in "TA_BUSY_avr TA_TA_BUSY_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TCC_NC_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum TCP_PERF_SEL_TOTAL_READ TCP_CACHE_MISS" \
           "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/tmp -o run -- python $REPO/bench.py --no-cpu-baseline --no-config5 --steps 6 --warmup 2 --preroll 2 > $O/pass_$i.log 2>&1
  cp "$(find $O/tmp -name '*counter_collection.csv' | head -1)" $O/pass_$i.csv 2>/dev/null; rm -rf $O/tmp
done
python - "$O" <<'P'
import csv, sys, collections, glob
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(O + "/pass_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("k_vel_", "k_scalar_", "k_confine", "k_conv3")): continue
        k = k.split("(")[0].replace("void tfl::", "")[:34]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    print("==", k)
    for c in sorted(acc[k]):
        print("  %-42s %16.0f per launch" % (c, acc[k][c] / n[k][c]))
P
