#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.txt
echo "== rsq A/B"; bash tools/ab_run.sh "vel|scalar" rsq3 rsq1 2>&1 | tee $O/ab_rsq.txt
echo "== stats consumer A/B"
for r in 1 2; do for v in 0 1; do echo "-- TFL_STATS_CONSUMER=$v"; TFL_STATS_CONSUMER=$v python bench.py --no-cpu-baseline --no-config5 --steps 30 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|conv3_in|reduce|bcs"; done; done 2>&1 | tee $O/ab_stats.txt
for r in 128 256; do timeout 300 python tools/slab_host_cost.py $r 8 --still --kernels > $O/slab_$r.txt 2>&1; echo "slab $r rc=$?"; done
timeout 200 python tools/slab_host_cost.py 128 2 --still > $O/slab_128_w2.txt 2>&1
cat $O/slab_*.txt | grep -v amdgpu.ids
