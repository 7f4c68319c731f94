#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06d; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_simulate.py -m gpu -q -x -k "stat or forward or native_simulate or zslab_decomposition_equals" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
echo "== stats consumer A/B (v2: batched loads, in-wave tree)"
for r in 1 2; do for v in 0 1; do echo "-- TFL_STATS_CONSUMER=$v"; TFL_STATS_CONSUMER=$v python bench.py --no-cpu-baseline --no-config5 --steps 30 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|conv3_in|reduce|bcs"; done; done 2>&1 | tee $O/ab_stats.txt
for w in 2 4; do timeout 300 python tools/slab_host_cost.py 128 $w --still --kernels --no-graph > $O/slab_128_w$w.txt 2>&1; done
cat $O/slab_*.txt | grep -v amdgpu.ids
