#!/bin/bash
# round 6, first GPU session: the suite, the slab rank-step (eager / graph, one / two streams), a bench line
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r06a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
for r in 128 256; do
  timeout 300 python tools/slab_host_cost.py $r 8 --still --kernels > $O/slab_$r.txt 2>&1; echo "slab $r rc=$?"
  TFL_SIDE_STREAM=0 timeout 300 python tools/slab_host_cost.py $r 8 --still > $O/slab_${r}_onestream.txt 2>&1
done
timeout 200 python tools/slab_host_cost.py 128 8 --still --python-null > $O/slab_128_pynull.txt 2>&1
cat $O/slab_*.txt | grep -v amdgpu.ids
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 900 $O/bench.json; echo
