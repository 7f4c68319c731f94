#!/bin/bash
# A/B timing on the GPU box: bench.py once per library in ab/ (alternating, two rounds), kernel table filtered.
# usage: tools/ab_run.sh "<grep pattern for kernels>" name1 name2 ...   (TFL_AB_ARGS: extra bench.py arguments)
pat=$1; shift
for round in 1 2; do
  for n in "$@"; do
    echo "== $n (round $round)"
    TFL_LIBRARY=$PWD/ab/$n.so python bench.py --no-cpu-baseline --no-config5 --steps 30 $TFL_AB_ARGS 2>/dev/null | python tools/bench_kernels.py | grep -E "ms/step|$pat"
  done
done
