#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
O=$REPO/gpurun_out/r04convabl; rm -rf $O; mkdir -p $O
for round in 1 2; do
for n in base nodma nomfma nostore nofrag noepi nobar noepi_nodma mfmaonly; do
  echo "== $n: $(TFL_LIBRARY=$PWD/ab/$n.so timeout 120 python tools/conv_abl.py 2>&1 | tail -2 | tr '\n' ' ')"
done
done 2>&1 | tee $O/abl.txt
