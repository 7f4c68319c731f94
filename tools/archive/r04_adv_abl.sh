#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
for round in 1 2; do
for n in "$@"; do
  echo "== $n: $(TFL_LIBRARY=$PWD/ab/$n.so timeout 120 python tools/adv_abl.py 2>&1 | tail -2 | tr '\n' ' ')"
done
done
