#!/bin/bash
# conv kernels on THIN z-windows (one rank of an 8-rank slab layout): z-marched vs tiled forms, chunk lengths
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
run() { local label=$1 res=$2 world=$3; shift; shift; shift
  echo "== $label res $res world $world"
  env "$@" timeout 100 python tools/slab_host_cost.py $res $world --kernels --still 2>&1 | grep -E "rank|k_conv|sum "
}
for res in 128 256; do
  run default $res 8 TFL_X=1
  run cz4 $res 8 TFL_M16_CZ=4 TFL_M16_CZ_IN=4
  run cz6 $res 8 TFL_M16_CZ=6 TFL_M16_CZ_IN=6
  run tiled_all $res 8 TFL_M16_KPACK=2 TFL_M16_TILED=3
  run tiled_in $res 8 TFL_M16_KPACK=2
done
run default 128 4 TFL_X=1
run cz6 128 4 TFL_M16_CZ=6 TFL_M16_CZ_IN=6
run tiled_all 128 4 TFL_M16_KPACK=2 TFL_M16_TILED=3
run default 128 2 TFL_X=1
run tiled_all 128 2 TFL_M16_KPACK=2 TFL_M16_TILED=3
