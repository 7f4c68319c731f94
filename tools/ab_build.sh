#!/bin/bash
# Build a variant of the library into ab/<name>.so (for A/B timing of kernel variants inside ONE gpurun session:
# `TFL_LIBRARY=ab/<name>.so python bench.py`): the listed source files are recompiled with the extra flags, every other
# object comes from the regular build (fluidnet_amd/csrc/build, `make` first).
# usage: tools/ab_build.sh <name> <file.hip[,file2.hip...]|all> [extra hipcc flags]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
name=$1; files=$2; shift; shift
mkdir -p "$REPO/ab/build_$name"
cd "$REPO/fluidnet_amd/csrc"
make -s
SRCS=$(sed -n 's/^SRCS := //p' Makefile)
[ "$files" = all ] && files=$(echo $SRCS | tr ' ' ',')
objs=()
for f in $SRCS; do
  if [[ ",$files," == *",$f,"* ]]; then
    o="$REPO/ab/build_$name/${f%.*}.o"
    extra=""; case $f in advect.hip|advect_vel3.hip|advect_scalar3.hip) extra="-fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function $extra -I"$REPO/include" "$@" -x hip -c -o "$o" "$f" &
  else
    o="build/${f%.*}.o"
  fi
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$REPO/ab/$name.so" "${objs[@]}" -ldl
ls -la "$REPO/ab/$name.so"
