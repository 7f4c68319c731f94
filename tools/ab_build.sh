#!/bin/bash
# Build the current csrc tree into ab/<name>.so (for A/B timing of kernel variants inside ONE gpurun session:
# `TFL_LIBRARY=ab/<name>.so python bench.py`). usage: tools/ab_build.sh <name> [extra hipcc flags]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$REPO/ab/build_$name"
cd "$REPO/fluidnet_amd/csrc"
objs=()
for f in abi.cpp simulate.cpp comm_rccl.cpp advect.hip advect_vel3.hip stencil.hip vorticity.hip jacobi.hip pcg.hip model.hip conv.hip conv_mfma.hip conv_valu.hip conv2d_mfma.hip backward.hip; do
  o="$REPO/ab/build_$name/${f%.*}.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -I"$REPO/include" "$@" -x hip -c -o "$o" "$f" &
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$REPO/ab/$name.so" "${objs[@]}"
ls -la "$REPO/ab/$name.so"
