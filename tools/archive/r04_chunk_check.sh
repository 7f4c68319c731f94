#!/bin/bash
# after pick_chunk(): parity of the conv / slab tests, the rule's picks on thin windows, the slab rank-step
REPO=$(cd "$(dirname "$0")/.." && pwd); cd "$REPO"; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_hip_simulate.py tests/test_native_transport.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
TFL_DEBUG=1 timeout 200 python - <<'PY' 2>&1 | grep -E "conv|x"
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from fluidnet_amd import FluidNetModel, tfluids
dev = torch.device("cuda:0")
model = FluidNetModel.default_3d(seed=1)
for Z, Y, X in [(24, 128, 128), (40, 128, 128), (68, 128, 128), (128, 128, 128), (40, 256, 256), (72, 256, 256), (136, 256, 256), (256, 256, 256)]:
    p = torch.randn(1, 1, Z, Y, X, device=dev); U = torch.randn(1, 3, Z, Y, X, device=dev); f = torch.ones(1, 1, Z, Y, X, device=dev)
    for _ in range(3): model.forward([p, U, f])
    with tfluids.profile(U) as prof:
        for _ in range(20): model.forward([p, U, f])
    print("%3dx%dx%d  " % (Z, Y, X) + "  ".join("%s %6.1f" % (k[8:], v["ms"] / v["calls"] * 1e3) for k, v in sorted(prof.kernels.items()) if "conv" in k), flush=True)
PY
for w in 8 4 2; do timeout 100 python tools/slab_host_cost.py 128 $w --kernels --still 2>&1 | grep -E "rank|k_conv|sum "; done
timeout 100 python tools/slab_host_cost.py 256 8 --kernels --still 2>&1 | grep -E "rank|k_conv|sum "
