"""Chunk length (planes a block of the z-marched conv kernels walks) against per-kernel time, on full grids and on the thin
z-windows of slab ranks. TFL_M16_CZ / TFL_M16_CZ_IN are read at every launch, so one process sweeps them.
usage: python tools/conv_cz_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fluidnet_amd import FluidNetModel, tfluids
dev = torch.device("cuda:0")
model = FluidNetModel.default_3d(seed=1)
SHAPES = [(24, 128, 128), (40, 128, 128), (68, 128, 128), (128, 128, 128), (40, 256, 256), (72, 256, 256), (136, 256, 256)]
CZS = [0, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 32, 40, 64]
for Z, Y, X in SHAPES:
    p = torch.randn(1, 1, Z, Y, X, device=dev); U = torch.randn(1, 3, Z, Y, X, device=dev); f = torch.ones(1, 1, Z, Y, X, device=dev)
    for rep in range(2):
        for cz in CZS:
            if cz > Z + 8: continue
            for k in ("TFL_M16_CZ", "TFL_M16_CZ_IN"):
                if cz: os.environ[k] = str(cz)
                else: os.environ.pop(k, None)
            for _ in range(3): out = model.forward([p, U, f])
            if cz == 0 and rep == 0: ref = out[0].clone()
            diff = float((out[0] - ref).abs().max())
            n = 20
            with tfluids.profile(U) as prof:
                for _ in range(n): model.forward([p, U, f])
            print("%3dx%dx%d rep %d cz %2d  " % (Z, Y, X, rep, cz) + "  ".join("%s %6.1f" % (k[8:], v["ms"] / v["calls"] * 1e3) for k, v in sorted(prof.kernels.items()) if "conv" in k) + "  maxdiff %.1e" % diff, flush=True)
