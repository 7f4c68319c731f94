"""Per-kernel times of the ConvNet projection alone (tfl_model_forward) at 128^3 and 256^3, for the ablation builds of
conv_mfma16.hip (results may be garbage there: nothing is checked). usage: TFL_LIBRARY=ab/<name>.so python tools/conv_abl.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fluidnet_amd import FluidNetModel, tfluids
dev = torch.device("cuda:0")
model = FluidNetModel.default_3d(seed=1)
for res in (128, 256):
    p = torch.randn(1, 1, res, res, res, device=dev); U = torch.randn(1, 3, res, res, res, device=dev); f = torch.ones(1, 1, res, res, res, device=dev)
    for _ in range(3): model.forward([p, U, f])
    n = 10
    with tfluids.profile(U) as prof:
        for _ in range(n): model.forward([p, U, f])
    print("%d^3 " % res + "  ".join("%s %.1f" % (k[2:], v["ms"] / v["calls"] * 1e3) for k, v in sorted(prof.kernels.items()) if "conv" in k))
