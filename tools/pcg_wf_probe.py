"""Cost of one IC(0) preconditioner application (skew + forward sweep + backward sweep + unskew) on box domains chosen to
isolate the parts of the pipelined-wavefront sweeps: one sub-box, a chain of slabs (z), a chain of strips (y).
usage: pcg_wf_probe.py [Z,Y,X ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidnet_amd import tfluids  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(18, 66, 128), (130, 66, 128), (18, 130, 128), (130, 130, 128)]
for Z, Y, X in shapes:
    flags = torch.full((1, 1, Z, Y, X), 2.0, device=dev)
    flags[:, :, 1:-1, 1:-1, 1:-1] = 1.0
    g = torch.Generator(device="cpu").manual_seed(1)
    div = torch.zeros_like(flags)
    div[:, :, 1:-1, 1:-1, 1:-1] = torch.randn((Z - 2, Y - 2, X - 2), generator=g).to(dev)
    p = torch.zeros_like(div)
    tot = {}
    for iters in (8, 40):
        tfluids.solveLinearSystemPCG(p, flags, div, True, 1e-30, iters, "ic0")
        torch.cuda.synchronize()
        with tfluids.profile(p) as prof:
            tfluids.solveLinearSystemPCG(p, flags, div, True, 1e-30, iters, "ic0")
        tot[iters] = (prof.kernels["k_pcg_precond"]["ms"], prof.kernels["k_pcg_precond"]["calls"], prof.kernels["k_pcg"]["ms"], prof.kernels["k_pcg"]["calls"])
    dpre = (tot[40][0] - tot[8][0]) / (tot[40][1] - tot[8][1]) * 1e3
    dcg = (tot[40][2] - tot[8][2]) / (tot[40][3] - tot[8][3]) * 1e3
    print(f"{Z}x{Y}x{X}: preconditioner application {dpre:8.1f} us   CG iteration without it {dcg:6.1f} us")
