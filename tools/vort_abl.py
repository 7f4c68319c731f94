"""Per-kernel times of the vorticity confinement alone at 128^3, 192^3 and 256^3: the two launches (in place) and the fused kernel
(tfl_vorticityConfinementFrom; TFL_VORT_FUSED=1 forces it below its size threshold), smooth random velocity, a border of
obstacle cells. usage: [RES=128,192,256] [TFL_VORT_FUSED=1] [TFL_VORT_PIPE=0|1] [TFL_VORT_CZ=n] python tools/vort_abl.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fluidnet_amd import tfluids
dev = torch.device("cuda:0")
for res in [int(x) for x in os.environ.get("RES", "128,192,256").split(",")]:
    g = torch.Generator(device=dev); g.manual_seed(1)
    U = torch.randn(1, 3, res, res, res, device=dev, generator=g)
    for _ in range(2):
        U = torch.nn.functional.avg_pool3d(U, 3, 1, 1)
    U = U.contiguous()
    fl = torch.ones(1, 1, res, res, res, device=dev)
    fl[:, :, 0] = 2; fl[:, :, -1] = 2; fl[:, :, :, 0] = 2; fl[:, :, :, -1] = 2; fl[..., 0] = 2; fl[..., -1] = 2
    a, b = U.clone(), torch.empty_like(U)
    for _ in range(2):
        tfluids.vorticityConfinement(a, fl, 0.05); tfluids.vorticityConfinement(b, fl, 0.05, USrc=U)
    n = 10
    with tfluids.profile(U) as prof:
        for _ in range(n):
            tfluids.vorticityConfinement(a, fl, 0.05)
            tfluids.vorticityConfinement(b, fl, 0.05, USrc=U)
    print("%d^3 " % res + "  ".join("%s %.1f" % (k[2:], v["ms"] / v["calls"] * 1e3) for k, v in sorted(prof.kernels.items())))
