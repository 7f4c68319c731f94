"""Per-kernel times of advectVel / advectScalar alone at 128^3 and 256^3 on a smooth random velocity field (|u| dt < 1 cell),
for the ablation builds of advect_vel3.hip. usage: TFL_LIBRARY=ab/<name>.so python tools/adv_abl.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fluidnet_amd import tfluids
dev = torch.device("cuda:0")
for res in (128, 256):
    g = torch.Generator(device=dev); g.manual_seed(1)
    U = torch.randn(1, 3, res, res, res, device=dev, generator=g)
    for _ in range(3):      # smooth it a little
        U = torch.nn.functional.avg_pool3d(U, 3, 1, 1)
    U = (U / U.abs().max() * 6.0).contiguous()
    fl = torch.ones(1, 1, res, res, res, device=dev)
    fl[:, :, 0] = 2; fl[:, :, -1] = 2; fl[:, :, :, 0] = 2; fl[:, :, :, -1] = 2; fl[..., 0] = 2; fl[..., -1] = 2
    rho = torch.rand(1, 1, res, res, res, device=dev, generator=g)
    for _ in range(2):
        tfluids.advectVel(0.1, U.clone(), fl, "maccormackOurs", maccormackStrength=0.6)
    n = 10
    Uc, rc = U.clone(), rho.clone()
    with tfluids.profile(U) as prof:
        for _ in range(n):
            tfluids.advectVel(0.1, Uc, fl, "maccormackOurs", maccormackStrength=0.6)
            tfluids.advectScalar(0.1, rc, U, fl, "maccormackOurs", maccormackStrength=0.6)
            Uc.copy_(U)
    print("%d^3 " % res + "  ".join("%s %.1f" % (k[2:], v["ms"] / v["calls"] * 1e3) for k, v in sorted(prof.kernels.items())))
