"""The halo-reading operators alone (advectScalar, advectVel, vorticity confinement: two launches and fused) on a smooth random
velocity field at RES^3 (env RES, default 128), a few calls each: the workload of tools/xcd_traffic.sh's counter passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fluidnet_amd import tfluids
dev = torch.device("cuda:0")
res = int(os.environ.get("RES", "128"))
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(1, 3, res, res, res, device=dev, generator=g)
for _ in range(3):
    U = torch.nn.functional.avg_pool3d(U, 3, 1, 1)
U = (U / U.abs().max() * 6.0).contiguous()
fl = torch.ones(1, 1, res, res, res, device=dev)
fl[:, :, 0] = 2; fl[:, :, -1] = 2; fl[:, :, :, 0] = 2; fl[:, :, :, -1] = 2; fl[..., 0] = 2; fl[..., -1] = 2
rho = torch.rand(1, 1, res, res, res, device=dev, generator=g)
Uc, rc, b = U.clone(), rho.clone(), torch.empty_like(U)
for _ in range(int(os.environ.get("CALLS", "4"))):
    tfluids.advectVel(0.1, Uc, fl, "maccormackOurs", maccormackStrength=0.6)
    tfluids.advectScalar(0.1, rc, U, fl, "maccormackOurs", maccormackStrength=0.6)
    Uc.copy_(U)
    tfluids.vorticityConfinement(Uc, fl, 0.05)
    tfluids.vorticityConfinement(b, fl, 0.05, USrc=U)
torch.cuda.synchronize()
