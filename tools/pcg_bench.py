"""Time the baseline PCG pressure solve (tfl_solveLinearSystemPCG) on the bench scene's divergence. usage: pcg_bench.py [res]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_scene  # noqa: E402
from fluidnet_amd import FluidNetModel, tfluids  # noqa: E402
from fluidnet_amd.simulate import simulate  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
model = FluidNetModel.default_3d(seed=1)
batch, mconf = build_scene(res, res, None, dev)
for _ in range(8):
    simulate(None, mconf, batch, model)
U, flags = batch["UDiv"].clone(), batch["flags"]
tfluids.setWallBcsForward(U, flags)
div = torch.empty_like(batch["pDiv"])
tfluids.velocityDivergenceForward(U, flags, div)
p = torch.zeros_like(div)
rhs = float(div.norm())
for pc, tol in (("none", 1e-3), ("none", 1e-4), ("ic0", 1e-3), ("ilu0", 1e-3)):
    tfluids.solveLinearSystemPCG(p, flags, div, True, tol, 3000, pc)   # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = tfluids.solveLinearSystemPCG(p, flags, div, True, tol, 3000, pc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    Un = U.clone()
    tfluids.velocityUpdateForward(Un, flags, p)
    d2 = torch.empty_like(div)
    tfluids.velocityDivergenceForward(Un, flags, d2)
    with tfluids.profile(p) as prof:
        tfluids.solveLinearSystemPCG(p, flags, div, True, tol, 3000, pc)
    kk = "  ".join("%s %d x %.1f us" % (k, v["calls"], v["ms"] / v["calls"] * 1e3) for k, v in sorted(prof.kernels.items()))
    print(f"   kernels: {kk}")
    print(f"{res}^3 precond {pc:5s} tol {tol:g}: {dt * 1e3:8.1f} ms  residual {r:.3e} (|rhs| {rhs:.2f})  max|div| {float(div.abs().max()):.3f} -> {float(d2.abs().max()):.2e}")
