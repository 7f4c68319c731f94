"""Per-operator timing on one MI355X (development aid; bench.py is the judged harness)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import scenes
from fluidnet_amd import tfluids

BYTES3D = dict(advectScalar=52, advectVel=68, addBuoyancy=32, vorticityConfinement=72, setWallBcs=28,
               divergence=20, velocityUpdate=32, jacobi=16)

def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    sc = scenes.make_scene((res, res, res), seed=1, vel_cells=1.5)
    f = torch.from_numpy(sc["flags"]).to(dev); U = torch.from_numpy(sc["U"]).to(dev)
    rho = torch.from_numpy(sc["density"]).to(dev); p = torch.from_numpy(sc["p"]).to(dev)
    div = torch.zeros_like(p); Ud = torch.zeros_like(U); sd = torch.zeros_like(rho)
    N = res ** 3
    ops = {
        "advectScalar": lambda: tfluids.advectScalar(0.1, rho, U, f, "maccormackOurs", sd),
        "advectVel": lambda: tfluids.advectVel(0.1, U, f, "maccormackOurs", Ud),
        "addBuoyancy": lambda: tfluids.addBuoyancy(Ud, f, rho, [0, -0.01, 0], 0.1),
        "vorticityConfinement": lambda: tfluids.vorticityConfinement(Ud, f, 0.01),
        "setWallBcs": lambda: tfluids.setWallBcsForward(Ud, f),
        "divergence": lambda: tfluids.velocityDivergenceForward(U, f, div),
        "velocityUpdate": lambda: tfluids.velocityUpdateForward(Ud, f, p),
    }
    for name, fn in ops.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = BYTES3D[name] * N / (ms * 1e-3) / 1e9
        print("%-22s %8.3f ms  %8.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (name, ms, gbs, gbs / 80.0))
    print("trace errors:", tfluids.traceErrors(U))

main()
