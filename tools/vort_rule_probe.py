"""Wall time of the native step on non-cubic grids whose planes fill k_vort_pipe's 64 x 16 tiles exactly (the round-6 rule of
vorticity_confinement_fused_ok: such grids take the fused kernel from 0.6 M cells on). Run once as it is and once under
TFL_VORT_FUSED=0 (two launches) and compare. usage: vort_rule_probe.py Z,Y,X [Z,Y,X ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_hip_simulate as T
from fluidnet_amd import FluidNetModel
from fluidnet_amd.simulate import simulate_native
dev = torch.device("cuda:0")
for arg in sys.argv[1:]:
    Z, Y, X = (int(v) for v in arg.split(","))
    b = T._to_dev(T._plume_batch((Z, Y, X), 0.15, 0.6, obstacles_seed=11), dev)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0, gravityScale=0.0,
                 vorticityConfinementAmp=2.0, simMethod="convnet")
    model = FluidNetModel.default_3d(seed=1)
    for _ in range(10):
        simulate_native(None, mconf, b, model)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(40):
            simulate_native(None, mconf, b, model)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40)
    print("%dx%dx%d (%.2f M cells) TFL_VORT_FUSED=%s: %.4f ms per step" % (Z, Y, X, Z * Y * X / 1e6, os.environ.get("TFL_VORT_FUSED", "-"), best * 1e3))
