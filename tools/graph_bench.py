"""eager vs HIP-graph step time for the BASELINE configs (development aid)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
from oracle import simulate_np as S
from test_hip_simulate import _plume_batch, _to_dev, _layers2d
from fluidnet_amd import FluidNetModel
from fluidnet_amd.simulate import GraphedSimulate, simulate, simulate_native
dev=torch.device('cuda:0')
def run(name, dims, mconf, model, rad, usc, steps=200):
    b=_to_dev(_plume_batch(dims, rad, usc), dev)
    for _ in range(20): simulate(None, mconf, b, model)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(steps): simulate(None, mconf, b, model)
    torch.cuda.synchronize(); te=(time.perf_counter()-t0)/steps
    for _ in range(5): simulate_native(None, mconf, b, model)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(steps): simulate_native(None, mconf, b, model)
    torch.cuda.synchronize(); tn=(time.perf_counter()-t0)/steps
    g=GraphedSimulate(None, mconf, b, model)
    for _ in range(5): g.step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(steps): g.step()
    torch.cuda.synchronize(); tg=(time.perf_counter()-t0)/steps
    n=np.prod(dims)
    print("%-34s eager %7.3f ms (%7.0f steps/s)   native step %7.3f ms   graph %7.3f ms (%7.0f steps/s, %7.1f Mcells/s)"%(name, te*1e3, 1/te, tn*1e3, tg*1e3, 1/tg, n/tg/1e6))
m2=dict(dt=4/60, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=1.0, gravityScale=0, vorticityConfinementAmp=0)
run("cfg1 2D 64^2 jacobi20", (1,64,64), dict(m2, simMethod="jacobi", maxIter=20), None, 0.05, 10.0)
run("cfg2 2D 128^2 convnet", (1,128,128), dict(m2, simMethod="convnet"), FluidNetModel(_layers2d(), False), 0.05, 10.0)
m3=dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, gravityScale=0, simMethod="convnet")
mod=FluidNetModel(S.default_3d_layers(seed=1), True)
run("cfg3 3D 64^3 convnet", (64,64,64), dict(m3, buoyancyScale=1.0, vorticityConfinementAmp=0), mod, 0.15, 0.5)
run("cfg4-like 3D 128^3 convnet+vort", (128,128,128), dict(m3, buoyancyScale=2.0, vorticityConfinementAmp=3.0), mod, 0.15, 1.0, steps=100)
run("cfg5-size 3D 256^3 convnet 1 GPU", (256,256,256), dict(m3, buoyancyScale=4.0, vorticityConfinementAmp=0), mod, 0.15, 2.0, steps=20)
