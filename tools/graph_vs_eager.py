"""bench.py's 128^3 step, eager (ten launches per tfl_simulate_step call) against the same call replayed as ONE HIP graph
(GraphedSimulate native=True), alternating blocks in one process. Round 6: the boxes of the pool differ in what lies BETWEEN
the kernels -- the same kernel times give 0.240 ms per step on some and 0.256 on others."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from fluidnet_amd import FluidNetModel
from fluidnet_amd.simulate import GraphedSimulate, simulate_native
res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
batch, mconf = bench.build_scene(res, res, None, dev)
model = FluidNetModel.default_3d(seed=1)
for _ in range(16):
    simulate_native(None, mconf, batch, model)
g = GraphedSimulate(None, mconf, batch, model, native=True)
def block(fn, n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eager = lambda: simulate_native(None, mconf, batch, model)
for _ in range(10): g.step()
te, tg = [], []
for _ in range(5):
    te.append(block(eager)); tg.append(block(g.step))
print("res %d: eager %s  median %.4f ms | graph replay %s  median %.4f ms | copy rate %.0f GB/s" % (
    res, " ".join("%.4f" % v for v in te), sorted(te)[2], " ".join("%.4f" % v for v in tg), sorted(tg)[2], bench.measured_hbm_GBps(dev)))
assert bool(torch.isfinite(batch["UDiv"]).all())
