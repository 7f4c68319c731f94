"""Host enqueue time vs GPU time of one eager simulate() step (bench scene), and the HIP-graph replay time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_scene  # noqa: E402
from fluidnet_amd import FluidNetModel  # noqa: E402
from fluidnet_amd.simulate import GraphedSimulate, simulate  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
model = FluidNetModel.default_3d(seed=1)
batch, mconf = build_scene(res, res, None, dev)
for _ in range(20):
    simulate(None, mconf, batch, model)
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    simulate(None, mconf, batch, model)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: host enqueue {1e3 * (t1 - t0) / N:.4f} ms/step, wall {1e3 * (t2 - t0) / N:.4f} ms/step")
g = GraphedSimulate(None, mconf, batch, model)
for _ in range(10):
    g.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    g.step()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"graph: wall {1e3 * (t2 - t0) / N:.4f} ms/step")
