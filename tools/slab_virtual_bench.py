"""Per-rank compute of the z-slab step estimated on ONE GPU: W virtual ranks (threads, ThreadComm: halos move by device
copies) advance the bench scene; all ranks share one stream, so wall time / W ~ the kernel time one real rank spends per
step (no RCCL, no overlap). Also prints rank 0's per-kernel table. usage: slab_virtual_bench.py [res] [world] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluidnet_amd import FluidNetModel, tfluids  # noqa: E402
from fluidnet_amd.dist import SlabLayout, SlabSimulation, ThreadComm, run_virtual_ranks  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
model = FluidNetModel.default_3d(seed=1)
hub = ThreadComm.Hub(world)
sims = []
for r in range(world):
    lay = SlabLayout(res, world, r)
    batch, mconf = bench.build_scene(res, res, lay, dev)
    sims.append(SlabSimulation(batch, mconf, model, lay, ThreadComm(hub, r), own_context=True))
run_virtual_ranks(sims, 6)
torch.cuda.synchronize()
t0 = time.time()
run_virtual_ranks(sims, steps)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("res %d, %d virtual ranks: %.3f ms per step for all ranks = %.3f ms per rank-step (single GPU un-split: see bench.py)"
      % (res, world, dt * 1e3, dt * 1e3 / world))
with tfluids.profile(sims[0].batch["UDiv"]) as prof:
    run_virtual_ranks(sims, 5)
tot = 0.0
for name, rec in sorted(prof.kernels.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-28s %7.1f us/step  (%.1f launches)" % (name, rec["ms"] / 5 * 1e3, rec["calls"] / 5))
    tot += rec["ms"] / 5
print("  rank 0 kernels: %.1f us/step" % (tot * 1e3))
