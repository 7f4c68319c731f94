"""The measured difference between the native z-slab step on virtual ranks and the unsplit step at BASELINE's sizes (what
tests/test_hip_fullsize.py::test_zslab_decomposition_at_baseline_size bounds by 1e-7): max over ranks of rel-L2 per field, and
the number of differing cells. usage: python tools/zslab_measure.py  (one GPU; ~1 min)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_hip_fullsize as T
from fluidnet_amd import FluidNetModel
from fluidnet_amd.dist import SlabLayout, SlabSimulation, ThreadComm, run_virtual_ranks
from fluidnet_amd.simulate import simulate_native

for res, world, config in [(256, 8, 5), (128, 8, 4), (128, 2, 4)]:
    ref, mconf = T._scene(res, config)
    model = FluidNetModel.default_3d(seed=1)
    for _ in range(8):
        simulate_native(None, mconf, ref, model)
    hub = ThreadComm.Hub(world)
    sims = []
    for r in range(world):
        lay = SlabLayout(res, world, r)
        loc = {k: (lay.extract(v) if torch.is_tensor(v) else v) for k, v in ref.items()}
        sims.append(SlabSimulation(loc, mconf, FluidNetModel.default_3d(seed=1), lay, ThreadComm(hub, r), own_context=True))
    for _ in range(3):
        simulate_native(None, mconf, ref, model)
    run_virtual_ranks(sims, 3)
    out = []
    for k in ("pDiv", "UDiv", "density"):
        worst, ndiff = 0.0, 0
        for s in sims:
            got, want = s.lay.owned(s.batch[k]), ref[k][:, :, s.lay.z0:s.lay.z1]
            worst = max(worst, float((got - want).norm() / want.norm().clamp_min(1e-30)))
            ndiff += int((got != want).sum())
        out.append("%s rel-L2 %.3g, %d cells differ" % (k, worst, ndiff))
    for s in sims:
        s.close()
    print("%d^3 in %d z-slabs (config %d), 3 steps after 8 of warm-up: " % (res, world, config) + "; ".join(out))
