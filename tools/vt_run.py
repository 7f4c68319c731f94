"""per-step clock of k_vort_pipe (library built with -DTFL_VORT_TIMING: tools/ab_build.sh vt vorticity.hip -DTFL_VORT_TIMING)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from fluidnet_amd import FluidNetModel
from fluidnet_amd.simulate import simulate_native
res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
batch, mconf = bench.build_scene(res, res, None, dev)
model = FluidNetModel.default_3d(seed=1)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    simulate_native(None, mconf, batch, model)
torch.cuda.synchronize()
