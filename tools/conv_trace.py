"""Per-block phase timestamps of the 3-D conv layers (TFL_CONV_TRACE=1; development aid). usage: conv_trace.py [res]"""
import os
import sys

os.environ["TFL_CONV_TRACE"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_scene  # noqa: E402
from fluidnet_amd import FluidNetModel  # noqa: E402
from fluidnet_amd.simulate import simulate  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
model = FluidNetModel.default_3d(seed=1)
batch, mconf = build_scene(res, res, None, dev)
for _ in range(4):
    simulate(None, mconf, batch, model)
torch.cuda.synchronize()
