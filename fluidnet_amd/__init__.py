"""fluidnet_amd -- MI355X-native tfluids.simulate() hot path behind FluidNet's tfluids.* API.

  fluidnet_amd.tfluids    host mirror of torch/tfluids/init.lua (operators; ctypes over the C ABI)
  fluidnet_amd.simulate   host mirror of torch/lib/simulate.lua (simulate, setConstVals, createPlumeBCs)
  fluidnet_amd.model      the `default` projection ConvNet (lib/model.lua) over tfl_model_forward
  fluidnet_amd.dist       z-slab decomposition across GPUs: halo exchange + 1 all-reduce per step (RCCL)
  fluidnet_amd.csrc/      hand-written HIP kernels for gfx950 + the C ABI (include/tfluids_hip.h)
"""
from . import tfluids  # noqa: F401
from . import simulate  # noqa: F401  (module: simulate.simulate, .createPlumeBCs, .setConstVals)
from .model import FluidNetModel  # noqa: F401
from . import modules  # noqa: F401  (the tfluids nn.Modules as torch.nn.Modules with autograd)
from ._lib import TfluidsError  # noqa: F401
