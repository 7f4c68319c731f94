"""Step time of the ConvNet projection alone (tfl_model_forward) on small / slab-shaped 3-D grids (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fluidnet_amd import FluidNetModel
dev = torch.device("cuda:0")
model = FluidNetModel.default_3d(seed=1)
for dims in [(32, 32, 32), (64, 64, 64), (24, 128, 128), (40, 128, 128), (40, 256, 256), (128, 128, 128)]:
    Z, Y, X = dims
    p = torch.randn(1, 1, Z, Y, X, device=dev); U = torch.randn(1, 3, Z, Y, X, device=dev); f = torch.ones(1, 1, Z, Y, X, device=dev)
    for _ in range(5): model.forward([p, U, f])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n): model.forward([p, U, f])
    torch.cuda.synchronize()
    print("%-16s model.forward %.1f us" % ("x".join(map(str, dims)), (time.perf_counter() - t0) / n * 1e6))
