"""Where does the z-marched fp16-MFMA kernel differ from the tile kernel? (TFL_M16_TILED bit 0: mid, bit 1: tail)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import scenes
from oracle import simulate_np as S
from fluidnet_amd import FluidNetModel
os.environ["TFL_CONV_PATH"] = "mfma16"
layers = S.default_3d_layers(seed=5)
dev = torch.device("cuda:0")
dims = tuple(int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 32, 32)))
sc = scenes.make_scene(dims, seed=51, vel_cells=0.4, B=1)
tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
out = {}
for mode in ("3", "2", "1", "0"):
    os.environ["TFL_M16_TILED"] = mode
    p, U = FluidNetModel(layers, True).forward([tp, tU, tf])
    out[mode] = p.cpu().numpy()[0, 0]
ref = out["3"]
for mode in ("2", "1", "0"):
    e = np.abs(out[mode] - ref)
    print("mode", mode, "(bit0 mid tiled, bit1 tail tiled): max err %.3e rel-l2 %.3e" % (e.max(), np.linalg.norm(e) / np.linalg.norm(ref)))
    bad = e > 1e-4 * np.abs(ref).max()
    print("  bad fraction %.4f; per-z bad counts" % bad.mean(), bad.sum(axis=(1, 2)).tolist())
    print("  per-y bad counts", bad.sum(axis=(0, 2)).tolist())
    print("  per-x bad counts", bad.sum(axis=(0, 1)).tolist())
