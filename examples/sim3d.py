#!/usr/bin/env python
"""examples/sim3d.py -- the reference's 3-D plume driver (torch/fluid_net_3d_sim.lua) on the MI355X path.

Same sequence as the Lua script: an empty res^3 domain (:62-69), a voxel model loaded from a .binvox at HALF the grid
resolution, lined up with two diagonal flips, padded to the grid and written into the flags inside the 1-cell border
(:90-132, lib/obstacles_import_binvox.lua:52-119, lib/voxel_utils.lua), plume boundary conditions (:149-152), then
numFrames x tfluids.simulate() -- here ONE C-ABI call per step, tfl_simulate_step -- dumping the obstacle occupancy
(geom_output.vbox, geom_output_blender.vbox with the border shell cleared) after the first frame and the density every
outputDecimation frames (density_output.vbox), in the .vbox layout the Blender importer reads (:155-190, :266-291).

The bunny / arch .binvox files are not in the reference tree: `--model procedural` writes a stand-in model (sphere on
a pedestal + torus) with fluidnet_amd.io.saveVoxelData and loads THAT file back through the reference-faithful reader,
so the whole binvox -> flags -> simulate -> vbox path runs. No projection model ships for 3-D either (the reference's
was trained by the user): seeded weights of the 3-D default topology, or --sim jacobi / pcg.

  python examples/sim3d.py --res 64 --frames 24 --out /tmp/sim3d [--model path.binvox] [--sim convnet|jacobi|pcg]
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidnet_amd import FluidNetModel, io, tfluids  # noqa: E402
from fluidnet_amd import simulate as sim  # noqa: E402


def procedural_model(path, model_res):
    """A capped model in a model_res^3 box: a sphere standing on a short pedestal, a torus above it."""
    n = model_res
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    c = (n - 1) / 2.0
    sphere = (x - c) ** 2 + (y - 0.42 * n) ** 2 + (z - c) ** 2 <= (0.22 * n) ** 2
    pedestal = ((x - c) ** 2 + (z - c) ** 2 <= (0.10 * n) ** 2) & (y >= 0.10 * n) & (y <= 0.30 * n)
    rho = np.sqrt((x - c) ** 2 + (z - c) ** 2) - 0.30 * n
    torus = rho ** 2 + (y - 0.78 * n) ** 2 <= (0.06 * n) ** 2
    vox = (sphere | pedestal | torus).astype(np.float32)
    io.saveVoxelData(path, vox)


def build(res, model_path, dev, offsets=(0.04, 0.0, 0.04)):
    """batch + flags as fluid_net_3d_sim.lua:62-132 builds them (batch size 1)."""
    flags = torch.empty(1, 1, res, res, res, device=dev)
    tfluids.emptyDomain(flags, True)
    batch = dict(pDiv=torch.zeros(1, 1, res, res, res, device=dev), UDiv=torch.zeros(1, 3, res, res, res, device=dev),
                 flags=flags, density=torch.zeros(1, 1, res, res, res, device=dev))
    if model_path:
        vox = io.loadVoxelData(model_path)["data"]
        vox = io.flipDiagonal(vox, 2)
        vox = io.flipDiagonal(vox, 0)
        vox = io.padVoxelsToDims(res, res, res, vox, offsets[0] * res, offsets[1] * res, offsets[2] * res)
        f = flags.cpu().numpy()
        io.voxelsToFlags(f, vox)
        batch["flags"] = torch.from_numpy(f).to(dev)
    return batch


def mconf_of(res, sim_method):
    # fluid_net_3d_sim.lua:72-86
    return dict(dt=0.1, buoyancyScale=2.0 * (res / 128.0), gravityScale=0, maccormackStrength=0.6, maxIter=34,
                vorticityConfinementAmp=3, advectionMethod="maccormackOurs", simMethod=sim_method)


def run(res, frames, out_dir, model_path, sim_method="convnet", decimation=3, model=None, device="cuda:0", quiet=False):
    dev = torch.device(device)
    os.makedirs(out_dir, exist_ok=True)
    if model_path == "procedural":
        model_res = 2 ** (int(math.floor(math.log2(res))) - 1)       # HALF the grid resolution, :95
        model_path = os.path.join(out_dir, "procedural_%d.binvox" % model_res)
        procedural_model(model_path, model_res)
    batch = build(res, model_path if model_path != "none" else None, dev)
    mconf = mconf_of(res, sim_method)
    if model is None and sim_method == "convnet":
        model = FluidNetModel.default_3d(seed=1)
    sim.createPlumeBCs(batch, [1.0], 1.0 * (res / 128.0), 0.15)       # :149-152
    dens = io.VboxWriter(os.path.join(out_dir, "density_output.vbox"), res, res, res, frames)
    geom = io.VboxWriter(os.path.join(out_dir, "geom_output.vbox"), res, res, res, 1)
    geom_b = io.VboxWriter(os.path.join(out_dir, "geom_output_blender.vbox"), res, res, res, 1)
    t0 = None
    for i in range(1, frames + 1):
        if i == 2:
            torch.cuda.synchronize(dev)
            t0 = time.time()              # the first frame is not timed, :232-234
        sim.simulate_native(None, mconf, batch, model)
        if i == 1:
            occ = torch.empty_like(batch["flags"])
            tfluids.flagsToOccupancy(batch["flags"], occ)             # :268-273
            o = occ.cpu().numpy()[0, 0]
            geom.write(o)
            o = o.copy()
            o[0], o[-1], o[:, 0], o[:, -1], o[:, :, 0], o[:, :, -1] = 0, 0, 0, 0, 0, 0   # :277-280
            geom_b.write(o)
        if i % decimation == 0:
            dens.write(batch["density"].mean(dim=1).cpu().numpy())    # greyscale density, :286-290
    torch.cuda.synchronize(dev)
    ms = 1000.0 * (time.time() - t0) / max(frames - 1, 1) if t0 else float("nan")
    for w in (dens, geom, geom_b):
        w.close()
    if not quiet:
        print("All done!  Processing time: %.3f ms per frame (%d^3, %s, obstacle cells %d)"
              % (ms, res, sim_method, int((batch["flags"] == 2).sum()) - (res ** 3 - (res - 2) ** 3)))
    return batch, mconf, model


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--out", default="/tmp/sim3d")
    ap.add_argument("--model", default="procedural", help="a .binvox file, 'procedural' or 'none'")
    ap.add_argument("--sim", default="convnet", choices=["convnet", "jacobi", "pcg"])
    a = ap.parse_args()
    run(a.res, a.frames, a.out, a.model, a.sim)
