"""Would advectScalar (3 kernels) and advectVel (2 kernels) -- independent of each other -- finish sooner on two streams?
Times the two operators back to back on one stream and concurrently on two, on a developed 128^3 plume state.
usage: overlap_probe.py [res]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluidnet_amd import FluidNetModel, tfluids  # noqa: E402
from fluidnet_amd.simulate import simulate_native  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
batch, mconf = bench.build_scene(res, res, None, dev)
model = FluidNetModel.default_3d(seed=1)
for _ in range(16):
    simulate_native(None, mconf, batch, model)
U, flags, rho = batch["UDiv"], batch["flags"], batch["density"]
lib, ctx = tfluids._context(U)
N = flags.numel()
tmpA = torch.empty(9 * N, device=dev)            # advectScalar: fwd, bwd, fwdPos[3], bwdPos[3], out
tmpB = torch.empty(9 * N, device=dev)            # advectVel: fwd[3], bwd[3], out[3]
tt = tfluids._tt


def view(buf, off, C):
    return buf[off * N:(off + C) * N].view(1, C, res, res, res)


def scalar():
    lib2, c2 = tfluids._context(U)               # binds the ctx to the CURRENT torch stream
    rc = lib2.tfl_advectScalar(c2, 0.1, tt(rho), tt(U), tt(flags), tt(view(tmpA, 0, 1)), tt(view(tmpA, 1, 1)), 1,
                               b"maccormackOurs", tt(view(tmpA, 2, 3)), tt(view(tmpA, 5, 3)), 1, 0, 0.6, tt(view(tmpA, 8, 1)))
    assert rc == 0


def vel():
    lib2, c2 = tfluids._context(U)
    rc = lib2.tfl_advectVel(c2, 0.1, tt(U), tt(flags), tt(view(tmpB, 0, 3)), tt(view(tmpB, 3, 3)), 1, b"maccormackOurs", 1,
                            0.6, tt(view(tmpB, 6, 3)))
    assert rc == 0


s2 = torch.cuda.Stream()
e0, e1, ej = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()


def seq():
    scalar(); vel()


def par():
    ej.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ej)
        scalar()
        done = torch.cuda.Event(); done.record()
    vel()
    torch.cuda.current_stream().wait_event(done)


for name, fn in (("one stream", seq), ("two streams", par), ("one stream", seq), ("two streams", par)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-12s advectScalar + advectVel: %.1f us" % (name, e0.elapsed_time(e1) / 50 * 1e3))
