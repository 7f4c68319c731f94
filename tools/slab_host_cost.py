"""Host cost of one z-slab rank-step: rank 0 of a W-rank layout with a transport that does NOTHING (results are wrong,
the enqueue path is the real one). With a 16-plane slab the GPU work is ~0.07 ms, so wall time per step ~ host time of
the native step + its four transport callbacks (without the transport's own cost). usage: slab_host_cost.py [res] [world]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fluidnet_amd import FluidNetModel  # noqa: E402
from fluidnet_amd.dist import SlabLayout, SlabSimulation, _CommBase  # noqa: E402


class NullComm(_CommBase):
    """--packed: offers only the staged exchange (the library packs / unpacks, as for the Python transports); default: also
    the in-place one (exchange_start_v, what the native RCCL transport offers), so that no pack / unpack kernels run."""

    def __init__(self, in_place=True):
        super().__init__()
        if in_place:
            import ctypes
            from fluidnet_amd._lib import COMM_START_V
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]

            def start_v(user, tag, n_lo, send_lo, recv_lo, n_hi, send_hi, recv_hi):
                # the halo planes a neighbour would have delivered are ZEROED (round 5): the divergence halos
                # live in the step's shared workspace and otherwise hold whatever the advection left there (+-inf clamp bounds),
                # which the first conv layer reports as fp16 range errors through one atomic counter -- and since round 5 the
                # next step is refused (TFL_ERANGE)
                if tag not in (2, 3):   # tag 0 (U, p): the halos keep what a still scene's neighbours would have sent; tags 2 / 3
                    return 0            # (advected U + density, divergence) land in workspace memory: garbage unless delivered
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for n, chunks in ((n_lo, recv_lo), (n_hi, recv_hi)):
                    for q in range(n):
                        hip.hipMemsetAsync(ctypes.c_void_p(chunks[q].ptr), 0, int(chunks[q].n) * 4, st)
                return 0
            self._cbv = COMM_START_V(start_v)
            self.struct.exchange_start_v = self._cbv

    def start(self, tag, send_lo, recv_lo, send_hi, recv_hi):
        pass

    def wait(self, tag):
        pass

    def allreduce(self, stats):
        # what the other ranks would have contributed: without it a rank whose slab holds no moving cell (the plume's disc spans
        # a third of z) has sum u^2 = 0, a scale of 0, an infinite net input -- every block of the conv stack reports a range
        # error through one atomic counter (the "12 M range errors" of profiles/r04_slab_host_cost.txt), and since round 5 the
        # next step is refused. One small in-place add on the step's stream, like the real all-reduce's kernel.
        stats.view(-1, 2)[:, 1] += 1.0e5


_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
res = int(_pos[0]) if len(_pos) > 0 else 128
world = int(_pos[1]) if len(_pos) > 1 else 8
dev = torch.device("cuda:0")
model = FluidNetModel.default_3d(seed=1)
for rank in (0, world // 2):
    lay = SlabLayout(res, world, rank)
    batch, mconf = bench.build_scene(res, res, lay, dev)
    mconf = dict(mconf, buoyancyScale=0.0)          # keep the (wrong) state tame: nothing rises into the missing halos
    if "--still" in sys.argv:
        # dt = 0: nothing moves, so the halos the null transport never refreshes stay what the neighbours would have sent
        # (except p, which only feeds the conv stack) and every kernel runs on the data it would see in a real run.
        # Without it the in-place exchange leaves stale velocities in the halos and the advection kernels take their
        # slow paths there, which is an artefact of the null transport, not of the step.
        mconf = dict(mconf, dt=0.0)
    sim = SlabSimulation(batch, mconf, model, lay, NullComm("--packed" not in sys.argv), check_reach=False)
    # --still also zeroes p before every step: with the null transport the net would otherwise iterate on its own output
    # (stale p halos), leave the fp16 range within a few steps, and every block of the first conv layer would report a
    # range error through one atomic counter -- 20 us of serialised atomics that no real run has
    still = "--still" in sys.argv
    # ... and (round 5) puts U and the density back to the initial state: the confinement force is added every step whatever dt
    # is, and with a projection that iterates on undelivered halos nothing removes it again -- the run blew up within the
    # warm-up (the 12 M range errors of profiles/r04_slab_host_cost.txt; since round 5 a refused step, TFL_ERANGE). The three
    # small copies are torch's, outside the library's kernel profile; they are inside the wall-clock figure.
    saved = {k: batch[k].clone() for k in ("UDiv", "density")}

    def one_step():
        if still:
            batch["pDiv"].zero_()
            for k, v in saved.items():
                batch[k].copy_(v)
        sim.step()

    for _ in range(5):
        one_step()
    torch.cuda.synchronize()
    for n in (50,):
        t0 = time.time()
        for _ in range(n):
            one_step()
        t_host = (time.time() - t0) / n
        torch.cuda.synchronize()
        t_all = (time.time() - t0) / n
    print("res %d, rank %d of %d (%d planes): enqueue %.3f ms per step, with GPU drain %.3f ms per step"
          % (res, rank, world, lay.hi - lay.lo, t_host * 1e3, t_all * 1e3))
    if rank == world // 2 and "--kernels" in sys.argv:
        from fluidnet_amd import tfluids
        with tfluids.profile(batch["UDiv"]) as prof:
            for _ in range(10):
                one_step()
        print("   fp16 range errors reported by the conv stack: %d" % model.range_errors(batch["pDiv"]))
        tot, cnt = 0.0, 0
        for name, rec in sorted(prof.kernels.items(), key=lambda kv: -kv[1]["ms"]):
            print("   %-26s %5.1f launches/step  %7.1f us/step  (%.1f us each)" % (name, rec["calls"] / 10, rec["ms"] * 100, rec["ms"] / rec["calls"] * 1e3))
            tot += rec["ms"] * 100; cnt += rec["calls"] / 10
        print("   sum %.1f us/step over %.0f launches" % (tot, cnt))
    sim.close()
