"""Cost of one z-slab rank-step on ONE GPU: rank r of a W-rank layout whose neighbours are not there (results are wrong, the
enqueue path and the kernels are the real ones). Round 6: the transport is the library's own (csrc/comm_rccl.cpp) over
tests/stub_rccl.cpp in STUB_RCCL_NULL mode -- a send is dropped, a receive is a zero-fill on the stream, the all-reduce adds
the other ranks' share -- so the host path is the one a real run takes (event record / wait, group calls) minus RCCL's own
enqueue cost, and the step can be recorded into a HIP graph (tfl_slab_graph_create): both the eager and the replayed step are
timed. --python-null = the round-5 transport (Python callbacks, in-place chunks; --packed = staged buffers): its callbacks
cost the host ~50 us per step that no native host pays.
usage: slab_host_cost.py [res] [world] [--still] [--kernels] [--python-null [--packed]] [--no-graph] [--check-reach]"""
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
python_null = "--python-null" in sys.argv
if not python_null:
    import subprocess
    import tempfile
    so = os.path.join(tempfile.mkdtemp(prefix="stub_rccl_"), "libstub_rccl.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-o", so, os.path.join(ROOT, "tests", "stub_rccl.cpp")])
    os.environ["TFL_RCCL_LIBRARY"], os.environ["STUB_RCCL_NULL"] = so, "1"
    from fluidnet_amd import tfluids  # noqa: E402
    from fluidnet_amd.dist import RcclComm  # noqa: E402


def run(rank, graph):
    lay = SlabLayout(res, world, rank)
    batch, mconf = bench.build_scene(res, res, lay, dev)
    mconf = dict(mconf, buoyancyScale=0.0)          # keep the (wrong) state tame: nothing rises into the missing halos
    still = "--still" in sys.argv
    if still:
        # dt = 0: nothing moves, so the halos the null transport never refreshes stay what the neighbours would have sent
        # (except p, which only feeds the conv stack) and every kernel runs on the data it would see in a real run.
        # Without it the in-place exchange leaves stale velocities in the halos and the advection kernels take their
        # slow paths there, which is an artefact of the null transport, not of the step.
        mconf = dict(mconf, dt=0.0)
    if python_null:
        comm = NullComm("--packed" not in sys.argv)
    else:
        lib, ctx = tfluids._context(batch["UDiv"])
        comm = RcclComm(ctx, RcclComm.unique_id(ctx), rank, world)
    sim = SlabSimulation(batch, mconf, model, lay, comm, check_reach=("--check-reach" in sys.argv), graph=graph)
    # --still also zeroes p before every step: with the null transport the net would otherwise iterate on its own output
    # (stale p halos), leave the fp16 range within a few steps, and every block of the first conv layer would report a
    # range error through one atomic counter -- 20 us of serialised atomics that no real run has -- and puts U and the
    # density back to the initial state (the confinement force is added every step whatever dt is, and with a projection that
    # iterates on undelivered halos nothing removes it again). The three small copies run as ONE captured torch graph so that
    # their host cost (3 x ~6 us of dispatch) stays out of the enqueue figure; their GPU time (~6 us) is inside the drained one.
    saved = {k: batch[k].clone() for k in ("UDiv", "density")}
    restore = None
    if still:
        restore = torch.cuda.CUDAGraph()
        with torch.cuda.graph(restore):
            batch["pDiv"].zero_()
            for k, v in saved.items():
                batch[k].copy_(v)

    host = [0.0]

    def one_step():
        if restore is not None:
            restore.replay()
        t = time.perf_counter()
        sim.step()
        host[0] += time.perf_counter() - t      # the library's call alone (the restore graph's replay is the tool's, not the step's)

    for _ in range(6):
        one_step()
    torch.cuda.synchronize()
    # what the tool's own restore graph costs the GPU per step (three small copies), to be taken off the drained figure
    t_restore = 0.0
    if restore is not None:
        reps = []
        for _ in range(5):      # (the fastest of five: one hiccup of the box must not go into the figure with the wrong sign)
            t0 = time.perf_counter()
            for _ in range(40):
                restore.replay()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 40)
        t_restore = min(reps)
    # the fastest of five blocks of 60 steps: one pause of the box (a scheduler hiccup of 50-100 ms on these shared hosts showed
    # up as a rank-step of 0.3-0.5 ms in a single 200-step block) must not go into either figure
    n, hosts, walls = 60, [], []
    for _ in range(5):
        host[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            one_step()
        hosts.append(host[0] / n)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / n)
    t_host = min(hosts)
    t_all = min(walls) - t_restore
    how = ("HIP graph, %d nodes" % sim.graph_nodes) if sim.graph is not None else ("eager" + (" (graph refused: %s)" % sim.graph_error if sim.graph_error else ""))
    print("res %d, rank %d of %d (%d planes), %s: host %.3f ms per step() call, rank-step %.3f ms (wall per step with the GPU drained, minus %.3f ms of the tool's restore copies)"
          % (res, rank, world, lay.hi - lay.lo, how, t_host * 1e3, t_all * 1e3, t_restore * 1e3))
    if rank == world // 2 and "--kernels" in sys.argv and sim.graph is None:
        from fluidnet_amd import tfluids as T
        with T.profile(batch["UDiv"]) as prof:
            for _ in range(10):
                one_step()
        print("   fp16 range errors reported by the conv stack: %d" % model.range_errors(batch["pDiv"]))
        tot, cnt = 0.0, 0
        for name, rec in sorted(prof.kernels.items(), key=lambda kv: -kv[1]["ms"]):
            print("   %-26s %5.1f launches/step  %7.1f us/step  (%.1f us each)" % (name, rec["calls"] / 10, rec["ms"] * 100, rec["ms"] / rec["calls"] * 1e3))
            tot += rec["ms"] * 100; cnt += rec["calls"] / 10
        print("   sum %.1f us/step over %.0f launches (the scalar and the velocity advection run on two streams: their times overlap)" % (tot, cnt))
    sim.close()


for rank in (0, world // 2):
    for graph in ((False,) if (python_null or "--no-graph" in sys.argv) else (False, True)):
        run(rank, graph)
