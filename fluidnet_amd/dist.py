"""z-slab decomposition of simulate() across the GPUs of one node (BASELINE config 5; 128^3 strong scaling).

The reference is single-GPU (no communication layer); this is new work defined by
BASELINE.json. One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI). The grid is cut along z --
the slowest spatial dimension, so a slab and every halo plane are contiguous in HBM -- and rank r owns planes
[z0, z1), storing `halo` = max(4, 2R+1) more planes next to each neighbour (R = back-trace reach in cells, 1 unless
the flow moves more than a cell per step).

The step itself is native: tfl_simulate_step_slab (fluidnet_amd/csrc/simulate.cpp) runs every phase of simulate() under
the narrowest z-window that keeps the owned planes exact (a few redundant planes per phase instead of a fixed wide
halo), packs the halo messages, and calls back into THIS module only to move them:
    U (max(R+1, 2R) planes) and p (4 / 3 planes)  leave at the end of a step, are consumed by the next one
    advected U (3 / 4) + density (max(4, 2R+1))  after MacCormack pass B, overlapped with the interior of pass B
    divergence (4 / 3)                        overlapped with the interior of the first conv layer
plus one 2-double all-reduce for the ConvNet's global std(U) normaliser (lib/model.lua:93-117). xGMI is point-to-point:
a slab only ever talks to ranks r-1 and r+1, each over its own link, so every exchange is one grouped send/recv pair
per neighbour -- no ring, no all-to-all.
"""
import ctypes
import threading

import torch

from . import _lib, tfluids
from ._lib import COMM_ALLREDUCE, COMM_START, COMM_WAIT, TfluidsError, tfl_comm, tfl_slab

TFL_EREACH = -5     # include/tfluids_hip.h: check_reach = 2 refused the step, nothing written


def slab_halo(reach=1):
    """Planes a slab stores next to each neighbour (tfl_slab_halo): max(4, 2*reach + 1)."""
    r = max(int(reach), 1)
    return max(4, 2 * r + 1)


class SlabLayout:
    """Owned planes [z0, z1) of a z_total grid and the extended local range [lo, hi)."""

    def __init__(self, z_total, world, rank, reach=1):
        if z_total % world != 0:
            raise ValueError("z extent %d is not divisible by %d ranks" % (z_total, world))
        per = z_total // world
        halo = slab_halo(reach) if world > 1 else 0
        if world > 1 and per < halo:
            raise ValueError("slab thickness %d is smaller than the halo %d" % (per, halo))
        self.z_total, self.world, self.rank, self.reach, self.halo = z_total, world, rank, max(int(reach), 1), halo
        self.z0, self.z1 = rank * per, (rank + 1) * per
        self.lo, self.hi = max(self.z0 - halo, 0), min(self.z1 + halo, z_total)
        self.c0, self.c1 = self.z0 - self.lo, self.z1 - self.lo   # owned planes in local indices
        self.has_lower, self.has_upper = rank > 0, rank < world - 1

    def extract(self, t):
        """Local extended COPY of a global [B, C, Z, Y, X] tensor (clone: a single-channel z-range is
        already contiguous, and .contiguous() would hand back a view aliasing the global tensor)."""
        return t[:, :, self.lo:self.hi].clone(memory_format=torch.contiguous_format)

    def owned(self, t):
        return t[:, :, self.c0:self.c1]


class _CommBase:
    """Turns (pointer, count) pairs of the C callbacks into views of the step's workspace tensor and keeps the
    ctypes trampolines alive. Subclasses implement start / wait / allreduce on torch tensors."""

    def __init__(self):
        self.ws = None
        self.error = None
        self._cb = (COMM_START(self._c_start), COMM_WAIT(self._c_wait), COMM_ALLREDUCE(self._c_allreduce))
        self.struct = tfl_comm(ctypes.sizeof(tfl_comm), None, *self._cb)

    def bind(self, ws):
        self.ws = ws

    def _view(self, ptr, n, dtype=torch.float32):
        if not ptr or n <= 0:
            return None
        off = (int(ptr) - self.ws.data_ptr()) // 4
        if dtype == torch.float64:
            return self.ws[off:off + 2 * n].view(torch.float64)
        return self.ws[off:off + n]

    def _guard(self, fn, *a):
        try:
            fn(*a)
            return 0
        except Exception as e:   # never let an exception cross the C frames: report through the return code
            self.error = e
            return 1

    def _c_start(self, _user, tag, slo, nslo, rlo, nrlo, shi, nshi, rhi, nrhi):
        return self._guard(self.start, int(tag), self._view(slo, nslo), self._view(rlo, nrlo), self._view(shi, nshi),
                           self._view(rhi, nrhi))

    def _c_wait(self, _user, tag):
        return self._guard(self.wait, int(tag))

    def _c_allreduce(self, _user, ptr, n):
        return self._guard(self.allreduce, self._view(ptr, int(n), torch.float64))


class DistComm(_CommBase):
    """Halo exchange + all-reduce over torch.distributed: nccl (= RCCL over xGMI) on GPU tensors; with the gloo
    backend messages are staged through the host (CPU tensors directly) -- that path only validates the multi-rank
    control flow on a box with fewer GPUs than ranks (TFL_DIST_BACKEND=gloo), it is not a measured configuration.

    A neighbour that never arrives: with gloo, `wait` gives up after `timeout_s` and raises TimeoutError naming the
    exchange (the native step turns that into TFL_ECOMM, see _CommBase). With nccl the wait only orders streams and
    cannot time out on the host; there the process group's own timeout applies (init_process_group(timeout=...); the
    NCCL/RCCL watchdog aborts the job when a send/recv has not completed by then) -- bench.py sets it explicitly."""

    def __init__(self, rank, world, group=None, timeout_s=300.0):
        super().__init__()
        import torch.distributed as dist
        self.dist, self.group, self.rank, self.world = dist, group, rank, world
        self.stage_host = dist.get_backend(group) == "gloo"
        self.timeout_s = float(timeout_s)
        self._pending = {}

    def start(self, tag, send_lo, recv_lo, send_hi, recv_hi):
        dist = self.dist
        ops, back = [], []

        def add(kind, t, peer):
            if t is None:
                return
            if self.stage_host and t.is_cuda:
                h = t.cpu() if kind is dist.isend else torch.empty(t.shape, dtype=t.dtype)
                if kind is dist.irecv:
                    back.append((t, h))
                t = h
            ops.append(dist.P2POp(kind, t, peer, self.group))

        add(dist.isend, send_lo, self.rank - 1)
        add(dist.irecv, recv_lo, self.rank - 1)
        add(dist.isend, send_hi, self.rank + 1)
        add(dist.irecv, recv_hi, self.rank + 1)
        # nccl: the grouped send/recv runs on the process group's own stream behind the packing kernels already queued
        # on the current stream; nothing blocks the host, and compute enqueued next overlaps the transfer
        self._pending[tag] = (dist.batch_isend_irecv(ops) if ops else [], back)

    def wait(self, tag):
        import datetime
        works, back = self._pending.pop(tag)
        for w in works:
            if not self.stage_host:
                w.wait()      # nccl: only the current STREAM waits (an explicit timeout would block the host thread instead)
                continue
            try:
                ok = w.wait(datetime.timedelta(seconds=self.timeout_s))   # gloo: the host waits
            except RuntimeError as e:
                raise TimeoutError("halo exchange %d of rank %d did not complete within %.0f s (%s)"
                                   % (tag, self.rank, self.timeout_s, str(e).splitlines()[0])) from e
            if ok is False:
                raise TimeoutError("halo exchange %d of rank %d did not complete within %.0f s" % (tag, self.rank, self.timeout_s))
        for dev_t, h in back:
            dev_t.copy_(h)

    def allreduce(self, stats):
        if self.stage_host and stats.is_cuda:
            h = stats.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            stats.copy_(h)
            return
        self.dist.all_reduce(stats, op=self.dist.ReduceOp.SUM, group=self.group)


class RcclComm:
    """The library's own transport (csrc/comm_rccl.cpp): ncclSend / ncclRecv / ncclAllReduce issued natively on a
    communication stream of the communicator -- no Python in the step's data path, the same object a LuaJIT or C host
    uses (include/tfluids_hip.h, "native transport"). `unique_id` = the 128 bytes rank 0 got from `unique_id()`,
    handed to the other ranks by the caller (torch.distributed broadcast in bench.py; any channel works).
    `ctx` = the tfl_ctx the slab steps run on (SlabSimulation passes its own)."""

    ID_BYTES = 128

    def __init__(self, ctx, unique_id, rank, world):
        self.lib = _lib.load()
        self.ctx, self.rank, self.world = ctx, rank, world
        buf = (ctypes.c_char * self.ID_BYTES).from_buffer_copy(bytes(unique_id))
        self.handle = self.lib.tfl_rccl_comm_create(ctx, buf, rank, world)
        if not self.handle:
            raise TfluidsError(self.lib.tfl_last_error(ctx).decode() or "tfl_rccl_comm_create failed")
        self.struct = self.lib.tfl_rccl_comm_callbacks(self.handle).contents
        self.error = None

    @staticmethod
    def unique_id(ctx):
        lib = _lib.load()
        buf = (ctypes.c_char * RcclComm.ID_BYTES)()
        if lib.tfl_rccl_get_unique_id(ctx, buf) != 0:
            raise TfluidsError(lib.tfl_last_error(ctx).decode() or "tfl_rccl_get_unique_id failed")
        return bytes(buf)

    def bind(self, ws):      # the native callbacks take device pointers as they are
        pass

    def set_inline(self, on):
        """every nccl* call on the step's own stream (slabs without overlap: tfl_rccl_comm_set_inline)"""
        self.lib.tfl_rccl_comm_set_inline(self.handle, int(bool(on)))

    def close(self):
        if self.handle:
            self.lib.tfl_rccl_comm_destroy(self.ctx, self.handle)
            self.handle = None


class ThreadComm(_CommBase):
    """Transport between VIRTUAL ranks living in threads of one process on one GPU (tests, single-GPU verification
    of the decomposition): a shared mailbox and barriers. All ranks enqueue on the same HIP stream, so device-side
    ordering follows the host-side order the barriers enforce."""

    class Hub:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.box = {}

    def __init__(self, hub, rank):
        super().__init__()
        self.hub, self.rank = hub, rank

    def start(self, tag, send_lo, recv_lo, send_hi, recv_hi):
        hub = self.hub
        hub.box[(tag, self.rank)] = (send_lo, send_hi)
        hub.barrier.wait()                      # every rank has queued its packing kernels
        if recv_lo is not None:
            recv_lo.copy_(hub.box[(tag, self.rank - 1)][1])
        if recv_hi is not None:
            recv_hi.copy_(hub.box[(tag, self.rank + 1)][0])
        hub.barrier.wait()                      # every copy is queued before a sender may repack its buffer

    def wait(self, tag):
        pass

    def allreduce(self, stats):
        hub = self.hub
        hub.box[("sum", self.rank)] = stats
        hub.barrier.wait()
        total = sum(hub.box[("sum", r)] for r in range(hub.world))
        hub.barrier.wait()                      # everyone has read the inputs before anyone overwrites its own
        stats.copy_(total)
        hub.barrier.wait()


class SlabSimulation:
    """tfluids.simulate() (lib/simulate.lua:175-327, ConvNet projection) on one z-slab through ONE native call per
    step (tfl_simulate_step_slab). `batch` holds the EXTENDED local tensors (SlabLayout.extract of the global pDiv,
    UDiv, flags, density and BC tensors). With world == 1 the result is exactly simulate_native()'s."""

    def __init__(self, batch, mconf, model, layout, comm=None, check_reach=True, overlap=None, own_context=False, graph=None):
        """graph: True = replay the rank-step as ONE HIP-graph launch (tfl_slab_graph_create, recorded after `graph_after`
        eager steps; raises if it cannot be recorded), False = always step eagerly, None (default) = try when the transport's
        calls are stream operations (tfl_comm.capturable: the native RCCL transport, or a slab without neighbours) and fall
        back to the eager step, with the reason in `graph_error`, when recording fails. mconf is frozen by the recording."""
        self.batch, self.mconf, self.model, self.lay, self.comm = batch, mconf, model, layout, comm
        U = batch["UDiv"]
        if (mconf.get("simMethod") or "convnet") != "convnet":
            raise TfluidsError("the z-slab path implements the ConvNet projection")
        if layout.world > 1 and comm is None:
            raise TfluidsError("a slab with neighbours needs a transport (RcclComm / DistComm / ThreadComm)")
        self.lib = _lib.load()
        self._own_ctx = None
        if own_context:      # virtual ranks in threads: the context carries per-step state (window, stream, reach word)
            dev = U.device.index if U.device.index is not None else torch.cuda.current_device()
            self._own_ctx = self.lib.tfl_create(dev)
            if not self._own_ctx:
                raise TfluidsError("tfl_create(%d) failed" % dev)
        self._check_reach, self._overlap = check_reach, overlap
        if graph is None:      # opt-in (TFL_SLAB_GRAPH=1): round 6 measured the replayed step level with the eager one on the device
            import os          # (profiles/r06_slab_host_cost.txt); what it buys is the host's time (0.015 against 0.07 ms per step)
            graph = None if (os.environ.get("TFL_SLAB_GRAPH", "0") == "1" and (comm is None or isinstance(comm, RcclComm))) else False
        self.graph_mode, self.graph_after, self.graph, self.graph_error, self.graph_nodes, self._steps = graph, 2, None, None, 0, 0
        self.relayouts = []        # reaches the simulation had to widen itself to (check_reach = "exact")
        self._setup()

    def _setup(self):
        """native arguments, slab description and workspace for the CURRENT batch tensors and layout"""
        from .simulate import _native_args
        batch, layout, comm = self.batch, self.lay, self.comm
        U = batch["UDiv"]
        lib, ctx = self._context()
        self.prm, self.st, self._keep = _native_args(lib, ctx, self.mconf, batch, self.model)
        cells = (layout.c1 - layout.c0) * U.size(3) * U.size(4)
        cr = 2 if self._check_reach == "exact" else int(bool(self._check_reach))
        self.slab = tfl_slab(layout.z_total, layout.lo, layout.c0, layout.c1, layout.reach,
                             int(cells >= (1 << 22)) if self._overlap is None else int(bool(self._overlap)), cr, 0)
        # (overlap: boundary strips first so that a message travels beside the interior's kernels. It doubles the launches of
        # three phases and needs the communication stream -- eight event hops per step at 12-15 us of device-side latency each
        # (tools/ubench/host_costs.hip) -- so it pays only where a message is long against ~100 us: from 4 M cells per rank on.
        # Round 6, profiles/r06_slab_host_cost.txt: 128^3 on 2 ranks 0.254 ms with it, kernels 0.177; 256^3 on 8 0.379 / 0.292.)
        n = int(lib.tfl_simulate_slab_workspace_floats(ctx, ctypes.byref(self.prm), ctypes.byref(self.st),
                                                       ctypes.byref(self.slab)))
        if n <= 0:
            raise TfluidsError(lib.tfl_last_error(ctx).decode() or "bad slab description")
        self.ws = torch.zeros(n, dtype=torch.float32, device=U.device)    # persistent: messages live here across steps
        if comm is not None and hasattr(comm, "struct"):
            comm.bind(self.ws)
        if isinstance(comm, RcclComm):
            comm.set_inline(not self.slab.overlap)     # a thin slab has nothing for a transfer to overlap with: no stream hops

    def _relayout(self, reach):
        """check_reach = "exact": the step was refused on every rank (TFL_EREACH, nothing written) because the flow needs a
        back-trace reach the halos do not cover. Every 5-D tensor of the batch (state, flags, BC pairs) gets the wider halo:
        a new array, the owned planes copied, the halo planes fetched from the neighbours' owned planes through the transport
        (tfl_slab_exchange); then the native arguments are rebuilt and the step is taken again. The batch DICT keeps its
        identity, its tensors are replaced. Collective: every rank arrives here at the same step with the same reach."""
        old = self.lay
        try:
            new = SlabLayout(old.z_total, old.world, old.rank, reach)
        except ValueError as e:
            raise TfluidsError("the flow needs a back-trace reach of %d planes: %s" % (reach, e))
        if self.graph is not None:
            self.lib.tfl_slab_graph_destroy(self._context()[1], self.graph)
            self.graph = None
        lib, ctx = self._context()
        names, tensors = [], []

        def walk(key, v):
            if torch.is_tensor(v) and v.dim() == 5 and v.size(2) == old.hi - old.lo:
                names.append(key); tensors.append(v)
            elif isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    walk((key, i), x)
        for k, v in list(self.batch.items()):
            walk(k, v)
        grown = []
        for t in tensors:
            nt = torch.zeros(t.size(0), t.size(1), new.hi - new.lo, t.size(3), t.size(4), dtype=t.dtype, device=t.device)
            nt[:, :, new.c0:new.c1] = t[:, :, old.c0:old.c1]
            grown.append(nt)
        slab = tfl_slab(new.z_total, new.lo, new.c0, new.c1, new.reach, 0, 0, 0)
        below = above = new.halo      # the same counts on every rank: what I send up is what my upper neighbour stores below
        for i in range(0, len(grown), 4):
            grp = grown[i:i + 4]
            descs = [tfluids._desc5(t) for t in grp]
            arr = (ctypes.POINTER(_lib.tfl_tensor) * len(grp))(*[ctypes.pointer(d) for d in descs])
            lo = (ctypes.c_int32 * len(grp))(*([below] * len(grp)))
            hi = (ctypes.c_int32 * len(grp))(*([above] * len(grp)))
            n = int(lib.tfl_slab_exchange_floats(len(grp), arr, lo, hi, ctypes.byref(slab)))
            scratch = torch.empty(max(n, 4), dtype=torch.float32, device=grp[0].device)
            if self.comm is not None:
                self.comm.bind(scratch)
                self.comm.error = None
            cptr = ctypes.byref(self.comm.struct) if self.comm is not None else None
            rc = lib.tfl_slab_exchange(ctx, len(grp), arr, lo, hi, ctypes.byref(slab), cptr, ctypes.c_void_p(scratch.data_ptr()), scratch.numel())
            if rc != 0:
                if self.comm is not None and self.comm.error is not None:
                    raise self.comm.error
                raise TfluidsError(lib.tfl_last_error(ctx).decode())
            torch.cuda.current_stream(grp[0].device).synchronize()      # scratch is released below
        for key, nt in zip(names, grown):
            if isinstance(key, tuple):
                seq = list(self.batch[key[0]]); seq[key[1]] = nt; self.batch[key[0]] = seq
            else:
                self.batch[key] = nt
        self.lay = new
        self.relayouts.append(reach)
        self._setup()

    def _context(self):
        if self._own_ctx is None:
            return tfluids._context(self.batch["UDiv"])
        dev = self.batch["UDiv"].device.index
        self.lib.tfl_set_stream(self._own_ctx, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        return self.lib, self._own_ctx

    def _call(self, fn, *args, ok_codes=()):
        lib, ctx = self._context()
        if self.comm is not None and not hasattr(self.comm, "struct"):
            # a factory: the transport needs this simulation's context and -- communicator creation being collective --
            # the rank's own thread / process (RcclComm)
            self.comm = self.comm(ctx)
            self.comm.bind(self.ws)
            if isinstance(self.comm, RcclComm):
                self.comm.set_inline(not self.slab.overlap)
        cptr = ctypes.byref(self.comm.struct) if self.comm is not None else None
        if self.comm is not None:
            self.comm.error = None
        rc = fn(ctx, *args, ctypes.byref(self.slab), cptr, ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel())
        if rc in ok_codes:
            return rc
        if rc != 0:
            if self.comm is not None and self.comm.error is not None:
                raise self.comm.error
            raise TfluidsError(lib.tfl_last_error(ctx).decode())

    def _record(self):
        """the rank-step as a HIP graph (include/tfluids_hip.h tfl_slab_graph_create); every rank reaches this at the same step"""
        lib, ctx = self._context()
        cptr = ctypes.byref(self.comm.struct) if self.comm is not None else None
        h = lib.tfl_slab_graph_create(ctx, ctypes.byref(self.prm), ctypes.byref(self.st), ctypes.byref(self.slab), cptr,
                                      ctypes.c_void_p(self.ws.data_ptr()), self.ws.numel())
        if h:
            self.graph, self.graph_nodes = ctypes.c_void_p(h), int(lib.tfl_slab_graph_nodes(ctypes.c_void_p(h)))
        else:
            self.graph_error = lib.tfl_last_error(ctx).decode()
            if self.graph_mode is True:
                raise TfluidsError(self.graph_error)
        self.graph_mode = False if not h else self.graph_mode

    def _capturable(self):
        if self.comm is None or not (self.lay.has_lower or self.lay.has_upper):
            return True
        return hasattr(self.comm, "struct") and bool(getattr(self.comm.struct, "capturable", 0))

    def step(self, eager=False):
        """One rank-step. eager=True: through tfl_simulate_step_slab even when a recorded step exists (the per-kernel profiler
        sees nothing inside a graph launch); every rank must make the same choice, and `drain()` comes before the next recorded
        step (a recorded step starts and ends with no message in flight, an eager one leaves the p / U halos travelling)."""
        if not eager and self.graph is None and self.graph_mode is not False and self._steps >= self.graph_after:
            if self.graph_mode is True or self._capturable():
                self._record()
            else:
                self.graph_mode, self.graph_error = False, "the transport runs host code per message (tfl_comm.capturable = 0)"
        self._steps += 1
        if self.graph is not None and not eager:
            self.drain()
            lib, ctx = self._context()
            rc = lib.tfl_slab_graph_step(ctx, self.graph)
            if rc != 0:
                raise TfluidsError(lib.tfl_last_error(ctx).decode())
            return
        for _ in range(8):
            rc = self._call(self.lib.tfl_simulate_step_slab, ctypes.byref(self.prm), ctypes.byref(self.st), ok_codes=(TFL_EREACH,))
            if rc != TFL_EREACH:
                return
            # nothing of the step has been written and every rank is here with the same number: widen the halos, take the step again
            self._relayout(int(self.lib.tfl_slab_needed_reach(self._context()[1])))
        raise TfluidsError("the reach kept growing while the halos were being widened")

    def drain(self):
        """Finish the p / U halo messages the last step left in flight (before reading halo planes)."""
        if self.slab.in_flight & 0xF:      # (bits 0-3: messages; bit 8 is the library's own)
            self._call(self.lib.tfl_slab_drain, ctypes.byref(self.st))

    def close(self):
        if self.graph is not None:
            lib, ctx = self._context()
            lib.tfl_slab_graph_destroy(ctx, self.graph)
            self.graph = None
        if isinstance(self.comm, RcclComm):
            self.comm.close()
        if self._own_ctx is not None:
            self.lib.tfl_destroy(self._own_ctx)
            self._own_ctx = None


def run_virtual_ranks(sims, steps):
    """Advance several SlabSimulation objects (ThreadComm transport, own_context=True) `steps` steps, one thread each."""
    errs = []
    dev = sims[0].batch["UDiv"].device

    def work(sim):
        try:
            torch.cuda.set_device(dev)
            for _ in range(steps):
                sim.step()
            sim.drain()
        except Exception as e:   # noqa: BLE001
            errs.append(e)
            try:
                sim.comm.hub.barrier.abort()
            except Exception:
                pass

    ts = [threading.Thread(target=work, args=(s,)) for s in sims]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize(dev)
    if errs:
        raise errs[0]
