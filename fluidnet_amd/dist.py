"""z-slab decomposition of simulate() across the GPUs of one node (BASELINE config 5).

The reference is single-GPU (no communication layer); this is new work defined by
BASELINE.json. One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI). The grid is
cut along z -- the slowest spatial dimension, so a slab and every halo plane are contiguous in HBM --
and rank r owns planes [z0, z1). It stores [z0 - h, z1 + h) (clipped to the domain), runs the
UNMODIFIED single-GPU kernels on that extended array, and refreshes the h halo planes from its two
z-neighbours at two points of the step. xGMI is point-to-point: a slab only ever talks to ranks r-1
and r+1, each over its own link, so the exchange is two grouped send/recv pairs -- no ring, no
all-to-all -- plus ONE 2-double all-reduce per step for the ConvNet's global std(U) normaliser
(lib/model.lua:93-117).

Why the unmodified kernels give the exact single-GPU answer on the owned planes: every operator reads a
bounded z-neighbourhood, so errors that enter at the artificial ends of the extended array (treated by
the kernels as the domain's border shell) travel inward by a bounded number of planes per phase:
  advection (MacCormack, two passes): 2*Rt + 3 planes, Rt = ceil(max|u_z|*dt) the back-trace reach
  buoyancy 1, vorticity confinement 3 (+1 border plane), ConvNet projection 5 (div 1 + three 3^3 convs
  3 + pressure gradient 1)
With h = 10: exchange {U, density} -> advect (<= 7 planes for Rt <= 2) -> exchange {U, density, p} ->
forces + projection (9 planes) keeps [z0, z1) exact. `check_reach=True` verifies Rt on the device.
Message size per neighbour and direction: planes * X * Y * 4 B * channels (128^2: 64 KiB per plane and
channel; 10 planes x 5 channels = 3.1 MiB), far above the latency-bound regime, sent as one buffer.
"""
import math

import torch

from . import tfluids
from .simulate import _f32, _gravity, _sparse_bc, setConstVals

DEFAULT_HALO = 10


class SlabLayout:
    """Owned planes [z0, z1) of a Z_total grid and the extended local range [lo, hi)."""

    def __init__(self, z_total, world, rank, halo=DEFAULT_HALO):
        if z_total % world != 0:
            raise ValueError("z extent %d is not divisible by %d ranks" % (z_total, world))
        per = z_total // world
        if world > 1 and per < halo:
            raise ValueError("slab thickness %d is smaller than the halo %d" % (per, halo))
        self.z_total, self.world, self.rank, self.halo = z_total, world, rank, halo
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


def _msg_numel(fields, a, b):
    return sum(f.size(0) * f.size(1) * (b - a) * f.size(3) * f.size(4) for f in fields)


def _hip_pack(fields, a, b, buf, unpack):
    """One kernel launch (tfl_packPlanes) instead of a dozen strided torch copies per message."""
    import ctypes
    from ._lib import tfl_tensor
    lib, ctx = tfluids._context(fields[0])
    descs = [tfl_tensor(f.data_ptr(), *f.shape) for f in fields]
    arr = (ctypes.POINTER(tfl_tensor) * len(descs))(*[ctypes.pointer(d) for d in descs])
    tfluids._call(lib, ctx, lib.tfl_packPlanes(ctx, len(descs), arr, int(a), int(b), ctypes.c_void_p(buf.data_ptr()),
                                               int(unpack)))


def _pack(fields, a, b, out=None):
    """Planes [a, b) of every field -> one contiguous message, layout [field][b][c][plane][Y][X]."""
    if fields[0].is_cuda:
        buf = out if out is not None else torch.empty(_msg_numel(fields, a, b), dtype=torch.float32,
                                                       device=fields[0].device)
        _hip_pack(fields, a, b, buf, 0)
        return buf
    return torch.cat([f[:, :, a:b].reshape(-1) for f in fields])     # CPU tensors (gloo tests)


def _unpack(buf, fields, a, b):
    if fields[0].is_cuda:
        _hip_pack(fields, a, b, buf, 1)
        return
    off = 0
    for f in fields:
        view = f[:, :, a:b]
        n = view.numel()
        view.copy_(buf[off:off + n].view(view.shape))
        off += n


class DistComm:
    """Halo exchange + all-reduce over torch.distributed (nccl = RCCL on GPUs, gloo on CPU tensors)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self._bufs = {}   # (direction, numel) -> (send, recv) message buffers, allocated once
        # gloo moves host memory: GPU messages are staged through the host. Only used to validate the multi-rank
        # control flow on a box without one GPU per rank (TFL_DIST_BACKEND=gloo); production = nccl (RCCL).
        self.stage_host = dist.get_backend(group) == "gloo"

    def _buffers(self, key, fields, a, b):
        n = _msg_numel(fields, a, b)
        hit = self._bufs.get((key, n))
        if hit is None:
            hit = (torch.empty(n, dtype=torch.float32, device=fields[0].device),
                   torch.empty(n, dtype=torch.float32, device=fields[0].device))
            self._bufs[(key, n)] = hit
        return hit

    def exchange(self, lay, fields):
        dist, h = self.dist, lay.halo
        ops, recvs = [], []
        if lay.has_lower:   # my lowest h owned planes -> rank-1's upper halo; its top planes -> my lower halo
            sbuf, recv = self._buffers("lo", fields, lay.c0, lay.c0 + h)
            send = _pack(fields, lay.c0, lay.c0 + h, out=sbuf if fields[0].is_cuda else None)
            ops += [dist.P2POp(dist.isend, send, lay.rank - 1, self.group),
                    dist.P2POp(dist.irecv, recv, lay.rank - 1, self.group)]
            recvs.append((recv, lay.c0 - h, lay.c0))
        if lay.has_upper:
            sbuf, recv = self._buffers("hi", fields, lay.c1 - h, lay.c1)
            send = _pack(fields, lay.c1 - h, lay.c1, out=sbuf if fields[0].is_cuda else None)
            ops += [dist.P2POp(dist.isend, send, lay.rank + 1, self.group),
                    dist.P2POp(dist.irecv, recv, lay.rank + 1, self.group)]
            recvs.append((recv, lay.c1, lay.c1 + h))
        if ops and self.stage_host and fields[0].is_cuda:
            host_ops, pairs = [], []
            for op in ops:
                h = op.tensor.cpu() if op.op == dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
                host_ops.append(dist.P2POp(op.op, h, op.peer, self.group))
                if op.op == dist.irecv:
                    pairs.append((op.tensor, h))
            for w in dist.batch_isend_irecv(host_ops):
                w.wait()
            for dev_t, h in pairs:
                dev_t.copy_(h)
        elif ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for buf, a, b in recvs:
            _unpack(buf, fields, a, b)

    def allreduce_sum(self, t):
        if self.stage_host and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


def run_lockstep(gens_layouts):
    """Drive several SlabSimulation.step_gen() generators of VIRTUAL ranks living in one process (tests,
    single-GPU verification of the decomposition): advance all to their next communication request,
    perform it among them with plain copies, repeat."""
    gens = [g for g, _ in gens_layouts]
    lays = [l for _, l in gens_layouts]
    reqs = [next(g, None) for g in gens]
    while any(r is not None for r in reqs):
        kinds = {r[0] for r in reqs}
        assert len(kinds) == 1, "virtual ranks diverged: %s" % kinds
        kind = kinds.pop()
        if kind == "halo":
            h = lays[0].halo
            sends = []
            for (_, fields), lay in zip(reqs, lays):
                lo_buf = _pack(fields, lay.c0, lay.c0 + h) if lay.has_lower else None
                hi_buf = _pack(fields, lay.c1 - h, lay.c1) if lay.has_upper else None
                sends.append((lo_buf, hi_buf))
            for r, ((_, fields), lay) in enumerate(zip(reqs, lays)):
                if lay.has_lower:
                    _unpack(sends[r - 1][1], fields, lay.c0 - h, lay.c0)
                if lay.has_upper:
                    _unpack(sends[r + 1][0], fields, lay.c1, lay.c1 + h)
        elif kind == "allreduce":
            total = sum(r[1] for r in reqs)
            for r in reqs:
                r[1].copy_(total)
        else:
            raise AssertionError(kind)
        reqs = [next(g, None) for g in gens]


class SlabSimulation:
    """tfluids.simulate() (lib/simulate.lua:175-327, ConvNet projection) on one z-slab.

    `batch` holds the EXTENDED local tensors (SlabLayout.extract of the global pDiv, UDiv, flags, density
    and BC tensors). With world == 1 this is exactly fluidnet_amd.simulate.simulate."""

    def __init__(self, batch, mconf, model, layout, comm=None, check_reach=False):
        self.batch, self.mconf, self.model, self.lay = batch, mconf, model, layout
        self.comm = comm
        self.check_reach = check_reach
        U = batch["UDiv"]
        _, C, _, Y, X = U.shape
        self.dx = 1.0 / max(X, Y, layout.z_total)
        self.count = float(C * layout.z_total * Y * X)            # global sample count of std(U)
        self.stats = torch.zeros(U.size(0), 2, dtype=torch.float64, device=U.device)
        if (mconf.get("simMethod") or "convnet") != "convnet":
            raise tfluids.TfluidsError("the z-slab path implements the ConvNet projection")

    def step_gen(self):
        """One simulate() step as a generator of communication requests ('halo', fields) / ('allreduce', stats).
        Several virtual ranks may interleave in one process (run_lockstep), and advectVel's deferred copy keeps
        a scratch tensor alive across a yield: each SlabSimulation therefore works in its own scratch scope."""
        gen = self._step_gen()
        while True:
            prev, tfluids._scratch_scope = tfluids._scratch_scope, ("slab", id(self))
            try:
                req = next(gen, None)
            finally:
                tfluids._scratch_scope = prev
            if req is None:
                return
            yield req

    def _step_gen(self):
        b, m, lay = self.batch, self.mconf, self.lay
        p, U, flags, rho = b["pDiv"], b["UDiv"], b["flags"], b["density"]
        dt, method, strength = m["dt"], m.get("advectionMethod"), m.get("maccormackStrength")
        multi = lay.world > 1
        if multi:
            tfluids.setDxOverride(U, self.dx)
            yield ("halo", [U, rho])
            if self.check_reach:
                reach = float(U[:, 2].abs().max()) * dt
                if 2 * math.ceil(reach) + 3 > lay.halo:
                    raise tfluids.TfluidsError("back-trace reach %.2f planes exceeds what halo %d covers"
                                               % (reach, lay.halo))
        tfluids.advectScalar(dt, rho, U, flags, method, None, False, strength)
        # as in simulate(): the advected field stays in advectVel's scratch (halo exchange and BCs happen there)
        # and addBuoyancy writes U = scratch + buoyancy, which replaces the U:copy sweep
        buoyant = m.get("buoyancyScale", 0) > 0
        Uadv = tfluids.advectVel(dt, U, flags, method, None, strength, _deferCopy=buoyant)
        Ucur = Uadv if buoyant else U
        if multi:
            yield ("halo", [Ucur, rho, p])
        setConstVals(b, p, Ucur, flags, rho)
        dx = self.dx if multi else tfluids.getDx(flags)
        if buoyant:
            s = _f32(-(dx / 4) * m["buoyancyScale"])
            tfluids.addBuoyancy(U, flags, rho, [_f32(v) * s for v in _gravity(m)], dt, USrc=Uadv)
        if m.get("gravityScale", 0) > 0:
            s = _f32((-dx / 4) * m["gravityScale"])
            tfluids.addGravity(U, flags, [_f32(v) * s for v in _gravity(m)], dt)
        if m.get("vorticityConfinementAmp", 0) > 0:
            tfluids.vorticityConfinement(U, flags, dx * m["vorticityConfinementAmp"])
        setConstVals(b, p, U, flags, rho, unchanged=("p", "density"))
        self.model.begin(U, flags, lay.c0, lay.c1, self.stats)
        if multi:
            yield ("allreduce", self.stats)
        ubc, umask = b.get("UBC"), b.get("UBCInvMask")
        sp = _sparse_bc(U, ubc, umask)
        late_ubc = sp is not None and sp[1]      # as in simulate(): sparse idempotent U BCs go after the projection
        self.model.finish(p, U, flags, self.stats, self.count, UBC=None if late_ubc else ubc,
                          UBCInvMask=None if late_ubc else umask, clamp=(-1e6, 1e6))
        rest = b if late_ubc else {k: v for k, v in b.items() if k not in ("UBC", "UBCInvMask")}
        setConstVals(rest, p, U, flags, rho, unchanged=("density",))
        if multi:
            tfluids.setDxOverride(U, None)

    def step(self):
        for req in self.step_gen():
            if req[0] == "halo":
                self.comm.exchange(self.lay, req[1])
            else:
                self.comm.allreduce_sum(req[1])
