"""Host mirror of torch/lib/simulate.lua: tfluids.simulate, setConstVals, createPlumeBCs.

`batch` is a dict with the reference's keys (pDiv, UDiv, flags, density [+ UBC, UBCInvMask,
densityBC, densityBCInvMask, pBC, pBCInvMask]) holding torch tensors on an MI355X; `mconf` is a dict
with the reference's model-config names (dt, advectionMethod, maccormackStrength, buoyancyScale,
gravityScale, vorticityConfinementAmp, simMethod, maxIter, gravity). State is updated in place in the
*Div slots exactly like the Lua (simulate.lua:180).
"""
import ctypes
import os
import math

import torch

from . import tfluids
from ._lib import TfluidsError


def getPUFlagsDensityReference(batch):
    return batch["pDiv"], batch["UDiv"], batch["flags"], batch.get("density")


def createPlumeBCs(batch, densityVal, uScale, rad, zOffset=0, zTotal=None):
    """simulate.lua:47-123 (1-based loops restated as index arithmetic; same cells, same values).
    zOffset/zTotal (extension for z-slab ranks, fluidnet_amd.dist): the tensors hold planes
    [zOffset, zOffset + Z) of a zTotal-deep grid; the plume geometry is that of the whole grid."""
    U = batch["UDiv"]
    batch["pBC"] = None
    batch["pBCInvMask"] = None
    batch["UBC"] = torch.zeros_like(U)
    batch["UBCInvMask"] = torch.ones_like(U)
    dens = batch.get("density")
    if dens is None:
        raise TfluidsError("plume BCs require a density field to be specified")
    multi = isinstance(dens, (list, tuple))
    chans = list(dens) if multi else [dens]
    if len(densityVal) != len(chans):
        raise TfluidsError("Need a density val per channel")
    if U.dim() != 5 or U.size(0) != 1:
        raise TfluidsError("Only single batch allowed.")
    _, C, zdim, ydim, xdim = U.shape
    is3D = C == 3
    if not is3D and zdim != 1:
        raise TfluidsError("2D plume needs zdim == 1")
    centerX = xdim // 2
    centerZ = max((zTotal if zTotal is not None else zdim) // 2, 1)
    plumeRad = int(math.floor(xdim * rad))
    x = torch.arange(1, xdim + 1, device=U.device).view(1, 1, xdim)
    z = torch.arange(1 + zOffset, zdim + 1 + zOffset, device=U.device).view(zdim, 1, 1)
    inside = ((centerX - x) ** 2 + (centerZ - z) ** 2) <= plumeRad * plumeRad   # [Z, 1, X]
    rows = min(4, ydim)
    inside = inside.expand(zdim, rows, xdim)
    batch["UBCInvMask"][0, :, :, :rows, :] = 0.0                                   # in or out of the plume
    batch["UBC"][0, 1, :, :rows, :] = torch.where(inside, float(uScale), 0.0).to(U.dtype)
    dbc, dmask = [], []
    for c, val in zip(chans, densityVal):
        bc, mk = torch.zeros_like(c), torch.ones_like(c)
        bc[0, 0, :, :rows, :] = torch.where(inside, float(val), 0.0).to(c.dtype)
        mk[0, 0, :, :rows, :] = torch.where(inside, 0.0, 1.0).to(c.dtype)
        dbc.append(bc)
        dmask.append(mk)
    batch["densityBC"] = dbc if multi else dbc[0]
    batch["densityBCInvMask"] = dmask if multi else dmask[0]


class _PairCache:
    """Per-(BC, BCInvMask) cache. Entries are keyed by the IDENTITY of the two tensor objects (held through
    weak references) and validated by torch's in-place version counters, so neither an address re-used by the
    caching allocator after the tensors were freed nor an interactive edit (the 2-D demo) can resurrect a
    stale payload; an entry is dropped -- and `on_evict(payload)` called -- when either tensor dies."""

    def __init__(self, on_evict=None):
        self._entries = {}
        self._on_evict = on_evict

    def _drop(self, key):
        ent = self._entries.pop(key, None)
        if ent is not None and self._on_evict is not None:
            self._on_evict(ent[4])

    def get(self, bc, inv):
        ent = self._entries.get((id(bc), id(inv)))
        if ent is None:
            return None
        if ent[0]() is bc and ent[1]() is inv and ent[2] == bc._version and ent[3] == inv._version:
            return ent[4]
        self._drop((id(bc), id(inv)))
        return None

    def put(self, bc, inv, payload):
        import weakref
        key = (id(bc), id(inv))
        self._drop(key)
        dead = lambda _ref, key=key: self._drop(key)
        self._entries[key] = (weakref.ref(bc, dead), weakref.ref(inv, dead), bc._version, inv._version, payload)

    def clear(self):
        for key in list(self._entries):
            self._drop(key)

    def __len__(self):
        return len(self._entries)


_bc_index_cache = _PairCache()   # -> (int32 index tensor, idempotent)


def _bc_indices(bc, inv):
    """(idx, idempotent): idx = the element indices where the BC pair is not the identity (invMask != 1 or
    bc != 0); idempotent = every such element has invMask == 0 (and |bc| <= 1e6), i.e. x*0 + bc applied twice
    equals applied once, bit for bit, and commutes with the step's final clamp to +-1e6. Cached per tensor pair
    (_PairCache: object identity + in-place version counters; the 2-D demo edits its BCs interactively)."""
    hit = _bc_index_cache.get(bc, inv)
    if hit is not None:
        return hit
    flat_inv = inv.reshape(-1)
    idx = torch.nonzero((flat_inv != 1) | (bc.reshape(-1) != 0)).reshape(-1)
    idem = bool((flat_inv[idx] == 0).all().item())
    if idem and idx.numel():     # "idempotent" also promises |bc| <= 1e6: the final U:clamp(-1e6, 1e6) leaves them alone
        idem = bool((bc.reshape(-1)[idx].abs() <= 1e6).all().item())
    idx = idx.to(torch.int32)
    _bc_index_cache.put(bc, inv, (idx, idem))
    return idx, idem


def _sparse_bc(x, bc, inv):
    """The index list of a BC pair if it is worth using (fewer than a quarter of the elements), else None."""
    if bc is None or inv is None or x.numel() >= 2 ** 31:
        return None
    idx, idem = _bc_indices(bc, inv)
    return (idx, idem) if idx.numel() * 4 < x.numel() else None


def _apply(x, bc, inv, clamp=None):
    lib, ctx = tfluids._context(x)
    sp = _sparse_bc(x, bc, inv) if clamp is None else None
    if sp is not None:
        idx = sp[0]
        tfluids._call(lib, ctx, lib.tfl_applyBCsIndexed(ctx, tfluids._tt5(x), tfluids._tt5(bc), tfluids._tt5(inv),
                                                        ctypes.c_void_p(idx.data_ptr()), idx.numel()))
        return
    lo, hi = clamp if clamp is not None else (0.0, 0.0)
    tfluids._call(lib, ctx, lib.tfl_applyBCs(ctx, tfluids._tt5(x), tfluids._tt5(bc) if bc is not None else None,
                                             tfluids._tt5(inv) if inv is not None else None,
                                             int(clamp is not None), lo, hi))


def _apply_many(triples, unchanged=False):
    """x = x*invMask + bc for several (x, bc, invMask) triples: the sparse ones go out as ONE launch
    (tfl_applyBCsIndexedMulti), dense ones one launch each. unchanged=True: x still holds the result of an
    earlier application of the same pair -- an idempotent pair (invMask == 0 on all its cells) is then skipped."""
    sparse = []
    for x, bc, inv in triples:
        sp = _sparse_bc(x, bc, inv)
        if sp is None:
            _apply(x, bc, inv)
        elif not (unchanged and sp[1]):
            sparse.append((x, bc, inv, sp[0]))
    if not sparse:
        return
    if len(sparse) == 1:
        x, bc, inv, _ = sparse[0]
        _apply(x, bc, inv)
        return
    for lo in range(0, len(sparse), 8):
        part = sparse[lo:lo + 8]
        lib, ctx = tfluids._context(part[0][0])
        n = len(part)
        kx, ax = tfluids._desc_array([t[0] for t in part])
        kb, ab = tfluids._desc_array([t[1] for t in part])
        km, am = tfluids._desc_array([t[2] for t in part])
        ai = (ctypes.c_void_p * n)(*[t[3].data_ptr() for t in part])
        an = (ctypes.c_int64 * n)(*[t[3].numel() for t in part])
        tfluids._call(lib, ctx, lib.tfl_applyBCsIndexedMulti(ctx, n, ax, ab, am, ai, an))
        del kx, kb, km


def setConstVals(batch, p, U, flags, density, unchanged=()):
    """simulate.lua:130-160: X = X*invMask + BC for p, U and each density channel. `unchanged` names the fields
    ('p', 'U', 'density') that nothing has written since the previous setConstVals of this step (simulate()
    knows; idempotent BC pairs on them are skipped -- the result is identical)."""
    fresh, stale = [], []
    def add(name, x, bc, inv):
        (stale if name in unchanged else fresh).append((x, bc, inv))
    if batch.get("pBC") is not None or batch.get("pBCInvMask") is not None:
        add("p", p, batch["pBC"], batch["pBCInvMask"])
    if batch.get("UBC") is not None or batch.get("UBCInvMask") is not None:
        add("U", U, batch["UBC"], batch["UBCInvMask"])
    if batch.get("densityBC") is not None or batch.get("densityBCInvMask") is not None:
        if isinstance(density, (list, tuple)):
            if len(density) != len(batch["densityBC"]):
                raise TfluidsError("density / densityBC channel mismatch")
            for i in range(len(density)):
                add("density", density[i], batch["densityBC"][i], batch["densityBCInvMask"][i])
        else:
            add("density", density, batch["densityBC"], batch["densityBCInvMask"])
    _apply_many(fresh)
    _apply_many(stale, unchanged=True)


def _gravity(mconf):
    g = mconf.get("gravity")
    if g is None:
        return [0.0, 1.0, 0.0]
    return [float(v) for v in (g.tolist() if torch.is_tensor(g) else g)]


def _f32(v):
    return torch.tensor(v, dtype=torch.float32).item()


def simulate(conf, mconf, batch, model, outputDiv=False):
    """tfluids.simulate, simulate.lua:175-327 (same call order, same in-place state update)."""
    p, U, flags, density = getPUFlagsDensityReference(batch)
    dt = mconf["dt"]
    method = mconf.get("advectionMethod")
    strength = mconf.get("maccormackStrength")
    if density is not None:
        chans = density if isinstance(density, (list, tuple)) else [density]
        for chan in chans:
            tfluids.advectScalar(dt, chan, U, flags, method, None, False, strength)
    # U:copy(advected) (init.lua:216-218) is folded into addBuoyancy when buoyancy is on: the advected field
    # stays in advectVel's scratch, takes the BCs there, and addBuoyancy writes U = scratch + buoyancy.
    buoyant = density is not None and mconf.get("buoyancyScale", 0) > 0
    Uadv = tfluids.advectVel(dt, U, flags, method, None, strength, _deferCopy=buoyant)
    setConstVals(batch, p, Uadv if buoyant else U, flags, density)

    if buoyant:
        s = _f32(-(tfluids.getDx(flags) / 4) * mconf["buoyancyScale"])   # gravity:mul(...) on a float tensor
        g = [_f32(v) * s for v in _gravity(mconf)]
        d0 = density[0] if isinstance(density, (list, tuple)) else density
        tfluids.addBuoyancy(U, flags, d0, g, dt, USrc=Uadv)
    if mconf.get("gravityScale", 0) > 0:
        s = _f32((-tfluids.getDx(flags) / 4) * mconf["gravityScale"])
        g = [_f32(v) * s for v in _gravity(mconf)]
        tfluids.addGravity(U, flags, g, dt)
    if mconf.get("vorticityConfinementAmp", 0) > 0:
        tfluids.vorticityConfinement(U, flags, tfluids.getDx(flags) * mconf["vorticityConfinementAmp"])
    if outputDiv:
        return

    simMethod = mconf.get("simMethod") or "convnet"
    if simMethod != "convnet":
        tfluids.setWallBcsForward(U, flags)
    # only U has been written since the first setConstVals (buoyancy, gravity, confinement, wall BCs)
    setConstVals(batch, p, U, flags, density, unchanged=("p", "density"))

    fused_tail = False
    if simMethod == "convnet":
        # model:forward + p:copy(pPred); U:copy(UPred) (simulate.lua:262-272): the prediction is written straight
        # into the state tensors, and the trailing setConstVals(U) + U:clamp (simulate.lua:321-326) ride in the
        # projection's last kernel. When the U BC pair is a
        # sparse idempotent one (the plume: 4 of 128 rows, invMask 0), reading two dense BC fields there costs
        # more than touching the BC cells afterwards: clamp(0*u + bc) = bc, so the index-list form is exact.
        ubc, umask = batch.get("UBC"), batch.get("UBCInvMask")
        sp = _sparse_bc(U, ubc, umask)
        late_ubc = sp is not None and sp[1]
        model.forward([p, U, flags], out=[p, U], UBC=None if late_ubc else ubc, UBCInvMask=None if late_ubc else umask,
                      clamp=(-1e6, 1e6))
        fused_tail = True
        # p was rewritten by the model, density has not changed since setConstVals #2
        rest = batch if late_ubc else {k: v for k, v in batch.items() if k not in ("UBC", "UBCInvMask")}
        setConstVals(rest, p, U, flags, density, unchanged=("density",))
    elif simMethod == "jacobi":
        div = batch.get("div")
        if div is None or div.shape != p.shape:
            div = torch.empty_like(p)
            batch["div"] = div
        tfluids.velocityDivergenceForward(U, flags, div)
        is3D = U.size(1) == 3
        tfluids.solveLinearSystemJacobi(p, flags, div, is3D, 0, mconf.get("maxIter") or 100, residual=False)
        tfluids.velocityUpdateForward(U, flags, p)
    elif simMethod == "pcg":
        # simulate.lua:281-286: tol 1e-4, maxIter (default 100), 'ic0' -- the default here too (pipelined wavefront
        # sweeps, pcg.hip); mconf.pcgPrecond = 'none' | 'ilu0' | 'ic0' overrides ('none' is ~1.7x faster per solve at
        # 128^3 on this machine but needs ~3x the iterations, which matters under a small maxIter).
        div = batch.get("div")
        if div is None or div.shape != p.shape:
            div = torch.empty_like(p)
            batch["div"] = div
        tfluids.velocityDivergenceForward(U, flags, div)
        tfluids.solveLinearSystemPCG(p, flags, div, U.size(1) == 3, 1e-4, mconf.get("maxIter") or 100,
                                     mconf.get("pcgPrecond") or "ic0")
        tfluids.velocityUpdateForward(U, flags, p)
    else:
        raise TfluidsError("mconf.simMethod (%s) is not a valid option" % simMethod)

    if not fused_tail:
        setConstVals(batch, p, U, flags, density, unchanged=("density",))
        _apply(U, None, None, clamp=(-1e6, 1e6))


_dead_plans = []      # plans whose tensors have died, waiting for a safe moment
_WALL_PLANS = os.environ.get("TFL_WALL_PLAN", "1") != "0"


def _destroy_plan(payload):
    """Called from a weak-reference callback, i.e. whenever the garbage collector gets round to a dead BC tensor -- possibly in
    the middle of somebody's HIP-graph capture (GraphedSimulate, SlabSimulation's recorded rank-step, the user's own), where the
    hipFree inside tfl_bc_plan_destroy invalidates the capture (round 6: the whole GPU suite in ONE process failed its 153rd
    test that way). So the plan is only queued here; _flush_dead_plans() frees the queue at the next plan look-up outside a
    capture."""
    _dead_plans.append(payload)


def _flush_dead_plans():
    if not _dead_plans:
        return
    try:
        if torch.cuda.is_current_stream_capturing():
            return
    except Exception:      # noqa: BLE001  (no device: nothing can be capturing)
        pass
    while _dead_plans:
        ent = _dead_plans.pop()
        if len(ent) == 4:
            ent[0].tfl_wall_plan_destroy(ent[1], ent[2])
        else:
            ent[0].tfl_bc_plan_destroy(ent[1], ent[2])


_plan_cache = _PairCache(on_evict=_destroy_plan)   # -> (lib, ctx, tfl_bc_plan*)


def _bc_plan(lib, ctx, bc, inv):
    """tfl_bc_plan of a (BC, BCInvMask) pair. The plan keeps raw device pointers into the two tensors, so it lives
    exactly as long as they do: cached by tensor identity (_PairCache), re-created when either tensor is edited in
    place, destroyed (tfl_bc_plan_destroy) when either is freed or replaced."""
    if bc is None or inv is None:
        return None
    _flush_dead_plans()
    # plans are context-independent (they hold device pointers and an index list; tfl_bc_plan_destroy ignores ctx):
    # a second context on the same tensors (SlabSimulation(own_context=True)) shares the plan instead of evicting it
    # from under a holder of the raw pointer
    hit = _plan_cache.get(bc, inv)
    if hit is not None:
        return hit[2]
    plan = lib.tfl_bc_plan_create(ctx, tfluids._tt5(bc), tfluids._tt5(inv))
    if not plan:
        raise TfluidsError("tfl_bc_plan_create failed")
    _plan_cache.put(bc, inv, (lib, ctx, plan))
    return plan


def _destroy_wall_plans(payload):
    for (lib, ctx, plan) in payload.values():
        lib.tfl_wall_plan_retire(plan)      # out of the context's registry NOW (no HIP call): the flags' address may be handed out again
        _dead_plans.append((lib, ctx, plan, "wall"))      # ... and freed at the next safe moment


_wall_cache = _PairCache(on_evict=_destroy_wall_plans)   # flags tensor -> {ctx: (lib, ctx, tfl_wall_plan*)}


def wall_plan(lib, ctx, flags):
    """The tfl_wall_plan of a flags tensor on this context (include/tfluids_hip.h: the setWallBcs decisions of the scene as one
    code per cell, computed once; the projection's first and last kernel then read codes instead of re-deriving them from ten rows
    of flag words every step). Created at the SECOND call with the same flags, cached by the tensor's identity and torch's in-place version counter like the BC plans, so an edited or a
    new flags tensor gets a new plan; tfluids.emptyDomain (which writes flags through the library) drops the entry itself.
    TFL_WALL_PLAN=0 switches the plans off (every step decodes the flags, as before round 6)."""
    if not _WALL_PLANS or flags is None or not flags.is_cuda:
        return None
    _flush_dead_plans()
    hit = _wall_cache.get(flags, flags)
    if hit is None:
        # first sighting of this tensor (or of this version of it): only remember it. A plan costs an allocation, a kernel and a
        # device synchronisation -- worth it for a scene that is stepped again and again, a loss for a caller that brings new
        # flags with every call (a training loop over random batches); the SECOND call with the same flags creates the plan
        _wall_cache.put(flags, flags, {})
        return None
    ent = hit.get(ctx)
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            return None          # (creating one allocates and synchronises: not inside somebody's graph capture)
        plan = lib.tfl_wall_plan_create(ctx, tfluids._tt(flags))
        if not plan:
            return None          # (not fatal: the step decodes the flags itself)
        ent = hit[ctx] = (lib, ctx, plan)
    return ent[2]


def drop_wall_plan(flags):
    """Forget the plan of a flags tensor that was just written through a raw pointer (torch's version counter does not see it)."""
    _wall_cache._drop((id(flags), id(flags)))


def _native_args(lib, ctx, mconf, batch, model, outputDiv=False):
    """(tfl_sim_params, tfl_sim_state, keep-alive list) of a batch / mconf pair for the native step entry points."""
    from ._lib import tfl_sim_params, tfl_sim_state
    p, U, flags, density = getPUFlagsDensityReference(batch)
    chans = [] if density is None else (list(density) if isinstance(density, (list, tuple)) else [density])
    prm = tfl_sim_params()
    prm.dt = float(mconf["dt"])
    method = mconf.get("advectionMethod")
    prm.advectionMethod = method.encode() if method else None
    strength = mconf.get("maccormackStrength")
    prm.maccormackStrength = 0.75 if strength is None else float(strength)
    prm.buoyancyScale = float(mconf.get("buoyancyScale", 0) or 0)
    prm.gravityScale = float(mconf.get("gravityScale", 0) or 0)
    for i, v in enumerate(_gravity(mconf)):
        prm.gravity[i] = v
    prm.vorticityConfinementAmp = float(mconf.get("vorticityConfinementAmp", 0) or 0)
    sm = mconf.get("simMethod")
    prm.simMethod = sm.encode() if sm else None
    prm.maxIter = int(mconf.get("maxIter") or 0)
    pc = mconf.get("pcgPrecond")
    prm.pcgPrecond = pc.encode() if pc else None
    prm.outputDiv = int(bool(outputDiv))
    keep = [tfluids._desc5(t) for t in (p, U, flags)] + [tfluids._desc5(c) for c in chans]
    st = tfl_sim_state()
    st.p, st.U, st.flags = (ctypes.pointer(k) for k in keep[:3])
    st.n_density = len(chans)
    for i in range(len(chans)):
        st.density[i] = ctypes.pointer(keep[3 + i])
    if model is not None and str(mconf.get("simMethod") or "") == "convnet":
        wall_plan(lib, ctx, flags)      # registered with the context: tfl_model_begin finds it by the flags' address
    st.pBC = _bc_plan(lib, ctx, batch.get("pBC"), batch.get("pBCInvMask"))
    st.UBC = _bc_plan(lib, ctx, batch.get("UBC"), batch.get("UBCInvMask"))
    if chans and batch.get("densityBC") is not None:
        dbc, dmk = batch["densityBC"], batch["densityBCInvMask"]
        if not isinstance(dbc, (list, tuple)):
            dbc, dmk = [dbc], [dmk]
        if len(dbc) != len(chans):
            raise TfluidsError("density / densityBC channel mismatch")
        for i in range(len(chans)):
            st.densityBC[i] = _bc_plan(lib, ctx, dbc[i], dmk[i])
    st.model = model._handle(lib, ctx, U.device.index) if model is not None else None
    return prm, st, keep


def simulate_native(conf, mconf, batch, model, outputDiv=False):
    """The same step through ONE C-ABI call, tfl_simulate_step (fluidnet_amd/csrc/simulate.cpp): what a LuaJIT / cgo
    host would bind instead of re-implementing this file's orchestration. Bit-identical to simulate()."""
    U = batch["UDiv"]
    lib, ctx = tfluids._context(U)
    prm, st, keep = _native_args(lib, ctx, mconf, batch, model, outputDiv)
    nws = int(lib.tfl_simulate_workspace_floats(ctx, ctypes.byref(prm), ctypes.byref(st)))
    ws = tfluids.getTempStorage(U, [(nws,)])[0]
    tfluids._call(lib, ctx, lib.tfl_simulate_step(ctx, ctypes.byref(prm), ctypes.byref(st), ctypes.c_void_p(ws.data_ptr()), nws))
    del keep


class GraphedSimulate:
    """simulate() captured once into a HIP graph and replayed: one host call per step instead of ~25
    kernel launches + Python dispatch. The reference's 2-D path is launch-bound (~70
    launches, 0.95 ms FPROP at 128^2); so is ours below ~64^3. State tensors keep their identity (the
    graph works in place on batch['pDiv'|'UDiv'|'density']), so callers observe the same in-place
    semantics as lib/simulate.lua. Re-create (or call .capture()) after changing mconf or BC tensors'
    sparsity pattern; scalar parameters are baked into the captured launches.

    Only capturable configurations: convnet, or jacobi with a fixed iteration count (pTol = 0) -- both
    are what lib/simulate.lua does (simulate.lua:287-291)."""

    def __init__(self, conf, mconf, batch, model=None, warmup=2, native=False):
        """native=True captures the ONE-call native step (tfl_simulate_step: the sparse setConstVals pairs and, in 3-D, the
        buoyancy force folded into the kernels that produce the fields -- fewer launches in the graph) instead of the
        operator-by-operator Python orchestration; same results (tests/test_hip_simulate.py)."""
        self.conf, self.mconf, self.batch, self.model = conf, mconf, batch, model
        self.native = bool(native)
        self.graph = None
        self._pinned = []     # every buffer the captured launches point into, kept alive with the graph
        self.capture(warmup)

    def _simulate(self):
        # the captured launches hold raw pointers into the tfluids scratch and the model workspace: run in a
        # scratch scope of our own (no other caller can grow / replace that buffer) -- see capture()
        prev, tfluids._scratch_scope = tfluids._scratch_scope, ("graph", id(self))
        try:
            if self.native:
                simulate_native(self.conf, self.mconf, self.batch, self.model)
            else:
                simulate(self.conf, self.mconf, self.batch, self.model)
        finally:
            tfluids._scratch_scope = prev

    def capture(self, warmup=2):
        dev = self.batch["UDiv"].device
        keep = {k: v.clone() for k, v in self._state().items()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):     # sizes scratch buffers, builds BC index caches, loads code
                self._simulate()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for k, v in self._state().items():      # warm-up must not advance the simulation
            v.copy_(keep[k])
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._simulate()
        torch.cuda.synchronize(dev)
        # Pin what the graph points into. The scratch is ours alone (private scope); the model workspace is shared
        # with eager callers of the same model, which may REPLACE it by a larger one: holding the reference keeps
        # the captured pointers valid (the old buffer cannot go back to the allocator), and step() re-captures
        # when it sees a different workspace object.
        key = (dev.index, ("graph", id(self)))
        self._pinned = [tfluids._tmp.get(key)]
        self._work = self.model._work.get(dev.index) if self.model is not None else None
        self._pinned.append(self._work)
        for t in self._state().values():
            self._pinned.append(t)
        for k in ("flags", "UBC", "UBCInvMask", "densityBC", "densityBCInvMask", "pBC", "pBCInvMask", "div"):
            v = self.batch.get(k)
            self._pinned.extend(v if isinstance(v, (list, tuple)) else [v])
        for k, v in self._state().items():      # neither must the capture pass (it does not execute, but
            v.copy_(keep[k])                    # keep the contract explicit)
        self.graph = g

    def _state(self):
        out = {}
        for k in ("pDiv", "UDiv", "density"):
            v = self.batch.get(k)
            if torch.is_tensor(v):
                out[k] = v
            elif isinstance(v, (list, tuple)):
                for i, t in enumerate(v):
                    out["%s%d" % (k, i)] = t
        return out

    def step(self):
        if self.model is not None and self.model._work.get(self.batch["UDiv"].device.index) is not self._work:
            self.capture(1)      # an eager forward on a larger grid replaced the model workspace
        self.graph.replay()

    def __del__(self):
        tmp = getattr(tfluids, "_tmp", None) if tfluids is not None else None      # both are None during interpreter shutdown
        if tmp is not None:
            tmp.pop((self.batch["UDiv"].device.index, ("graph", id(self))), None)

    __call__ = step
