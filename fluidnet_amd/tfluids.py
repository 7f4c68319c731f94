"""Host-side mirror of torch/tfluids/init.lua on torch (ROCm) tensors.

Same operator names, argument order, defaults, argument checks and in-place / copy-back behaviour
as the Lua wrappers (init.lua:89-735), so code written against `tfluids.*` reads the same. Tensors
are contiguous fp32 [B, C, Z, Y, X] on an MI355X; every call is asynchronous on torch's current
stream, like the reference's CUDA path. The work is done by libtfluids_hip.so through its C ABI
(include/tfluids_hip.h) -- torch only owns the memory and the stream.
"""
import ctypes

import torch

from . import _lib
from ._lib import TfluidsError, tfl_tensor

# tfluids.CellType (init.cu:108-120 exporting third_party/cell_type.h:22-33)
CellType = dict(TypeNone=0, TypeFluid=1, TypeObstacle=2, TypeEmpty=4, TypeInflow=8,
                TypeOutflow=16, TypeOpen=32, TypeStick=128)

_ctx = {}   # device index -> tfl_ctx*
_tmp = {}   # (device index, scope) -> grow-only flat scratch tensor (init.lua:22,35-64)
_scratch_scope = None   # virtual z-slab ranks sharing one process each get their own scratch (dist.py)


def _check(cond, msg):
    if not cond:
        raise TfluidsError(msg)


def _context(t):
    _check(t.is_cuda, "tfluids_hip operators need tensors on an MI355X (got a CPU tensor); "
                      "there is no CPU fallback")
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    lib = _lib.load()
    ctx = _ctx.get(dev)
    if ctx is None:
        ctx = lib.tfl_create(dev)
        if not ctx:
            raise TfluidsError("tfl_create(%d) failed" % dev)
        _ctx[dev] = ctx
    lib.tfl_set_stream(ctx, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    return lib, ctx


ADVECT_EXACT, ADVECT_FAST = 0, 1


def set_advect_mode(t, mode):
    """Arithmetic of the LDS-tiled 3-D advection kernels on t's device (include/tfluids_hip.h tfl_set_advect_mode):
    "exact" (default; bit-equal to the reference CPU path) or "fast" (the tolerance mode, rel-L2 <= 1e-5).
    Returns the previous mode's name."""
    lib, ctx = _context(t)
    names = {"exact": ADVECT_EXACT, "fast": ADVECT_FAST, ADVECT_EXACT: ADVECT_EXACT, ADVECT_FAST: ADVECT_FAST}
    _check(mode in names, "advect mode must be 'exact' or 'fast'")
    prev = lib.tfl_get_advect_mode(ctx)
    rc = lib.tfl_set_advect_mode(ctx, names[mode])
    if rc != 0:
        raise TfluidsError(lib.tfl_last_error(ctx).decode())
    return "fast" if prev == ADVECT_FAST else "exact"


def _tt(t):
    _check(t.dtype == torch.float32, "tensors must be float32")
    b, c, z, y, x = t.shape
    return ctypes.byref(tfl_tensor(t.data_ptr(), b, c, z, y, x))


def _tt5(t):
    """Any contiguous fp32 tensor with <= 5 dims as a 5-D descriptor (leading dims padded with 1)."""
    _check(t.dtype == torch.float32 and t.is_contiguous(), "tensors must be contiguous float32")
    sh = [1] * (5 - t.dim()) + list(t.shape)
    return ctypes.byref(tfl_tensor(t.data_ptr(), *sh))


def _desc5(t):
    """_tt5 as a tfl_tensor struct (for arrays of descriptors)."""
    _check(t.dtype == torch.float32 and t.is_contiguous(), "tensors must be contiguous float32")
    sh = [1] * (5 - t.dim()) + list(t.shape)
    return tfl_tensor(t.data_ptr(), *sh)


def _desc_array(tensors):
    """(keepalive, array of tfl_tensor*) for a `const tfl_tensor* const*` argument."""
    descs = [_desc5(t) for t in tensors]
    arr = (ctypes.POINTER(tfl_tensor) * len(descs))(*[ctypes.pointer(d) for d in descs])
    return descs, arr


def _call(lib, ctx, rc):
    if rc != 0:
        raise TfluidsError(lib.tfl_last_error(ctx).decode())


def getTempStorage(like, sizes):
    """init.lua:35-64: one grow-only buffer per device, carved into views; contents undefined."""
    dev = like.device.index
    total = 0
    numels = []
    for s in sizes:
        n = 1
        for v in s:
            _check(v > 0, "tensor sizes must be positive non-zero")
            n *= v
        numels.append(n)
        total += n
    key = (dev, _scratch_scope)
    buf = _tmp.get(key)
    if buf is None or buf.numel() < total:
        buf = torch.empty(total, dtype=torch.float32, device=like.device)
        _tmp[key] = buf
    out, off = [], 0
    for s, n in zip(sizes, numels):
        out.append(buf[off:off + n].view(*s))
        off += n
    return out


def _dims(U, flags):
    _check(U.dim() == 5 and flags.dim() == 5, "Dimension mismatch")
    _check(flags.size(1) == 1, "flags is not scalar")
    bsz, _, d, h, w = flags.shape
    is3D = U.size(1) == 3
    if not is3D:
        _check(d == 1, "2D velocity field but zdepth > 1")
        _check(U.size(1) == 2, "2D velocity field must have only 2 channels")
    _check(U.size(0) == bsz and U.size(2) == d and U.size(3) == h and U.size(4) == w,
           "Size mismatch")
    _check(U.is_contiguous() and flags.is_contiguous(), "Input is not contiguous")
    return bsz, d, h, w, is3D


def advectScalar(dt, s, U, flags, method=None, sDst=None, sampleOutsideFluid=None,
                 maccormackStrength=None, boundaryWidth=None):
    """init.lua:89-149. In place on `s` unless sDst is given."""
    method = method or "maccormackOurs"
    boundaryWidth = boundaryWidth or 1
    sampleOutsideFluid = bool(sampleOutsideFluid) if sampleOutsideFluid is not None else False
    maccormackStrength = 0.75 if maccormackStrength is None else maccormackStrength
    _check(s.dim() == 5, "Dimension mismatch")
    bsz, d, h, w, is3D = _dims(U, flags)
    _check(s.shape == flags.shape, "Size mismatch")
    _check(s.is_contiguous(), "Input is not contiguous")
    C = U.size(1)
    sizes = [(bsz, 1, d, h, w), (bsz, 1, d, h, w), (bsz, C, d, h, w), (bsz, C, d, h, w)]
    # maccormackOurs' second pass reads s only at the cell it writes: the library accepts sDst == s,
    # which makes init.lua:146-148's copy-back unnecessary for the default method.
    inplace = sDst is None and method == "maccormackOurs"
    if sDst is None and not inplace:
        sizes.append((bsz, 1, d, h, w))
    elif sDst is not None:
        _check(sDst.dim() == 5 and sDst.shape == s.shape, "Size mismatch")
        _check(sDst.is_contiguous(), "Input is not contiguous")
    tmp = getTempStorage(s, sizes)
    fwd, bwd, fwdPos, bwdPos = tmp[:4]
    out = sDst if sDst is not None else (s if inplace else tmp[4])
    lib, ctx = _context(s)
    _call(lib, ctx, lib.tfl_advectScalar(ctx, dt, _tt(s), _tt(U), _tt(flags), _tt(fwd), _tt(bwd),
                                         int(is3D), method.encode(), _tt(fwdPos), _tt(bwdPos),
                                         int(boundaryWidth), int(sampleOutsideFluid),
                                         maccormackStrength, _tt(out)))
    if sDst is None and not inplace:
        s.copy_(out)


def advectVel(dt, U, flags, method=None, UDst=None, maccormackStrength=None, boundaryWidth=None, _deferCopy=False):
    """init.lua:170-219. In place on `U` unless UDst is given.
    _deferCopy (simulate() only): with UDst None, skip the trailing U:copy and return the scratch tensor that
    holds the result -- the caller folds the copy into its next pass over U (addBuoyancy(..., USrc=))."""
    method = method or "maccormackOurs"
    boundaryWidth = boundaryWidth or 1
    maccormackStrength = 0.75 if maccormackStrength is None else maccormackStrength
    bsz, d, h, w, is3D = _dims(U, flags)
    C = U.size(1)
    if UDst is None:
        tmp = getTempStorage(U, [(bsz, C, d, h, w)] * 3)
    else:
        tmp = getTempStorage(U, [(bsz, C, d, h, w)] * 2)
        _check(UDst.dim() == 5 and UDst.shape == U.shape, "Size mismatch")
        _check(UDst.is_contiguous(), "Input is not contiguous")
    fwd, bwd = tmp[0], tmp[1]
    out = UDst if UDst is not None else tmp[2]
    lib, ctx = _context(U)
    _call(lib, ctx, lib.tfl_advectVel(ctx, dt, _tt(U), _tt(flags), _tt(fwd), _tt(bwd), int(is3D),
                                      method.encode(), int(boundaryWidth), maccormackStrength,
                                      _tt(out)))
    if UDst is None:
        if _deferCopy:
            return out
        U.copy_(out)
    return None


def setWallBcsForward(U, flags):
    """init.lua:226-247 (in place)."""
    _, _, _, _, is3D = _dims(U, flags)
    lib, ctx = _context(U)
    _call(lib, ctx, lib.tfl_setWallBcsForward(ctx, _tt(U), _tt(flags), int(is3D)))


def setWallBcsBackward(flags, gradOutput, gradU=None):
    """tfluids.SetWallBcs:updateGradInput (tfluids/set_wall_bcs.lua:50-66): gradient w.r.t. U of the wall-BC module
    (= mask * gradOutput, the forward operator applied to the gradient); the gradient w.r.t. flags is zero."""
    _, _, _, _, is3D = _dims(gradOutput, flags)
    if gradU is None:
        gradU = torch.empty_like(gradOutput)
    _check(gradU.shape == gradOutput.shape and gradU.is_contiguous(), "Size mismatch")
    lib, ctx = _context(gradOutput)
    _call(lib, ctx, lib.tfl_setWallBcsBackward(ctx, _tt(flags), _tt(gradOutput), int(is3D), _tt(gradU)))
    return gradU


def velocityDivergenceForward(U, flags, UDiv):
    """init.lua:255-279."""
    _, _, _, _, is3D = _dims(U, flags)
    _check(UDiv.dim() == 5 and UDiv.shape == flags.shape, "Size mismatch")
    _check(UDiv.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(U)
    _call(lib, ctx, lib.tfl_velocityDivergenceForward(ctx, _tt(U), _tt(flags), _tt(UDiv),
                                                      int(is3D)))


def velocityUpdateForward(U, flags, p):
    """init.lua:322-347 (in place on U)."""
    _, _, _, _, is3D = _dims(U, flags)
    _check(p.dim() == 5 and p.shape == flags.shape, "Size mismatch")
    _check(p.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(U)
    _call(lib, ctx, lib.tfl_velocityUpdateForward(ctx, _tt(U), _tt(flags), _tt(p), int(is3D)))


def velocityDivergenceBackward(U, flags, gradOutput, gradU):
    """init.lua:288-313."""
    _, _, _, _, is3D = _dims(U, flags)
    _check(gradOutput.dim() == 5 and gradU.dim() == 5, "Dimension mismatch")
    _check(gradOutput.shape == flags.shape and gradU.shape == U.shape, "Size mismatch")
    _check(gradOutput.is_contiguous() and gradU.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(U)
    _call(lib, ctx, lib.tfl_velocityDivergenceBackward(ctx, _tt(U), _tt(flags), _tt(gradOutput), int(is3D),
                                                       _tt(gradU)))


def velocityUpdateBackward(U, flags, p, gradOutput, gradP):
    """init.lua:358-383."""
    _, _, _, _, is3D = _dims(U, flags)
    _check(p.dim() == 5 and gradOutput.dim() == 5 and gradP.dim() == 5, "Dimension mismatch")
    _check(gradP.shape == p.shape and p.shape == flags.shape and gradOutput.shape == U.shape, "Size mismatch")
    _check(p.is_contiguous() and gradOutput.is_contiguous() and gradP.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(U)
    _call(lib, ctx, lib.tfl_velocityUpdateBackward(ctx, _tt(U), _tt(flags), _tt(p), _tt(gradOutput), int(is3D),
                                                   _tt(gradP)))


def volumetricUpSamplingNearestForward(ratio, input, output):
    """init.lua:618-622."""
    _check(input.dim() == 5 and output.dim() == 5, "ERROR: input and output must be dim 5")
    _check(input.is_contiguous() and output.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(input)
    _call(lib, ctx, lib.tfl_volumetricUpSamplingNearestForward(ctx, int(ratio), _tt(input), _tt(output)))


def volumetricUpSamplingNearestBackward(ratio, input, gradOutput, gradInput):
    """init.lua:623-627."""
    _check(input.dim() == 5 and gradOutput.dim() == 5 and gradInput.dim() == 5,
           "ERROR: input, gradOutput and gradInput must be dim 5")
    _check(gradOutput.is_contiguous() and gradInput.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(input)
    _call(lib, ctx, lib.tfl_volumetricUpSamplingNearestBackward(ctx, int(ratio), _tt(input), _tt(gradOutput),
                                                                _tt(gradInput)))


def vorticityConfinement(U, flags, strength, USrc=None):
    """init.lua:394-430 (in place on U).
    USrc (extension): U = USrc + confinement(USrc), every cell of U written (tfl_vorticityConfinementFrom: one fused
    z-marched launch on 3-D grids from 2 M cells per batch item on (arrays 64+ planes deep), the two launches reading USrc below); U must not alias USrc."""
    bsz, d, h, w, is3D = _dims(U, flags)
    _check(isinstance(strength, (int, float)), "strength must be a number")
    C = U.size(1)
    if USrc is not None:
        _check(USrc.shape == U.shape and USrc.is_contiguous(), "USrc must have U's shape")
        curl, curlNorm = getTempStorage(U, [(bsz, 3, d, h, w), (bsz, 1, d, h, w)])
        lib, ctx = _context(U)
        _call(lib, ctx, lib.tfl_vorticityConfinementFrom(ctx, _tt(USrc), _tt(U), _tt(flags), float(strength), _tt(curl),
                                                         _tt(curlNorm), int(is3D)))
        return
    centered, curl, curlNorm, force = getTempStorage(
        U, [(bsz, C, d, h, w), (bsz, 3, d, h, w), (bsz, 1, d, h, w), (bsz, C, d, h, w)])
    lib, ctx = _context(U)
    _call(lib, ctx, lib.tfl_vorticityConfinement(ctx, _tt(U), _tt(flags), float(strength),
                                                 _tt(centered), _tt(curl), _tt(curlNorm),
                                                 _tt(force), int(is3D)))


def _vec3(g, name):
    if torch.is_tensor(g):
        _check(g.dim() == 1 and g.size(0) == 3, "%s must be a 3D vector (even in 2D)." % name)
        g = g.detach().cpu().tolist()
    _check(len(g) == 3, "%s must be a 3D vector (even in 2D)." % name)
    return (ctypes.c_float * 3)(*[float(v) for v in g])


def addBuoyancy(U, flags, density, gravity, dt, USrc=None):
    """init.lua:442-470 (in place on U). gravity: 3 floats (host tensor, list or tuple).
    USrc (extension): U = USrc + buoyancy, every cell of U written (tfl_addBuoyancyFrom)."""
    _, _, _, _, is3D = _dims(U, flags)
    _check(density.dim() == 5 and density.shape == flags.shape, "Size mismatch")
    _check(density.is_contiguous(), "Input is not contiguous")
    _check(isinstance(dt, (int, float)), "time step must be a number")
    lib, ctx = _context(U)
    if USrc is not None and USrc.data_ptr() != U.data_ptr():
        _check(USrc.shape == U.shape and USrc.is_contiguous(), "Size mismatch")
        _call(lib, ctx, lib.tfl_addBuoyancyFrom(ctx, _tt(USrc), _tt(U), _tt(flags), _tt(density),
                                                _vec3(gravity, "gravity"), float(dt), int(is3D)))
        return
    _call(lib, ctx, lib.tfl_addBuoyancy(ctx, _tt(U), _tt(flags), _tt(density), _vec3(gravity, "gravity"),
                                        None, float(dt), int(is3D)))


def addGravity(U, flags, gravity, dt):
    """init.lua:481-506 (in place on U)."""
    _, _, _, _, is3D = _dims(U, flags)
    _check(isinstance(dt, (int, float)), "time step must be a number")
    lib, ctx = _context(U)
    _call(lib, ctx, lib.tfl_addGravity(ctx, _tt(U), _tt(flags), _vec3(gravity, "gravity"), float(dt),
                                       int(is3D), None))


def emptyDomain(flags, is3D, bnd=None):
    """init.lua:545-553."""
    bnd = bnd or 1
    _check(flags.dim() == 5 and flags.size(1) == 1, "Flags should be 5D and scalar")
    _check(flags.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(flags)
    _call(lib, ctx, lib.tfl_emptyDomain(ctx, _tt(flags), int(bool(is3D)), int(bnd)))
    from .simulate import drop_wall_plan
    drop_wall_plan(flags)      # written through the library: torch's version counter has not moved
    return flags


_dx_override = {}  # device index -> global dx of a z-slab-decomposed grid (fluidnet_amd.dist)


def setDxOverride(like, dx):
    """A z-slab rank holds part of the grid; getDx = 1/max(X,Y,Z) is a property of the whole grid.
    dx > 0 pins it for this device (Python side and inside libtfluids_hip.so); None/0 clears it."""
    lib, ctx = _context(like)
    dev = like.device.index
    if dx:
        _dx_override[dev] = float(dx)
    else:
        _dx_override.pop(dev, None)
    _call(lib, ctx, lib.tfl_set_dx_override(ctx, float(dx or 0.0)))


def getDx(flags):
    """init.lua:560-564 | grid.cc:37-40."""
    _check(flags.dim() == 5, "Dimension mismatch")
    ov = _dx_override.get(flags.device.index) if flags.is_cuda else None
    if ov:
        return ov
    return 1.0 / max(flags.size(2), flags.size(3), flags.size(4))


def flagsToOccupancy(flags, occupancy):
    """init.lua:567-576."""
    _check(flags.dim() == 5 and flags.size(1) == 1, "Flags should be 5D and scalar")
    _check(occupancy.shape == flags.shape, "Size mismatch")
    _check(flags.is_contiguous() and occupancy.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(flags)
    _call(lib, ctx, lib.tfl_flagsToOccupancy(ctx, _tt(flags), _tt(occupancy)))


def rectangularBlur(src, blurRad, is3D, dst):
    """init.lua:578-595: O(n) box blur of radius blurRad with clamped edges (per line a running sum)."""
    _check(src.dim() == 5 and dst.dim() == 5, "Dimension mismatch")
    _check(src.shape == dst.shape, "Size mismatch")
    _check(blurRad > 0 and int(blurRad) == blurRad, "blurRad must be a positive, non-zero integer")
    _check(src.is_contiguous() and dst.is_contiguous(), "Input is not contiguous")
    tmp = getTempStorage(src, [tuple(src.shape)])[0]
    lib, ctx = _context(src)
    _call(lib, ctx, lib.tfl_rectangularBlur(ctx, _tt(src), int(blurRad), int(bool(is3D)), _tt(dst), _tt(tmp)))


def signedDistanceField(flags, searchRad, is3D, dst):
    """init.lua:597-613: distance to the nearest obstacle within searchRad (clamped there), O(pix * searchRad^dim)."""
    _check(flags.dim() == 5 and dst.dim() == 5, "Dimension mismatch")
    _check(flags.shape == dst.shape, "Size mismatch")
    _check(flags.is_contiguous() and dst.is_contiguous(), "Input is not contiguous")
    _check(flags.size(1) == 1, "flags must be scalar")
    _check(searchRad > 0 and int(searchRad) == searchRad, "searchRad must be a positive, non-zero integer")
    lib, ctx = _context(flags)
    _call(lib, ctx, lib.tfl_signedDistanceField(ctx, _tt(flags), int(searchRad), int(bool(is3D)), _tt(dst)))


def solveLinearSystemPCG(p, flags, div, is3D, tol=None, maxIter=None, precondType=None, verbose=None):
    """init.lua:645-677: the baseline (P)CG pressure solve, A p = div per connected fluid component.
    precondType 'none' | 'ilu0' | 'ic0' (default 'ic0'), tol default 1e-6, maxIter default 1000.
    Returns the max residual over the solved systems (host sync, as in the reference)."""
    _check(p.dim() == 5 and flags.dim() == 5 and div.dim() == 5, "Dimension mismatch")
    _check(flags.size(1) == 1, "flags is not scalar")
    bsz, _, d, h, w = flags.shape
    _check(p.shape == flags.shape, "size mismatch")
    _check(div.shape == flags.shape, "size mismatch")
    if not is3D:
        _check(d == 1, "d > 1 for a 2D domain")
    verbose = bool(verbose)
    precondType = precondType or "ic0"
    tol = 1e-6 if tol is None else tol
    maxIter = 1000 if maxIter is None else maxIter
    _check(p.is_contiguous() and flags.is_contiguous() and div.is_contiguous(), "Input is not contiguous")
    lib, ctx = _context(p)
    nws = int(lib.tfl_pcg_workspace_floats(d, h, w))
    ws = getTempStorage(p, [(nws,)])[0]
    res = ctypes.c_float(0.0)
    _call(lib, ctx, lib.tfl_solveLinearSystemPCG(ctx, _tt(p), _tt(flags), _tt(div), int(bool(is3D)),
                                                 str(precondType).encode(), float(tol), int(maxIter), int(verbose),
                                                 ctypes.c_void_p(ws.data_ptr()), nws, ctypes.byref(res)))
    return res.value


def normalizePressureMean(p, flags, is3D):
    """init.lua:747-764: subtract from p the mean over each connected fluid component (in place; on the device,
    where the reference round-trips through the host)."""
    _check(p.dim() == 5 and flags.dim() == 5 and flags.size(1) == 1, "Dimension mismatch")
    _check(p.shape == flags.shape, "size mismatch")
    _check(p.is_contiguous() and flags.is_contiguous(), "Input is not contiguous")
    _, _, d, h, w = flags.shape
    if not is3D:
        _check(d == 1, "d > 1 for a 2D domain")
    lib, ctx = _context(p)
    nws = int(lib.tfl_normalize_workspace_floats(d, h, w))
    ws = getTempStorage(p, [(nws,)])[0]
    _call(lib, ctx, lib.tfl_normalizePressureMean(ctx, _tt(p), _tt(flags), int(bool(is3D)), ctypes.c_void_p(ws.data_ptr()), nws))


def solveLinearSystemJacobi(p, flags, div, is3D, pTol=None, maxIter=None, verbose=None, residual=True):
    """init.lua:693-735. Returns the final residual (a Python float => one host sync at the end);
    residual=False skips the readback (returns None) when pTol <= 0, keeping the call fully async."""
    pTol = 1e-5 if pTol is None else pTol
    maxIter = 1000 if maxIter is None else maxIter
    verbose = bool(verbose)
    _check(p.dim() == 5 and flags.dim() == 5 and div.dim() == 5, "Dimension mismatch")
    _check(flags.size(1) == 1, "flags is not scalar")
    bsz, _, d, h, w = flags.shape
    _check(p.shape == flags.shape and div.shape == flags.shape, "size mismatch")
    if not is3D:
        _check(d == 1, "d > 1 for a 2D domain")
    _check(p.is_contiguous() and flags.is_contiguous() and div.is_contiguous(),
           "Input is not contiguous")
    pPrev, pDelta, pDeltaNorm = getTempStorage(p, [(bsz, 1, d, h, w), (bsz, 1, d, h, w), (bsz,)])
    lib, ctx = _context(p)
    res = ctypes.c_float(0.0)
    dn = tfl_tensor(pDeltaNorm.data_ptr(), bsz, 1, 1, 1, 1)
    _call(lib, ctx, lib.tfl_solveLinearSystemJacobi(ctx, _tt(p), _tt(flags), _tt(div), _tt(pPrev),
                                                    _tt(pDelta), ctypes.byref(dn), int(bool(is3D)),
                                                    float(pTol), int(maxIter), int(verbose),
                                                    ctypes.byref(res) if (residual or pTol > 0) else None))
    return res.value if (residual or pTol > 0) else None


def traceErrors(like):
    """Back-traces that hit a calcLineTrace invariant path since the last call (diagnostic)."""
    lib, ctx = _context(like)
    return int(lib.tfl_trace_errors(ctx))


class profile:
    """with tfluids.profile(tensor) as prof: ...; prof.kernels -> {name: {"calls": n, "ms": total}}.
    Per-kernel HIP-event timing inside the library (tfl_profile_begin/end)."""

    def __init__(self, like):
        self._like = like
        self.kernels = {}

    def __enter__(self):
        lib, ctx = _context(self._like)
        _call(lib, ctx, lib.tfl_profile_begin(ctx))
        return self

    def __exit__(self, *exc):
        import json
        lib, ctx = _context(self._like)
        buf = ctypes.create_string_buffer(1 << 16)
        n = lib.tfl_profile_end(ctx, buf, len(buf))
        if n < 0:
            raise TfluidsError(lib.tfl_last_error(ctx).decode())
        self.kernels = json.loads(buf.value.decode() or "{}")
        return False
