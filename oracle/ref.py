"""ctypes binding to oracle/_ref/libtfluids_ref.so -- TEST INFRASTRUCTURE ONLY.

The .so is the REFERENCE's own CPU tfluids code (torch/tfluids/init.cu compiled as C++ against a
fake TH/luaT shim, recipe: oracle/Makefile target `ref`). This module re-creates the Lua-side
wrappers of torch/tfluids/init.lua:89-735 on numpy arrays (same names, argument order, defaults,
temp-buffer shapes and the copy-back of in-place results) so tests can call the reference exactly
like the Lua drivers do.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class _RefArg(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("num", ctypes.c_double), ("str", ctypes.c_char_p),
                ("data", ctypes.c_void_p), ("ndim", ctypes.c_int), ("size", ctypes.c_long * 5)]


class RefError(RuntimeError):
    pass


def available(fast=False):
    return os.path.exists(_path(fast))


_JACOBI = None


def jacobi_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libtfluids_ref_jacobi.so"))


def _jacobi_lib():
    global _JACOBI
    if _JACOBI is None:
        _JACOBI = ctypes.CDLL(os.path.join(_HERE, "_ref", "libtfluids_ref_jacobi.so"))
        _JACOBI.tfluids_ref_jacobi.restype = ctypes.c_int
    return _JACOBI


_PCG = None


def pcg_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libtfluids_ref_pcg.so"))


def _pcg_lib():
    global _PCG
    if _PCG is None:
        _PCG = ctypes.CDLL(os.path.join(_HERE, "_ref", "libtfluids_ref_pcg.so"))
        _PCG.tfluids_ref_pcg.restype = ctypes.c_int
    return _PCG


def _path(fast):
    return os.path.join(_HERE, "_ref", "libtfluids_ref_fast.so" if fast else "libtfluids_ref.so")


class RefTfluids:
    """The reference's tfluids ops (float or double instantiation) on numpy arrays."""

    def __init__(self, dtype=np.float32, fast=False):
        self.lib = ctypes.CDLL(_path(fast))
        self.dtype = np.dtype(dtype)
        self._dt = 0 if self.dtype == np.float32 else 1
        self.lib.tfluids_ref_call.restype = ctypes.c_int

    # -- raw call --------------------------------------------------------------------------
    def call(self, op, *args):
        arr = (_RefArg * len(args))()
        keep = []
        for i, a in enumerate(args):
            r = arr[i]
            if isinstance(a, bool):
                r.kind, r.num = 1, float(a)
            elif isinstance(a, (int, float, np.floating, np.integer)):
                r.kind, r.num = 0, float(a)
            elif isinstance(a, str):
                b = a.encode()
                keep.append(b)
                r.kind, r.str = 2, b
            elif isinstance(a, np.ndarray):
                assert a.flags["C_CONTIGUOUS"], "reference needs contiguous tensors"
                if a.dtype == np.int32:
                    r.kind = 4
                else:
                    assert a.dtype == self.dtype, (a.dtype, self.dtype)
                    r.kind = 3
                r.data = a.ctypes.data
                r.ndim = a.ndim
                for d in range(a.ndim):
                    r.size[d] = a.shape[d]
            else:
                raise TypeError(type(a))
        ret = ctypes.c_double(0.0)
        nret = ctypes.c_int(0)
        err = ctypes.create_string_buffer(512)
        rc = self.lib.tfluids_ref_call(op.encode(), self._dt, len(args), arr, ctypes.byref(ret),
                                       ctypes.byref(nret), err, 512)
        if rc == -1:
            raise RefError("unknown reference op " + op)
        if rc != 0:
            raise RefError(err.value.decode())
        return ret.value if nret.value else None

    def _tmp(self, *shapes):
        # init.lua:35-64 getTempStorage: contents undefined on entry -> fill with noise.
        rng = np.random.RandomState(12345)
        return [rng.randn(*s).astype(self.dtype) for s in shapes]

    @staticmethod
    def _dims(flags):
        b, _, d, h, w = flags.shape
        return b, d, h, w

    # -- init.lua wrappers -----------------------------------------------------------------
    def advectScalar(self, dt, s, U, flags, method="maccormackOurs", sDst=None,
                     sampleOutsideFluid=False, maccormackStrength=0.75, boundaryWidth=1):
        b, d, h, w = self._dims(flags)
        C = U.shape[1]
        is3D = C == 3
        fwd, bwd, fwdPos, bwdPos, out = self._tmp((b, 1, d, h, w), (b, 1, d, h, w),
                                                  (b, C, d, h, w), (b, C, d, h, w),
                                                  (b, 1, d, h, w))
        self.call("advectScalar", dt, s, U, flags, fwd, bwd, is3D, method, fwdPos, bwdPos,
                  boundaryWidth, sampleOutsideFluid, maccormackStrength,
                  sDst if sDst is not None else out)
        if sDst is None:
            s[...] = out
        return {"fwd": fwd, "bwd": bwd, "fwdPos": fwdPos, "bwdPos": bwdPos}

    def advectVel(self, dt, U, flags, method="maccormackOurs", UDst=None,
                  maccormackStrength=0.75, boundaryWidth=1):
        is3D = U.shape[1] == 3
        fwd, bwd, out = self._tmp(U.shape, U.shape, U.shape)
        self.call("advectVel", dt, U, flags, fwd, bwd, is3D, method, boundaryWidth,
                  maccormackStrength, UDst if UDst is not None else out)
        if UDst is None:
            U[...] = out
        return {"fwd": fwd, "bwd": bwd}

    def setWallBcsForward(self, U, flags):
        self.call("setWallBcsForward", U, flags, U.shape[1] == 3)

    def velocityDivergenceForward(self, U, flags, UDiv):
        self.call("velocityDivergenceForward", U, flags, UDiv, U.shape[1] == 3)

    def velocityUpdateForward(self, U, flags, p):
        self.call("velocityUpdateForward", U, flags, p, U.shape[1] == 3)

    def vorticityConfinement(self, U, flags, strength):
        b, d, h, w = self._dims(flags)
        C = U.shape[1]
        centered, curl, curlNorm, force = self._tmp((b, C, d, h, w), (b, 3, d, h, w),
                                                    (b, 1, d, h, w), (b, C, d, h, w))
        self.call("vorticityConfinement", U, flags, float(strength), centered, curl, curlNorm,
                  force, C == 3)

    def addBuoyancy(self, U, flags, density, gravity, dt):
        strength = self._tmp((3,))[0]
        self.call("addBuoyancy", U, flags, density, gravity, strength, dt, U.shape[1] == 3)

    def addGravity(self, U, flags, gravity, dt):
        force = self._tmp((3,))[0]
        self.call("addGravity", U, flags, gravity, dt, U.shape[1] == 3, force)

    def velocityDivergenceBackward(self, U, flags, gradOutput, gradU):
        self.call("velocityDivergenceBackward", U, flags, gradOutput, U.shape[1] == 3, gradU)

    def velocityUpdateBackward(self, U, flags, p, gradOutput, gradP):
        # the reference scatters with `omp atomic`: one thread makes the summation order (and the last bit) fixed
        omp = ctypes.CDLL("libgomp.so.1")
        nthreads = omp.omp_get_max_threads()
        omp.omp_set_num_threads(1)
        try:
            self.call("velocityUpdateBackward", U, flags, p, gradOutput, U.shape[1] == 3, gradP)
        finally:
            omp.omp_set_num_threads(nthreads)

    def volumetricUpSamplingNearestForward(self, ratio, inp, out):
        self.call("volumetricUpSamplingNearestForward", int(ratio), inp, out)

    def volumetricUpSamplingNearestBackward(self, ratio, inp, gradOutput, gradInput):
        self.call("volumetricUpSamplingNearestBackward", int(ratio), inp, gradOutput, gradInput)

    def emptyDomain(self, flags, is3D, bnd=1):
        self.call("emptyDomain", flags, bool(is3D), bnd)
        return flags

    def normalizePressureMean(self, p, flags, is3D):
        """init.lua:747-764 (CPU path): inds is an IntTensor temp the size of p."""
        inds = np.zeros(p.shape[:1] + p.shape[2:], dtype=np.int32)
        self.call("normalizePressureMean", p, flags, bool(is3D), inds)

    def flagsToOccupancy(self, flags, occupancy):
        self.call("flagsToOccupancy", flags, occupancy)

    def rectangularBlur(self, src, blurRad, is3D, dst):
        """init.lua:583-595: the wrapper supplies the temp buffer (contents undefined on entry)."""
        tmp = self._tmp(src.shape)[0]
        self.call("rectangularBlur", src, int(blurRad), bool(is3D), dst, tmp)

    def signedDistanceField(self, flags, searchRad, is3D, dst):
        self.call("signedDistanceField", flags, int(searchRad), bool(is3D), dst)

    def solveLinearSystemJacobi(self, p, flags, div, is3D, pTol=1e-5, maxIter=1000, verbose=False):
        """init.lua:693-735. CUDA only in the reference: runs the reference's own kernel and host loop
        (generic/tfluids.cu:1765-1927) compiled for the host (oracle/ref_jacobi.cc, `make ref_jacobi`)."""
        assert self._dt == 0, "the reference's Jacobi solver is float only"
        lib = _jacobi_lib()
        b, d, h, w = self._dims(flags)
        pPrev, pDelta = self._tmp(p.shape, p.shape)
        norm = self._tmp((b,))[0]
        res = ctypes.c_double(0.0)
        err = ctypes.create_string_buffer(512)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = lib.tfluids_ref_jacobi(vp(p), vp(flags), vp(div), vp(pPrev), vp(pDelta), vp(norm), b, d, h, w,
                                    int(bool(is3D)), ctypes.c_float(pTol), int(maxIter), ctypes.byref(res), err, 512)
        if rc != 0:
            raise RefError(err.value.decode())
        return float(res.value)

    def solveLinearSystemPCG(self, p, flags, div, is3D, tol=1e-6, maxIter=1000, precondType="ic0", verbose=False):
        """init.lua:637-691. CUDA + cuSPARSE / cuBLAS only in the reference: runs the reference's own host function
        (generic/tfluids.cu:864-1759: components, reduced indices, setupLaplacian, the CG loop, the mean subtraction)
        compiled for the host (oracle/ref_pcg.cc, `make ref_pcg`) over host restatements of the five cuSPARSE / cuBLAS
        primitives it calls (oracle/ref_shim/cusparse_host.h). Returns the max residual over batches and components."""
        assert self._dt == 0, "the reference's PCG solver is float only"
        lib = _pcg_lib()
        b, d, h, w = self._dims(flags)
        res = ctypes.c_double(0.0)
        err = ctypes.create_string_buffer(512)
        for a in (p, flags, div):
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = lib.tfluids_ref_pcg(vp(p), vp(flags), vp(div), b, d, h, w, int(bool(is3D)), precondType.encode(),
                                 ctypes.c_float(tol), int(maxIter), int(bool(verbose)), ctypes.byref(res), err, 512)
        if rc != 0:
            raise RefError(err.value.decode())
        return float(res.value)

    @staticmethod
    def getDx(flags):
        return 1.0 / max(flags.shape[2], flags.shape[3], flags.shape[4])

    # -- direct line trace (generic/calc_line_trace.cc:313) ---------------------------------
    def calcLineTrace(self, pos, delta, flags3d, is3D=True):
        assert self._dt == 0
        f = np.ascontiguousarray(flags3d, dtype=np.float32)
        zs, ys, xs = f.shape
        p = np.asarray(pos, dtype=np.float32)
        d = np.asarray(delta, dtype=np.float32)
        out = np.zeros(3, dtype=np.float32)
        hit = ctypes.c_int(0)
        err = ctypes.create_string_buffer(512)
        rc = self.lib.tfluids_ref_calcLineTrace(
            p.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p),
            f.ctypes.data_as(ctypes.c_void_p), zs, ys, xs, int(is3D),
            out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(hit), err, 512)
        if rc != 0:
            raise RefError(err.value.decode())
        return out, bool(hit.value)
