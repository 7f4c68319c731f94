"""ctypes binding to oracle/libtfluids_oracle.so (tfluids_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Same Python surface as oracle/ref.py::RefTfluids (which mirrors torch/tfluids/init.lua:89-735) so
tests can run either the restatement or the compiled reference through identical code.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtfluids_oracle.so")

METHODS = {"euler": 0, "maccormack": 1, "eulerOurs": 2, "rk2Ours": 3, "rk3Ours": 4,
           "maccormackOurs": 5}  # generic/advect_type.cc:18-37


class OracleError(RuntimeError):
    pass


def available():
    return os.path.exists(_SO)


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleTfluids:
    dtype = np.dtype(np.float32)

    def __init__(self):
        self.lib = ctypes.CDLL(_SO)
        self.lib.ora_solveLinearSystemJacobi.restype = ctypes.c_float
        self.strict = True   # raise when the reference would have THError'd inside a trace

    def _tmp(self, *shapes):
        rng = np.random.RandomState(12345)
        return [rng.randn(*s).astype(np.float32) for s in shapes]

    @staticmethod
    def _dims(flags):
        b, _, d, h, w = flags.shape
        return b, d, h, w

    def advectScalar(self, dt, s, U, flags, method="maccormackOurs", sDst=None,
                     sampleOutsideFluid=False, maccormackStrength=0.75, boundaryWidth=1):
        b, d, h, w = self._dims(flags)
        C = U.shape[1]
        fwd, bwd, fwdPos, bwdPos, out = self._tmp((b, 1, d, h, w), (b, 1, d, h, w),
                                                  (b, C, d, h, w), (b, C, d, h, w),
                                                  (b, 1, d, h, w))
        dst = sDst if sDst is not None else out
        rc = self.lib.ora_advectScalar(
            ctypes.c_float(dt), _p(s), _p(U), _p(flags), _p(fwd), _p(bwd), int(C == 3),
            METHODS[method], _p(fwdPos), _p(bwdPos), int(bool(sampleOutsideFluid)),
            ctypes.c_float(maccormackStrength), _p(dst), b, d, h, w)
        if rc != 0 and self.strict:
            raise OracleError("advectScalar: %d traces hit a reference THError path" % -rc)
        if sDst is None:
            s[...] = out
        return {"fwd": fwd, "bwd": bwd, "fwdPos": fwdPos, "bwdPos": bwdPos}

    def advectVel(self, dt, U, flags, method="maccormackOurs", UDst=None,
                  maccormackStrength=0.75, boundaryWidth=1):
        b, d, h, w = self._dims(flags)
        fwd, bwd, out = self._tmp(U.shape, U.shape, U.shape)
        dst = UDst if UDst is not None else out
        rc = self.lib.ora_advectVel(
            ctypes.c_float(dt), _p(U), _p(flags), _p(fwd), _p(bwd), int(U.shape[1] == 3),
            METHODS[method], ctypes.c_float(maccormackStrength), _p(dst), b, d, h, w)
        if rc != 0 and self.strict:
            raise OracleError("advectVel: %d traces hit a reference THError path" % -rc)
        if UDst is None:
            U[...] = out
        return {"fwd": fwd, "bwd": bwd}

    def setWallBcsForward(self, U, flags):
        b, d, h, w = self._dims(flags)
        self.lib.ora_setWallBcsForward(_p(U), _p(flags), int(U.shape[1] == 3), b, d, h, w)

    def velocityDivergenceForward(self, U, flags, UDiv):
        b, d, h, w = self._dims(flags)
        self.lib.ora_velocityDivergenceForward(_p(U), _p(flags), _p(UDiv),
                                               int(U.shape[1] == 3), b, d, h, w)

    def velocityUpdateForward(self, U, flags, p):
        b, d, h, w = self._dims(flags)
        self.lib.ora_velocityUpdateForward(_p(U), _p(flags), _p(p), int(U.shape[1] == 3),
                                           b, d, h, w)

    def vorticityConfinement(self, U, flags, strength):
        b, d, h, w = self._dims(flags)
        C = U.shape[1]
        centered, curl, curlNorm, force = self._tmp((b, C, d, h, w), (b, 3, d, h, w),
                                                    (b, 1, d, h, w), (b, C, d, h, w))
        self.lib.ora_vorticityConfinement(_p(U), _p(flags), ctypes.c_float(strength),
                                          _p(centered), _p(curl), _p(curlNorm), _p(force),
                                          int(C == 3), b, d, h, w)

    def addBuoyancy(self, U, flags, density, gravity, dt):
        b, d, h, w = self._dims(flags)
        g = np.ascontiguousarray(gravity, dtype=np.float32)
        self.lib.ora_addBuoyancy(_p(U), _p(flags), _p(density), _p(g), ctypes.c_float(dt),
                                 int(U.shape[1] == 3), b, d, h, w)

    def addGravity(self, U, flags, gravity, dt):
        b, d, h, w = self._dims(flags)
        g = np.ascontiguousarray(gravity, dtype=np.float32)
        self.lib.ora_addGravity(_p(U), _p(flags), _p(g), ctypes.c_float(dt),
                                int(U.shape[1] == 3), b, d, h, w)

    def emptyDomain(self, flags, is3D, bnd=1):
        b, d, h, w = self._dims(flags)
        self.lib.ora_emptyDomain(_p(flags), int(bool(is3D)), int(bnd), b, d, h, w)
        return flags

    def flagsToOccupancy(self, flags, occupancy):
        rc = self.lib.ora_flagsToOccupancy(_p(flags), _p(occupancy), ctypes.c_long(flags.size))
        if rc != 0:
            raise OracleError("ERROR: unsupported flag cell found!")

    def rectangularBlur(self, src, blurRad, is3D, dst):
        """init.lua:583-595 (the temp buffer is the wrapper's)."""
        tmp = np.empty_like(src)
        B, C, Z, Y, X = src.shape
        self.lib.ora_rectangularBlur(_p(src), int(blurRad), int(bool(is3D)), _p(dst), _p(tmp), B, C, Z, Y, X)

    def signedDistanceField(self, flags, searchRad, is3D, dst):
        b, d, h, w = self._dims(flags)
        self.lib.ora_signedDistanceField(_p(flags), int(searchRad), _p(dst), b, d, h, w)

    @staticmethod
    def getDx(flags):
        return 1.0 / max(flags.shape[2], flags.shape[3], flags.shape[4])

    def velocityDivergenceBackward(self, U, flags, gradOutput, gradU):
        b, d, h, w = self._dims(flags)
        self.lib.ora_velocityDivergenceBackward(_p(flags), _p(gradOutput), _p(gradU), int(U.shape[1] == 3), b, d, h, w)

    def velocityUpdateBackward(self, U, flags, p, gradOutput, gradP):
        b, d, h, w = self._dims(flags)
        self.lib.ora_velocityUpdateBackward(_p(flags), _p(gradOutput), _p(gradP), int(U.shape[1] == 3), b, d, h, w)

    def volumetricUpSamplingNearestForward(self, ratio, inp, out):
        b, f, d, h, w = inp.shape
        self.lib.ora_volumetricUpSamplingNearestForward(int(ratio), _p(inp), _p(out), ctypes.c_long(b * f), d, h, w)

    def volumetricUpSamplingNearestBackward(self, ratio, inp, gradOutput, gradInput):
        b, f, d, h, w = inp.shape
        self.lib.ora_volumetricUpSamplingNearestBackward(int(ratio), _p(gradOutput), _p(gradInput),
                                                         ctypes.c_long(b * f), d, h, w)

    def solveLinearSystemJacobi(self, p, flags, div, is3D, pTol=1e-5, maxIter=1000,
                                verbose=False):
        b, d, h, w = self._dims(flags)
        if maxIter < 1:
            raise OracleError("At least 1 iteration is needed (maxIter < 1)")
        prev = self._tmp(p.shape)[0]
        return float(self.lib.ora_solveLinearSystemJacobi(
            _p(p), _p(flags), _p(div), _p(prev), int(bool(is3D)), ctypes.c_float(pTol),
            int(maxIter), b, d, h, w))

    def normalizePressureMean(self, p, flags, is3D):
        b, d, h, w = self._dims(flags)
        self.lib.ora_normalizePressureMean(_p(p), _p(flags), int(bool(is3D)), b, d, h, w)

    def solveLinearSystemPCG(self, p, flags, div, is3D, tol=1e-6, maxIter=1000, precondType="ic0", verbose=False):
        """init.lua:645-677; restated from the CUDA-only generic/tfluids.cu:864-1759 (see tfluids_oracle.c)."""
        b, d, h, w = self._dims(flags)
        pc = {"none": 0, "ilu0": 1, "ic0": 2}.get(precondType)
        if pc is None:
            raise OracleError("precondType is not supported.")
        res = ctypes.c_float(0.0)
        rc = self.lib.ora_solveLinearSystemPCG(_p(p), _p(flags), _p(div), int(bool(is3D)), pc, ctypes.c_float(tol),
                                               int(maxIter), b, d, h, w, ctypes.byref(res))
        if rc == -1:
            raise OracleError("Non fluid cell found in a connected component or fluid cell found on the domain border")
        if rc == -2:
            raise OracleError("ERROR: r_norm_sq1 is nan!")
        return float(res.value)

    def calcLineTrace(self, pos, delta, flags3d, is3D=True):
        f = np.ascontiguousarray(flags3d, dtype=np.float32)
        zs, ys, xs = f.shape
        p = np.asarray(pos, dtype=np.float32)
        dl = np.asarray(delta, dtype=np.float32)
        out = np.zeros(3, dtype=np.float32)
        rc = self.lib.ora_calcLineTrace(_p(p), _p(dl), _p(f), zs, ys, xs, int(is3D), _p(out))
        if rc < 0:
            raise OracleError("calcLineTrace: reference THError path %d" % rc)
        return out, bool(rc)
