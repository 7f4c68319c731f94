"""Adapter giving fluidnet_amd.tfluids (torch tensors on the GPU) the numpy surface of
oracle.oracle.OracleTfluids / oracle.ref.RefTfluids, so one test body drives all three."""
import numpy as np
import torch

from fluidnet_amd import tfluids


class HipTfluids:
    dtype = np.dtype(np.float32)

    def __init__(self, device="cuda:0"):
        self.dev = torch.device(device)

    def _up(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def advectScalar(self, dt, s, U, flags, method="maccormackOurs", sDst=None,
                     sampleOutsideFluid=False, maccormackStrength=0.75, boundaryWidth=1):
        ts = self._up(s)
        td = self._up(sDst) if sDst is not None else None
        tfluids.advectScalar(dt, ts, self._up(U), self._up(flags), method, td, sampleOutsideFluid,
                             maccormackStrength, boundaryWidth)
        if sDst is not None:
            sDst[...] = td.cpu().numpy()
        else:
            s[...] = ts.cpu().numpy()

    def advectVel(self, dt, U, flags, method="maccormackOurs", UDst=None, maccormackStrength=0.75,
                  boundaryWidth=1):
        tu = self._up(U)
        td = self._up(UDst) if UDst is not None else None
        tfluids.advectVel(dt, tu, self._up(flags), method, td, maccormackStrength, boundaryWidth)
        if UDst is not None:
            UDst[...] = td.cpu().numpy()
        else:
            U[...] = tu.cpu().numpy()

    def _inplace_u(self, fn, U, *args):
        tu = self._up(U)
        fn(tu, *args)
        U[...] = tu.cpu().numpy()

    def setWallBcsForward(self, U, flags):
        self._inplace_u(tfluids.setWallBcsForward, U, self._up(flags))

    def velocityDivergenceForward(self, U, flags, UDiv):
        td = self._up(UDiv)
        tfluids.velocityDivergenceForward(self._up(U), self._up(flags), td)
        UDiv[...] = td.cpu().numpy()

    def velocityUpdateForward(self, U, flags, p):
        self._inplace_u(tfluids.velocityUpdateForward, U, self._up(flags), self._up(p))

    def vorticityConfinement(self, U, flags, strength):
        self._inplace_u(tfluids.vorticityConfinement, U, self._up(flags), float(strength))

    def addBuoyancy(self, U, flags, density, gravity, dt):
        self._inplace_u(tfluids.addBuoyancy, U, self._up(flags), self._up(density),
                        [float(g) for g in gravity], float(dt))

    def addGravity(self, U, flags, gravity, dt):
        self._inplace_u(tfluids.addGravity, U, self._up(flags), [float(g) for g in gravity],
                        float(dt))

    def velocityDivergenceBackward(self, U, flags, gradOutput, gradU):
        tg = self._up(gradU)
        tfluids.velocityDivergenceBackward(self._up(U), self._up(flags), self._up(gradOutput), tg)
        gradU[...] = tg.cpu().numpy()

    def velocityUpdateBackward(self, U, flags, p, gradOutput, gradP):
        tg = self._up(gradP)
        tfluids.velocityUpdateBackward(self._up(U), self._up(flags), self._up(p), self._up(gradOutput), tg)
        gradP[...] = tg.cpu().numpy()

    def volumetricUpSamplingNearestForward(self, ratio, inp, out):
        to = self._up(out)
        tfluids.volumetricUpSamplingNearestForward(ratio, self._up(inp), to)
        out[...] = to.cpu().numpy()

    def volumetricUpSamplingNearestBackward(self, ratio, inp, gradOutput, gradInput):
        tg = self._up(gradInput)
        tfluids.volumetricUpSamplingNearestBackward(ratio, self._up(inp), self._up(gradOutput), tg)
        gradInput[...] = tg.cpu().numpy()

    def emptyDomain(self, flags, is3D, bnd=1):
        tf = self._up(flags)
        tfluids.emptyDomain(tf, is3D, bnd)
        flags[...] = tf.cpu().numpy()
        return flags

    def flagsToOccupancy(self, flags, occupancy):
        to = self._up(occupancy)
        tfluids.flagsToOccupancy(self._up(flags), to)
        occupancy[...] = to.cpu().numpy()

    def rectangularBlur(self, src, blurRad, is3D, dst):
        td = self._up(dst)
        tfluids.rectangularBlur(self._up(src), blurRad, is3D, td)
        dst[...] = td.cpu().numpy()

    def signedDistanceField(self, flags, searchRad, is3D, dst):
        td = self._up(dst)
        tfluids.signedDistanceField(self._up(flags), searchRad, is3D, td)
        dst[...] = td.cpu().numpy()

    def solveLinearSystemJacobi(self, p, flags, div, is3D, pTol=1e-5, maxIter=1000, verbose=False):
        tp = self._up(p)
        r = tfluids.solveLinearSystemJacobi(tp, self._up(flags), self._up(div), is3D, pTol, maxIter,
                                            verbose)
        p[...] = tp.cpu().numpy()
        return r

    def normalizePressureMean(self, p, flags, is3D):
        tp = self._up(p)
        tfluids.normalizePressureMean(tp, self._up(flags), is3D)
        p[...] = tp.cpu().numpy()

    def solveLinearSystemPCG(self, p, flags, div, is3D, tol=1e-6, maxIter=1000, precondType="ic0", verbose=False):
        tp = self._up(p)
        r = tfluids.solveLinearSystemPCG(tp, self._up(flags), self._up(div), is3D, tol, maxIter, precondType, verbose)
        p[...] = tp.cpu().numpy()
        return r

    def traceErrors(self):
        return tfluids.traceErrors(torch.empty(1, device=self.dev))
