"""The pressure-projection ConvNet: host mirror of torch/lib/model.lua's `default` model.

`FluidNetModel.forward({pDiv, UDiv, flags}) -> {p, U}` has the call shape of the reference's
`model:forward(torch.getModelInput(batch))` (lib/model.lua:398, 421-450; lib/simulate.lua:262-272).
The graph (SetWallBcs -> divergence -> std-normalise -> conv stack -> VelocityUpdate -> un-scale ->
SetWallBcs) runs inside libtfluids_hip.so (tfl_model_forward); this class only owns the weights'
host copy, the device handle and the scratch tensor.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib, tfluids, torch7
from ._lib import TfluidsError


# the mconf switches that change the forward graph, with torch/lib/default_conf.lua:60-100's defaults
DEFAULT_OPTS = dict(inputChannels=dict(pDiv=True, UDiv=False, div=True, flags=True), normalizeInput=True,
                    normalizeInputChan="UDiv", normalizeInputFunc="std", nonlinType="relu", addPressureSkip=False)


def _resolve_opts(opts):
    o = {k: (dict(v) if isinstance(v, dict) else v) for k, v in DEFAULT_OPTS.items()}
    for k, v in (opts or {}).items():
        if k == "inputChannels":
            o[k].update(v)
        elif k in o:
            o[k] = v
        else:
            raise TfluidsError("unknown model option %r" % (k,))
    if not o["inputChannels"]["flags"]:
        raise TfluidsError("Are you sure you dont want flags?")                       # lib/model.lua:81
    for key, allowed in (("normalizeInputChan", ("UDiv", "pDiv", "div")), ("normalizeInputFunc", ("std", "norm")),
                         ("nonlinType", ("relu", "relu6", "sigmoid"))):
        if o[key] not in allowed:
            raise TfluidsError("bad %s %r (one of %s)" % (key, o[key], ", ".join(allowed)))
    return o


class FluidNetModel:
    def __init__(self, layers, is3D, pool=None, up=None, opts=None):
        """layers: [(weight[nOut, nIn, k(,k),k], bias[nOut])] in forward order (numpy float32),
        weight layout as cudnn.{Spatial,Volumetric}Convolution.weight. pool / up: per-layer psize / usize of
        lib/model.lua's layer tables (1 or 2; None = all 1): 2x average pooling after a layer, or the layer is an
        nn.{Spatial,Volumetric}ConvolutionUpsample (its weight then has nOut * 2^dim output channels).
        opts: mconf fields that change the forward graph (lib/model.lua:27-160, 356-387), defaults as
        default_conf.lua: inputChannels={pDiv,UDiv,div,flags}, normalizeInput, normalizeInputChan ('UDiv'|'pDiv'|'div'),
        normalizeInputFunc ('std'|'norm'), nonlinType ('relu'|'relu6'|'sigmoid'), addPressureSkip."""
        self.opts = _resolve_opts(opts)
        self.is3D = bool(is3D)
        self.layers = [(np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32))
                       for w, b in layers]
        n = len(self.layers)
        self.pool = [1] * n if pool is None else [int(v) for v in pool]
        self.up = [1] * n if up is None else [int(v) for v in up]
        if len(self.pool) != n or len(self.up) != n:
            raise TfluidsError("pool / up need one entry per layer")
        for w, _ in self.layers:
            if w.ndim != (5 if self.is3D else 4):
                raise TfluidsError("weight rank does not match is3D")
        self._handles = {}   # device index -> tfl_model*
        self._work = {}      # device index -> scratch tensor

    # -- constructors ------------------------------------------------------------------------
    @classmethod
    def from_torch7(cls, path):
        """Load a model saved by torch.saveModel (lib/model.lua:463-478), e.g. data/models/myModel2D."""
        layers = torch7.conv_layers(torch7.load(path))
        if not layers:
            raise TfluidsError("no convolution layers found in " + path)
        return cls(layers, is3D=layers[0][0].ndim == 5)

    @classmethod
    def from_npz(cls, path):
        z = np.load(path)
        n = len([k for k in z.files if k.startswith("w")])
        layers = [(z["w%d" % i], z["b%d" % i]) for i in range(n)]
        return cls(layers, is3D=layers[0][0].ndim == 5)

    @classmethod
    def default_3d(cls, seed=1, scale=0.35):
        """Seeded stand-in for a trained 3-D `default` model (the reference ships none): topology of
        lib/model.lua:219-226 (3->8 k3, 8->8 k3, 8->8 k3, 8->8 k1, 8->1 k1)."""
        rng = np.random.RandomState(seed)
        layers = []
        for co, ci, k in [(8, 3, 3), (8, 8, 3), (8, 8, 3), (8, 8, 1), (1, 8, 1)]:
            fan = ci * k ** 3
            w = (rng.randn(co, ci, k, k, k) * scale * math.sqrt(2.0 / fan)).astype(np.float32)
            b = (rng.randn(co) * 0.01).astype(np.float32)
            layers.append((w, b))
        return cls(layers, True)

    @classmethod
    def tog(cls, is3D, seed=1, scale=0.5):
        """Seeded stand-in for a trained `tog` model (none is shipped), single bank: the layer tables of
        lib/model.lua:163-178 (2-D) / :211-218 (3-D) -- pooling after the first layer(s), ConvolutionUpsample at the end."""
        if is3D:
            osize, ksize = [16, 16, 16, 16, 32, 32, 1], [3, 3, 3, 3, 1, 1, 3]
            psize, usize = [2, 2, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 2, 2]
        else:
            osize, ksize = [16, 32, 32, 64, 64, 32, 1], [5, 5, 5, 5, 1, 1, 3]
            psize, usize = [2, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 2]
        dim = 3 if is3D else 2
        rng = np.random.RandomState(seed)
        layers, ci = [], 3
        for co, k, u in zip(osize, ksize, usize):
            fan = ci * k ** dim
            w = (rng.randn(co * u ** dim, ci, *([k] * dim)) * scale * math.sqrt(2.0 / fan)).astype(np.float32)
            b = (rng.randn(co * u ** dim) * 0.01).astype(np.float32)
            layers.append((w, b))
            ci = co
        return cls(layers, is3D, pool=psize, up=usize)

    # -- device handle -----------------------------------------------------------------------
    def _handle(self, lib, ctx, dev):
        h = self._handles.get(dev)
        if h is None:
            n = len(self.layers)
            I32 = ctypes.c_int32 * n
            dim = 3 if self.is3D else 2
            cin = I32(*[w.shape[1] for w, _ in self.layers])
            cout = I32(*[w.shape[0] // (u ** dim) for (w, _), u in zip(self.layers, self.up)])
            ks = I32(*[w.shape[-1] for w, _ in self.layers])
            FP = ctypes.POINTER(ctypes.c_float)
            ws = (FP * n)(*[w.ctypes.data_as(FP) for w, _ in self.layers])
            bs = (FP * n)(*[b.ctypes.data_as(FP) for _, b in self.layers])
            o, ic = self.opts, self.opts["inputChannels"]
            copts = _lib.tfl_model_opts(int(ic["pDiv"]), int(ic["UDiv"]), int(ic["div"]), int(o["normalizeInput"]),
                                        ("UDiv", "pDiv", "div").index(o["normalizeInputChan"]),
                                        ("std", "norm").index(o["normalizeInputFunc"]),
                                        ("relu", "relu6", "sigmoid").index(o["nonlinType"]), int(o["addPressureSkip"]))
            h = lib.tfl_model_create_opts(ctx, int(self.is3D), n, cin, cout, ks, I32(*self.pool), I32(*self.up), ws, bs,
                                          ctypes.byref(copts))
            if not h:
                raise TfluidsError(lib.tfl_last_error(ctx).decode())
            self._handles[dev] = h
        return h

    def forward(self, inputs, out=None, UBC=None, UBCInvMask=None, clamp=None):
        """inputs = [pDiv, UDiv, flags] -> [p, U]. `out=[p, U]` writes into given tensors (they may
        be the inputs themselves: simulate() copies the prediction back into the state anyway).
        UBC/UBCInvMask/clamp fuse simulate()'s trailing setConstVals + clamp into the last kernel."""
        pDiv, UDiv, flags = inputs
        tfluids._dims(UDiv, flags)
        tfluids._check(pDiv.shape == flags.shape and pDiv.is_contiguous(), "Size mismatch")
        tfluids._check((UDiv.size(1) == 3) == self.is3D, "model / input dimensionality mismatch")
        lib, ctx = tfluids._context(UDiv)
        dev = UDiv.device.index
        h = self._handle(lib, ctx, dev)
        B, _, Z, Y, X = flags.shape
        need = lib.tfl_model_workspace_floats(h, B, Z, Y, X)
        work = self._work.get(dev)
        if work is None or work.numel() < need:
            work = torch.empty(need, dtype=torch.float32, device=UDiv.device)
            self._work[dev] = work
        if out is None:
            p, U = torch.empty_like(pDiv), torch.empty_like(UDiv)
        else:
            p, U = out
        lo, hi = clamp if clamp is not None else (0.0, 0.0)
        from .simulate import wall_plan
        wall_plan(lib, ctx, flags)      # (the context finds it again by the flags' address: include/tfluids_hip.h tfl_wall_plan)
        rc = lib.tfl_model_forward(ctx, h, tfluids._tt(pDiv), tfluids._tt(UDiv), tfluids._tt(flags),
                                   tfluids._tt(p), tfluids._tt(U), ctypes.c_void_p(work.data_ptr()),
                                   work.numel(), tfluids._tt(UBC) if UBC is not None else None,
                                   tfluids._tt(UBCInvMask) if UBCInvMask is not None else None,
                                   int(clamp is not None), lo, hi)
        tfluids._call(lib, ctx, rc)
        return [p, U]

    __call__ = forward

    def range_errors(self, like):
        """Thread blocks of the fp16-MFMA convolution path that clamped an activation at the fp16 range since the last
        call (tfl_model_range_errors; synchronises). 0 for any working simulation; always 0 on the fp32 paths."""
        lib, ctx = tfluids._context(like)
        h = self._handles.get(like.device.index)
        return 0 if h is None else int(lib.tfl_model_range_errors(ctx, h))

    def range_flag(self, like):
        """The same count as far as the device has reported it, WITHOUT a stream synchronisation and without resetting it
        (tfl_model_range_flag): non-zero means a forward pass clamped activations at the fp16 range, and the next
        forward / simulate step of this model is refused (TFL_ERANGE) until range_errors() has been called. The strict-fp32
        stack without a range limit is TFL_CONV_PATH=winograd at model creation."""
        lib, ctx = tfluids._context(like)
        h = self._handles.get(like.device.index)
        return 0 if h is None else int(lib.tfl_model_range_flag(ctx, h))

    # -- the same forward in two halves, for z-slab decomposition (fluidnet_amd.dist) --------------
    def _prep(self, flags):
        lib, ctx = tfluids._context(flags)
        dev = flags.device.index
        h = self._handle(lib, ctx, dev)
        B, _, Z, Y, X = flags.shape
        need = lib.tfl_model_workspace_floats(h, B, Z, Y, X)
        work = self._work.get(dev)
        if work is None or work.numel() < need:
            work = torch.empty(need, dtype=torch.float32, device=flags.device)
            self._work[dev] = work
        return lib, ctx, h, work

    def begin(self, U, flags, zlo, zhi, stats):
        """SetWallBcs(U) in place + divergence + {sum u, sum u^2} over z-planes [zlo, zhi) into
        `stats` (float64 tensor [B, 2] on the device) -- the caller all-reduces it across ranks."""
        tfluids._dims(U, flags)
        tfluids._check(stats.dtype == torch.float64 and stats.is_contiguous() and stats.numel() >= 2 * U.size(0),
                       "stats must be a contiguous float64 [B, 2] tensor")
        lib, ctx, h, work = self._prep(flags)
        from .simulate import wall_plan
        wall_plan(lib, ctx, flags)
        tfluids._call(lib, ctx, lib.tfl_model_begin(ctx, h, tfluids._tt(U), tfluids._tt(flags), tfluids._tt(U),
                                                    ctypes.c_void_p(work.data_ptr()), work.numel(), int(zlo),
                                                    int(zhi), ctypes.c_void_p(stats.data_ptr())))

    def finish(self, p, U, flags, stats, count, UBC=None, UBCInvMask=None, clamp=None):
        """Everything after the (all-reduced) statistics: net input, conv stack, velocity update,
        un-scale, wall BCs (+ the fused setConstVals/clamp tail); p and U are updated in place."""
        lib, ctx, h, work = self._prep(flags)
        lo, hi = clamp if clamp is not None else (0.0, 0.0)
        tfluids._call(lib, ctx, lib.tfl_model_finish(
            ctx, h, tfluids._tt(p), tfluids._tt(flags), tfluids._tt(p), tfluids._tt(U),
            ctypes.c_void_p(work.data_ptr()), work.numel(), ctypes.c_void_p(stats.data_ptr()), float(count),
            tfluids._tt(UBC) if UBC is not None else None,
            tfluids._tt(UBCInvMask) if UBCInvMask is not None else None, int(clamp is not None), lo, hi))
