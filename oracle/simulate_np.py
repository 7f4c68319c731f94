"""CPU restatement of the Lua side of the hot path -- TEST INFRASTRUCTURE ONLY.

  create_plume_bcs   torch/lib/simulate.lua:47-123   (tfluids.createPlumeBCs)
  set_const_vals     torch/lib/simulate.lua:130-160  (setConstVals)
  model_forward      torch/lib/model.lua:27-401 graph of the `default` model:
                     SetWallBcs -> VelocityDivergence -> std-normalise -> {pDiv, div, occupancy}
                     -> conv stack (+ReLU) -> VelocityUpdate -> un-scale -> SetWallBcs
  simulate           torch/lib/simulate.lua:175-327  (tfluids.simulate)

The tfluids operators come from an `ops` object: oracle.oracle.OracleTfluids (restatement) or
oracle.ref.RefTfluids (the compiled reference). Convolutions are PyTorch-CPU conv2d/conv3d
(cross-correlation, zero padding (k-1)/2, like cudnn.Spatial/VolumetricConvolution as used by
lib/model_utils.lua:80-116). PARITY UNPINNED at that boundary: cuDNN / cudnn.torch are not in the
reference tree and no reference test pins conv outputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

import numpy as np


def create_plume_bcs(batch, density_val, u_scale, rad):
    """simulate.lua:47-123. 1-based loops restated 0-based; single scalar density grid or list."""
    U = batch["UDiv"]
    assert U.ndim == 5 and U.shape[0] == 1, "Only single batch allowed."
    batch["pBC"] = None
    batch["pBCInvMask"] = None
    batch["UBC"] = np.zeros_like(U)
    batch["UBCInvMask"] = np.ones_like(U)
    dens = batch["density"]
    multi = isinstance(dens, (list, tuple))
    chans = list(dens) if multi else [dens]
    assert len(density_val) == len(chans)
    dbc = [np.zeros_like(c) for c in chans]
    dmask = [np.ones_like(c) for c in chans]
    _, C, zdim, ydim, xdim = U.shape
    is3d = C == 3
    if not is3d:
        assert zdim == 1
    cx = xdim // 2
    cz = max(zdim // 2, 1)
    prad = int(math.floor(xdim * rad))
    for z in range(1, zdim + 1):
        for y in range(1, 5):
            for x in range(1, xdim + 1):
                dx, dz = cx - x, cz - z
                if dx * dx + dz * dz <= prad * prad:
                    batch["UBC"][0, :, z - 1, y - 1, x - 1] = 0.0
                    batch["UBC"][0, 1, z - 1, y - 1, x - 1] = 1.0 * u_scale
                    batch["UBCInvMask"][0, :, z - 1, y - 1, x - 1] = 0.0
                    for i in range(len(chans)):
                        dbc[i][0, :, z - 1, y - 1, x - 1] = density_val[i]
                        dmask[i][0, :, z - 1, y - 1, x - 1] = 0.0
                else:
                    batch["UBC"][0, :, z - 1, y - 1, x - 1] = 0.0
                    batch["UBCInvMask"][0, :, z - 1, y - 1, x - 1] = 0.0
    batch["densityBC"] = dbc if multi else dbc[0]
    batch["densityBCInvMask"] = dmask if multi else dmask[0]


def set_const_vals(batch, p, U, flags, density):
    """simulate.lua:130-160 (cmul then add, fp32)."""
    if batch.get("pBC") is not None or batch.get("pBCInvMask") is not None:
        p *= batch["pBCInvMask"]
        p += batch["pBC"]
    if batch.get("UBC") is not None or batch.get("UBCInvMask") is not None:
        U *= batch["UBCInvMask"]
        U += batch["UBC"]
    if batch.get("densityBC") is not None or batch.get("densityBCInvMask") is not None:
        if isinstance(density, (list, tuple)):
            for i in range(len(density)):
                density[i] *= batch["densityBCInvMask"][i]
                density[i] += batch["densityBC"][i]
        else:
            density *= batch["densityBCInvMask"]
            density += batch["densityBC"]


def input_scale(U_bc):
    """model.lua:93-117 + lib/modules/variance.lua:44-76: per-sample std with n-1, i.e.
    sqrt((n*sum(x^2) - sum(x)^2) / (n*(n-1))); the Clamp is a no-op (threshold typo, model.lua:106).
    Sums are accumulated in float64 (THC's float reductions are order-dependent; fp64 is the
    order-free statement of the same quantity) and the scale is rounded to float32."""
    B = U_bc.shape[0]
    x = U_bc.reshape(B, -1).astype(np.float64)
    n = x.shape[1]
    var = (n * (x * x).sum(1) - x.sum(1) ** 2) / (n * (n - 1.0))
    return np.sqrt(var).astype(np.float32)


def conv_stack(x, layers, is3d, dtype="float32", pool=None, up=None, nonlin="relu", skip=None):
    """x: [B, C, Z, Y, X] float32; layers: [(w[nOut, nIn, k..], b)], the non-linearity (model_utils.lua:20-34: relu |
    relu6 | sigmoid) after all but the last. pool / up: the per-layer psize / usize of lib/model.lua's `tog` tables
    (:163-178, :211-218). skip: addPressureSkip (model.lua:356-360) -- a [B, 1, Z, Y, X] field joined to the input of
    the LAST layer as its last channel."""
    import torch
    import torch.nn.functional as F
    td = getattr(torch, dtype)
    h = torch.from_numpy(np.ascontiguousarray(x)).to(td)
    if not is3d:
        h = h[:, :, 0]
    for li, (w, b) in enumerate(layers):
        wt, bt = torch.from_numpy(np.asarray(w)).to(td), torch.from_numpy(np.asarray(b)).to(td)
        if skip is not None and li + 1 == len(layers):
            sk = torch.from_numpy(np.ascontiguousarray(skip)).to(td)
            h = torch.cat([h, sk if is3d else sk[:, :, 0]], dim=1)      # nn.JoinTable(2)({hl, pDiv})
        pad = (w.shape[-1] - 1) // 2
        h = F.conv3d(h, wt, bt, padding=pad) if is3d else F.conv2d(h, wt, bt, padding=pad)
        u = 1 if up is None else up[li]
        if u > 1:   # nn.{Spatial,Volumetric}ConvolutionUpsample: view (b, nO, s.., dims..) -> interleave (pixel shuffle)
            bsz = h.shape[0]
            if is3d:
                no, (t, hh, ww) = h.shape[1] // u ** 3, h.shape[2:]
                h = h.view(bsz, no, u, u, u, t, hh, ww).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(bsz, no, t * u, hh * u, ww * u)
            else:
                no, (hh, ww) = h.shape[1] // u ** 2, h.shape[2:]
                h = h.view(bsz, no, u, u, hh, ww).permute(0, 1, 4, 2, 5, 3).reshape(bsz, no, hh * u, ww * u)
        if li + 1 < len(layers):
            h = {"relu": torch.relu, "relu6": lambda t: torch.clamp(t, 0.0, 6.0), "sigmoid": torch.sigmoid}[nonlin](h)
        if pool is not None and pool[li] > 1:   # cudnn average pooling, window = stride (model_utils.lua:184-208)
            h = F.avg_pool3d(h, pool[li]) if is3d else F.avg_pool2d(h, pool[li])
    if not is3d:
        h = h.unsqueeze(2)
    return h.to(torch.float32).numpy()


DEFAULT_MODEL_OPTS = dict(inputChannels=dict(pDiv=True, UDiv=False, div=True, flags=True), normalizeInput=True,
                          normalizeInputChan="UDiv", normalizeInputFunc="std", nonlinType="relu", addPressureSkip=False)


def model_opts(opts=None):
    """default_conf.lua:60-100's forward-graph switches, overridden by `opts` (same field names)."""
    o = {k: (dict(v) if isinstance(v, dict) else v) for k, v in DEFAULT_MODEL_OPTS.items()}
    for k, v in (opts or {}).items():
        if k == "inputChannels":
            o[k].update(v)
        elif k in o:
            o[k] = v
        else:
            raise KeyError("unknown model option %r" % k)
    return o


def model_forward(ops, layers, pDiv, UDiv, flags, conv_dtype="float32", pool=None, up=None, opts=None):
    """Model FPROP, lib/model.lua:27-160 + 356-390; returns (p, U) and leaves the inputs untouched. opts: the mconf
    switches of model_opts()."""
    o = model_opts(opts)
    ic = o["inputChannels"]
    assert ic["flags"], "Are you sure you dont want flags?"                      # model.lua:81
    assert ic["div"] or ic["pDiv"] or ic["UDiv"], "Are you sure you dont want any (U, div or p) fields?"
    is3d = UDiv.shape[1] == 3
    U_bc = UDiv.copy()
    ops.setWallBcsForward(U_bc, flags)                       # tfluids/set_wall_bcs.lua:29-48
    div = np.zeros_like(pDiv)
    ops.velocityDivergenceForward(U_bc, flags, div)          # tfluids/velocity_divergence.lua:28-37
    if o["normalizeInput"]:                                  # model.lua:93-128
        src = {"UDiv": U_bc, "pDiv": pDiv, "div": div}[o["normalizeInputChan"]]
        if o["normalizeInputFunc"] == "std":
            scale = input_scale(src)                         # nn.StandardDeviation (n-1); the Clamp is a no-op (typo)
        elif o["normalizeInputFunc"] == "norm":
            x2 = src.reshape(src.shape[0], -1).astype(np.float64)
            scale = np.sqrt((x2 * x2).sum(1)).astype(np.float32)     # Power(2) -> Sum -> Sqrt
        else:
            raise ValueError("Incorrect normalize input function")
    else:
        scale = np.ones(pDiv.shape[0], np.float32)
    sc = scale.reshape(-1, 1, 1, 1, 1)
    occ = np.zeros_like(pDiv)
    ops.flagsToOccupancy(flags, occ)
    chans = []                                               # model.lua:130-148: pDiv, UDiv, div, occupancy
    if ic["pDiv"]:
        chans.append(pDiv / sc)
    if ic["UDiv"]:
        chans.append(U_bc / sc)
    if ic["div"]:
        chans.append(div / sc)
    chans.append(occ)
    x = np.concatenate(chans, axis=1).astype(np.float32)     # apply_scale.lua (CDivTable), JoinTable
    skip = (pDiv / sc).astype(np.float32) if o["addPressureSkip"] else None
    p_pred = conv_stack(x, layers, is3d, conv_dtype, pool, up, nonlin=o["nonlinType"], skip=skip)
    U = (U_bc / sc).astype(np.float32)
    ops.velocityUpdateForward(U, flags, np.ascontiguousarray(p_pred))  # velocity_update.lua:29-39
    p = (p_pred * sc).astype(np.float32)                     # model.lua:383-387
    U = (U * sc).astype(np.float32)
    ops.setWallBcsForward(U, flags)                          # model.lua:390
    return p, U


def simulate(ops, mconf, batch, layers=None, output_div=False, conv_dtype="float32"):
    """tfluids.simulate, simulate.lua:175-327. State lives in batch['pDiv'], ['UDiv'], ['flags'],
    ['density'] (numpy, modified in place). mconf keys: dt, advectionMethod, maccormackStrength,
    buoyancyScale, gravityScale, vorticityConfinementAmp, simMethod, maxIter, gravity (optional)."""
    p, U, flags, density = batch["pDiv"], batch["UDiv"], batch["flags"], batch.get("density")
    is3d = U.shape[1] == 3
    dt = mconf["dt"]
    if density is not None:
        for chan in (density if isinstance(density, (list, tuple)) else [density]):
            ops.advectScalar(dt, chan, U, flags, mconf["advectionMethod"], None, False,
                             mconf["maccormackStrength"])
    ops.advectVel(dt, U, flags, mconf["advectionMethod"], None, mconf["maccormackStrength"])
    set_const_vals(batch, p, U, flags, density)

    def gravity():
        g = mconf.get("gravity")
        return np.array([0, 1, 0], np.float32) if g is None else np.array(g, np.float32)

    dx = np.float32(ops.getDx(flags))
    if density is not None and mconf.get("buoyancyScale", 0) > 0:
        # gravity:mul(-(getDx / 4) * buoyancyScale): Lua doubles, stored into a float tensor
        g = (gravity() * np.float32(-(float(ops.getDx(flags)) / 4) * mconf["buoyancyScale"])).astype(np.float32)
        d0 = density[0] if isinstance(density, (list, tuple)) else density
        ops.addBuoyancy(U, flags, d0, g, dt)
    if mconf.get("gravityScale", 0) > 0:
        g = (gravity() * np.float32((-float(ops.getDx(flags)) / 4) * mconf["gravityScale"])).astype(np.float32)
        ops.addGravity(U, flags, g, dt)
    if mconf.get("vorticityConfinementAmp", 0) > 0:
        ops.vorticityConfinement(U, flags, float(ops.getDx(flags)) * mconf["vorticityConfinementAmp"])
    del dx
    if output_div:
        return
    sim = mconf.get("simMethod") or "convnet"
    if sim != "convnet":
        ops.setWallBcsForward(U, flags)
    set_const_vals(batch, p, U, flags, density)
    if sim == "convnet":
        p_pred, U_pred = model_forward(ops, layers, p, U, flags, conv_dtype)
        p[...] = p_pred
        U[...] = U_pred
    elif sim == "jacobi":
        div = np.zeros_like(p)
        ops.velocityDivergenceForward(U, flags, div)
        ops.solveLinearSystemJacobi(p, flags, div, is3d, 0.0, mconf.get("maxIter") or 100)
        ops.velocityUpdateForward(U, flags, p)
    elif sim == "pcg":      # simulate.lua:281-286: tol 1e-4, ic0
        div = np.zeros_like(p)
        ops.velocityDivergenceForward(U, flags, div)
        ops.solveLinearSystemPCG(p, flags, div, is3d, 1e-4, mconf.get("maxIter") or 100, "ic0")
        ops.velocityUpdateForward(U, flags, p)
    else:
        raise ValueError("mconf.simMethod (%s) is not a valid option" % sim)
    set_const_vals(batch, p, U, flags, density)
    np.clip(U, -1e6, 1e6, out=U)


def default_3d_layers(seed=1, scale=0.35):
    """Seeded stand-in for a trained 3-D `default` model (none is shipped): topology of
    model.lua:219-226 (3->8 k3, 8->8 k3, 8->8 k3, 8->8 k1, 8->1 k1), He-style scaled weights."""
    rng = np.random.RandomState(seed)
    shapes = [(8, 3, 3), (8, 8, 3), (8, 8, 3), (8, 8, 1), (1, 8, 1)]
    layers = []
    for co, ci, k in shapes:
        fan = ci * k ** 3
        w = (rng.randn(co, ci, k, k, k) * scale * math.sqrt(2.0 / fan)).astype(np.float32)
        b = (rng.randn(co) * 0.01).astype(np.float32)
        layers.append((w, b))
    return layers
