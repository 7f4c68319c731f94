"""CPU checks of the host logic around the path and of the oracle-side restatement of the Lua
(oracle/simulate_np.py): plume BCs, setConstVals ordering, the std normaliser, and that the shipped
2-D model's weights -- fed the way SURVEY.md section 5 describes -- actually project (divergence
drops), which pins channel order / sign conventions of the restatement."""
import os

import numpy as np
import pytest
import torch

import scenes
from oracle import simulate_np as S

HERE = os.path.dirname(os.path.abspath(__file__))


def _layers2d():
    z = np.load(os.path.join(HERE, "golden", "myModel2D_weights.npz"))
    return [(z["w%d" % i], z["b%d" % i]) for i in range(5)]


@pytest.mark.parametrize("dims,rad", [((1, 64, 64), 0.05), ((1, 128, 128), 0.05), ((16, 16, 16), 0.15),
                                       ((20, 24, 32), 0.15)])
def test_plume_bcs_host_mirror_matches_restatement(dims, rad):
    """fluidnet_amd.simulate.createPlumeBCs (vectorised torch) == the literal 1-based loops of
    lib/simulate.lua:47-123 restated in oracle/simulate_np.py."""
    from fluidnet_amd.simulate import createPlumeBCs
    Z, Y, X = dims
    C = 3 if Z > 1 else 2
    nb = dict(UDiv=np.zeros((1, C, Z, Y, X), np.float32), density=np.zeros((1, 1, Z, Y, X), np.float32))
    S.create_plume_bcs(nb, [0.7], 1.5, rad)
    tb = dict(UDiv=torch.zeros(1, C, Z, Y, X), density=torch.zeros(1, 1, Z, Y, X))
    createPlumeBCs(tb, [0.7], 1.5, rad)
    for k in ("UBC", "UBCInvMask", "densityBC", "densityBCInvMask"):
        assert np.array_equal(nb[k], tb[k].numpy()), k
    assert nb["UBC"].sum() > 0 and tb["pBC"] is None


def test_plume_bcs_multichannel_density():
    from fluidnet_amd.simulate import createPlumeBCs
    nb = dict(UDiv=np.zeros((1, 2, 1, 32, 32), np.float32),
              density=[np.zeros((1, 1, 1, 32, 32), np.float32) for _ in range(3)])
    S.create_plume_bcs(nb, [1.0, 0.5, 0.25], 10, 0.05)
    tb = dict(UDiv=torch.zeros(1, 2, 1, 32, 32), density=[torch.zeros(1, 1, 1, 32, 32) for _ in range(3)])
    createPlumeBCs(tb, [1.0, 0.5, 0.25], 10, 0.05)
    for i in range(3):
        assert np.array_equal(nb["densityBC"][i], tb["densityBC"][i].numpy())
        assert np.array_equal(nb["densityBCInvMask"][i], tb["densityBCInvMask"][i].numpy())


def test_input_scale_is_sample_std():
    rng = np.random.RandomState(0)
    U = rng.randn(2, 3, 5, 6, 7).astype(np.float32)
    s = S.input_scale(U)
    assert np.allclose(s, U.reshape(2, -1).astype(np.float64).std(axis=1, ddof=1), rtol=1e-6)


def test_shipped_2d_model_projects(oracle):
    """One FPROP of data/models/myModel2D's weights through the restated graph must cut ||div||
    (SURVEY.md 8c-2 probe: 24.2 -> 12.4 on a comparable scene); a wrong channel order does not."""
    sc = scenes.make_scene((1, 128, 128), seed=31, vel_cells=0.3, dt=0.1)
    f = sc["flags"]
    U = sc["U"].copy()
    oracle.setWallBcsForward(U, f)
    div0 = np.zeros_like(sc["p"])
    oracle.velocityDivergenceForward(U, f, div0)
    p0 = np.zeros_like(sc["p"])
    p, U1 = S.model_forward(oracle, _layers2d(), p0, U, f)
    div1 = np.zeros_like(div0)
    oracle.velocityDivergenceForward(U1, f, div1)
    assert np.linalg.norm(div1) < 0.7 * np.linalg.norm(div0)
    for _ in range(5):
        p, U1 = S.model_forward(oracle, _layers2d(), p, U1, f)
    oracle.velocityDivergenceForward(U1, f, div1)
    assert np.linalg.norm(div1) < 0.3 * np.linalg.norm(div0)


def test_simulate_restatement_runs_jacobi_2d(oracle):
    """BASELINE config 1 (2-D 64x64 plume, Jacobi 20 iters) on the CPU oracle: finite, plume rises."""
    X = 64
    batch = dict(pDiv=np.zeros((1, 1, 1, X, X), np.float32), UDiv=np.zeros((1, 2, 1, X, X), np.float32),
                 flags=scenes.empty_domain(1, 1, X, X, False), density=np.zeros((1, 1, 1, X, X), np.float32))
    S.create_plume_bcs(batch, [1.0], 10.0, 0.05)
    mconf = dict(dt=4 / 60, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=1.0,
                 gravityScale=0, vorticityConfinementAmp=0, simMethod="jacobi", maxIter=20)
    for _ in range(8):
        S.simulate(oracle, mconf, batch)
    assert np.isfinite(batch["UDiv"]).all() and np.isfinite(batch["density"]).all()
    assert batch["density"][0, 0, 0, 4:, :].sum() > 0   # smoke left the inflow rows


def test_plume_bcs_slab_construction_matches_global_slice():
    from fluidnet_amd.simulate import createPlumeBCs
    Zt, Y, X = 24, 12, 20
    g = dict(UDiv=torch.zeros(1, 3, Zt, Y, X), density=torch.zeros(1, 1, Zt, Y, X))
    createPlumeBCs(g, [1.0], 2.0, 0.15)
    for lo, hi in [(0, 10), (7, 19), (14, 24)]:
        l = dict(UDiv=torch.zeros(1, 3, hi - lo, Y, X), density=torch.zeros(1, 1, hi - lo, Y, X))
        createPlumeBCs(l, [1.0], 2.0, 0.15, zOffset=lo, zTotal=Zt)
        for k in ("UBC", "UBCInvMask", "densityBC", "densityBCInvMask"):
            assert torch.equal(l[k], g[k][:, :, lo:hi]), (k, lo)


def _direct_sum_conv(x, w, b, relu):
    """Independent witness for the oracle's convolutions: cross-correlation, stride 1, zero padding (k-1)/2, written
    as an explicit fp64 sum over taps with numpy slicing only (no torch, no FFT, no im2col)."""
    x = np.asarray(x, np.float64); w = np.asarray(w, np.float64)
    nd = w.ndim - 2
    k = w.shape[-1]; r = (k - 1) // 2
    xp = np.pad(x, [(0, 0), (0, 0)] + [(r, r)] * nd)
    out = np.zeros((x.shape[0], w.shape[0]) + x.shape[2:], np.float64)
    for tap in np.ndindex(*([k] * nd)):
        sl = tuple(slice(t, t + n) for t, n in zip(tap, x.shape[2:]))
        patch = xp[(slice(None), slice(None)) + sl]                       # [B, Cin, ...]
        out += np.tensordot(w[(slice(None), slice(None)) + tap], patch, axes=([1], [1])).swapaxes(0, 1)
    out += np.asarray(b, np.float64).reshape((1, -1) + (1,) * nd)
    return np.maximum(out, 0.0) if relu else out


@pytest.mark.parametrize("is3d", [False, True])
def test_conv_stack_against_fp64_direct_sum(is3d):
    """the ConvNet's arithmetic lives in cuDNN (absent), the oracle uses PyTorch-CPU convolutions;
    a second, independent evaluation (explicit fp64 tap sums in numpy) pins what `conv_stack` computes -- padding,
    cross-correlation orientation, bias, ReLU placement -- so PyTorch is not the only witness. 2-D: the SHIPPED model's
    weights (tests/golden/myModel2D_weights.npz); 3-D: the seeded default topology."""
    rng = np.random.RandomState(3)
    if is3d:
        layers = S.default_3d_layers(seed=5)
        x = rng.randn(2, 3, 9, 12, 14).astype(np.float32)
    else:
        layers = _layers2d()
        x = rng.randn(2, 3, 1, 21, 30).astype(np.float32)
    got32 = S.conv_stack(x, layers, is3d)
    got64 = S.conv_stack(x, layers, is3d, dtype="float64")
    h = x[:, :, 0] if not is3d else x
    for li, (w, b) in enumerate(layers):
        h = _direct_sum_conv(h, w, b, relu=li + 1 < len(layers))
    want = h[:, :, None] if not is3d else h
    assert got64.shape == want.shape
    # conv_stack returns float32 even for the fp64 run: one rounding of the final value
    assert scenes.rel_l2(got64, want) <= 1e-7
    assert scenes.rel_l2(got32, want) <= 2e-6


def test_wall_plan_cache_host_logic(monkeypatch):
    """fluidnet_amd.simulate.wall_plan on the host side alone (a stand-in for the library): nothing is created at the first
    sighting of a flags tensor, the plan at the second; an in-place edit (torch's version counter) retires the old plan at once --
    without a HIP call -- and queues its destruction; a dead tensor does the same from its weak-reference callback; the queue is
    emptied at the next look-up outside a graph capture and never inside one; TFL_WALL_PLAN=0 creates nothing."""
    import gc
    from fluidnet_amd import simulate as SIM, tfluids

    class Lib:
        def __init__(self):
            self.created, self.retired, self.destroyed, self.next = [], [], [], 1000
        def tfl_wall_plan_create(self, ctx, t):
            self.next += 8
            self.created.append(self.next)
            return self.next
        def tfl_wall_plan_retire(self, plan):
            self.retired.append(plan)
        def tfl_wall_plan_destroy(self, ctx, plan):
            self.destroyed.append(plan)

    class Flags:      # what wall_plan asks of a tensor: identity, a version counter, is_cuda
        is_cuda = True
        def __init__(self):
            self._version = 0

    capturing = [False]
    monkeypatch.setattr(tfluids, "_tt", lambda t: t)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing[0])
    monkeypatch.setattr(SIM, "_WALL_PLANS", True)
    lib, ctx, f = Lib(), 77, Flags()
    assert SIM.wall_plan(lib, ctx, f) is None and lib.created == []          # first sighting: remembered only
    p1 = SIM.wall_plan(lib, ctx, f)
    assert p1 == lib.created[0] and SIM.wall_plan(lib, ctx, f) == p1 and len(lib.created) == 1
    f._version += 1                                                          # an in-place edit
    assert SIM.wall_plan(lib, ctx, f) is None                                # ... is a first sighting of the new version
    assert lib.retired == [p1] and lib.destroyed == []                       # out of the registry at once, freed later
    capturing[0] = True
    assert SIM.wall_plan(lib, ctx, f) is None and lib.destroyed == [] and len(lib.created) == 1      # nothing inside a capture
    capturing[0] = False
    p2 = SIM.wall_plan(lib, ctx, f)
    assert p2 == lib.created[1] and lib.destroyed == [p1]                    # the queue was emptied on the way
    del f
    gc.collect()
    assert lib.retired == [p1, p2] and lib.destroyed == [p1]                 # the dead tensor's plan: retired by the callback ...
    g = Flags()
    SIM.wall_plan(lib, ctx, g)
    assert lib.destroyed == [p1, p2]                                         # ... and freed at the next look-up
    monkeypatch.setattr(SIM, "_WALL_PLANS", False)
    h = Flags()
    assert SIM.wall_plan(lib, ctx, h) is None and SIM.wall_plan(lib, ctx, h) is None and len(lib.created) == 2
    SIM.drop_wall_plan(g)
