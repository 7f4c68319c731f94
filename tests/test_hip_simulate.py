"""GPU parity of the assembled path: ConvNet projection (tfl_model_forward) and whole simulate()
steps for the BASELINE.json configs at oracle-sized grids, against oracle/simulate_np.py driving the
C oracle + PyTorch-CPU convolutions.

Tolerance: rel-L2 <= 1e-5 (BASELINE.json north_star). The tfluids operators themselves are bit-exact
(test_hip_parity.py); the convolutions sum in a different order than PyTorch-CPU (both exact-fp32
fma chains), which is an O(1e-7) relative effect, so multi-step ConvNet runs are compared with the
north-star tolerance while Jacobi runs must stay bit-exact.
"""
import os

import numpy as np
import pytest

import scenes
from oracle import simulate_np as S

from flavours import child_env, experiments_flavour  # noqa: E402

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-5


def _layers2d():
    z = np.load(os.path.join(HERE, "golden", "myModel2D_weights.npz"))
    return [(z["w%d" % i], z["b%d" % i]) for i in range(5)]


def _to_dev(batch, dev):
    import torch
    out = {}
    for k, v in batch.items():
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(v.copy()).to(dev)
        elif isinstance(v, list):
            out[k] = [torch.from_numpy(a.copy()).to(dev) for a in v]
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("dims,seed,which", [((1, 128, 128), 41, "2d"), ((1, 70, 90), 42, "2d"),
                                              ((32, 32, 32), 43, "3d"), ((20, 36, 68), 44, "3d")])
def test_model_forward_matches_restatement(oracle, dims, seed, which):
    import torch
    from fluidnet_amd import FluidNetModel
    layers = _layers2d() if which == "2d" else S.default_3d_layers(seed=3)
    model = FluidNetModel(layers, is3D=which == "3d")
    sc = scenes.make_scene(dims, seed=seed, vel_cells=0.4)
    p_ref, U_ref = S.model_forward(oracle, layers, sc["p"], sc["U"], sc["flags"])
    p64, U64 = S.model_forward(oracle, layers, sc["p"], sc["U"], sc["flags"], conv_dtype="float64")
    dev = torch.device("cuda:0")
    tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
    p, U = model.forward([tp, tU, tf])
    assert torch.equal(tp.cpu(), torch.from_numpy(sc["p"])) and torch.equal(tU.cpu(), torch.from_numpy(sc["U"]))
    rp, rU = scenes.rel_l2(p.cpu().numpy(), p_ref), scenes.rel_l2(U.cpu().numpy(), U_ref)
    assert rp <= TOL and rU <= TOL, (rp, rU)
    # error bars against the fp64-conv answer. 3-D (the default path: fp32 operands as fp16 hi/lo pairs on the matrix cores,
    # conv_mfma16.hip): at least as close as PyTorch-CPU's fp32 convolution is -- the claim DESIGN 3.3b makes for the `f32`
    # label (the test allowed 4x); 2-D (fp32-operand MFMA, another summation order only): within 4x of it
    e_ours, e_torch = scenes.rel_l2(p.cpu().numpy(), p64), scenes.rel_l2(p_ref, p64)
    print("conv witness %s %s: ours %.3e, PyTorch fp32 %.3e, ratio %.2f" % (which, dims, e_ours, e_torch, e_ours / max(e_torch, 1e-300)))
    assert e_ours <= (1.0 if which == "3d" else 4.0) * e_torch + 1e-8, (e_ours, e_torch)
    # in-place form used by simulate(): outputs alias the inputs
    model.forward([tp, tU, tf], out=[tp, tU])
    assert torch.equal(tp, p) and torch.equal(tU, U)


def _run_both(oracle, batch_np, mconf, layers, steps, model=None):
    import torch
    from fluidnet_amd.simulate import simulate
    dev = torch.device("cuda:0")
    tb = _to_dev(batch_np, dev)
    for _ in range(steps):
        S.simulate(oracle, mconf, batch_np, layers)
        simulate(None, mconf, tb, model)
    return batch_np, tb


def _plume_batch(dims, rad, uscale, obstacles_seed=None):
    Z, Y, X = dims
    is3d = Z > 1
    C = 3 if is3d else 2
    flags = scenes.empty_domain(1, Z, Y, X, is3d)
    if obstacles_seed is not None:
        rng = np.random.RandomState(obstacles_seed)
        scenes.add_obstacles(flags[:, :, :, Y // 3:, :], is3d, rng, n_sphere=2, n_box=1)
    b = dict(pDiv=np.zeros((1, 1, Z, Y, X), np.float32), UDiv=np.zeros((1, C, Z, Y, X), np.float32),
             flags=np.ascontiguousarray(flags), density=np.zeros((1, 1, Z, Y, X), np.float32))
    S.create_plume_bcs(b, [1.0], uscale, rad)
    return b


def test_simulate_config1_2d_jacobi_bit_exact(oracle):
    """BASELINE config 1: 2-D 64x64 smoke plume, Jacobi 20 iterations (fluid_net_2d_demo.lua settings)."""
    b = _plume_batch((1, 64, 64), 0.05, 10.0)
    mconf = dict(dt=4 / 60, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=1.0,
                 gravityScale=0, vorticityConfinementAmp=0, simMethod="jacobi", maxIter=20)
    nb, tb = _run_both(oracle, b, mconf, None, 12)
    for k in ("pDiv", "UDiv", "density"):
        assert np.array_equal(tb[k].cpu().numpy(), nb[k]), k
    assert nb["density"][0, 0, 0, 4:].sum() > 0


@pytest.mark.parametrize("dims,rad,usc", [((1, 48, 48), 0.08, 4.0), ((20, 24, 24), 0.15, 1.0)])
def test_simulate_pcg_projection(oracle, dims, rad, usc):
    """simMethod = 'pcg' (simulate.lua:281-286: tol 1e-4, ic0): the baseline solver inside the step. The solve is
    iterative in fp32 on both sides (different summation orders), so fields are held to the north-star tolerance
    scaled by the solver's own tol, and the projected velocity must be divergence-free to that tol."""
    from fluidnet_amd import tfluids
    b = _plume_batch(dims, rad, usc, obstacles_seed=5)
    # an inflow into a closed box makes A p = div singular AND inconsistent (what (P)CG then returns is rounding);
    # open the top: a row of empty cells under the upper wall (p = 0 there) makes the system well posed
    Y = dims[1]
    top = b["flags"][:, :, 1:-1, Y - 2, 1:-1] if dims[0] > 1 else b["flags"][:, :, :, Y - 2, 1:-1]
    top[...] = 4.0
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=1.0,
                 gravityScale=0, vorticityConfinementAmp=0.5, simMethod="pcg", maxIter=400)
    nb, tb = _run_both(oracle, b, mconf, None, 5)
    for k in ("UDiv", "density"):
        r = scenes.rel_l2(tb[k].cpu().numpy(), nb[k])
        assert np.isfinite(tb[k].cpu().numpy()).all() and r <= 2e-4, (k, r)
    assert np.abs(tb["pDiv"].cpu().numpy() - nb["pDiv"]).max() <= 1e-3 * max(1.0, np.abs(nb["pDiv"]).max())
    import torch
    U = tb["UDiv"].clone()
    tfluids.setWallBcsForward(U, tb["flags"])
    div = torch.empty_like(tb["pDiv"])
    tfluids.velocityDivergenceForward(U, tb["flags"], div)
    inner = div[..., 5:Y - 3, :]       # rows 0..3 carry the plume inflow BC (re-imposed after the projection)
    assert float(inner.abs().max()) < 5e-4, float(inner.abs().max())
    assert nb["density"][0, 0, :, 4:].sum() > 0


def test_simulate_config2_2d_convnet(oracle):
    """BASELINE config 2: 2-D 128x128, ConvNet projection with the shipped myModel2D weights."""
    from fluidnet_amd import FluidNetModel
    layers = _layers2d()
    b = _plume_batch((1, 128, 128), 0.05, 10.0)
    mconf = dict(dt=4 / 60, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=1.0,
                 gravityScale=0, vorticityConfinementAmp=0, simMethod="convnet")
    nb, tb = _run_both(oracle, b, mconf, layers, 6, FluidNetModel(layers, False))
    for k in ("pDiv", "UDiv", "density"):
        r = scenes.rel_l2(tb[k].cpu().numpy(), nb[k])
        assert np.isfinite(tb[k].cpu().numpy()).all() and r <= TOL, (k, r)


@pytest.mark.parametrize("res,vort,obst", [(32, 0.0, None), (40, 3.0, 7)])
def test_simulate_config3_4_3d_convnet(oracle, res, vort, obst):
    """BASELINE configs 3/4 at oracle size: 3-D plume (fluid_net_3d_sim.lua:62-87 settings scaled by
    res/128), MacCormack + ConvNet projection; config 4 adds a voxel obstacle + vorticity confinement."""
    from fluidnet_amd import FluidNetModel
    layers = S.default_3d_layers(seed=1)
    b = _plume_batch((res, res, res), 0.15, 1.0 * res / 128, obst)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=2.0 * res / 128,
                 gravityScale=0, vorticityConfinementAmp=vort, simMethod="convnet")
    nb, tb = _run_both(oracle, b, mconf, layers, 4, FluidNetModel(layers, True))
    for k in ("pDiv", "UDiv", "density"):
        r = scenes.rel_l2(tb[k].cpu().numpy(), nb[k])
        assert np.isfinite(tb[k].cpu().numpy()).all() and r <= TOL, (k, r)
    assert np.abs(nb["UDiv"]).max() > 0


def test_simulate_gravity_and_rgb_density(oracle):
    """The 2-D demo's RGB density table + gravity branch (simulate.lua:183-195, 229-233)."""
    b = _plume_batch((1, 48, 48), 0.1, 5.0)
    b["density"] = [b["density"].copy() for _ in range(3)]
    S.create_plume_bcs(b, [1.0, 0.5, 0.25], 5.0, 0.1)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=0.5,
                 gravityScale=0.3, vorticityConfinementAmp=1.0, simMethod="jacobi", maxIter=7)
    nb, tb = _run_both(oracle, b, mconf, None, 5)
    assert np.array_equal(tb["UDiv"].cpu().numpy(), nb["UDiv"])
    for i in range(3):
        assert np.array_equal(tb["density"][i].cpu().numpy(), nb["density"][i])


@experiments_flavour
def test_conv_paths_agree_3d(oracle, monkeypatch):
    """The split-operand fp16 MFMA kernels of conv_mfma16.hip (the default: z-marched 32x8 columns, and the 32x4x4 tile
    form kept beside them), the vector-ALU kernels of conv_valu.hip (Winograd F(2,3) along x), the fp32-MFMA implicit GEMM
    (conv_mfma.hip) and the shape-generic direct kernels (conv.hip) are evaluations of the same sums: they must agree to
    rounding, including on grids that are ragged (and odd in x: the Winograd lanes own x-pairs) against the tiles; the
    fp16 path must not have clamped anything."""
    import torch
    from fluidnet_amd import FluidNetModel
    layers = S.default_3d_layers(seed=5)
    dev = torch.device("cuda:0")
    for dims, seed in [((32, 32, 32), 51), ((13, 21, 45), 52), ((5, 9, 33), 53), ((6, 7, 130), 54)]:
        sc = scenes.make_scene(dims, seed=seed, vel_cells=0.4, B=2)
        tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
        monkeypatch.setenv("TFL_CONV_PATH", "direct")
        pd, Ud = FluidNetModel(layers, True).forward([tp, tU, tf])
        monkeypatch.setenv("TFL_CONV_PATH", "mfma")
        pm, Um = FluidNetModel(layers, True).forward([tp, tU, tf])
        rp, rU = scenes.rel_l2(pm.cpu().numpy(), pd.cpu().numpy()), scenes.rel_l2(Um.cpu().numpy(), Ud.cpu().numpy())
        assert rp <= 2e-6 and rU <= 2e-6, (dims, rp, rU)
        monkeypatch.setenv("TFL_CONV_PATH", "winograd")      # conv_valu.hip, Winograd along x
        pw, Uw = FluidNetModel(layers, True).forward([tp, tU, tf])
        rp, rU = scenes.rel_l2(pw.cpu().numpy(), pd.cpu().numpy()), scenes.rel_l2(Uw.cpu().numpy(), Ud.cpu().numpy())
        assert rp <= 2e-6 and rU <= 2e-6, ("wino", dims, rp, rU)
        monkeypatch.delenv("TFL_CONV_PATH")      # the default: conv_mfma16.hip
        for tiled in (None, "3"):
            if tiled:
                monkeypatch.setenv("TFL_M16_TILED", tiled)
            m16 = FluidNetModel(layers, True)
            ph, Uh = m16.forward([tp, tU, tf])
            rp, rU = scenes.rel_l2(ph.cpu().numpy(), pd.cpu().numpy()), scenes.rel_l2(Uh.cpu().numpy(), Ud.cpu().numpy())
            assert rp <= 2e-6 and rU <= 2e-6, ("mfma16", tiled, dims, rp, rU)
            assert m16.range_errors(tp) == 0
            if tiled:
                monkeypatch.delenv("TFL_M16_TILED")
        p_ref, U_ref = S.model_forward(oracle, layers, sc["p"], sc["U"], sc["flags"])
        assert scenes.rel_l2(pm.cpu().numpy(), p_ref) <= TOL and scenes.rel_l2(Um.cpu().numpy(), U_ref) <= TOL


@experiments_flavour
def test_conv_fused_layers_and_mfma_tail(monkeypatch):
    """Round 5: layers 1 + 2 of the 3-D default net in ONE launch (k_conv3_m16p_f2: layer 1's planes stay in an LDS ring, its
    x / y halo is recomputed) issue the same MFMAs in the same order per voxel as k_conv3_m16p_in + k_conv3_m16p -- the
    projection must be BIT-identical with the fusion on and off, on ragged grids, with several chunks per column
    (TFL_M16_CZ_F2) and B = 2. The tail's 1x1x1 layers on the matrix cores (one 16x16x16 MFMA per row) sum in another order
    than the vector-ALU epilogue: equal to rounding, and nothing clamped."""
    import torch
    from fluidnet_amd import FluidNetModel
    layers = S.default_3d_layers(seed=7)
    dev = torch.device("cuda:0")
    for dims, seed, B in [((32, 32, 32), 61, 1), ((13, 21, 45), 62, 2), ((5, 9, 33), 63, 1), ((6, 7, 130), 64, 1),
                          ((40, 24, 70), 65, 2), ((3, 8, 32), 66, 1), ((2, 6, 17), 67, 1)]:
        sc = scenes.make_scene(dims, seed=seed, vel_cells=0.4, B=B)
        tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
        monkeypatch.delenv("TFL_M16_FUSE12", raising=False)     # the default: one launch per layer
        m0 = FluidNetModel(layers, True)
        p0, U0 = m0.forward([tp, tU, tf])
        assert m0.range_errors(tp) == 0
        monkeypatch.setenv("TFL_M16_FUSE12", "1")
        for cz in (None, "3", "1", "64"):
            if cz:
                monkeypatch.setenv("TFL_M16_CZ_F2", cz)
            m1 = FluidNetModel(layers, True)
            p1, U1 = m1.forward([tp, tU, tf])
            assert torch.equal(p0, p1) and torch.equal(U0, U1), (dims, cz, scenes.rel_l2(p1.cpu().numpy(), p0.cpu().numpy()))
            assert m1.range_errors(tp) == 0
            if cz:
                monkeypatch.delenv("TFL_M16_CZ_F2")
        monkeypatch.delenv("TFL_M16_FUSE12")
        # the software-pipelined 8 -> 8 layers (k_conv3_m16q: the epilogue of plane q - 1 between the MFMAs of plane q, the
        # default) against a plane's epilogue behind its own MFMAs (k_conv3_m16p, TFL_M16_PIPE=0): the same MFMAs in the same
        # order per accumulator, the same epilogue arithmetic -- bit-identical
        monkeypatch.setenv("TFL_M16_PIPE", "0")
        m3 = FluidNetModel(layers, True)
        p3, U3 = m3.forward([tp, tU, tf])
        monkeypatch.delenv("TFL_M16_PIPE")
        assert torch.equal(p0, p3) and torch.equal(U0, U3), ("pipe", dims, scenes.rel_l2(p3.cpu().numpy(), p0.cpu().numpy()))
        monkeypatch.setenv("TFL_M16_TAIL_MFMA", "0")
        m2 = FluidNetModel(layers, True)
        p2, U2 = m2.forward([tp, tU, tf])
        monkeypatch.delenv("TFL_M16_TAIL_MFMA")
        rp, rU = scenes.rel_l2(p0.cpu().numpy(), p2.cpu().numpy()), scenes.rel_l2(U0.cpu().numpy(), U2.cpu().numpy())
        assert rp <= 1e-6 and rU <= 1e-6, (dims, rp, rU)
    # a blown-up input is still reported by the fused kernel (its own range check) and by the tail's split
    sc = scenes.make_scene((12, 16, 40), seed=77, vel_cells=0.4)
    tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
    monkeypatch.setenv("TFL_M16_FUSE12", "1")
    m = FluidNetModel(layers, True)
    m.forward([tp * 1e9, tU, tf])
    assert m.range_errors(tp) > 0


@experiments_flavour
def test_stats_reduction_folded_into_its_producer(monkeypatch):
    """Round 5: the fp64 reduction of the per-block {sum u, sum u^2} partials (the std normaliser of the net input) runs in
    the last block of k_bcs_div_stats to finish (two-level ticket counters, chip-coherent partials) instead of a launch of
    its own (k_reduce_stats) under TFL_STATS_FOLD=1 -- opt-in: measured slower than the launch it saves (model.hip). Same
    summation order, so the scale -- and with it every bit of p and U -- must be identical; several calls in a row (the
    counters re-arm), 2-D and 3-D, B = 3, ragged grids that take the one-cell kernels."""
    import torch
    from fluidnet_amd import FluidNetModel
    dev = torch.device("cuda:0")
    for dims, seed, B, which in [((24, 40, 72), 71, 1, "3d"), ((9, 13, 30), 72, 3, "3d"), ((1, 70, 90), 73, 2, "2d"),
                                 ((64, 64, 64), 74, 1, "3d")]:
        layers = _layers2d() if which == "2d" else S.default_3d_layers(seed=3)
        sc = scenes.make_scene(dims, seed=seed, vel_cells=0.4, B=B)
        tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
        monkeypatch.delenv("TFL_STATS_FOLD", raising=False)      # the default: k_reduce_stats as a launch of its own
        p0, U0 = FluidNetModel(layers, which == "3d").forward([tp, tU, tf])
        monkeypatch.setenv("TFL_STATS_FOLD", "1")
        m = FluidNetModel(layers, which == "3d")
        for rep in range(3):
            p1, U1 = m.forward([tp, tU, tf])
            assert torch.equal(p0, p1) and torch.equal(U0, U1), (dims, rep)


def test_first_conv_layer_sums_the_stat_partials_itself(monkeypatch):
    """Round 6: in tfl_model_forward the blocks of the first conv layer reduce k_bcs_div_stats' partial
    pairs themselves, in k_reduce_stats' order (tfl_device.hpp block_sum_pairs) -- the launch between the two kernels is gone.
    Same bits as with the separate launch (TFL_STATS_CONSUMER=0), B = 1 and 3, ragged grids (one-cell kernels), repeated
    calls; and the launch really is gone from the profile."""
    import torch
    from fluidnet_amd import FluidNetModel, tfluids
    dev = torch.device("cuda:0")
    for dims, seed, B in [((24, 40, 72), 71, 1), ((9, 13, 30), 72, 3), ((64, 64, 64), 74, 1), ((40, 128, 128), 75, 2)]:
        layers = S.default_3d_layers(seed=3)
        sc = scenes.make_scene(dims, seed=seed, vel_cells=0.4, B=B)
        tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
        monkeypatch.setenv("TFL_STATS_CONSUMER", "0")
        m0 = FluidNetModel(layers, True)
        with tfluids.profile(tU) as prof0:
            p0, U0 = m0.forward([tp, tU, tf])
        monkeypatch.delenv("TFL_STATS_CONSUMER")
        m = FluidNetModel(layers, True)
        for rep in range(3):
            with tfluids.profile(tU) as prof1:
                p1, U1 = m.forward([tp, tU, tf])
            assert torch.equal(p0, p1) and torch.equal(U0, U1), (dims, rep)
        assert "k_reduce_stats" in prof0.kernels and "k_reduce_stats" not in prof1.kernels, (sorted(prof0.kernels), sorted(prof1.kernels))


def test_fp16_range_errors_are_counted(oracle):
    """conv_mfma16.hip clamps activations at the fp16 range and COUNTS the blocks that did (tfl_model_range_errors): a net
    input far outside it (pressure 1e9 times the velocity scale) must be reported, an ordinary one must not."""
    import torch
    from fluidnet_amd import FluidNetModel
    dev = torch.device("cuda:0")
    layers = S.default_3d_layers(seed=3)
    sc = scenes.make_scene((12, 16, 40), seed=77, vel_cells=0.4)
    tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
    m = FluidNetModel(layers, True)
    m.forward([tp, tU, tf])
    assert m.range_errors(tp) == 0
    m.forward([tp * 1e9, tU, tf])
    assert m.range_errors(tp) > 0
    m.forward([tp, tU, tf])
    assert m.range_errors(tp) == 0          # the counter is read-and-reset
    # an overflow nobody polled for must not stay silent -- the projection kernel copies the count to pinned
    # host memory, and the next forward pass (no synchronisation needed to see it) is refused until it has been read
    from fluidnet_amd import TfluidsError
    m.forward([tp * 1e9, tU, tf])
    torch.cuda.synchronize()
    assert m.range_flag(tp) > 0
    with pytest.raises(TfluidsError, match="fp16 range"):
        m.forward([tp, tU, tf])
    assert m.range_errors(tp) > 0           # acknowledged ...
    assert m.range_flag(tp) == 0
    m.forward([tp, tU, tf])                 # ... and the model runs again
    assert m.range_errors(tp) == 0


MODEL_OPTS = [
    dict(inputChannels=dict(pDiv=True, UDiv=True, div=True)),                          # every field feeds the net
    dict(inputChannels=dict(pDiv=False, UDiv=True, div=False), nonlinType="relu6"),
    dict(inputChannels=dict(pDiv=False, UDiv=False, div=True), normalizeInputChan="div", nonlinType="sigmoid"),
    dict(normalizeInputChan="pDiv", normalizeInputFunc="norm"),
    dict(normalizeInputFunc="norm", addPressureSkip=True),
    dict(normalizeInput=False, addPressureSkip=True, nonlinType="relu6"),
]


@pytest.mark.parametrize("is3d", [False, True])
@pytest.mark.parametrize("oi", range(len(MODEL_OPTS)))
def test_model_options_match_restatement(oracle, is3d, oi):
    """The mconf switches of lib/model.lua:27-160, 356-387 that change the forward graph (tfl_model_opts): input channel
    sets, the normaliser's channel / function / absence, relu6 / sigmoid, the pressure skip -- each against the oracle's
    restatement of the same graph (PyTorch-CPU convolutions), seeded weights, hidden width 6 (zero-padded to 8) so that
    the skip channel lands behind padding."""
    import torch
    from fluidnet_amd import FluidNetModel
    opts = MODEL_OPTS[oi]
    o = S.model_opts(opts)
    ic = o["inputChannels"]
    C = 3 if is3d else 2
    in_c = int(ic["pDiv"]) + C * int(ic["UDiv"]) + int(ic["div"]) + 1
    skip = 1 if o["addPressureSkip"] else 0
    rng = np.random.RandomState(100 + oi)
    shapes = [(6, in_c, 3), (6, 6, 3), (1, 6 + skip, 1)]
    layers = []
    for co, ci, k in shapes:
        taps = k ** (3 if is3d else 2)
        shape = (co, ci) + ((k, k, k) if is3d else (k, k))
        layers.append(((rng.randn(*shape) * np.sqrt(2.0 / (ci * taps))).astype(np.float32), (rng.randn(co) * 0.05).astype(np.float32)))
    dims = (9, 12, 20) if is3d else (1, 33, 41)
    sc = scenes.make_scene(dims, seed=31 + oi, vel_cells=0.4, B=2)
    dev = torch.device("cuda:0")
    tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
    pm, Um = FluidNetModel(layers, is3d, opts=opts).forward([tp, tU, tf])
    p_ref, U_ref = S.model_forward(oracle, layers, sc["p"], sc["U"], sc["flags"], opts=opts)
    assert scenes.rel_l2(pm.cpu().numpy(), p_ref) <= TOL and scenes.rel_l2(Um.cpu().numpy(), U_ref) <= TOL


def test_model_option_errors():
    """Option combinations the reference rejects (model.lua:50-52, 81, 103, 121) or that do not chain."""
    import torch
    from fluidnet_amd import FluidNetModel, tfluids
    dev = torch.device("cuda:0")
    sc = scenes.make_scene((1, 16, 16), seed=3, B=1)
    tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
    w = lambda co, ci, k: (np.zeros((co, ci, k, k), np.float32), np.zeros(co, np.float32))
    with pytest.raises(tfluids.TfluidsError, match="flags"):
        FluidNetModel([w(4, 3, 3), w(1, 4, 1)], False, opts=dict(inputChannels=dict(flags=False)))
    with pytest.raises(tfluids.TfluidsError, match="normalizeInputFunc"):
        FluidNetModel([w(4, 3, 3), w(1, 4, 1)], False, opts=dict(normalizeInputFunc="max"))
    with pytest.raises(tfluids.TfluidsError, match="any"):
        FluidNetModel([w(4, 1, 3), w(1, 4, 1)], False, opts=dict(inputChannels=dict(pDiv=False, div=False))).forward([tp, tU, tf])
    with pytest.raises(tfluids.TfluidsError, match="input channels"):      # UDiv on: 2 more input channels than given
        FluidNetModel([w(4, 3, 3), w(1, 4, 1)], False, opts=dict(inputChannels=dict(UDiv=True))).forward([tp, tU, tf])
    with pytest.raises(tfluids.TfluidsError, match="chain"):               # the skip channel is missing from the last layer
        FluidNetModel([w(4, 3, 3), w(1, 4, 1)], False, opts=dict(addPressureSkip=True)).forward([tp, tU, tf])


@pytest.mark.parametrize("is3d,shapes", [(False, [(6, 3, 3), (6, 6, 1), (6, 6, 1), (1, 6, 1)]),       # `yang`, model.lua:188-205
                                         (True, [(6, 3, 3), (6, 6, 1), (6, 6, 1), (1, 6, 1)]),
                                         (True, [(5, 3, 3), (12, 5, 3), (1, 12, 1)]),
                                         (False, [(3, 3, 5), (20, 3, 3), (1, 20, 1)])])
def test_other_topologies_through_the_generic_kernels(oracle, is3d, shapes):
    """Any `default`-style conv stack (odd kernel sizes, up to 32 channels) runs through conv.hip; channel counts
    the kernels are not instantiated for (the `yang` model's 6) are zero-padded at model creation. Seeded weights
    (no such model is shipped), same non-conv graph, against the oracle's PyTorch-CPU convolutions."""
    import torch
    from fluidnet_amd import FluidNetModel
    rng = np.random.RandomState(17)
    layers = []
    for co, ci, k in shapes:
        taps = k ** (3 if is3d else 2)
        shape = (co, ci) + ((k, k, k) if is3d else (k, k))
        layers.append(((rng.randn(*shape) * np.sqrt(2.0 / (ci * taps))).astype(np.float32),
                       (rng.randn(co) * 0.05).astype(np.float32)))
    dev = torch.device("cuda:0")
    dims = (11, 14, 18) if is3d else (1, 37, 45)
    sc = scenes.make_scene(dims, seed=23, vel_cells=0.4, B=2)
    tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
    pm, Um = FluidNetModel(layers, is3d).forward([tp, tU, tf])
    p_ref, U_ref = S.model_forward(oracle, layers, sc["p"], sc["U"], sc["flags"])
    assert scenes.rel_l2(pm.cpu().numpy(), p_ref) <= TOL and scenes.rel_l2(Um.cpu().numpy(), U_ref) <= TOL


@pytest.mark.parametrize("is3d,dims", [(False, (1, 64, 96)), (True, (16, 24, 32)), (False, (1, 35, 70))])
def test_tog_topology_pooling_and_convolution_upsample(oracle, is3d, dims):
    """modelType = 'tog' (lib/model.lua:163-178 2-D, :211-218 3-D): 2x average pooling after the first
    layer(s), nn.{Spatial,Volumetric}ConvolutionUpsample (conv to 2^dim * nOut channels + pixel shuffle,
    lib/modules/*_convolution_upsample.lua) at the end; up to 64 channels, 5x5 kernels. Seeded weights (no such model is
    shipped) against PyTorch-CPU conv / avg_pool / the module's view-permute restated in oracle/simulate_np.py."""
    import torch
    from fluidnet_amd import FluidNetModel, TfluidsError
    dev = torch.device("cuda:0")
    model = FluidNetModel.tog(is3d, seed=9)
    sc = scenes.make_scene(dims, seed=31, vel_cells=0.4, B=2)
    tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
    if dims[2] % 2 or dims[1] % 2:      # 2-D tog pools once: odd sizes are refused, like cudnn's shape check would
        with pytest.raises(TfluidsError):
            model.forward([tp, tU, tf])
        return
    p, U = model.forward([tp, tU, tf])
    p_ref, U_ref = S.model_forward(oracle, model.layers, sc["p"], sc["U"], sc["flags"], pool=model.pool, up=model.up)
    rp, rU = scenes.rel_l2(p.cpu().numpy(), p_ref), scenes.rel_l2(U.cpu().numpy(), U_ref)
    assert rp <= TOL and rU <= TOL, (rp, rU)
    assert float(np.abs(p_ref).max()) > 0
    p64, _ = S.model_forward(oracle, model.layers, sc["p"], sc["U"], sc["flags"], conv_dtype="float64", pool=model.pool, up=model.up)
    assert scenes.rel_l2(p.cpu().numpy(), p64) <= 4 * scenes.rel_l2(p_ref, p64) + 1e-7


@pytest.mark.parametrize("which", ["2d_jacobi", "2d_convnet_rgb", "3d_convnet_vort_obstacle", "3d_gravity_rk2", "2d_pcg",
                                   "3d_buoyancy_any_gravity", "3d_buoyancy_no_vorticity", "3d_buoyancy_jacobi", "3d_buoyancy_fast_mode"])
def test_native_simulate_step_equals_python_orchestration(which):
    """tfl_simulate_step (csrc/simulate.cpp: lib/simulate.lua in native code behind one C-ABI call) against
    fluidnet_amd.simulate.simulate() over several steps: identical state, bit for bit."""
    import torch
    from fluidnet_amd import FluidNetModel
    from fluidnet_amd.simulate import simulate, simulate_native
    dev = torch.device("cuda:0")
    base = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0, gravityScale=0,
                vorticityConfinementAmp=0)
    model = None
    if which == "2d_jacobi":
        b, mconf = _plume_batch((1, 40, 44), 0.08, 5.0, obstacles_seed=3), dict(base, simMethod="jacobi", maxIter=15)
    elif which == "2d_convnet_rgb":
        b = _plume_batch((1, 36, 40), 0.1, 4.0)
        d = b["density"]
        b["density"] = [d.copy(), d.copy(), d.copy()]
        S.create_plume_bcs(b, [1.0, 0.5, 0.25], 4.0, 0.1)
        layers = _layers2d()
        model, mconf = FluidNetModel(layers, False), dict(base, simMethod="convnet", vorticityConfinementAmp=0.4)
    elif which == "3d_convnet_vort_obstacle":
        b = _plume_batch((20, 24, 28), 0.15, 1.0, obstacles_seed=7)
        model = FluidNetModel(S.default_3d_layers(seed=4), True)
        mconf = dict(base, simMethod="convnet", vorticityConfinementAmp=2.0, buoyancyScale=2.0)
    elif which == "3d_gravity_rk2":
        b = _plume_batch((12, 14, 16), 0.15, 1.0)
        model = FluidNetModel(S.default_3d_layers(seed=5), True)
        mconf = dict(base, simMethod="convnet", advectionMethod="rk2Ours", gravityScale=0.3, buoyancyScale=0, gravity=[0.2, 1.0, -0.3])
    elif which.startswith("3d_buoyancy"):
        # round 5: the buoyancy force inside pass B of advectVel (BuoyFold) and the velocity delivered into U by whoever writes
        # every cell next -- a gravity with three components (the generic BUOY mask), no vorticity (the ConvNet projection reads
        # the advection's scratch array), a Jacobi projection (the copy path), and the advection's tolerance mode
        b = _plume_batch((20, 24, 68), 0.15, 1.0, obstacles_seed=7)
        model = FluidNetModel(S.default_3d_layers(seed=5), True)
        if which == "3d_buoyancy_any_gravity":
            mconf = dict(base, simMethod="convnet", buoyancyScale=1.5, gravity=[0.3, 1.0, -0.4], vorticityConfinementAmp=2.0)
        elif which == "3d_buoyancy_no_vorticity":
            mconf = dict(base, simMethod="convnet", buoyancyScale=1.5)
        elif which == "3d_buoyancy_jacobi":
            model, mconf = None, dict(base, simMethod="jacobi", maxIter=12, buoyancyScale=1.5, gravity=[0.0, 0.0, 1.0])
        else:
            mconf = dict(base, simMethod="convnet", buoyancyScale=1.5, vorticityConfinementAmp=1.0)
    else:
        b = _plume_batch((1, 32, 32), 0.1, 3.0, obstacles_seed=5)
        b["flags"][:, :, :, 30, 1:-1] = 4.0      # open top: a well-posed system
        mconf = dict(base, simMethod="pcg", maxIter=300, pcgPrecond="none")
    ta, tb = _to_dev(b, dev), _to_dev(b, dev)
    if which == "3d_buoyancy_fast_mode":
        from fluidnet_amd import tfluids
        tfluids.set_advect_mode(ta["UDiv"], "fast")
    try:
        for _ in range(5):
            simulate(None, mconf, ta, model)
            simulate_native(None, mconf, tb, model)
    finally:
        if which == "3d_buoyancy_fast_mode":
            tfluids.set_advect_mode(ta["UDiv"], "exact")
    for k in ("pDiv", "UDiv"):
        assert torch.equal(ta[k], tb[k]), (which, k)
    da, db = ta["density"], tb["density"]
    for x, y in zip(da if isinstance(da, list) else [da], db if isinstance(db, list) else [db]):
        assert torch.equal(x, y), (which, "density")
    assert float(ta["UDiv"].abs().max()) > 0.1


def _slab_sims(ref, mconf, world, layers_seed, reach=1, overlap=None, check_reach=True, transport="thread"):
    """Cut the global state `ref` (dict of device tensors) into `world` virtual z-slab ranks (threads). transport:
    "thread" = ThreadComm (Python callbacks), "native" = the library's RCCL transport (csrc/comm_rccl.cpp; needs an
    RCCL that accepts several ranks per device, i.e. tests/stub_rccl.cpp through TFL_RCCL_LIBRARY)."""
    import torch
    from fluidnet_amd import FluidNetModel, tfluids
    from fluidnet_amd.dist import RcclComm, SlabLayout, SlabSimulation, ThreadComm
    Zt = ref["flags"].size(2)
    hub = ThreadComm.Hub(world)
    uid = RcclComm.unique_id(tfluids._context(ref["flags"])[1]) if transport == "native" and world > 1 else None
    sims = []
    for r in range(world):
        lay = SlabLayout(Zt, world, r, reach)
        loc = {k: (lay.extract(v) if torch.is_tensor(v) else v) for k, v in ref.items()}
        model = FluidNetModel(layers_seed, True) if isinstance(layers_seed, list) else FluidNetModel.default_3d(seed=layers_seed)
        comm = None
        if world > 1:
            comm = ThreadComm(hub, r) if transport == "thread" else (lambda ctx, r=r: RcclComm(ctx, uid, r, world))
        sims.append(SlabSimulation(loc, mconf, model, lay, comm, check_reach=check_reach, overlap=overlap, own_context=True))
    return sims


def _assert_slabs_equal(sims, ref, tol=1e-6):
    for s in sims:
        for k in ("pDiv", "UDiv", "density"):
            got = s.lay.owned(s.batch[k])
            want = ref[k][:, :, s.lay.z0:s.lay.z1]
            rel = float((got - want).norm() / want.norm().clamp_min(1e-30))
            assert rel <= tol, (s.lay.rank, k, rel)


_BC_FOLD_AB = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import test_hip_simulate as T
from fluidnet_amd import FluidNetModel
from fluidnet_amd.simulate import simulate_native
dev = torch.device("cuda:0")
model = FluidNetModel.default_3d(seed=1)
mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=2.0, gravityScale=0,
             vorticityConfinementAmp=3.0, simMethod="convnet")
for dims in [(20, 40, 72), (12, 28, 128)]:
    res = {}
    for fold in ("1", "0"):
        os.environ["TFL_BC_FOLD"] = fold
        tb = T._to_dev(T._plume_batch(dims, 0.15, 1.0, obstacles_seed=5), dev)
        for _ in range(4):
            simulate_native(None, mconf, tb, model)
        res[fold] = {k: tb[k].cpu().numpy() for k in ("pDiv", "UDiv", "density")}
        assert np.isfinite(res[fold]["UDiv"]).all() and float(np.abs(res[fold]["UDiv"]).max()) > 0
    for k in res["1"]:
        a, b = res["1"][k], res["0"][k]
        assert np.array_equal(a, b), (dims, k, int((a != b).sum()))      # IEEE ==: -0 and +0 compare equal
print("BC_FOLD_AB_OK")
"""


@pytest.mark.parametrize("env", [{}, {"TFL_VEL3_KZ": "2", "TFL_VORT_FUSED": "1", "TFL_SCAL3_TZ": "14"}], ids=["small-grid-kernels", "big-grid-kernels"])
def test_bc_fold_on_and_off_agree(env):
    """the sparse setConstVals pairs are applied either by their own index-list launch or, folded, inside the
    kernel that produces the field (pass B of the advection, the last force, the projection) -- which of the two happens
    depends on the grid size (the big-grid advectVel / fused-confinement kernels never fold), on the alignment of the
    pair and on TFL_BC_FOLD. The two forms must give the same state: IEEE-equal (the folded form evaluates x*inv+bc on
    every cell of the pair's bounding box like the reference's dense cmul/add, the index list leaves identity cells alone:
    that can turn a -0 into +0 and nothing else). Native step, 3-D plume + obstacles + vorticity, fold on and off, with the
    small-grid and (forced: child process, the switches are read once) the big-grid kernel variants."""
    import subprocess, sys
    code = _BC_FOLD_AB % (os.path.dirname(HERE), HERE)
    e = dict(os.environ)
    for k in ("TFL_VEL3_KZ", "TFL_SCAL3_TZ", "TFL_VORT_FUSED", "TFL_BC_FOLD"):
        e.pop(k, None)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "BC_FOLD_AB_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_native_step_equals_python_with_the_big_grid_kernels():
    """The native step picks its kernels by grid size (two-plane advectVel from 6 M cells, the fused confinement from 2 M), and
    with a buoyancy fold pending it sends pass B of advectVel through the ONE-plane kernel that carries the fold while pass A
    stays two-plane (advect_vel3.hip, round 5: 256^3). Those switches are read once per process, so the 3-D scenes of
    test_native_simulate_step_equals_python_orchestration run again in a child process with the big-grid variants forced."""
    import subprocess, sys
    e = dict(os.environ)
    e.update({"TFL_VEL3_KZ": "2", "TFL_VORT_FUSED": "1", "TFL_SCAL3_TZ": "14"})
    e.pop("TFL_VEL3_KZ_B", None)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_hip_simulate.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          "-k", "test_native_simulate_step_equals_python_orchestration and 3d_"], env=e, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("world,overlap", [(1, False), (2, False), (2, True), (4, True), (3, False)])
def test_zslab_decomposition_equals_single_gpu(world, overlap):
    """tfl_simulate_step_slab, verified on ONE GPU with virtual ranks (threads + an in-process transport): every rank
    runs the native slab step on its planes + 4 halo planes, each phase under its own z-window; after 6 steps the owned
    planes must equal the unsplit tfl_simulate_step (bit-exact up to the fp64 summation order of the std all-reduce;
    world = 1 must be bit-exact). overlap = boundary strips first, interior while the message is in flight."""
    import torch
    from fluidnet_amd import FluidNetModel
    from fluidnet_amd.dist import run_virtual_ranks
    from fluidnet_amd.simulate import simulate_native
    dev = torch.device("cuda:0")
    Zt, Y, X = 12 * world, 24, 32
    b = _plume_batch((Zt, Y, X), 0.15, 0.6, obstacles_seed=11)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                 gravityScale=0.2, vorticityConfinementAmp=2.0, simMethod="convnet")
    layers = S.default_3d_layers(seed=2)
    ref = _to_dev(b, dev)
    model = FluidNetModel(layers, True)
    sims = _slab_sims(ref, mconf, world, layers, overlap=overlap)
    for _ in range(3):                       # two rounds: messages left in flight by one call feed the next
        for _ in range(2):
            simulate_native(None, mconf, ref, model)
        run_virtual_ranks(sims, 2)
        # bit for bit: back-trace positions are formed in global z (tfl_set_z_origin) and dx comes from the whole grid; only
        # the fp64 summation order of the std all-reduce differs, which the fp32 scale almost never sees
        _assert_slabs_equal(sims, ref, 0.0 if world == 1 else 1e-7)
    assert float(ref["UDiv"].abs().max()) > 0
    for s in sims:       # after drain() the halo planes the messages refresh are valid too: U (2, 2), p (4 below, 3 above)
        lay = s.lay
        for k, below, above in (("UDiv", 2, 2), ("pDiv", 4, 3)):
            a = lay.c0 - (below if lay.has_lower else 0)
            b = lay.c1 + (above if lay.has_upper else 0)
            assert torch.equal(s.batch[k][:, :, a:b], ref[k][:, :, lay.lo + a:lay.lo + b]), (lay.rank, k)
        s.close()


def test_zslab_reach_violation_is_reported():
    """A flow faster than the slab's back-trace reach (max|u_z|*dt >= R) breaks the halo contract: the step after the
    offending one must raise instead of silently returning wrong planes (unchecked precondition)."""
    import torch
    from fluidnet_amd import tfluids
    from fluidnet_amd.dist import run_virtual_ranks
    dev = torch.device("cuda:0")
    b = _plume_batch((24, 16, 16), 0.15, 0.6)
    b["UDiv"][:, 2, 4:20, 4:12, 4:12] = 15.0          # 1.5 cells per step along z
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                 gravityScale=0, vorticityConfinementAmp=0, simMethod="convnet")
    sims = _slab_sims(_to_dev(b, dev), mconf, 2, S.default_3d_layers(seed=2))
    with pytest.raises(tfluids.TfluidsError, match="reach"):
        run_virtual_ranks(sims, 3)
    for s in sims:
        s.close()
    # the same flow is fine for a slab that was laid out for reach 2 (5 halo planes)
    sims = _slab_sims(_to_dev(b, dev), dict(mconf, buoyancyScale=0.0), 2, S.default_3d_layers(seed=2), reach=2)
    run_virtual_ranks(sims, 1)
    for s in sims:
        s.close()


def test_zslab_reach_check_rides_on_the_projection_kernel():
    """Round 6: on grids whose planes fill k_project's blocks (X a multiple of 128, Y of 8) the step's own projection kernel folds
    max |u_z| of the planes it writes into the sticky reach word, and from the second step on the slab step launches no k_absmax
    of its own. A flow that only becomes too fast LATER in the run (written into the state after two quiet steps) must still
    be reported -- through that route --, and a quiet run must stay equal to the un-cut one."""
    import torch
    from fluidnet_amd import FluidNetModel, tfluids
    from fluidnet_amd.dist import run_virtual_ranks
    from fluidnet_amd.simulate import simulate_native
    dev = torch.device("cuda:0")
    b = _plume_batch((24, 16, 128), 0.15, 0.6)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                 gravityScale=0, vorticityConfinementAmp=0, simMethod="convnet")
    layers = S.default_3d_layers(seed=2)
    ref = _to_dev(b, dev)
    model = FluidNetModel(layers, True)
    sims = _slab_sims(ref, mconf, 2, layers)
    for _ in range(3):
        simulate_native(None, mconf, ref, model)
        run_virtual_ranks(sims, 1)
        _assert_slabs_equal(sims, ref, 1e-7)
    torch.cuda.synchronize()
    for s in sims:
        s.batch["UDiv"][:, 2, :, 4:12, 32:96] = 15.0          # 1.5 cells per step along z from now on
    with pytest.raises(tfluids.TfluidsError, match="reach"):
        run_virtual_ranks(sims, 8)
    for s in sims:
        s.close()


@pytest.mark.parametrize("world", [2, 3])
def test_zslab_reach_fallback_keeps_the_cut_run_exact(world):
    """VERDICT r05 item 7 / SURVEY 7 "semi-Lagrangian reach": with check_reach = "exact" (tfl_slab.check_reach = 2) the reach a
    step needs is found BEFORE its advection, agreed by all ranks (one-hot flags through the transport's all-reduce), and a
    step the halos do not cover is refused on every rank with nothing written (TFL_EREACH); SlabSimulation then widens the
    halos of every tensor of the batch through the transport (tfl_slab_exchange) and takes the step again. The scene of
    test_zslab_reach_violation_is_reported -- 1.5 cells per step along z through the cut(s) -- must end equal to the un-cut
    step instead of raising, and the simulation must report that it re-laid itself out for reach 2."""
    import torch
    from fluidnet_amd import FluidNetModel
    from fluidnet_amd.dist import run_virtual_ranks
    from fluidnet_amd.simulate import simulate_native
    dev = torch.device("cuda:0")
    Zt = 12 * world
    b = _plume_batch((Zt, 16, 16), 0.15, 0.6)
    b["UDiv"][:, 2, 4:Zt - 4, 4:12, 4:12] = 15.0          # 1.5 cells per step along z
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                 gravityScale=0, vorticityConfinementAmp=1.0, simMethod="convnet")
    layers = S.default_3d_layers(seed=2)
    ref = _to_dev(b, dev)
    model = FluidNetModel(layers, True)
    sims = _slab_sims(ref, mconf, world, layers, check_reach="exact")
    for step in range(3):
        simulate_native(None, mconf, ref, model)
        run_virtual_ranks(sims, 1)
        _assert_slabs_equal(sims, ref, 1e-7)
    assert float(ref["UDiv"][:, 2].abs().max()) * 0.1 > 1.0
    for sim in sims:
        assert sim.relayouts == [2] and sim.lay.reach == 2 and sim.lay.halo == 5, (sim.lay.rank, sim.relayouts)
        assert sim.batch["UDiv"].size(2) == sim.lay.hi - sim.lay.lo
        sim.close()


def test_zslab_reach2_equals_single_gpu():
    """with reach R = 2 the velocity's self-advection samples U up to 2R planes from the owned range, so the
    U message must refresh max(R+1, 2R) planes, not R+1. A flow with |u_z|*dt in (1, 2) across the slab boundaries,
    several steps (the planes beyond R+1 go stale only from the second step on), against the unsplit step."""
    import torch
    from fluidnet_amd import FluidNetModel
    from fluidnet_amd.dist import run_virtual_ranks
    from fluidnet_amd.simulate import simulate_native
    dev = torch.device("cuda:0")
    world, Zt, Y, X = 3, 36, 20, 24
    b = _plume_batch((Zt, Y, X), 0.15, 0.6)
    b["UDiv"][:, 2, 2:34, 4:16, 4:20] = 14.0          # 1.4 cells per step along z, through both cuts (z = 12, 24)
    b["UDiv"][:, 2, 8:28, 6:12, 6:14] = -12.0
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                 gravityScale=0, vorticityConfinementAmp=1.0, simMethod="convnet")
    layers = S.default_3d_layers(seed=2)
    ref = _to_dev(b, dev)
    model = FluidNetModel(layers, True)
    sims = _slab_sims(ref, mconf, world, layers, reach=2, overlap=False, check_reach=False)
    for _ in range(4):
        simulate_native(None, mconf, ref, model)
        run_virtual_ranks(sims, 1)
        _assert_slabs_equal(sims, ref, 1e-7)
    assert float(ref["UDiv"][:, 2].abs().max()) * 0.1 > 1.0     # the flow really exceeds one cell per step
    for s in sims:
        s.close()


@pytest.mark.parametrize("dims", [(1, 24, 32), (12, 16, 24), (20, 24, 128)], ids=["2d", "3d-ragged-blocks", "3d-full-blocks"])
def test_wall_plan_gives_the_same_bits(dims):
    """Round 6: with a tfl_wall_plan registered for the scene's flags (fluidnet_amd.simulate.wall_plan: created on first use,
    cached by the tensor's identity and version) the projection's first kernel reads one code byte per cell instead of decoding
    ten rows of flag words -- the same setWallBcs decisions, so every field of the step must come out bit for bit the same as
    without a plan. Obstacles with and without the stick bit, an outflow face, several steps; then an in-place edit of the flags
    must be noticed (a new plan), and the edited scene must again equal its plan-less twin."""
    import torch
    from fluidnet_amd import FluidNetModel, simulate as SIM
    from fluidnet_amd.simulate import simulate_native
    dev = torch.device("cuda:0")
    Z, Y, X = dims
    is3d = Z > 1
    b = _plume_batch(dims, 0.15, 0.6, obstacles_seed=7)
    rng = np.random.RandomState(3)
    scenes.add_obstacles(b["flags"][:, :, :, Y // 2:, :], is3d, rng, n_sphere=1, n_box=1, stick=True)
    b["flags"][:, :, :, -1, 1:-1][b["flags"][:, :, :, -1, 1:-1] == scenes.OBSTACLE] = scenes.EMPTY | scenes.OUTFLOW      # an open top
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0, gravityScale=0,
                 vorticityConfinementAmp=1.0 if is3d else 0, simMethod="convnet")
    layers = S.default_3d_layers(seed=2) if is3d else _layers2d()
    model = FluidNetModel(layers, is3d)
    plain, planned = _to_dev(b, dev), _to_dev(b, dev)

    def steps(batch, with_plans, n):
        prev, SIM._WALL_PLANS = SIM._WALL_PLANS, with_plans
        try:
            for _ in range(n):
                simulate_native(None, mconf, batch, model)
        finally:
            SIM._WALL_PLANS = prev
    steps(plain, False, 3)
    steps(planned, True, 3)
    assert SIM._wall_cache.get(planned["flags"], planned["flags"]) and not SIM._wall_cache.get(plain["flags"], plain["flags"])
    for k in ("pDiv", "UDiv", "density"):
        assert torch.equal(plain[k], planned[k]), k
    assert float(planned["UDiv"].abs().max()) > 0
    first = SIM._wall_cache.get(planned["flags"], planned["flags"])      # (the cache entry of this VERSION of the tensor)
    for batch in (plain, planned):      # an obstacle appears: torch's version counter moves, the plan must follow
        batch["flags"][:, :, :, Y // 4:Y // 4 + 2, X // 4:X // 2] = float(scenes.OBSTACLE)
    steps(plain, False, 2)
    steps(planned, True, 2)
    second = SIM._wall_cache.get(planned["flags"], planned["flags"])
    assert second and second is not first      # a new entry with a new plan (the old plan was queued for destruction)
    for k in ("pDiv", "UDiv", "density"):
        assert torch.equal(plain[k], planned[k]), k


@pytest.mark.parametrize("which", ["2d_convnet", "2d_jacobi", "3d_convnet"])
def test_graphed_simulate_equals_eager(which):
    """GraphedSimulate (one HIP-graph replay per step) must be bit-identical to eager simulate()."""
    import torch
    from fluidnet_amd import FluidNetModel
    from fluidnet_amd.simulate import GraphedSimulate, simulate
    dev = torch.device("cuda:0")
    if which == "3d_convnet":
        b = _plume_batch((24, 24, 32), 0.15, 0.5, obstacles_seed=5)
        model = FluidNetModel(S.default_3d_layers(seed=1), True)
        mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0,
                     gravityScale=0, vorticityConfinementAmp=2.0, simMethod="convnet")
    else:
        b = _plume_batch((1, 64, 64), 0.05, 10.0)
        model = FluidNetModel(_layers2d(), False) if which == "2d_convnet" else None
        mconf = dict(dt=4 / 60, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=1.0,
                     gravityScale=0, vorticityConfinementAmp=0, simMethod="convnet" if model else "jacobi", maxIter=20)
    eager, graphed, graphed_native = _to_dev(b, dev), _to_dev(b, dev), _to_dev(b, dev)
    g = GraphedSimulate(None, mconf, graphed, model)
    gn = GraphedSimulate(None, mconf, graphed_native, model, native=True)    # the one-call native step captured (round 5)
    for _ in range(6):
        simulate(None, mconf, eager, model)
        g.step()
        gn.step()
    for k in ("pDiv", "UDiv", "density"):
        assert torch.equal(eager[k], graphed[k]), k
        assert torch.equal(eager[k], graphed_native[k]), ("native", k)
    assert float(eager["UDiv"].abs().max()) > 0


def test_conv_paths_agree_2d(oracle, monkeypatch):
    """2-D: the 16-channel MFMA kernels (conv2d_mfma.hip) vs the shape-generic direct kernels (conv.hip) on the
    shipped myModel2D weights, including grids ragged against the 32x4 tile and unaligned row pitches."""
    import torch
    from fluidnet_amd import FluidNetModel
    layers = _layers2d()
    dev = torch.device("cuda:0")
    for dims, seed in [((1, 128, 128), 71), ((1, 37, 53), 72), ((1, 5, 130), 73)]:
        sc = scenes.make_scene(dims, seed=seed, vel_cells=0.4, B=2)
        tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
        monkeypatch.setenv("TFL_CONV_PATH", "direct")
        pd, Ud = FluidNetModel(layers, False).forward([tp, tU, tf])
        monkeypatch.setenv("TFL_CONV_PATH", "mfma")
        pm, Um = FluidNetModel(layers, False).forward([tp, tU, tf])
        rp, rU = scenes.rel_l2(pm.cpu().numpy(), pd.cpu().numpy()), scenes.rel_l2(Um.cpu().numpy(), Ud.cpu().numpy())
        assert rp <= 2e-6 and rU <= 2e-6, (dims, rp, rU)
        p_ref, U_ref = S.model_forward(oracle, layers, sc["p"], sc["U"], sc["flags"])
        assert scenes.rel_l2(pm.cpu().numpy(), p_ref) <= TOL and scenes.rel_l2(Um.cpu().numpy(), U_ref) <= TOL


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_simulate_long_horizon_parity(oracle, mode):
    """Drift check: 40 consecutive simulate() steps of the bench scene at 48^3 (MacCormack + buoyancy + vorticity
    confinement + obstacle + ConvNet projection) stay within the north-star tolerance of the CPU restatement, in both
    advection modes (tfl_set_advect_mode: `fast` = the tolerance mode of the LDS-tiled kernels).
    Measured r01 (exact): rel-L2 U 9e-8, p 1e-7, density 3e-8 after 40 steps (1.7e-7 / 1.4e-7 / 6.5e-8 after 60)."""
    import torch
    import bench
    from fluidnet_amd import FluidNetModel, tfluids
    from fluidnet_amd.simulate import simulate
    dev = torch.device("cuda:0")
    batch, mconf = bench.build_scene(48, 48, None, dev)
    model = FluidNetModel.default_3d(seed=1)
    nb = {k: (v.cpu().numpy().copy() if torch.is_tensor(v) else v) for k, v in batch.items()}
    tfluids.set_advect_mode(batch["UDiv"], mode)
    try:
        for _ in range(40):
            simulate(None, mconf, batch, model)
            S.simulate(oracle, mconf, nb, model.layers)
    finally:
        tfluids.set_advect_mode(batch["UDiv"], "exact")
    assert float(np.abs(nb["UDiv"]).max()) > 0.5          # the plume has developed
    for k in ("pDiv", "UDiv", "density"):
        r = scenes.rel_l2(batch[k].cpu().numpy(), nb[k])
        print("40 steps, %s advection: %s rel-L2 %.2e" % (mode, k, r))
        assert r <= TOL, (k, r)


def test_advect_fast_mode_is_a_tolerance_mode():
    """tfl_set_advect_mode(FAST) on single operators: advectVel / advectScalar (maccormackOurs, eulerOurs) of a 3-D scene
    with obstacles differ from the exact mode by rounding only (rel-L2 <= 2e-6, max abs <= 1e-5 of the field scale) and
    the exact mode is unaffected by having been in fast mode (same context)."""
    import torch
    from fluidnet_amd import tfluids
    dev = torch.device("cuda:0")
    sc = scenes.make_scene((20, 24, 40), seed=5, vel_cells=0.7)
    U, fl, rho = (torch.from_numpy(sc[k]).to(dev) for k in ("U", "flags", "density"))

    def run():
        out = {}
        for m in ("maccormackOurs", "eulerOurs"):
            Uc, sc_ = U.clone(), rho.clone()
            tfluids.advectVel(0.1, Uc, fl, m, maccormackStrength=0.6)
            tfluids.advectScalar(0.1, sc_, U, fl, m, maccormackStrength=0.6)
            out["U_" + m], out["s_" + m] = Uc.cpu().numpy(), sc_.cpu().numpy()
        return out
    exact = run()
    assert tfluids.set_advect_mode(U, "fast") == "exact"
    try:
        fast = run()
    finally:
        assert tfluids.set_advect_mode(U, "exact") == "fast"
    again = run()
    for k in exact:
        assert np.array_equal(exact[k], again[k]), k
        r = scenes.rel_l2(fast[k], exact[k])
        scale = float(np.abs(exact[k]).max())
        frac = float((fast[k] != exact[k]).mean())
        print("fast vs exact %s: rel-L2 %.2e, max abs %.2e of %.2e, cells that differ %.1f%%"
              % (k, r, float(np.abs(fast[k] - exact[k]).max()), scale, 100 * frac))
        assert r <= 2e-6 and float(np.abs(fast[k] - exact[k]).max()) <= 1e-5 * max(scale, 1.0), (k, r)


def test_batched_rollout_with_output_div(oracle):
    """The training-side caller of simulate() (lib/run_epoch.lua:240-266): a batch of B = 4 DIFFERENT samples
    (own obstacles, own flow speed, hence own std(U) input scale: lib/model.lua:93-117 normalises per batch item),
    no plume BCs, stepped numFutureSteps = 4 times with outputDiv = true on the last step (advect + forces, no
    projection). Native step vs the numpy/C restatement."""
    import torch
    from fluidnet_amd import FluidNetModel
    from fluidnet_amd.simulate import simulate, simulate_native
    dev = torch.device("cuda:0")
    sc = scenes.make_scene((12, 20, 24), seed=77, vel_cells=0.6, B=4)
    for bi, f in enumerate((1.0, 0.35, 2.0, 0.05)):
        sc["U"][bi] *= f
    layers = S.default_3d_layers(seed=6)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=1.0, gravityScale=0,
                 vorticityConfinementAmp=1.5, simMethod="convnet")
    nb = dict(pDiv=sc["p"].copy(), UDiv=sc["U"].copy(), flags=sc["flags"].copy(), density=sc["density"].copy())
    ta, tb = _to_dev(nb, dev), _to_dev(nb, dev)
    model = FluidNetModel(layers, True)
    scales = []
    for i in range(4):
        last = i == 3
        S.simulate(oracle, mconf, nb, layers, output_div=last)
        simulate_native(None, mconf, ta, model, outputDiv=last)
        simulate(None, mconf, tb, model, outputDiv=last)
        scales.append([float(x) for x in ta["UDiv"].flatten(1).std(dim=1)])
    assert max(scales[0]) / min(scales[0]) > 5            # the four samples really are normalised differently
    for k in ("pDiv", "UDiv", "density"):
        assert torch.equal(ta[k], tb[k]), k
        for bi in range(4):
            r = scenes.rel_l2(ta[k][bi].cpu().numpy(), nb[k][bi])
            assert r <= TOL, (k, bi, r)
    # the last step skipped the projection: the velocity is NOT divergence-free, p is the previous step's
    from fluidnet_amd import tfluids
    div = torch.empty_like(ta["pDiv"])
    tfluids.velocityDivergenceForward(ta["UDiv"], ta["flags"], div)
    assert float(div.abs().max()) > 1e-3


@pytest.mark.parametrize("dims,stick", [((1, 33, 47), False), ((12, 18, 22), True)])
def test_set_wall_bcs_module_backward(oracle, dims, stick):
    """tfluids.SetWallBcs:updateGradInput (tfluids/set_wall_bcs.lua:50-66): gradInput[1] = mask * gradOutput with
    mask = setWallBcsForward(ones, flags)."""
    import torch
    from fluidnet_amd import tfluids
    dev = torch.device("cuda:0")
    sc = scenes.make_scene(dims, seed=91, B=3, stick=stick, empty_cells=True)
    rng = np.random.RandomState(5)
    g = rng.randn(*sc["U"].shape).astype(np.float32)
    mask = np.ones_like(sc["U"])
    oracle.setWallBcsForward(mask, sc["flags"])
    want = mask * g
    assert 0 < int((mask == 0).sum()) < mask.size
    tg, tf = torch.from_numpy(g).to(dev), torch.from_numpy(sc["flags"]).to(dev)
    got = tfluids.setWallBcsBackward(tf, tg)
    assert np.array_equal(got.cpu().numpy(), want) and torch.equal(tg.cpu(), torch.from_numpy(g))
    tfluids.setWallBcsBackward(tf, tg, tg)             # in place
    assert np.array_equal(tg.cpu().numpy(), want)


def test_grid_limits_are_reported_not_crashed():
    """the launch-grid limit (B*Z <= 65535: the batch is folded into gridDim.z) and the 32-bit
    cell-offset guard (3*Z*Y*X < 2^31) must surface as TfluidsError; just inside the limits the operators work."""
    import torch
    from fluidnet_amd import tfluids
    dev = torch.device("cuda:0")
    U = torch.zeros(512, 3, 128, 4, 8, device=dev)
    flags = torch.ones(512, 1, 128, 4, 8, device=dev)
    with pytest.raises(tfluids.TfluidsError, match="launch grid"):
        tfluids.setWallBcsForward(U, flags)
    U, flags = U[:255, :, :].contiguous(), flags[:255].contiguous()        # 255 * 128 = 32640
    flags[:, :, :, 0] = 2.0
    U.fill_(1.0)
    tfluids.setWallBcsForward(U, flags)
    # y-faces above the wall zeroed in every sample; rows further up untouched
    assert float(U[:, 1, :, 1].abs().max()) == 0.0 and float(U[:, :, :, 2:].min()) == 1.0
    del U, flags
    big = torch.empty(1, 1, 900, 900, 900, device=dev)                      # 729M cells: 3*N >= 2^31
    bigU = torch.empty(1, 3, 900, 900, 900, device=dev)
    with pytest.raises(tfluids.TfluidsError, match="32-bit"):
        tfluids.setWallBcsForward(bigU, big)


def test_512_cubed_step_runs():
    """The reference driver's largest resolution (fluid_net_3d_sim.lua:62: res up to 512): two whole steps at 512^3
    (134M cells, ~21 GB of state + workspace) stay finite, move the plume and hit no line-trace error path."""
    import torch
    from fluidnet_amd import FluidNetModel, tfluids
    from fluidnet_amd.simulate import createPlumeBCs, simulate_native
    dev = torch.device("cuda:0")
    res = 512
    flags = torch.empty(1, 1, res, res, res, device=dev)
    tfluids.emptyDomain(flags, True)
    batch = dict(pDiv=torch.zeros_like(flags), UDiv=torch.zeros(1, 3, res, res, res, device=dev), flags=flags,
                 density=torch.zeros_like(flags))
    createPlumeBCs(batch, [1.0], 4.0, 0.15)
    mconf = dict(dt=0.1, advectionMethod="maccormackOurs", maccormackStrength=0.6, buoyancyScale=8.0, gravityScale=0,
                 vorticityConfinementAmp=3.0, simMethod="convnet")
    model = FluidNetModel.default_3d(seed=1)
    for _ in range(2):
        simulate_native(None, mconf, batch, model)
    assert bool(torch.isfinite(batch["UDiv"]).all()) and bool(torch.isfinite(batch["pDiv"]).all())
    assert float(batch["density"][0, 0, :, 4:].sum()) > 0 and float(batch["UDiv"].abs().max()) > 0.5
    assert tfluids.traceErrors(batch["UDiv"]) == 0


@pytest.mark.parametrize("dims", [(1, 20, 24), (7, 10, 12)])
def test_nn_modules_forward_and_autograd(oracle, dims):
    """fluidnet_amd.modules = tfluids/{velocity_divergence,velocity_update,set_wall_bcs,flags_to_occupancy,
    volumetric_up_sampling_nearest}.lua as torch.nn.Modules: forward == the operator, autograd gradients == the oracle's
    backward operators fed the same gradOutput (bit-exact: the HIP backward ops are), gradU of VelocityUpdate is zero and
    flags get no gradient, as in the reference modules. (The reference checks its modules with nn.Jacobian,
    test_tfluids.lua:428-432, 483-488, 536-542; the adjoint property itself is in test_oracle.py.)"""
    import torch
    from fluidnet_amd import modules as M
    dev = torch.device("cuda:0")
    sc = scenes.make_scene(dims, seed=77, vel_cells=1.0, B=2, empty_cells=True)
    rng = np.random.RandomState(78)
    f, U, p = sc["flags"], sc["U"], sc["p"]
    tf = torch.from_numpy(f).to(dev)
    # VelocityDivergence
    tU = torch.from_numpy(U).to(dev).requires_grad_(True)
    div = M.VelocityDivergence()([tU, tf])
    want = np.zeros_like(p); oracle.velocityDivergenceForward(U, f, want)
    assert np.array_equal(div.detach().cpu().numpy(), want)
    go = rng.randn(*p.shape).astype(np.float32)
    div.backward(torch.from_numpy(go).to(dev))
    gU = np.zeros_like(U); oracle.velocityDivergenceBackward(U, f, go, gU)
    assert np.array_equal(tU.grad.cpu().numpy(), gU)
    # VelocityUpdate
    tU = torch.from_numpy(U).to(dev).requires_grad_(True)
    tp = torch.from_numpy(p).to(dev).requires_grad_(True)
    Un = M.VelocityUpdate()([tp, tU, tf])
    want = U.copy(); oracle.velocityUpdateForward(want, f, p)
    assert np.array_equal(Un.detach().cpu().numpy(), want)
    goU = rng.randn(*U.shape).astype(np.float32)
    Un.backward(torch.from_numpy(goU).to(dev))
    gP = np.zeros_like(p); oracle.velocityUpdateBackward(U, f, p, goU, gP)
    assert np.array_equal(tp.grad.cpu().numpy(), gP)
    assert float(tU.grad.abs().max()) == 0.0                      # velocity_update.lua:47
    # SetWallBcs: gradient = mask * gradOutput
    tU = torch.from_numpy(U).to(dev).requires_grad_(True)
    Ub = M.SetWallBcs()([tU, tf])
    want = U.copy(); oracle.setWallBcsForward(want, f)
    assert np.array_equal(Ub.detach().cpu().numpy(), want)
    Ub.backward(torch.from_numpy(goU).to(dev))
    mask = np.ones_like(U); oracle.setWallBcsForward(mask, f)
    assert np.array_equal(tU.grad.cpu().numpy(), mask * goU)
    # FlagsToOccupancy (no gradient) and VolumetricUpSamplingNearest
    f2 = scenes.make_scene(dims, seed=79, B=2)["flags"]            # fluid / obstacle only: the CPU reference raises otherwise
    occ = M.FlagsToOccupancy()(torch.from_numpy(f2).to(dev))
    want = np.zeros_like(f2); oracle.flagsToOccupancy(f2, want)
    assert np.array_equal(occ.cpu().numpy(), want) and not occ.requires_grad
    tx = torch.from_numpy(U).to(dev).requires_grad_(True)
    up = M.VolumetricUpSamplingNearest(2)(tx)
    B, C, Z, Y, X = U.shape
    want = np.zeros((B, C, 2 * Z, 2 * Y, 2 * X), np.float32); oracle.volumetricUpSamplingNearestForward(2, U, want)
    assert np.array_equal(up.detach().cpu().numpy(), want)
    g = rng.randn(*want.shape).astype(np.float32)
    up.backward(torch.from_numpy(g).to(dev))
    gi = np.zeros_like(U); oracle.volumetricUpSamplingNearestBackward(2, U, g, gi)
    assert np.array_equal(tx.grad.cpu().numpy(), gi)
