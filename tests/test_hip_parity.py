"""GPU parity tests proper: the HIP path (through the C ABI) against
 (1) the committed golden fixtures produced by the compiled reference, and
 (2) the oracle on fresh seeded scenes (sizes the oracle finishes in seconds).
Bars: operators whose arithmetic is a fixed sequence of fp32 (+ the reference's one fp64 step) are
required to be BIT-EXACT; the hard tolerance from BASELINE.json's north_star (velocity / scalar
rel-L2 <= 1e-5) is asserted as well and would be the fallback bar if a compiler ever reordered
a rounding. Run with: pytest -m gpu
"""
import glob
import os

import numpy as np
import pytest

import scenes
from golden.make_golden import run_ops

from flavours import child_env  # noqa: E402

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "g[23]d_*.npz")))
REL_L2_TOL = 1e-5  # BASELINE.json north_star: "velocity rel-L2 <= 1e-5 vs reference"


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need an MI355X"
    from hip_adapter import HipTfluids
    return HipTfluids()


def _compare(got, want, what):
    bad = []
    for k in sorted(want):
        g, w = got[k], want[k]
        rel = scenes.rel_l2(g, w)
        nmis = int((g != w).sum())
        assert np.isfinite(g).all(), (what, k)
        assert rel <= REL_L2_TOL, (what, k, rel)
        if nmis:
            bad.append((k, nmis, float(np.abs(g - w).max()), rel))
    assert not bad, "%s: not bit-exact: %s" % (what, bad)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_matches_golden(hip, path):
    z = np.load(path)
    sc = dict(flags=z["flags"], U=z["U"], density=z["density"], p=z["p"], dt=float(z["dt"]),
              is3d=z["U"].shape[1] == 3)
    want = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    _compare(run_ops(hip, sc), want, os.path.basename(path))
    assert hip.traceErrors() == 0


@pytest.mark.parametrize("dims,seed,kw", [
    ((1, 64, 64), 21, dict(vel_cells=3.0)),
    ((1, 128, 128), 22, dict(vel_cells=6.0, empty_cells=True, stick=True, noise=2.0)),
    ((1, 70, 130), 23, dict(vel_cells=2.0, B=3)),                 # ragged vs the 64x4 tile
    ((32, 32, 32), 24, dict(vel_cells=2.5)),
    ((20, 36, 68), 25, dict(vel_cells=5.0, empty_cells=True, stick=True, noise=2.0, B=2)),
    ((64, 64, 64), 26, dict(vel_cells=1.5, obstacles=False)),
    ((3, 3, 3), 27, dict(vel_cells=1.0, obstacles=False)),        # smallest legal 3-D grid
    ((1, 3, 3), 28, dict(vel_cells=1.0, obstacles=False)),
])
def test_hip_matches_oracle(hip, oracle, dims, seed, kw):
    sc = scenes.make_scene(dims, seed=seed, **kw)
    _compare(run_ops(hip, sc), run_ops(oracle, sc), str(dims))
    assert hip.traceErrors() == 0


@pytest.mark.parametrize("is3d", [False, True])
def test_hip_empty_domain_and_occupancy(hip, is3d):
    # test_tfluids.lua:675-753
    Z = 9 if is3d else 1
    for bnd in (1, 2, 3):
        flags = np.full((2, 1, Z, 11, 12), -3.0, np.float32)
        hip.emptyDomain(flags, is3d, bnd)
        assert np.array_equal(flags, scenes.empty_domain(2, Z, 11, 12, is3d, bnd))
    sc = scenes.make_scene((Z, 9, 10), seed=3)
    occ = np.full_like(sc["flags"], 5.0)
    hip.flagsToOccupancy(sc["flags"], occ)
    assert np.array_equal(occ, (sc["flags"] == 2).astype(np.float32))


@pytest.mark.parametrize("is3d", [False, True])
def test_hip_blur_and_signed_distance_match_oracle_bitwise(hip, oracle, is3d):
    """tfluids.rectangularBlur / signedDistanceField (generic/tfluids.cc:642-821; the criterion-side helpers of
    init.lua:578-613): same arithmetic in the same order, so bit-exact; odd sizes, several radii, B = 2, C = 3."""
    rng = np.random.RandomState(8)
    shape = (2, 3, 13, 19, 70) if is3d else (2, 3, 1, 37, 131)
    src = rng.uniform(-1, 1, shape).astype(np.float32)
    for rad in (1, 2, 5):
        d_h = rng.rand(*shape).astype(np.float32)
        d_o = np.empty_like(src)
        hip.rectangularBlur(src, rad, is3d, d_h)
        oracle.rectangularBlur(src, rad, is3d, d_o)
        assert np.array_equal(d_h, d_o), rad
    flags = np.ones((2, 1) + shape[2:], np.float32)
    flags[rng.rand(*flags.shape) < 0.02] = 2.0
    for rad in (1, 3):
        s_h = rng.rand(*flags.shape).astype(np.float32)
        s_o = np.empty_like(s_h)
        hip.signedDistanceField(flags, rad, is3d, s_h)
        oracle.signedDistanceField(flags, rad, is3d, s_o)
        assert np.array_equal(s_h, s_o), rad


# (2-D grids of up to 16 K cells with a fixed iteration count take the one-launch LDS solve of jacobi.hip: 1, 4 or 16 cells per
# thread -- 24 x 33, 64 x 64, the ragged 100 x 90 --; 130 x 140 is past it and iterates launches like the 3-D grid)
@pytest.mark.parametrize("dims", [(1, 64, 64), (24, 20, 28), (1, 24, 33), (1, 100, 90), (1, 130, 140)])
def test_hip_jacobi_matches_oracle(hip, oracle, dims):
    sc = scenes.make_scene(dims, seed=9, vel_cells=1.0, B=2)
    f, U = sc["flags"], sc["U"].copy()
    oracle.setWallBcsForward(U, f)
    div = np.zeros_like(sc["density"])
    oracle.velocityDivergenceForward(U, f, div)
    for iters in (1, 2, 20, 33):
        pa, pb = np.full_like(div, 3.0), np.full_like(div, -1.0)
        ra = oracle.solveLinearSystemJacobi(pa, f, div, sc["is3d"], 0.0, iters)
        rb = hip.solveLinearSystemJacobi(pb, f, div, sc["is3d"], 0.0, iters)
        assert np.array_equal(pa, pb), iters
        assert abs(ra - rb) <= 1e-5 * max(abs(ra), 1e-30), (ra, rb)
    # tolerance-terminated solve stops at the same iterate
    pa, pb = np.zeros_like(div), np.zeros_like(div)
    tol = 0.5 * float(oracle.solveLinearSystemJacobi(pa.copy(), f, div, sc["is3d"], 0.0, 10))
    ra = oracle.solveLinearSystemJacobi(pa, f, div, sc["is3d"], tol, 5000)
    rb = hip.solveLinearSystemJacobi(pb, f, div, sc["is3d"], tol, 5000)
    assert np.array_equal(pa, pb) and ra < tol and rb < tol


def test_hip_argument_errors(hip):
    """Same failures the Lua asserts raise (init.lua:99-120), surfaced as TfluidsError."""
    import torch
    from fluidnet_amd import tfluids, TfluidsError
    dev = hip.dev
    flags = torch.ones(1, 1, 1, 8, 8, device=dev)
    U3 = torch.zeros(1, 3, 1, 8, 8, device=dev)
    with pytest.raises(TfluidsError):
        tfluids.setWallBcsForward(torch.zeros(1, 2, 2, 8, 8, device=dev), torch.ones(1, 1, 2, 8, 8, device=dev))
    with pytest.raises(TfluidsError):
        tfluids.advectVel(0.1, torch.zeros(1, 2, 1, 8, 9, device=dev), flags)
    with pytest.raises(TfluidsError):
        tfluids.advectScalar(0.1, torch.zeros(1, 1, 1, 8, 8, device=dev), torch.zeros(1, 2, 1, 8, 8, device=dev),
                             flags, "noSuchMethod")
    with pytest.raises(TfluidsError):
        tfluids.advectVel(0.1, torch.zeros(1, 2, 1, 8, 8), torch.ones(1, 1, 1, 8, 8))  # CPU tensors
    del U3


@pytest.mark.parametrize("dims,seed,kw", [((1, 70, 130), 61, dict(vel_cells=3.0)), ((20, 36, 68), 62, dict(vel_cells=4.0, B=2)),
                                          ((33, 17, 40), 63, dict(vel_cells=0.8))])
def test_maccormack_large_displacements_and_ragged_grids(hip, oracle, dims, seed, kw):
    """maccormackOurs with back-traces of several cells (multi-step line traces that cross obstacles and the domain
    wall) on grids ragged against the 64x4 / vec4 thread tiles, B = 2: bit-exact against the C oracle."""
    sc = scenes.make_scene(dims, seed=seed, **kw)
    for op, fld in (("advectScalar", "density"), ("advectVel", "U")):
        a, b = sc[fld].copy(), sc[fld].copy()
        if op == "advectScalar":
            hip.advectScalar(sc["dt"], a, sc["U"], sc["flags"], "maccormackOurs")
            oracle.advectScalar(sc["dt"], b, sc["U"], sc["flags"], "maccormackOurs")
        else:
            hip.advectVel(sc["dt"], a, sc["flags"], "maccormackOurs")
            oracle.advectVel(sc["dt"], b, sc["flags"], "maccormackOurs")
        assert np.array_equal(a, b), (dims, op, int((a != b).sum()))
    assert hip.traceErrors() == 0


@pytest.mark.parametrize("dims,seed,vel,B", [((6, 20, 200), 71, 0.3, 1), ((5, 13, 129), 72, 0.9, 2), ((9, 22, 70), 73, 2.5, 1),
                                             ((3, 9, 66), 74, 0.5, 1)])
def test_advectvel_fast_path_tiles_and_flag_words(hip, oracle, dims, seed, vel, B):
    """The LDS-tiled fast-path advectVel kernels (advect_vel3.hip) where their special cases live: grids with interior and
    edge 64x4 tiles, displacements below / around / above the fast path's 0.99-cell limit, flag words that are fluid but
    not the plain TypeFluid word (fluid | inflow = 9, fluid | open = 33: the fast path must hand those lanes to the generic
    code), the thinnest grid the tile takes (Z = 3), B = 2; eulerOurs (single pass) and maccormackOurs: bit-exact."""
    sc = scenes.make_scene(dims, seed=seed, vel_cells=vel, B=B)
    f = sc["flags"]
    rng = np.random.RandomState(seed)
    fl = np.flatnonzero(f == 1.0)
    pick = rng.choice(fl, size=max(1, fl.size // 40), replace=False)
    f.reshape(-1)[pick] = rng.choice([9.0, 33.0], size=pick.size)
    for m in ("maccormackOurs", "eulerOurs"):
        a, b = sc["U"].copy(), sc["U"].copy()
        hip.advectVel(sc["dt"], a, f, m)
        oracle.advectVel(sc["dt"], b, f, m)
        assert np.array_equal(a, b), (dims, m, int((a != b).sum()))
    assert hip.traceErrors() == 0


@pytest.mark.parametrize("dims,seed", [((1, 48, 40), 81), ((14, 18, 22), 82)])
def test_hip_sample_outside_fluid_and_explicit_dst(hip, oracle, dims, seed):
    """Non-default wrapper arguments of init.lua:89-219: sampleOutsideFluid=true (temperature-like fields)
    and explicit sDst / UDst (no copy-back) -- bit-exact against the oracle for every method."""
    sc = scenes.make_scene(dims, seed=seed, vel_cells=3.0, stick=True)
    f, dt = sc["flags"], sc["dt"]
    for m in ("maccormackOurs", "eulerOurs", "rk2Ours", "rk3Ours", "euler", "maccormack"):
        a, b = sc["density"].copy(), sc["density"].copy()
        hip.advectScalar(dt, a, sc["U"], f, m, None, True, 0.9)
        oracle.advectScalar(dt, b, sc["U"], f, m, None, True, 0.9)
        assert np.array_equal(a, b), ("outside", m)
        s0 = sc["density"].copy()
        da, db = np.full_like(s0, 9.0), np.full_like(s0, -9.0)
        hip.advectScalar(dt, s0, sc["U"], f, m, da, False, 0.5)
        oracle.advectScalar(dt, sc["density"].copy(), sc["U"], f, m, db, False, 0.5)
        assert np.array_equal(da, db) and np.array_equal(s0, sc["density"]), ("sDst", m)
        U0 = sc["U"].copy()
        ua, ub = np.full_like(U0, 9.0), np.full_like(U0, -9.0)
        hip.advectVel(dt, U0, f, m, ua, 0.5)
        oracle.advectVel(dt, sc["U"].copy(), f, m, ub, 0.5)
        assert np.array_equal(ua, ub) and np.array_equal(U0, sc["U"]), ("UDst", m)
    assert hip.traceErrors() == 0


@pytest.mark.parametrize("res", [128, 256])
def test_hip_fullsize_properties(res):
    """BASELINE.json's full sizes (128^3, 256^3), where the CPU oracle takes minutes: size-independent
    properties of the operators instead. All exact (bitwise) statements."""
    import torch
    from fluidnet_amd import tfluids
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    flags = tfluids.emptyDomain(torch.empty(1, 1, res, res, res, device=dev), True)
    assert float(flags.sum()) == float(res ** 3 + 6 * res * res - 12 * res + 8)       # 2 on the shell, 1 inside
    occ = torch.empty_like(flags)
    tfluids.flagsToOccupancy(flags, occ)
    assert torch.equal(occ, flags - 1)
    plane = torch.rand(res, res, generator=g)
    s = (plane[None, :, :] * torch.linspace(0.5, 1.5, res)[:, None, None]).view(1, 1, res, res, res).contiguous().to(dev)
    interior = torch.zeros_like(s, dtype=torch.bool)
    interior[..., 1:-1, 1:-1, 1:-1] = True
    # (1) zero velocity: MacCormack advection is the identity on the interior and zeroes the border shell
    U0 = torch.zeros(1, 3, res, res, res, device=dev)
    out = torch.empty_like(s)
    for m in ("maccormackOurs", "eulerOurs", "maccormack"):
        tfluids.advectScalar(0.1, s, U0, flags, m, out)
        assert torch.equal(out, torch.where(interior, s, torch.zeros_like(s))), m
    # (2) uniform whole-cell displacement: the result is the field shifted by exactly that many cells
    #     (trilinear weights degenerate to 1/0, the MacCormack correction cancels, the clamp is inactive)
    U1 = torch.zeros(1, 3, res, res, res, device=dev)
    U1[:, 0] = 2.0            # dt = 0.5 -> one cell in +x per step
    tfluids.advectScalar(0.5, s, U1, flags, "maccormackOurs", out)
    core = (slice(None), slice(None), slice(4, -4), slice(4, -4), slice(4, -4))
    shifted = torch.roll(s, shifts=1, dims=4)
    assert torch.equal(out[core], shifted[core])
    Uout = torch.empty_like(U1)
    tfluids.advectVel(0.5, U1, flags, "maccormackOurs", Uout)
    assert torch.equal(Uout[core], U1[core])                      # a uniform flow advects itself unchanged
    # (3) wall BCs are idempotent; divergence of a curl-free uniform flow is zero away from walls
    U = torch.rand(1, 3, res, res, res, generator=g).to(dev)
    a = U.clone()
    tfluids.setWallBcsForward(a, flags)
    b = a.clone()
    tfluids.setWallBcsForward(b, flags)
    assert torch.equal(a, b) and not torch.equal(a, U)
    div = torch.empty_like(s)
    tfluids.velocityDivergenceForward(U1, flags, div)
    assert float(div[core].abs().max()) == 0.0
    # (4) velocityUpdate with a linear pressure p = c*x subtracts exactly c from u_x on fluid/fluid faces
    p = (torch.arange(res, dtype=torch.float32) * 0.25).view(1, 1, 1, 1, res).expand(1, 1, res, res, res).contiguous().to(dev)
    Uu = torch.zeros(1, 3, res, res, res, device=dev)
    tfluids.velocityUpdateForward(Uu, flags, p)
    assert torch.equal(Uu[:, 0][(slice(None), slice(2, -2), slice(2, -2), slice(2, -1))],
                       torch.full((1, res - 4, res - 4, res - 3), -0.25, device=dev))
    assert float(Uu[:, 1:].abs().max()) == 0.0
    assert tfluids.traceErrors(s) == 0


from backward_cases import CASES as BWD_CASES, run_backward_ops  # noqa: E402


@pytest.mark.parametrize("dims,seed,kw", BWD_CASES)
def test_hip_backward_ops_bit_exact(hip, oracle, dims, seed, kw):
    """velocityDivergenceBackward / velocityUpdateBackward / volumetricUpSamplingNearest{Forward,Backward}: the
    gather kernels reproduce the reference's serial summation order bit for bit."""
    a, b = run_backward_ops(hip, dims, seed, **kw), run_backward_ops(oracle, dims, seed, **kw)
    for k in sorted(b):
        assert np.array_equal(a[k], b[k]), (k, int((a[k] != b[k]).sum()))


_VEC4_CASES = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import scenes
from hip_adapter import HipTfluids
from oracle.oracle import OracleTfluids
from oracle import simulate_np as S
import torch
from fluidnet_amd import FluidNetModel
hip, ora = HipTfluids(), OracleTfluids()
# rows wider than one 32-lane x 4-cell segment (X > 128): the x-1 / x+4 taps at a segment end come from memory,
# everywhere else from the neighbouring lane; X = 8, 24 exercise the narrow (8 / 16 lane) launch shapes
# (seeds chosen so that no back-trace runs into one of the reference's THError paths: the oracle raises there)
for dims, seed, B in [((5, 9, 136), 91, 1), ((1, 11, 264), 110, 2), ((3, 6, 132), 93, 1), ((4, 7, 8), 94, 1),
                      ((1, 9, 24), 95, 1), ((6, 8, 260), 93, 1), ((6, 8, 260), 106, 1)]:
    sc = scenes.make_scene(dims, seed=seed, B=B, vel_cells=1.5, stick=(seed %% 2 == 0))
    f, dt = sc["flags"], sc["dt"]
    a, b = sc["density"].copy(), sc["density"].copy()
    hip.advectScalar(dt, a, sc["U"], f, "maccormackOurs"); ora.advectScalar(dt, b, sc["U"], f, "maccormackOurs")
    assert np.array_equal(a, b), (dims, "advectScalar/minmax3", int((a != b).sum()))
    a, b = sc["U"].copy(), sc["U"].copy()
    hip.vorticityConfinement(a, f, 0.7); ora.vorticityConfinement(b, f, 0.7)
    assert np.array_equal(a, b), (dims, "vorticityConfinement", int((a != b).sum()))
    a, b = sc["U"].copy(), sc["U"].copy()
    g = [0.3, -1.0, 0.2] if sc["is3d"] else [0.3, -1.0, 0.0]
    hip.addBuoyancy(a, f, sc["density"], g, dt); ora.addBuoyancy(b, f, sc["density"], g, dt)
    assert np.array_equal(a, b), (dims, "addBuoyancy", int((a != b).sum()))
    if seed %% 2 == 1:   # the model path has no stick cells (flagsToOccupancy rejects them)
        if sc["is3d"]:
            layers = S.default_3d_layers(seed=seed)
            dev = torch.device("cuda:0")
            tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
            pm, Um = FluidNetModel(layers, True).forward([tp, tU, tf])
            p_ref, U_ref = S.model_forward(ora, layers, sc["p"], sc["U"], sc["flags"])
            assert scenes.rel_l2(pm.cpu().numpy(), p_ref) <= 1e-5 and scenes.rel_l2(Um.cpu().numpy(), U_ref) <= 1e-5, dims
print("VEC4_OK")
"""


@pytest.mark.parametrize("no_vec4", [False, True])
def test_vec4_kernels_and_their_fallbacks_match_the_oracle(no_vec4):
    """The four-cells-per-thread kernels (tfl_vec4.hpp: k_minmax3, k_curl, k_confine, k_add_buoyancy,
    k_bcs_div_stats, k_project) on rows wider than one lane segment and on the narrow launch shapes, and -- with
    TFL_NO_VEC4=1 -- the one-cell-per-thread kernels they replace, both against the oracle. Child processes:
    the switch is read from the environment once."""
    import subprocess, sys
    code = _VEC4_CASES % (os.path.dirname(HERE), HERE)
    env = dict(os.environ)
    env.pop("TFL_NO_VEC4", None)
    if no_vec4:
        env = child_env(env, {"TFL_NO_VEC4": "1"})      # (a switch of the EXPERIMENTS flavour of the library)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "VEC4_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


# ---- solveLinearSystemPCG ---------------------------------------------------------------------
@pytest.mark.parametrize("dims,seed,split,B,tol", [((1, 24, 28), 3, False, 1, 1e-5), ((9, 11, 13), 4, False, 2, 1e-5),
                                                  ((1, 30, 34), 5, True, 1, 1e-5), ((8, 10, 16), 6, True, 1, 1e-5),
                                                  ((24, 20, 36), 8, False, 1, 1e-4), ((1, 96, 128), 9, True, 1, 1e-4),
                                                  # 3 strips x 5 slabs of the pipelined wavefront sweeps (64 rows x 8 planes each)
                                                  ((36, 134, 22), 10, True, 1, 1e-4)])
def test_hip_pcg_matches_the_oracle_and_the_reference_properties(hip, oracle, dims, seed, split, B, tol):
    """The matrix-free device PCG (pcg.hip) against the reference's own PCG host function compiled for the host
    (oracle/_ref/libtfluids_ref_pcg.so; cuSPARSE / cuBLAS primitives restated) and its bit-equal C restatement:
    same converged pressure for all three preconditioners, plus what test_tfluids.lua:836-906 asserts (residual
    below 2 tol, no NaN, velocityUpdate leaves no divergence) and the component rules (zero outside the fluid, a
    one-cell component untouched)."""
    # (fp32 CG stalls near 1e-6 |rhs|: the two larger grids get a smaller velocity and a looser tol)
    from oracle import ref as refmod
    sc, f, U, div = scenes.pcg_problem(oracle, dims, seed, split=split, B=B, vel_cells=2.0 if tol < 5e-5 else 0.3)
    for pc in ("none", "ilu0", "ic0"):
        pa = np.random.RandomState(2).rand(*div.shape).astype(np.float32)
        pb = pa.copy()
        ra = hip.solveLinearSystemPCG(pa, f, div, sc["is3d"], tol, 1000, pc)
        rb = oracle.solveLinearSystemPCG(pb, f, div, sc["is3d"], tol, 1000, pc)
        if refmod.pcg_available():
            # the checker proper: the reference's own host function (generic/tfluids.cu:864-1759) compiled for the host
            # (oracle/ref_pcg.cc); the restatement is bit-equal to it (tests/test_oracle.py), asserted here again
            pr = np.random.RandomState(2).rand(*div.shape).astype(np.float32)
            rr = refmod.RefTfluids().solveLinearSystemPCG(pr, f, div, sc["is3d"], tol, 1000, pc)
            assert np.array_equal(pr, pb) and rr == rb, pc
        assert ra < 2 * tol and rb < 2 * tol and np.isfinite(pa).all(), (pc, ra, rb)
        scale = max(np.abs(pb).max(), 1e-6)
        assert np.abs(pa - pb).max() < max(5e-5 * scale, 50 * tol), (pc, np.abs(pa - pb).max(), scale)
        assert np.all(pa[f != 1.0] == 0.0)
        Un = U.copy()
        hip.velocityUpdateForward(Un, f, pa)
        d2 = np.zeros_like(div)
        hip.velocityDivergenceForward(Un, f, d2)
        assert np.abs(d2).max() < max(3e-5 * max(1.0, np.abs(div).max()), 2 * tol), (pc, np.abs(d2).max())


_PCG_FALLBACK = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import scenes
from hip_adapter import HipTfluids
from oracle.oracle import OracleTfluids
hip, ora = HipTfluids(), OracleTfluids()
for dims, seed, split in (((12, 20, 24), 11, False), ((36, 134, 22), 10, True)):
    sc, f, U, div = scenes.pcg_problem(ora, dims, seed, split=split, B=1, vel_cells=0.3)
    for pc in ("ilu0", "ic0"):
        pa = np.zeros_like(div); pb = np.zeros_like(div)
        ra = hip.solveLinearSystemPCG(pa, f, div, True, 1e-4, 1000, pc)
        rb = ora.solveLinearSystemPCG(pb, f, div, True, 1e-4, 1000, pc)
        scale = max(np.abs(pb).max(), 1e-6)
        assert ra < 2e-4 and np.abs(pa - pb).max() < max(5e-5 * scale, 5e-3), (dims, pc, ra, np.abs(pa - pb).max())
print("PCG_PATH_OK")
"""


@pytest.mark.parametrize("mode", ["wavefronts", "hyperplanes", "timeout", "chunks"])
def test_pcg_triangular_solves_both_schedules(mode):
    """IC(0) / ILU(0) on 3-D grids: the pipelined-wavefront sweeps (default) and, with TFL_PCG_HYPERPLANES=1, the
    one-launch-per-hyperplane sweeps they replace (still the path of 2-D grids and the fallback), both against
    the oracle; "timeout": the wavefront sweeps report that a sub-box never saw its predecessor (pretended:
    TFL_WF_TEST_TIMEOUT) and the solve must come out right through the automatic repeat with hyperplane sweeps;
    "chunks": at most 4 sub-boxes per launch (TFL_WF_MAX_BLOCKS), the way grids with more sub-boxes than fit the GPU
    at once run, one range of slabs per launch. Child processes: the switches are read from the environment once."""
    import subprocess, sys
    code = _PCG_FALLBACK % (os.path.dirname(HERE), HERE)
    env = dict(os.environ)
    env.pop("TFL_PCG_HYPERPLANES", None)
    env.pop("TFL_WF_TEST_TIMEOUT", None)
    if mode == "hyperplanes":
        env["TFL_PCG_HYPERPLANES"] = "1"
    env.pop("TFL_WF_MAX_BLOCKS", None)
    if mode == "timeout":
        env = child_env(env, {"TFL_WF_TEST_TIMEOUT": "1"})      # (test hooks of the EXPERIMENTS flavour of the library)
    if mode == "chunks":
        env = child_env(env, {"TFL_WF_MAX_BLOCKS": "4"})
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "PCG_PATH_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


_ADV_VARIANTS = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import scenes
from hip_adapter import HipTfluids
from oracle.oracle import OracleError, OracleTfluids
hip, ora = HipTfluids(), OracleTfluids()
rng = np.random.RandomState(%d)
done = 0
for dims in ((7, 13, 70), (9, 22, 129), (5, 9, 200), (8, 16, 64), (11, 10, 36), (6, 21, 133), (4, 12, 66), (13, 8, 20)):
    for rep in range(2):
        seed = int(rng.randint(1 << 30))
        kw = dict(B=1 + rep, vel_cells=float(rng.choice([0.3, 1.0, 2.5, 4.0])), stick=bool(rep), empty_cells=bool(rep) and dims[1] >= 10)
        sc = scenes.make_scene(dims, seed=seed, **kw)
        f, dt = sc["flags"], sc["dt"]
        if rep:           # fluid cells whose flag word is not the plain TypeFluid word: the generic path next to fast lanes
            fl = np.flatnonzero(f == 1.0)
            pick = rng.choice(fl, size=max(1, fl.size // 50), replace=False)
            f.reshape(-1)[pick] = rng.choice([9.0, 33.0], size=pick.size)
        try:
            for m in ("maccormackOurs", "eulerOurs"):
                a, b = sc["density"].copy(), sc["density"].copy()
                hip.advectScalar(dt, a, sc["U"], f, m); ora.advectScalar(dt, b, sc["U"], f, m)
                assert np.array_equal(a, b), ("advectScalar", m, dims, seed, int((a != b).sum()))
                a, b = sc["U"].copy(), sc["U"].copy()
                hip.advectVel(dt, a, f, m); ora.advectVel(dt, b, f, m)
                assert np.array_equal(a, b), ("advectVel", m, dims, seed, int((a != b).sum()))
            done += 1
        except OracleError:
            pass          # a back-trace ran into one of the reference's THError paths: not a comparable scene
assert done >= 12, done
print("ADV_VARIANTS_OK", done)
"""


@pytest.mark.parametrize("env", [{"TFL_VEL3_KZ": "2", "TFL_SCAL3_TZ": "14"}, {"TFL_VEL3_KZ": "2", "TFL_SCAL3_TZ": "12"},
                                 {"TFL_VEL3_KZ": "1", "TFL_SCAL3_TZ": "1"}, {"TFL_ADVECT_GATHER": "1"},
                                 {"TFL_SCAL3_MARCH": "1", "TFL_SCAL3M_CZ_A": "3", "TFL_SCAL3M_CZ_B": "2"},
                                 {"TFL_SCAL3_MARCH": "1", "TFL_SCAL3M_CZ_A": "1", "TFL_SCAL3M_CZ_B": "64"}, {"TFL_SCAL3_MARCH": "1"}],
                         ids=["big-grid-defaults", "kz2-scal-1x2", "kz1-scal-1x1", "gather-kernels", "marched-short-chunks",
                              "marched-1-and-64", "marched-scalar-kernels"])
def test_advection_kernel_variants_are_bit_exact(env):
    """The advection kernels pick their block shape from the grid size: from 6 M cells per batch item on (256^3, BASELINE
    config 5) advectVel runs two planes per block (advect_vel3.inc, kz2) and advectScalar's pass A four planes per thread
    (advect_scalar3.hip <1, 4>). The golden / oracle tests above run small grids and would never see those kernels, so
    the variants are FORCED here (TFL_VEL3_KZ, TFL_SCAL3_TZ; read once per process: child processes) onto small ragged
    grids -- partial 64 x 4 tiles, odd plane counts against the 2- and 4-plane blocks, B = 2, obstacles, stick and empty
    cells, exotic fluid words -- and held to the oracle bit for bit like the defaults. Also: the round-2 gather kernels
    (the fallback of grids beyond the 32-bit offset guard), which no default-size test reaches any more either."""
    import subprocess, sys
    code = _ADV_VARIANTS % (os.path.dirname(HERE), HERE, 4242)
    e = dict(os.environ)
    for k in ("TFL_VEL3_KZ", "TFL_SCAL3_TZ", "TFL_ADVECT_GATHER", "TFL_ADVECT_MODE", "TFL_SCAL3M_CZ_A", "TFL_SCAL3M_CZ_B", "TFL_SCAL3_MARCH"):
        e.pop(k, None)
    e = child_env(e, env)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ADV_VARIANTS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("env", [{"TFL_XCD_ORDER": "0"}, {"TFL_XCD_ORDER": "1"}, {"TFL_XCD_RUN": "3"}],
                         ids=["hardware-order", "one-run-per-xcd", "runs-of-3-tiles"])
def test_block_order_variants_are_bit_exact(env):
    """The halo-reading kernels (advectScalar's passes, k_curl / k_confine, the fused confinement) map blocks to tiles through
    tfl_device.hpp block_tile: runs of consecutive tiles per XCD (default: an eighth of a plane), decoded with exact
    multiply-high divisions. Any run length must give the same fields: the oracle / ragged-grid / fused-confinement cases run
    again in child processes (the switches are read once) with the hardware's order, one run per XCD, and runs of 3 tiles --
    which leaves launches whose block count is no multiple of 8 x 3 with a partial round."""
    import subprocess, sys
    e = dict(os.environ)
    for k in ("TFL_XCD_ORDER", "TFL_XCD_RUN"):
        e.pop(k, None)
    e = child_env(e, env)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                          "test_hip_matches_oracle or test_maccormack_large_displacements_and_ragged_grids or "
                          "test_fused_vorticity_confinement_equals_the_two_launch_form"], env=e, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_hip_pcg_errors_and_defaults(hip, oracle):
    from fluidnet_amd import TfluidsError
    sc, f, U, div = scenes.pcg_problem(oracle, (1, 12, 12), 7)
    with pytest.raises(TfluidsError):
        hip.solveLinearSystemPCG(np.zeros_like(div), f, div, False, 1e-5, 100, "cholesky")
    fb = f.copy()
    fb[0, 0, 0, 0, 5] = 1.0          # a fluid cell on the domain border: the reference raises (tfluids.cu:1083-1091)
    with pytest.raises(TfluidsError):
        hip.solveLinearSystemPCG(np.zeros_like(div), fb, div, False, 1e-5, 100, "none")
    # maxIter is honoured (iter <= maxIter: maxIter + 1 iterations) and the residual reported is the one reached
    p1 = np.zeros_like(div)
    r1 = hip.solveLinearSystemPCG(p1, f, div, False, 1e-12, 3, "none")
    p2 = np.zeros_like(div)
    r2 = oracle.solveLinearSystemPCG(p2, f, div, False, 1e-12, 3, "none")
    assert abs(r1 - r2) <= 1e-3 * r2 and np.abs(p1 - p2).max() < 1e-4 * np.abs(p2).max()
    # all-obstacle grid: nothing to solve, p zeroed, residual -inf like the reference's initial value
    fo = np.full_like(f, 2.0)
    p3 = np.ones_like(div)
    r3 = hip.solveLinearSystemPCG(p3, fo, div, False, 1e-5, 10, "ic0")
    assert r3 == -np.inf and not p3.any()


@pytest.mark.parametrize("dims,seed,split", [((1, 30, 34), 5, True), ((8, 10, 16), 6, True), ((20, 24, 40), 13, False)])
def test_hip_normalize_pressure_mean(hip, oracle, dims, seed, split):
    """generic/tfluids.cc:845-925 on the device (labelling shared with the PCG solver) vs the oracle: equal to
    rounding (fp64 component sums on both sides, different orders), non-fluid cells bit-untouched."""
    sc, f, U, div = scenes.pcg_problem(oracle, dims, seed, split=split, B=2)
    p0 = (np.random.RandomState(seed).randn(*div.shape) * 3 + 5).astype(np.float32)
    a, b = p0.copy(), p0.copy()
    hip.normalizePressureMean(a, f, sc["is3d"])
    oracle.normalizePressureMean(b, f, sc["is3d"])
    assert np.abs(a - b).max() < 2e-6 * np.abs(p0).max()
    assert np.array_equal(a[f != 1.0], p0[f != 1.0])


@pytest.mark.parametrize("dims,seed,B", [((20, 36, 68), 91, 1), ((5, 9, 70), 92, 2), ((33, 16, 64), 93, 1), ((3, 8, 8), 94, 1),
                                          ((40, 70, 130), 95, 1)])
def test_fused_vorticity_confinement_equals_the_two_launch_form(oracle, dims, seed, B):
    """tfl_vorticityConfinementFrom on a 3-D grid = one z-marched launch with curl / |curl| in LDS (vorticity.hip
    k_vort_fused): bit-equal to the in-place operator (itself bit-equal to the reference, test_hip_matches_golden) and to
    the oracle, on ragged grids (partial 64 x 8 columns), more than one batch item, obstacles and empty cells; a 2-D grid
    takes the copy + two-launch route of the same entry point."""
    import torch
    from fluidnet_amd import tfluids
    dev = torch.device("cuda:0")
    for is3d in (True, False):
        d = dims if is3d else (1, dims[1], dims[2])
        sc = scenes.make_scene(d, seed=seed, vel_cells=2.0, B=B, empty_cells=True)
        U, fl = torch.from_numpy(sc["U"]).to(dev), torch.from_numpy(sc["flags"]).to(dev)
        want = U.clone()
        tfluids.vorticityConfinement(want, fl, 0.7)
        got = torch.full_like(U, float("nan"))
        tfluids.vorticityConfinement(got, fl, 0.7, USrc=U)
        assert torch.equal(got, want), (d, int((got != want).sum()))
        ref = sc["U"].copy()
        oracle.vorticityConfinement(ref, sc["flags"], 0.7)
        assert np.array_equal(got.cpu().numpy(), ref), d
        assert torch.equal(U, torch.from_numpy(sc["U"]).to(dev))          # the source is left alone
        if is3d and d[0] >= 5:
            # under a compute window (tfl_set_z_window: two plane runs in one launch, what a z-slab rank's phases use) the
            # planes of the window -- and only those -- are written, with the whole-array values. The fused kernels take the
            # window as it is; the two-launch route (round 6) runs its curl pass on the window widened by two planes
            # either way -- the scratch arrays are poisoned first, so a confinement that tapped a curl plane outside that range
            # would show; a grid that only the one-cell kernels take is refused there instead of answered wrongly.
            lib, ctx = tfluids._context(U)
            Z = d[0]
            a0, a1, b0, b1 = 1, 3, Z - 2, Z
            win = torch.full_like(U, 7.0)
            for t in tfluids._tmp.values():
                if t is not None:
                    t.fill_(float("nan"))
            assert lib.tfl_set_z_window(ctx, a0, a1, b0, b1) == 0
            refused = False
            try:
                tfluids.vorticityConfinement(win, fl, 0.7, USrc=U)
            except tfluids.TfluidsError as e:
                refused = True
                assert "z-window" in str(e)
            finally:
                assert lib.tfl_set_z_window(ctx, 0, 0, 0, 0) == 0
            if refused:
                assert os.environ.get("TFL_VORT_FUSED") != "1" and d[2] % 4 != 0, d
                continue
            inside = torch.zeros(Z, dtype=torch.bool, device=dev)
            inside[a0:a1] = True; inside[b0:b1] = True
            assert torch.equal(win[:, :, inside], want[:, :, inside]), d
            assert bool((win[:, :, ~inside] == 7.0).all()), d


@pytest.mark.parametrize("env", [{"TFL_VORT_FUSED": "1", "TFL_VORT_PIPE": "1"}, {"TFL_VORT_FUSED": "1", "TFL_VORT_PIPE": "1", "TFL_VORT_CZ": "5"},
                                 {"TFL_VORT_FUSED": "1", "TFL_VORT_PIPE": "0"}, {"TFL_VORT_FUSED": "1", "TFL_VORT_PIPE": "0", "TFL_VORT_CZ": "5"},
                                 {"TFL_VORT_FUSED": "1", "TFL_XCD_ORDER": "0"}, {"TFL_VORT_FUSED": "1", "TFL_VORT_TILE": "32"},
                                 {"TFL_VORT_FUSED": "1", "TFL_VORT_TILE": "32", "TFL_VORT_CZ": "5"}],
                         ids=["pipelined", "pipelined-short-chunks", "three-barrier", "three-barrier-short-chunks", "pipelined-hardware-block-order",
                              "pipelined-32x16-tiles", "pipelined-32x16-short-chunks"])
def test_fused_vorticity_kernel_variants(env):
    """The fused confinement has two kernels -- k_vort_pipe (software-pipelined, one barrier per plane step; chosen when the
    default where the device holds its block) and k_vort_fused -- and tfl_vorticityConfinementFrom takes the fused route only
    from 2 M cells per batch item on (below, the two launches: the cases of the test above as they stand). The switches are
    read once per process: the cases run again in child processes with the fused route forced (TFL_VORT_FUSED=1) onto the small
    ragged grids, each kernel in turn, also with chunks shorter than the pipeline (5 planes against 9 / 6 steps of fill); there
    the test also runs the operator under a two-run z-window (the slab step's form of the call)."""
    import subprocess, sys
    e = dict(os.environ)
    for k in ("TFL_VORT_PIPE", "TFL_VORT_CZ", "TFL_VORT_FUSED", "TFL_XCD_ORDER", "TFL_VORT_TILE"):
        e.pop(k, None)
    e = child_env(e, env)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          "-k", "test_fused_vorticity_confinement_equals_the_two_launch_form"], env=e, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_round4_abi_additions(hip):
    """tfl_set_advect_mode / tfl_get_advect_mode, tfl_stream_copy and the tfl_comm size check (ABI 3): argument errors
    come back as codes, the copy copies, and a slab step refuses a transport struct that does not declare its size."""
    import ctypes
    import torch
    from fluidnet_amd import _lib, tfluids, TfluidsError
    dev = hip.dev
    lib, ctx = tfluids._context(torch.zeros(1, device=dev))
    assert lib.tfl_abi_version() == 4
    assert lib.tfl_get_advect_mode(ctx) == 0
    assert lib.tfl_set_advect_mode(ctx, 7) != 0 and b"unknown mode" in lib.tfl_last_error(ctx)
    assert lib.tfl_set_advect_mode(ctx, 1) == 0 and lib.tfl_get_advect_mode(ctx) == 1
    assert lib.tfl_set_advect_mode(ctx, 0) == 0 and lib.tfl_get_advect_mode(ctx) == 0
    with pytest.raises(TfluidsError):
        tfluids.set_advect_mode(torch.zeros(1, device=dev), "quick")
    # stream copy: 16-byte aligned, a multiple of four floats
    a = torch.randn(1 << 16, device=dev)
    b = torch.zeros_like(a)
    assert lib.tfl_stream_copy(ctx, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), a.numel()) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert lib.tfl_stream_copy(ctx, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), 6) != 0
    assert lib.tfl_stream_copy(ctx, ctypes.c_void_p(b.data_ptr() + 4), ctypes.c_void_p(a.data_ptr()), 8) != 0
    # vorticityConfinementFrom refuses an aliased destination
    U = torch.zeros(1, 3, 6, 8, 8, device=dev)
    with pytest.raises(TfluidsError):
        tfluids.vorticityConfinement(U, torch.ones(1, 1, 6, 8, 8, device=dev), 0.5, USrc=U)
    # a transport struct without its size is refused by the slab step instead of being read past its end
    import bench
    from fluidnet_amd import FluidNetModel
    from fluidnet_amd.dist import SlabLayout, SlabSimulation, ThreadComm
    lay = SlabLayout(16, 2, 0)
    batch, mconf = bench.build_scene(16, 16, lay, dev)
    comm = ThreadComm(ThreadComm.Hub(2), 0)
    comm.struct.size = 0
    sim = SlabSimulation(batch, mconf, FluidNetModel.default_3d(seed=1), lay, comm)
    with pytest.raises(TfluidsError, match="size"):
        sim.step()
    sim.close()
