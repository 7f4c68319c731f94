"""Pins the oracle (oracle/tfluids_oracle.c) before anything trusts it:
 (1) bit-for-bit against the reference's own CPU code compiled here (oracle/_ref),
 (2) bit-for-bit against the committed golden fixtures generated from that build,
 (3) against the reference's portable known-answer tests
     (generic/CalcLineTraceTest.m:26-160, test_tfluids.lua:675-753).
CPU only.
"""
import glob
import os

import numpy as np
import pytest

import scenes
from golden.make_golden import METHODS, run_ops

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "g[23]d_*.npz")))


def _load(path):
    z = np.load(path)
    sc = dict(flags=z["flags"], U=z["U"], density=z["density"], p=z["p"], dt=float(z["dt"]),
              is3d=z["U"].shape[1] == 3)
    outs = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return sc, outs


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_golden_bitwise(oracle, path):
    sc, outs = _load(path)
    got = run_ops(oracle, sc)
    assert set(got) == set(outs)
    for k in sorted(outs):
        assert np.array_equal(got[k], outs[k]), k


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_reference_reproduces_golden(ref, path):
    """The fixtures are what the compiled reference produces today (guards the generator)."""
    sc, outs = _load(path)
    got = run_ops(ref, sc)
    for k in sorted(outs):
        assert np.array_equal(got[k], outs[k]), k


@pytest.mark.parametrize("dims,seed,kw", [
    ((1, 48, 64), 1, dict(vel_cells=3.0)),
    ((1, 40, 40), 2, dict(vel_cells=6.0, empty_cells=True, stick=True, noise=2.0)),
    ((24, 20, 28), 3, dict(vel_cells=2.5)),
    ((16, 24, 20), 4, dict(vel_cells=5.0, empty_cells=True, stick=True, noise=2.0, B=2)),
])
def test_oracle_matches_reference_bitwise(oracle, ref, dims, seed, kw):
    sc = scenes.make_scene(dims, seed=seed, **kw)
    a, b = run_ops(oracle, sc), run_ops(ref, sc)
    for k in sorted(a):
        assert np.array_equal(a[k], b[k]), k


def test_advect_temps_match_reference(oracle, ref):
    """fwd / fwdPos side outputs (init.lua:127-136 temps) agree too: they feed our fused pass B."""
    sc = scenes.make_scene((10, 12, 14), seed=5, vel_cells=3.0)
    ta = oracle.advectScalar(sc["dt"], sc["density"].copy(), sc["U"], sc["flags"])
    tb = ref.advectScalar(sc["dt"], sc["density"].copy(), sc["U"], sc["flags"])
    for k in ("fwd", "bwd", "fwdPos", "bwdPos"):
        assert np.array_equal(ta[k], tb[k]), k


# ---- known-answer tests ported from generic/CalcLineTraceTest.m -----------------------------
def _flags_from_obs(obs_xyz):
    """obs indexed [x, y, z] (Matlab order) -> flags [z, y, x] with Obstacle=2 / Fluid=1."""
    return np.where(np.transpose(obs_xyz, (2, 1, 0)) > 0, 2.0, 1.0).astype(np.float32)


@pytest.fixture(params=["oracle", "ref"])
def tracer(request):
    return request.getfixturevalue(request.param)


def test_linetrace_single_voxel(tracer):
    # CalcLineTraceTest.m:26-59: from every free cell centre of a 3x4x5 grid go 90 % of the way
    # to the centre of the one occupied cell (1,2,3) -> must collide.
    dims = (3, 4, 5)
    obs = np.zeros(dims)
    obs[1, 2, 3] = 1
    flags = _flags_from_obs(obs)
    filled = np.array([1, 2, 3]) + 0.5
    n = 0
    for z in range(dims[2]):
        for y in range(dims[1]):
            for x in range(dims[0]):
                pos = np.array([x, y, z]) + 0.5
                if np.linalg.norm(filled - pos) <= 1e-5:
                    continue
                delta = 0.9 * (filled - pos) - np.array([0.001, 0, 0])
                _, hit = tracer.calcLineTrace(pos, delta, flags)
                assert hit, (x, y, z)
                n += 1
    assert n == 59


def _sphere_scene():
    w, h, d = 26, 33, 28
    cc = np.array([w / 2, h / 2, d / 2])
    r = min(w, d, h) / 4 + 0.5
    u, v, z = np.meshgrid(np.arange(1, w + 1), np.arange(1, h + 1), np.arange(1, d + 1),
                          indexing="ij")
    obs = ((u - cc[0]) ** 2 + (v - cc[1]) ** 2 + (z - cc[2]) ** 2) <= r * r
    return np.array([w, h, d], float), _flags_from_obs(obs)


def test_linetrace_borders_and_corners(tracer):
    dims, flags = _sphere_scene()
    # CalcLineTraceTest.m:103-126 -- +/- borders. (Starts sit next to, not inside, the sphere.)
    for dim in range(3):
        pos = dims / 2
        pos[(dim + 1) % 3] = 2.5  # keep the ray clear of the sphere; the .m plots through it
        pos[dim] = dims[dim] - 4.1
        delta = np.zeros(3)
        delta[dim] = 6.1
        exp = pos.copy()
        exp[dim] = dims[dim]
        new, hit = tracer.calcLineTrace(pos, delta, flags)
        assert hit and np.linalg.norm(new - exp) < 1e-4
        pos[dim] = 4.1
        delta[dim] = -6.1
        exp[dim] = 0
        new, hit = tracer.calcLineTrace(pos, delta, flags)
        assert hit and np.linalg.norm(new - exp) < 1e-4
    # :128-138 step off all borders
    pos = np.array([dims[0] - 5.2, dims[1] - 6.3, dims[2] - 7.4])
    new, hit = tracer.calcLineTrace(pos, np.array([13.0, 17.0, 19.0]), flags)
    assert hit and min(abs(new - dims)) < 1e-4
    # :140-146 exact corner
    new, hit = tracer.calcLineTrace(dims - 1.5, np.array([2.0, 2.0, 2.0]), flags)
    assert hit and np.linalg.norm(new - dims) < 1e-4
    # :148-154 mixed corner
    pos = np.array([dims[0] - 0.5, 0.5, dims[2] - 0.5])
    new, hit = tracer.calcLineTrace(pos, np.array([2.0, -2.0, 2.0]), flags)
    assert hit and np.linalg.norm(new - np.array([dims[0], 0, dims[2]])) < 1e-4
    # :156-160 collision with the sphere
    pos = np.array([5.5, dims[1] - 3.2, 11.1])
    new, hit = tracer.calcLineTrace(pos, dims / 3 - pos, flags)
    assert hit
    # no-collision control (the commented-out case 1 of the .m)
    pos = np.array([3.2, 3.3, 3.4])
    d = np.array([-1.5, -0.5, 1.0])
    new, hit = tracer.calcLineTrace(pos, d, flags)
    assert not hit and np.linalg.norm(new - (pos + d)) < 1e-5


# ---- analytic tests of test_tfluids.lua ------------------------------------------------------
@pytest.mark.parametrize("is3d", [False, True])
@pytest.mark.parametrize("bnd", [1, 2, 3])
def test_empty_domain(tracer, is3d, bnd):
    # test_tfluids.lua:675-708
    Z = 9 if is3d else 1
    flags = np.full((2, 1, Z, 11, 12), -3.0, np.float32)
    tracer.emptyDomain(flags, is3d, bnd)
    exp = scenes.empty_domain(2, Z, 11, 12, is3d, bnd)
    assert np.array_equal(flags, exp)


def test_flags_to_occupancy(tracer):
    # test_tfluids.lua:710-753
    sc = scenes.make_scene((6, 9, 10), seed=3)
    occ = np.full_like(sc["flags"], 5.0)
    tracer.flagsToOccupancy(sc["flags"], occ)
    assert np.array_equal(occ, (sc["flags"] == 2).astype(np.float32))
    bad = sc["flags"].copy()
    bad[0, 0, 2, 2, 2] = 4.0
    with pytest.raises(Exception):
        tracer.flagsToOccupancy(bad, occ)


# ---- Jacobi: no CPU reference exists (generic/tfluids.cc:836-839) -> analytic properties ------
@pytest.mark.parametrize("dims", [(1, 24, 24), (10, 12, 14)])
def test_jacobi_converges_to_divergence_free(oracle, dims):
    sc = scenes.make_scene(dims, seed=9, vel_cells=1.0)
    f, U = sc["flags"], sc["U"].copy()
    oracle.setWallBcsForward(U, f)
    div = np.zeros_like(sc["density"])
    oracle.velocityDivergenceForward(U, f, div)
    p = np.full_like(div, 3.0)
    r20 = oracle.solveLinearSystemJacobi(p.copy(), f, div, sc["is3d"], 0.0, 20)
    r200 = oracle.solveLinearSystemJacobi(p.copy(), f, div, sc["is3d"], 0.0, 200)
    assert r200 < r20
    res = oracle.solveLinearSystemJacobi(p, f, div, sc["is3d"], 1e-7, 20000)
    assert res < 1e-7 * 1.0001 or res < 1e-5
    oracle.velocityUpdateForward(U, f, p)
    oracle.velocityDivergenceForward(U, f, div)
    assert np.abs(div).max() < 2e-4 * max(1.0, np.abs(sc["U"]).max())


# ---- solveLinearSystemPCG: CUDA-only in the reference, so the oracle is a restatement ------------
@pytest.mark.parametrize("dims,seed,split", [((1, 24, 28), 3, False), ((9, 11, 13), 4, False), ((1, 30, 34), 5, True),
                                            ((8, 10, 16), 6, True)])
def test_pcg_oracle_has_the_properties_the_reference_tests(oracle, dims, seed, split):
    """test_tfluids.lua:836-906 without the Manta fixtures: for every preconditioner the residual is below 2 tol,
    p has no NaN, and velocityUpdate(p) leaves (almost) no divergence. Plus: the three preconditioners agree, p is
    zero outside the fluid, a one-cell component is skipped, each component's mean pressure is zero."""
    sc, f, U, div = scenes.pcg_problem(oracle, dims, seed, split=split)
    tol, sols = 1e-5, {}
    for pc in ("none", "ilu0", "ic0"):
        p = np.random.RandomState(1).rand(*div.shape).astype(np.float32)
        res = oracle.solveLinearSystemPCG(p, f, div, sc["is3d"], tol, 1000, pc)
        assert res < 2 * tol and np.isfinite(p).all()
        Un = U.copy()
        oracle.velocityUpdateForward(Un, f, p)
        d2 = np.zeros_like(div)
        oracle.velocityDivergenceForward(Un, f, d2)
        assert np.abs(d2).max() < 2e-5 * max(1.0, np.abs(div).max()), (pc, np.abs(d2).max())
        assert np.all(p[f != 1.0] == 0.0)
        sols[pc] = p
    scale = np.abs(sols["ic0"]).max()
    assert np.abs(sols["none"] - sols["ic0"]).max() < 1e-4 * scale and np.abs(sols["ilu0"] - sols["ic0"]).max() < 1e-4 * scale
    if split:
        Z, Y, X = dims
        assert sols["ic0"][0, 0, Z // 2 if Z > 1 else 0, Y // 2, 3] == 0.0      # the enclosed single cell
        left = sols["ic0"][..., :X // 2][f[..., :X // 2] == 1.0]
        assert abs(left.sum()) < 1e-3 * scale * left.size ** 0.5


@pytest.mark.parametrize("dims,seed,split", [((1, 30, 34), 5, True), ((8, 10, 16), 6, True), ((6, 7, 9), 12, False)])
def test_normalize_pressure_mean_oracle_matches_reference(oracle, ref, dims, seed, split):
    """generic/tfluids.cc:845-925 compiled from the reference vs the restatement: equal to rounding (the reference
    sums each component with unordered float atomics), non-fluid cells untouched, component means zero afterwards."""
    sc, f, U, div = scenes.pcg_problem(oracle, dims, seed, split=split, B=2)
    p0 = (np.random.RandomState(seed).randn(*div.shape) * 3 + 5).astype(np.float32)
    a, b = p0.copy(), p0.copy()
    oracle.normalizePressureMean(a, f, sc["is3d"])
    ref.normalizePressureMean(b, f, sc["is3d"])
    assert np.abs(a - b).max() < 1e-5 * np.abs(p0).max()
    assert np.array_equal(a[f != 1.0], p0[f != 1.0])
    assert abs(a[f == 1.0].mean()) < 1.0        # sanity: the bulk offset (5) is gone


def test_pcg_oracle_rejects_fluid_on_the_border(oracle):
    sc, f, U, div = scenes.pcg_problem(oracle, (1, 12, 12), 7)
    f[0, 0, 0, 0, 5] = 1.0
    from oracle.oracle import OracleError
    with pytest.raises(OracleError):
        oracle.solveLinearSystemPCG(np.zeros_like(div), f, div, False, 1e-5, 100, "none")


# ---- PCG pinned to the reference's own host function, compiled for the host ----------------------------------------
@pytest.mark.parametrize("dims,seed,split,B,kw", [((1, 24, 28), 3, False, 1, {}), ((9, 11, 13), 4, False, 2, {}),
                                                 ((1, 30, 34), 5, True, 1, {}), ((8, 10, 16), 6, True, 2, {}),
                                                 ((16, 20, 24), 8, True, 1, {}), ((7, 12, 10), 13, False, 2, {"empty_cells": True}),
                                                 ((1, 18, 22), 14, True, 1, {"empty_cells": True})])
def test_pcg_restatement_equals_compiled_reference_host_function(oracle, ref_pcg, dims, seed, split, B, kw):
    """generic/tfluids.cu:864-1759 -- findConnectedFluidComponents, createReducedSystemIndices, setupLaplacian, the CG
    loop (Golub & Van Loan 10.3.1) with clampToEpsilon, its `while (r2 > tol^2 && iter <= maxIter)` rule, the
    per-component mean subtraction and the copy kernels -- built by `make ref_pcg` from the file where it lies, over
    host stand-ins for the five cuSPARSE / cuBLAS primitives it calls (oracle/ref_shim/cusparse_host.h), vs
    oracle/tfluids_oracle.c: identical pressure and residual, BIT FOR BIT, for all three preconditioners, converged and
    cut off after a few iterations, with several components, a one-cell component, empty cells and B = 2."""
    sc, f, U, div = scenes.pcg_problem(oracle, dims, seed, split=split, B=B, **kw)
    for pc in ("none", "ilu0", "ic0"):
        for tol, max_iter in ((1e-5, 1000), (1e-12, 0), (1e-12, 1), (1e-12, 6), (1e3, 1000)):
            pa = np.random.RandomState(1).rand(*div.shape).astype(np.float32)
            pb = pa.copy()
            ra = oracle.solveLinearSystemPCG(pa, f, div, sc["is3d"], tol, max_iter, pc)
            rb = ref_pcg.solveLinearSystemPCG(pb, f, div, sc["is3d"], tol, max_iter, pc)
            assert np.array_equal(pa, pb), (pc, tol, max_iter, np.abs(pa - pb).max())
            assert ra == rb, (pc, tol, max_iter, ra, rb)
    assert np.abs(pb).max() == 0 and np.abs(div).max() < 1e3     # tol above |rhs|: no iteration, p stays zero


def _grid_laplacian(dims, seed):
    """CSR of the reference's pressure matrix on a box with random obstacles (setupLaplacian's stencil: diagonal = number of
    non-obstacle neighbours, -1 towards fluid neighbours), as scipy sparse float32 with sorted columns."""
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    fluid = np.ones(dims, bool)
    fluid[rng.rand(*dims) < 0.15] = False
    idx = -np.ones(dims, np.int64)
    idx[fluid] = np.arange(fluid.sum())
    rows, cols, vals = [], [], []
    for c in np.argwhere(fluid):
        diag = 0.0
        for ax in range(len(dims)):
            for s_ in (-1, 1):
                nb = c.copy(); nb[ax] += s_
                if nb[ax] < 0 or nb[ax] >= dims[ax]:
                    continue                      # domain wall: an obstacle
                diag += 1.0
                if fluid[tuple(nb)]:
                    rows.append(idx[tuple(c)]); cols.append(idx[tuple(nb)]); vals.append(-1.0)
                else:
                    diag -= 1.0
        rows.append(idx[tuple(c)]); cols.append(idx[tuple(c)]); vals.append(max(diag, 1.0) + 0.25)   # + shift: SPD, well away from singular
    A = sp.csr_matrix((np.array(vals, np.float32), (rows, cols)), shape=(int(fluid.sum()),) * 2)
    A.sort_indices()
    return A


@pytest.mark.parametrize("dims,seed", [((9, 11), 1), ((5, 6, 7), 2)])
def test_restated_cusparse_primitives_have_their_defining_properties(ref_pcg, dims, seed):
    """oracle/ref_shim/cusparse_host.h restates the five cuSPARSE entry points the reference's PCG calls (the toolkit is not in
    the tree). Independent witnesses, in numpy / scipy float64: csrilu0 -> (L U)_ij = A_ij on A's pattern (the definition of
    ILU(0)); csric0 on the stored upper triangle -> (R^T R)_ij = A_ij on the pattern; csrsv_solve -> op(T) x = f for the
    descriptor's triangle (unit / non-unit diagonal, N / T); csrmv with a SYMMETRIC descriptor and upper storage -> y = A x."""
    import ctypes
    import scipy.sparse as sp
    from oracle import ref as refmod
    lib = refmod._pcg_lib()
    A = _grid_laplacian(dims, seed)
    n = A.shape[0]
    ip = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(ctypes.c_void_p)
    fp = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    def call(what, M, val, type_=0, fill=0, diag=0, op=0, f=None, x=None):
        rc = lib.tfluids_ref_cusparse_primitive(what, n, int(M.nnz), ip(M.indptr), ip(M.indices), fp(val), type_, fill, diag, op,
                                                fp(f) if f is not None else None, fp(x) if x is not None else None)
        assert rc == 0, (what, rc)

    rng = np.random.RandomState(seed + 10)
    f = rng.randn(n).astype(np.float32)
    # ---- ILU(0) ----
    val = A.data.copy()
    call(0, A, val)
    F = sp.csr_matrix((val.astype(np.float64), A.indices, A.indptr), shape=A.shape)
    L = sp.tril(F, -1) + sp.identity(n)
    U = sp.triu(F, 0)
    LU = (L @ U).tocsr()
    pat = A.copy(); pat.data[:] = 1.0
    assert abs(LU.multiply(pat) - A.astype(np.float64)).max() < 2e-6 * abs(A).max()
    # unit-lower then upper solve = (L U)^-1 f
    y, z = np.zeros(n, np.float32), np.zeros(n, np.float32)
    call(2, A, val, type_=0, fill=0, diag=1, op=0, f=f, x=y)
    call(2, A, val, type_=0, fill=1, diag=0, op=0, f=y, x=z)
    assert np.abs(LU @ z.astype(np.float64) - f).max() < 2e-5 * max(1.0, np.abs(f).max())
    # ---- IC(0) on the stored upper triangle ----
    Au = sp.triu(A, 0).tocsr(); Au.sort_indices()
    valc = Au.data.copy()
    call(1, Au, valc, type_=1, fill=1)
    R = sp.csr_matrix((valc.astype(np.float64), Au.indices, Au.indptr), shape=A.shape)
    RtR = (R.T @ R).tocsr()
    assert abs(RtR.multiply(pat) - A.astype(np.float64)).max() < 2e-6 * abs(A).max()
    call(2, Au, valc, type_=3, fill=1, diag=0, op=1, f=f, x=y)      # R^T y = f
    call(2, Au, valc, type_=3, fill=1, diag=0, op=0, f=y, x=z)      # R z = y
    assert np.abs(RtR @ z.astype(np.float64) - f).max() < 2e-5 * max(1.0, np.abs(f).max())
    # ---- csrmv: general (full storage) and SYMMETRIC (upper storage) give the same y = A x ----
    y1, y2 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    call(3, A, A.data.copy(), type_=0, f=f, x=y1)
    call(3, Au, Au.data.copy(), type_=1, fill=1, f=f, x=y2)
    want = A.astype(np.float64) @ f
    assert np.abs(y1 - want).max() < 1e-5 * np.abs(want).max() and np.abs(y2 - want).max() < 1e-5 * np.abs(want).max()


def test_pcg_compiled_reference_raises_like_the_restatement(oracle, ref_pcg):
    """a fluid cell on the domain border: setupLaplacian raises (generic/tfluids.cu:1082-1090); an unknown preconditioner
    name: StringToPrecondType raises (:1216-1228)"""
    from oracle.oracle import OracleError
    from oracle.ref import RefError
    sc, f, U, div = scenes.pcg_problem(oracle, (1, 12, 12), 7)
    with pytest.raises(RefError):
        ref_pcg.solveLinearSystemPCG(np.zeros_like(div), f, div, False, 1e-5, 100, "cholesky")
    f[0, 0, 0, 0, 5] = 1.0
    with pytest.raises(RefError):
        ref_pcg.solveLinearSystemPCG(np.zeros_like(div), f, div, False, 1e-5, 100, "none")
    with pytest.raises(OracleError):
        oracle.solveLinearSystemPCG(np.zeros_like(div), f, div, False, 1e-5, 100, "none")


# ---- training-side operators + resampler -------------------------------------------------
from backward_cases import CASES as BWD_CASES, run_backward_ops  # noqa: E402


@pytest.mark.parametrize("dims,seed,kw", BWD_CASES)
def test_backward_ops_oracle_matches_reference_bitwise(oracle, ref, dims, seed, kw):
    a, b = run_backward_ops(oracle, dims, seed, **kw), run_backward_ops(ref, dims, seed, **kw)
    for k in sorted(a):
        assert np.array_equal(a[k], b[k]), k


def test_divergence_backward_is_the_adjoint(oracle):
    """<div(U), g> == <U, divBackward(g)> for any U: the backward op is the exact transpose of the forward."""
    sc = scenes.make_scene((7, 9, 11), seed=95, vel_cells=1.0)
    rng = np.random.RandomState(1)
    g = rng.randn(*sc["p"].shape).astype(np.float32)
    div = np.zeros_like(sc["p"])
    oracle.velocityDivergenceForward(sc["U"], sc["flags"], div)
    gU = np.zeros_like(sc["U"])
    oracle.velocityDivergenceBackward(sc["U"], sc["flags"], g, gU)
    lhs = float((div.astype(np.float64) * g).sum())
    rhs = float((sc["U"].astype(np.float64) * gU).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


# ---- Jacobi pinned to the reference's own CUDA kernel + host loop, compiled for the host ----------------------
@pytest.mark.parametrize("dims,seed", [((1, 33, 47), 61), ((14, 19, 23), 62), ((1, 64, 64), 63)])
def test_jacobi_restatement_equals_compiled_reference_kernel(oracle, ref, dims, seed):
    """generic/tfluids.cu:1765-1927 (kernel_jacobiIteration + the ping-pong loop) built by `make ref_jacobi`
    through oracle/ref_shim/cuda_host.h vs oracle/tfluids_oracle.c: identical pressure, bit for bit, for fixed
    iteration counts of both parities (the copy-back branch, :1913-1915) and for early termination on pTol."""
    import subprocess
    from oracle import ref as refmod
    if not refmod.jacobi_available():
        if not os.path.isdir("/root/reference/torch/tfluids"):
            pytest.skip("oracle/_ref/libtfluids_ref_jacobi.so not built and /root/reference absent")
        subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(HERE), "oracle"), "ref_jacobi"])
    sc, f, U, div = scenes.pcg_problem(oracle, dims, seed, B=2, empty_cells=True)
    for iters in (1, 2, 7, 20):
        pa, pb = np.full_like(sc["p"], 3.0), np.full_like(sc["p"], -1.0)   # the initial guess is ignored (zeroed)
        ra = oracle.solveLinearSystemJacobi(pa, f, div, sc["is3d"], 0.0, iters)
        rb = ref.solveLinearSystemJacobi(pb, f, div, sc["is3d"], 0.0, iters)
        assert np.array_equal(pa, pb), iters
        assert abs(ra - rb) <= 1e-5 * max(1.0, abs(rb)), (ra, rb)
    assert np.abs(pa).max() > 0
    tol = 0.5 * rb
    pa, pb = np.zeros_like(sc["p"]), np.zeros_like(sc["p"])
    ra = oracle.solveLinearSystemJacobi(pa, f, div, sc["is3d"], tol, 5000)
    rb = ref.solveLinearSystemJacobi(pb, f, div, sc["is3d"], tol, 5000)
    assert np.array_equal(pa, pb) and ra < tol and rb < tol
    with pytest.raises(Exception):
        ref.solveLinearSystemJacobi(pb, f, div, sc["is3d"], 0.0, 0)


@pytest.mark.parametrize("is3d", [False, True])
@pytest.mark.parametrize("rad", [1, 3, 4])
def test_rectangular_blur(tracer, oracle, ref, is3d, rad):
    """test_tfluids.lua:1072-1133 without nn: the blur equals a box filter of width 2 rad + 1 over the edge-clamped
    field (the reference checks that against a convolution at 1e-6); and the C restatement is the compiled reference
    bit for bit (the running sum is evaluated in the reference's order)."""
    rng = np.random.RandomState(11 + rad)
    shape = (2, 3, 17, 21, 22) if is3d else (2, 3, 1, 35, 36)
    src = rng.uniform(0, 1, shape).astype(np.float32)
    dst = rng.uniform(0, 1, shape).astype(np.float32)          # filled with noise, as the reference test does
    tracer.rectangularBlur(src, rad, is3d, dst)
    pads = [(0, 0), (0, 0)] + [((rad, rad) if (is3d or ax > 0) else (0, 0)) for ax in range(3)]
    padded = np.pad(src.astype(np.float64), pads, mode="edge")
    want = np.zeros(shape, np.float64)
    k = 2 * rad + 1
    for dz in range(k if is3d else 1):
        for dy in range(k):
            for dx in range(k):
                want += padded[:, :, dz:dz + shape[2], dy:dy + shape[3], dx:dx + shape[4]]
    want /= float(k ** (3 if is3d else 2))
    assert np.abs(dst - want).max() < 1e-5
    d_o, d_r = np.empty_like(src), np.empty_like(src)
    oracle.rectangularBlur(src, rad, is3d, d_o)
    ref.rectangularBlur(src, rad, is3d, d_r)
    assert np.array_equal(d_o, d_r)


@pytest.mark.parametrize("is3d", [False, True])
def test_signed_distance_field(tracer, oracle, ref, is3d):
    """test_tfluids.lua:1135-1240: against a brute-force search over ALL obstacle cells, clamped to the search radius;
    restatement == compiled reference bit for bit."""
    rng = np.random.RandomState(5)
    dims = (9, 12, 14) if is3d else (1, 20, 23)
    rad = 3
    flags = np.ones((2, 1) + dims, np.float32)
    flags[rng.rand(*flags.shape) < 0.03] = 2.0
    dist = rng.rand(*flags.shape).astype(np.float32)
    tracer.signedDistanceField(flags, rad, is3d, dist)
    zz, yy, xx = np.meshgrid(*[np.arange(n) for n in dims], indexing="ij")
    for b in range(2):
        obs = np.argwhere(flags[b, 0] == 2.0)
        d2 = np.full(dims, float(rad * rad))
        for (oz, oy, ox) in obs:
            d2 = np.minimum(d2, (zz - oz) ** 2 + (yy - oy) ** 2 + (xx - ox) ** 2)
        want = np.sqrt(d2)
        # the search window is a CUBE of half-width rad: an obstacle at (rad, rad) offset lies outside the radius anyway
        assert np.abs(dist[b, 0] - want).max() < 1e-6
    d_o, d_r = np.empty_like(dist), np.empty_like(dist)
    oracle.signedDistanceField(flags, rad, is3d, d_o)
    ref.signedDistanceField(flags, rad, is3d, d_r)
    assert np.array_equal(d_o, d_r)
