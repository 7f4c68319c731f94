"""Seeded synthetic scenes shared by the parity tests (CPU and GPU) and the golden generator.

Every field is a contiguous float32 [B, C, Z, Y, X] array, x fastest, like the reference's tensors
(torch/tfluids/third_party/grid.h:68-78). Flags hold Manta cell types as floats
(third_party/cell_type.h:22-33).
"""
import numpy as np

FLUID, OBSTACLE, EMPTY, OUTFLOW, STICK = 1, 2, 4, 16, 128


def empty_domain(B, Z, Y, X, is3d, bnd=1):
    """generic/tfluids.cc:136-167 restated in numpy (test helper)."""
    f = np.full((B, 1, Z, Y, X), FLUID, np.float32)
    f[..., :bnd] = OBSTACLE
    f[..., X - bnd:] = OBSTACLE
    f[..., :bnd, :] = OBSTACLE
    f[..., Y - bnd:, :] = OBSTACLE
    if is3d:
        f[:, :, :bnd] = OBSTACLE
        f[:, :, Z - bnd:] = OBSTACLE
    return f


def add_obstacles(flags, is3d, rng, n_sphere=2, n_box=1, stick=False):
    B, _, Z, Y, X = flags.shape
    zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    for b in range(B):
        for _ in range(n_sphere):
            c = [rng.uniform(0.25, 0.75) * s for s in (X, Y, Z)]
            r = rng.uniform(0.08, 0.16) * min(X, Y, Z if is3d else X)
            d2 = (xx - c[0]) ** 2 + (yy - c[1]) ** 2 + ((zz - c[2]) ** 2 if is3d else 0)
            flags[b, 0][d2 <= r * r] = OBSTACLE | (STICK if stick else 0)
        for _ in range(n_box):
            lo = [int(rng.uniform(0.15, 0.6) * s) for s in (X, Y, Z)]
            sz = [max(1, int(rng.uniform(0.05, 0.2) * s)) for s in (X, Y, Z)]
            ks = slice(lo[2], lo[2] + sz[2]) if is3d else slice(0, 1)
            flags[b, 0, ks, lo[1]:lo[1] + sz[1], lo[0]:lo[0] + sz[0]] = OBSTACLE
    return flags


def smooth_field(shape, rng, modes=4):
    """Sum of a few random sinusoids: smooth, non-trivial, deterministic for a seeded rng."""
    B, C, Z, Y, X = shape
    zz, yy, xx = np.meshgrid(np.arange(Z) / max(Z, 1), np.arange(Y) / Y, np.arange(X) / X,
                             indexing="ij")
    out = np.zeros(shape, np.float64)
    for b in range(B):
        for c in range(C):
            for _ in range(modes):
                k = rng.randint(1, 4, size=3)
                ph = rng.uniform(0, 2 * np.pi, size=3)
                a = rng.uniform(0.3, 1.0)
                out[b, c] += a * np.sin(2 * np.pi * k[0] * xx + ph[0]) * \
                    np.sin(2 * np.pi * k[1] * yy + ph[1]) * \
                    (np.sin(2 * np.pi * k[2] * zz + ph[2]) if Z > 1 else 1.0)
    return out


def make_scene(dims, seed=0, B=1, vel_cells=2.5, dt=0.1, obstacles=True, empty_cells=False,
               stick=False, noise=0.0):
    """dims = (Z, Y, X); Z == 1 means 2-D. vel_cells = max |u_c|*dt in cells."""
    Z, Y, X = dims
    is3d = Z > 1
    C = 3 if is3d else 2
    rng = np.random.RandomState(seed)
    flags = empty_domain(B, Z, Y, X, is3d)
    if obstacles:
        add_obstacles(flags, is3d, rng, stick=stick)
    if empty_cells:  # a slab of empty / outflow cells for velocityUpdate, addGravity coverage
        j0 = int(0.8 * Y)
        sl = flags[:, :, 1:-1, j0:Y - 1, 1:X - 1] if is3d else flags[:, :, :, j0:Y - 1, 1:X - 1]
        sl[sl == FLUID] = EMPTY
        sl[..., -1, :][sl[..., -1, :] == EMPTY] = EMPTY | OUTFLOW
    U = smooth_field((B, C, Z, Y, X), rng)
    U *= vel_cells / dt / max(np.abs(U).max(), 1e-9)
    if noise > 0:
        U += noise * rng.randn(*U.shape)
    s = np.abs(smooth_field((B, 1, Z, Y, X), rng)) + 0.1
    p = smooth_field((B, 1, Z, Y, X), rng)
    return dict(flags=np.ascontiguousarray(flags, np.float32), U=U.astype(np.float32),
                density=s.astype(np.float32), p=p.astype(np.float32), is3d=is3d, dt=dt)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def pcg_problem(tf, dims, seed, vel_cells=2.0, split=False, **kw):
    """flags + the divergence of a wall-BC'd random velocity field: the right-hand side of the reference's PCG test
    (test_tfluids.lua:836-906 does the same from its Manta fixtures). split=True walls off part of the domain (a
    second fluid component) and leaves one fluid cell enclosed on its own (a size-1 component, which the solver
    skips). tf = any object with the tfluids operator methods (oracle / ref / HIP adapter)."""
    sc = make_scene(dims, seed=seed, vel_cells=vel_cells, **kw)
    f = sc["flags"]
    if split:
        Z, Y, X = dims
        xs = X // 2
        f[..., xs] = 2.0                         # a wall across x
        f[..., xs + 1:xs + 4] = np.where(f[..., xs + 1:xs + 4] == 1.0, 1.0, f[..., xs + 1:xs + 4])
        j0, k0 = Y // 2, (Z // 2 if Z > 1 else 0)
        if Z > 1:
            f[:, :, k0 - 1:k0 + 2, j0 - 1:j0 + 2, 2:5] = 2.0
        else:
            f[:, :, :, j0 - 1:j0 + 2, 2:5] = 2.0
        f[:, :, k0, j0, 3] = 1.0                 # one fluid cell inside a 3^dim obstacle block
    U = sc["U"].copy()
    tf.setWallBcsForward(U, f)
    div = np.zeros_like(sc["p"])
    tf.velocityDivergenceForward(U, f, div)
    return sc, f, U, div
