"""Shared driver for the training-side operators (velocityDivergenceBackward, velocityUpdateBackward) and the
nearest-neighbour resampler: runs them through any implementation with the numpy surface of oracle.oracle."""
import numpy as np

import scenes


def run_backward_ops(t, dims, seed, **kw):
    sc = scenes.make_scene(dims, seed=seed, **kw)
    rng = np.random.RandomState(seed + 1000)
    f, U, p = sc["flags"], sc["U"], sc["p"]
    out = {}
    go = rng.randn(*p.shape).astype(np.float32)
    gU = np.full_like(U, 7.0)
    t.velocityDivergenceBackward(U, f, go, gU)
    out["divBwd"] = gU
    goU = rng.randn(*U.shape).astype(np.float32)
    gP = np.full_like(p, -3.0)
    t.velocityUpdateBackward(U, f, p, goU, gP)
    out["updBwd"] = gP
    for ratio in (1, 2, 3):
        B, C, Z, Y, X = U.shape
        up = np.full((B, C, Z * ratio, Y * ratio, X * ratio), 5.0, np.float32)
        t.volumetricUpSamplingNearestForward(ratio, U, up)
        out["upFwd%d" % ratio] = up
        g = rng.randn(*up.shape).astype(np.float32)
        gi = np.full_like(U, 9.0)
        t.volumetricUpSamplingNearestBackward(ratio, U, g, gi)
        out["upBwd%d" % ratio] = gi
    return out


CASES = [((1, 20, 24), 91, dict(vel_cells=1.0, empty_cells=True)),
         ((1, 33, 70), 92, dict(vel_cells=1.0, B=2)),
         ((9, 12, 14), 93, dict(vel_cells=1.0, empty_cells=True, B=2)),
         ((5, 6, 66), 94, dict(vel_cells=1.0))]
