"""Generate tests/golden/*.npz from the REFERENCE's own CPU code (oracle/_ref, i.e.
/root/reference/torch/tfluids compiled in this container by oracle/Makefile target `ref`).

Run here (needs /root/reference): python tests/golden/make_golden.py
The fixtures hold inputs AND reference outputs, so they can be checked anywhere (the GPU box has
no /root/reference). One file per (dimensionality, scene); every op of the hot path is applied to
the same seeded scene independently (each op starts from the pristine inputs).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.ref import RefTfluids  # noqa: E402
import scenes  # noqa: E402

CASES = {
    # name: (dims ZYX, seed, kwargs)
    "g2d_a": ((1, 20, 24), 11, dict(vel_cells=2.5)),
    "g2d_b": ((1, 33, 17), 12, dict(vel_cells=4.0, empty_cells=True, stick=True)),
    "g3d_a": ((12, 14, 16), 13, dict(vel_cells=2.0)),
    "g3d_b": ((9, 17, 13), 14, dict(vel_cells=3.5, empty_cells=True, stick=True)),
}
METHODS = ["maccormackOurs", "eulerOurs", "euler", "maccormack", "rk2Ours", "rk3Ours"]


def run_ops(t, sc):
    """Apply every hot-path op (init.lua wrapper semantics) to copies of the scene; returns dict."""
    out = {}
    f, dt = sc["flags"], sc["dt"]
    for m in METHODS:
        s = sc["density"].copy()
        t.advectScalar(dt, s, sc["U"].copy(), f, m, maccormackStrength=0.75)
        out["advectScalar_" + m] = s
        U = sc["U"].copy()
        t.advectVel(dt, U, f, m, maccormackStrength=0.6)
        out["advectVel_" + m] = U
    U = sc["U"].copy(); t.setWallBcsForward(U, f); out["setWallBcs"] = U
    d = np.full_like(sc["density"], 7.0); t.velocityDivergenceForward(sc["U"].copy(), f, d)
    out["divergence"] = d
    U = sc["U"].copy(); t.velocityUpdateForward(U, f, sc["p"].copy()); out["velocityUpdate"] = U
    U = sc["U"].copy(); t.vorticityConfinement(U, f, 0.35); out["vorticity"] = U
    g = np.array([0.3, -1.1, 0.7], np.float32)
    U = sc["U"].copy(); t.addBuoyancy(U, f, sc["density"].copy(), g, dt); out["buoyancy"] = U
    U = sc["U"].copy(); t.addGravity(U, f, g, dt); out["gravity"] = U
    return out


def main():
    t = RefTfluids()
    for name, (dims, seed, kw) in CASES.items():
        sc = scenes.make_scene(dims, seed=seed, **kw)
        res = run_ops(t, sc)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), flags=sc["flags"], U=sc["U"],
                            density=sc["density"], p=sc["p"], dt=np.float32(sc["dt"]),
                            **{"out_" + k: v for k, v in res.items()})
        print(name, {k: float(np.abs(v).sum()) for k, v in list(res.items())[:2]})


if __name__ == "__main__":
    main()
