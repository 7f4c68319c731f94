"""Randomised parity sweep of solveLinearSystemPCG (HIP vs the oracle's CSR restatement) over 3-D grids that span several
sub-boxes of the pipelined-wavefront triangular solves (ragged strips / slabs, obstacles, several components).
usage: fuzz_pcg.py [n] [seed0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from hip_adapter import HipTfluids  # noqa: E402
from oracle.oracle import OracleTfluids  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
hip, ora = HipTfluids(), OracleTfluids()
rng = np.random.RandomState(seed0)
bad = 0
for t in range(n):
    Z = int(rng.choice([4, 9, 10, 11, 18, 19, 27, 35]))
    Y = int(rng.choice([12, 66, 67, 70, 130, 131, 140]))
    X = int(rng.choice([6, 10, 17, 24, 33, 40]))
    seed = int(rng.randint(1 << 30))
    split = bool(rng.rand() < 0.5)
    tol = 1e-4
    sc, f, U, div = scenes.pcg_problem(ora, (Z, Y, X), seed, split=split, B=1, vel_cells=0.3)
    for pc in ("ic0", "ilu0"):
        pa = np.zeros_like(div); pb = np.zeros_like(div)
        ra = hip.solveLinearSystemPCG(pa, f, div, True, tol, 1000, pc)
        rb = ora.solveLinearSystemPCG(pb, f, div, True, tol, 1000, pc)
        scale = max(np.abs(pb).max(), 1e-6)
        err = np.abs(pa - pb).max()
        ok = ra < 2 * tol and np.isfinite(pa).all() and err < max(5e-5 * scale, 50 * tol)
        if not ok:
            bad += 1
            print("MISMATCH dims", (Z, Y, X), "seed", seed, "split", split, pc, "res", ra, rb, "err", err, "scale", scale)
print("fuzz_pcg: %d grids, %d mismatches" % (n, bad))
