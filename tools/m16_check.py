"""Split-operand fp16 MFMA conv path (conv_mfma16.hip) against the other three evaluations of the 3-D default net:
rel-L2 of the pressure vs the shape-generic direct kernels and vs an fp64 convolution (the witness), range errors.
usage (GPU box): python tools/m16_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import scenes
    from oracle import simulate_np as S
    from fluidnet_amd import FluidNetModel
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    from oracle.oracle import OracleTfluids
    oracle = OracleTfluids()
    layers = S.default_3d_layers(seed=5)
    dev = torch.device("cuda:0")
    for dims, seed in [((32, 32, 32), 51), ((13, 21, 45), 52), ((5, 9, 33), 53), ((6, 7, 130), 54), ((48, 48, 48), 55)]:
        sc = scenes.make_scene(dims, seed=seed, vel_cells=0.4, B=2)
        tp, tU, tf = (torch.from_numpy(sc[k]).to(dev) for k in ("p", "U", "flags"))
        res = {}
        for path in ("direct", "winograd", "mfma", "mfma16"):
            os.environ["TFL_CONV_PATH"] = path
            m = FluidNetModel(layers, True)
            p, U = m.forward([tp, tU, tf])
            res[path] = (p.cpu().numpy(), U.cpu().numpy(), m.range_errors(tp))
        p32, _ = S.model_forward(oracle, layers, sc["p"], sc["U"], sc["flags"])
        p64, _ = S.model_forward(oracle, layers, sc["p"], sc["U"], sc["flags"], conv_dtype="float64")
        line = "%-14s torch32-vs-64 %.2e |" % (dims, scenes.rel_l2(p32, p64))
        for path in ("direct", "winograd", "mfma", "mfma16"):
            line += " %s: vs64 %.2e vs-direct %.2e" % (path, scenes.rel_l2(res[path][0], p64), scenes.rel_l2(res[path][0], res["direct"][0]))
        print(line, "range_err", res["mfma16"][2], flush=True)
        print("    U: mfma16 vs direct %.2e" % scenes.rel_l2(res["mfma16"][1], res["direct"][1]), flush=True)


if __name__ == "__main__":
    main()
