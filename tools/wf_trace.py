"""Run a few IC(0)-preconditioned CG iterations on a box domain; with TFL_WF_TRACE=1 the library prints when every sub-box of
the last forward / backward triangular sweep started and finished (pcg.hip). usage: TFL_WF_TRACE=1 python tools/wf_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidnet_amd import tfluids
dev = torch.device("cuda", 0)
Z, Y, X = 130, 130, 128
flags = torch.full((1, 1, Z, Y, X), 2.0, device=dev); flags[:, :, 1:-1, 1:-1, 1:-1] = 1.0
div = torch.zeros_like(flags); div[:, :, 1:-1, 1:-1, 1:-1] = torch.randn((Z - 2, Y - 2, X - 2)).to(dev)
p = torch.zeros_like(div)
tfluids.solveLinearSystemPCG(p, flags, div, True, 1e-30, 6, "ic0")
