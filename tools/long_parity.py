"""Long-horizon parity of simulate(): GPU path vs the CPU restatement over many steps (development aid)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
os.environ.setdefault("OMP_WAIT_POLICY","PASSIVE"); os.environ.setdefault("GOMP_SPINCOUNT","0")
import numpy as np, torch
import scenes, bench
from oracle import simulate_np as S
from oracle.ref import RefTfluids, available
from oracle.oracle import OracleTfluids
from fluidnet_amd import FluidNetModel
from fluidnet_amd.simulate import simulate
res=int(sys.argv[1]) if len(sys.argv)>1 else 48
steps=int(sys.argv[2]) if len(sys.argv)>2 else 60
dev=torch.device('cuda:0')
batch, mconf = bench.build_scene(res, res, None, dev)
model = FluidNetModel.default_3d(seed=1)
nb = {k:(v.cpu().numpy().copy() if torch.is_tensor(v) else v) for k,v in batch.items()}
ops = RefTfluids(fast=False) if available() else OracleTfluids()
t0=time.time()
for s in range(1, steps+1):
    simulate(None, mconf, batch, model)
    S.simulate(ops, mconf, nb, model.layers)
    if s in (1,2,5,10,20,30,40,60,80,100) or s==steps:
        r={k: scenes.rel_l2(batch[k].cpu().numpy(), nb[k]) for k in ("pDiv","UDiv","density")}
        mism=int((batch["density"].cpu().numpy()!=nb["density"]).sum())
        print("step %3d  rel-L2 p %.2e  U %.2e  rho %.2e   max|U| %.3f  rho cells differing %d  (%.0fs)"%(s, r["pDiv"], r["UDiv"], r["density"], float(np.abs(nb["UDiv"]).max()), mism, time.time()-t0), flush=True)
