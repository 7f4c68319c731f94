import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
import scenes
from oracle import simulate_np as S
from test_hip_simulate import _plume_batch, _to_dev
from fluidnet_amd import FluidNetModel
from fluidnet_amd.dist import SlabLayout, SlabSimulation, run_lockstep
from fluidnet_amd.simulate import simulate
dev=torch.device('cuda:0'); world=2
Zt,Y,X=16*world,24,32
b=_plume_batch((Zt,Y,X),0.15,0.6,obstacles_seed=11)
mconf=dict(dt=0.1,advectionMethod="maccormackOurs",maccormackStrength=0.6,buoyancyScale=1.0,gravityScale=0,vorticityConfinementAmp=2.0,simMethod="convnet")
model=FluidNetModel(S.default_3d_layers(seed=2),True)
ref=_to_dev(b,dev)
lays=[SlabLayout(Zt,world,r,10) for r in range(world)]
sims=[]
for lay in lays:
    loc={k:(lay.extract(v) if torch.is_tensor(v) else v) for k,v in ref.items()}
    sims.append(SlabSimulation(loc,mconf,FluidNetModel(S.default_3d_layers(seed=2),True),lay,None,check_reach=True))
for step in range(4):
    simulate(None,mconf,ref,model)
    run_lockstep([(s.step_gen(),s.lay) for s in sims])
    for s in sims:
        for k in ("pDiv","UDiv","density"):
            got=s.batch[k]; want=ref[k][:,:,s.lay.lo:s.lay.hi]
            err=(got-want).abs().amax(dim=(0,1,3,4)).cpu().numpy()
            print("step",step,"rank",s.lay.rank,k,"owned",s.lay.c0,s.lay.c1,"max|ref|=%.3g"%float(want.abs().max()), np.array2string(err,precision=1,max_line_width=250))
    print("scale stats", [s.stats.tolist() for s in sims])
