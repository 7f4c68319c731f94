import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import torch
from test_hip_simulate import _plume_batch, _to_dev, _layers2d
from fluidnet_amd import FluidNetModel, tfluids
from fluidnet_amd.simulate import simulate
dev=torch.device('cuda:0')
m2=dict(dt=4/60, advectionMethod="maccormackOurs", maccormackStrength=0.75, buoyancyScale=1.0, gravityScale=0, vorticityConfinementAmp=0, simMethod="convnet")
b=_to_dev(_plume_batch((1,128,128),0.05,10.0),dev); model=FluidNetModel(_layers2d(),False)
for _ in range(30): simulate(None,m2,b,model)
with tfluids.profile(b["UDiv"]) as prof:
    for _ in range(20): simulate(None,m2,b,model)
tot=0
for k,v in sorted(prof.kernels.items(), key=lambda kv:-kv[1]["ms"]):
    print("%-22s calls/step %4.1f  avg %7.2f us  per step %7.2f us"%(k, v["calls"]/20, v["ms"]/v["calls"]*1e3, v["ms"]/20*1e3)); tot+=v["ms"]/20*1e3
print("sum", tot)
