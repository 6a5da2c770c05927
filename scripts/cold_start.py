"""Cold-start timing: import, library load (11 MB of gfx950 code objects), first plan, first launch."""
import time; t0=time.perf_counter()
import numpy as np, torch
t1=time.perf_counter()
import sys, os; sys.path.insert(0, os.getcwd())
from aggfly_amd import hip, synth
hip.load(); t2=time.perf_counter()
torch.cuda.init(); x=torch.zeros(1,device="cuda"); torch.cuda.synchronize(); t3=time.perf_counter()
T,ny,nx=24*30,16,24
cube=torch.from_numpy(synth.temperature_cube(T,ny,nx,dtype=np.float64,seed=5)).cuda()
ib=synth.hourly_bounds(T); ob=np.array([0,len(ib)-1])
plan=hip.FusedPlan(T,ny*nx,hip.F64,ib,ob,[dict(inner="mean",outer="sum")]); t4=time.perf_counter()
out=plan.run_temporal(cube); torch.cuda.synchronize(); t5=time.perf_counter()
out=plan.run_temporal(cube); torch.cuda.synchronize(); t6=time.perf_counter()
print("import torch %.2f | load lib %.2f | cuda init %.2f | plan create %.3f | first launch %.3f | second %.4f"%(t1-t0,t2-t1,t3-t2,t4-t3,t5-t4,t6-t5))
