import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_drake_amd import MPTCController, workloads
n=4096
b=workloads.make_batch(3,n=n)
ctrl=MPTCController(model=b["model"],max_batch=n,device=0)
up=lambda x: None if x is None else torch.tensor(x,device="cuda:0")
args=[up(b[k]) for k in ("q","v","targets","mask","mu","mass_scale")]
out=(torch.empty((12,n),dtype=torch.float64,device="cuda:0"),torch.empty((4,n),dtype=torch.float64,device="cuda:0"),torch.empty((n,),dtype=torch.int32,device="cuda:0"))
t0=time.perf_counter()
while time.perf_counter()-t0<1.0: ctrl.time_steps(100,*args,out=out)
for gap in (0.0, 0.0001, 0.001, 0.01, 0.1):
    for rep in range(3):
        ctrl.time_steps(20,*args,out=out); ctrl.sync(); torch.cuda.synchronize()
        time.sleep(gap)
        each,_=ctrl.time_steps_each(20,*args,out=out)
        ms,_=ctrl.time_steps(20,*args,out=out)
        print("idle gap %.4f s: back-to-back avg of the next 20: %.2f us | per-launch (events between): first %s ... median %.2f"%(gap, ms*1e3, np.round(each[:5]*1e3,1).tolist(), np.median(each)*1e3))
    # and directly after a gap, the plain 20-launch average
    ctrl.sync(); time.sleep(gap)
    ms,_=ctrl.time_steps(20,*args,out=out)
    ms2,_=ctrl.time_steps(200,*args,out=out)
    print("   after gap %.4f: 20 launches avg %.2f us; then 200 launches avg %.2f us"%(gap,ms*1e3,ms2*1e3))
