#!/usr/bin/env python3
"""GPU box: the HIP path against a dump of tools/reference_sweep.py (inputs + what the reference's executed controller
code returned).  The dump is scratch (not committed: it is regenerated from the reference in the build container and
travels with the gpurun snapshot).   python tools/reference_sweep_gpu.py gpurun_in/reference_sweep.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
z = np.load(sys.argv[1])
CLS = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}
cases = sorted({k.rsplit("_", 1)[0] for k in z.files if k.endswith("_tau")})
for c in cases:
    kind = c.split("_")[0]
    g = lambda k: z[c + "_" + k]
    n = g("q").shape[1]
    up = lambda x: None if x is None or x.size == 0 else torch.tensor(np.ascontiguousarray(x), device="cuda:0")
    ctrl = CLS[kind](model=str(g("model")), max_batch=n, device=0)
    vd = torch.zeros((18, n), dtype=torch.float64, device="cuda:0")
    ctrl.set_vdot_output(vd)
    tau, met, st = ctrl.step(up(g("q")), up(g("v")), up(g("targets")), up(g("mask")), up(g("mu")), up(g("mass_scale")))
    ctrl.sync()
    tau, met, st, vd = tau.cpu().numpy(), met.cpu().numpy(), st.cpu().numpy(), vd.cpu().numpy()
    ctrl.close()
    ref = g("tau")
    rel = np.abs(tau - ref).max(0) / np.maximum(np.abs(ref).max(0), 1e-3)
    dv = np.abs(vd - g("vd")).max(0) / (1.0 + np.abs(g("vd")).max(0))
    cols = [1] if kind == "id" else [0, 1, 3]
    dm = (np.abs(met[cols] - g("metrics")[cols]) / (1.0 + np.abs(g("metrics")[cols]))).max()
    print("%-12s %5d ticks: HIP vs executed reference: torque rel dev worst %.2e median %.2e | accelerations worst %.2e | metrics worst %.2e | status != 0: %d"
          % (c, n, rel.max(), np.median(rel), dv.max(), dm, int((st != 0).sum())), flush=True)
