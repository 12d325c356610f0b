#!/usr/bin/env python3
"""Analysis (GPU box): who is right when the HIP path and the double-precision oracle disagree?  Both are compared with the oracle
compiled in x87 extended precision (oracle/oracle_ld.py: same source, 64-bit mantissa).  The QPs are strictly convex, so the solution
is unique and the extended build is ~2000x closer to it than either.   python3 tools/lab/truth.py [n] [seeds] [case indices]
Prints one markdown row per law / state combination of tools/soak.py: worst relative torque error of each side against the extended
reference, instances above 1e-6 / 1e-5 / 1e-4, and on how many of the disagreeing instances (HIP vs oracle > 1e-6) the HIP path is
the closer one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle_py as orc, oracle_ld as old
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cores = len(os.sched_getaffinity(0))
cases = [("mptc", 3, MPTCController, {}), ("mptc", 5, MPTCController, {}), ("mptc", 4, MPTCController, {}), ("id", 2, IDController, {}),
         ("id", 3, IDController, {}), ("pc", 3, PCController, {}), ("clf", 3, CLFController, {}), ("mptc", 3, MPTCController, {"tau_max": 12.0}),
         ("mptc", 2, MPTCController, {}), ("pc", 2, PCController, {}), ("clf", 2, CLFController, {}), ("id", 4, IDController, {}),
         ("id", 2, IDController, {"tau_max": 12.0}), ("clf", 3, CLFController, {"tau_max": 12.0}), ("pc", 3, PCController, {"tau_max": 10.0})]
if len(sys.argv) > 3:
    cases = [c for i, c in enumerate(cases) if str(i) in sys.argv[3].split(",")]
rel = lambda a, ref: np.abs(a - ref).max(0) / np.maximum(np.abs(ref).max(0), 1e-3)
cnt = lambda r: "%d / %d / %d" % ((r > 1e-6).sum(), (r > 1e-5).sum(), (r > 1e-4).sum())
print("| law, config | instances | HIP vs extended: worst, above 1e-6 / 1e-5 / 1e-4 | oracle (double) vs extended | HIP vs oracle | disagreeing instances (> 1e-6): HIP closer to the extended reference | status mismatches HIP / oracle vs extended |")
print("|---|---|---|---|---|---|---|")
for kind, cfg, cls, prm in cases:
    rh, ro, rho, closer, dis, tot, mism_h, mism_o = [], [], [], 0, 0, 0, 0, 0
    for s in range(seeds):
        b = workloads.make_batch(cfg, n=n, seed=50000 + 97 * s + cfg)
        ctrl = cls(model=b["model"], max_batch=n, device=0, params=prm or None)
        up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
        tau, met, st = ctrl.step(*(up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")))
        ctrl.sync(); tau = tau.cpu().numpy(); st = st.cpu().numpy(); ctrl.close()
        p = orc.params(kind)
        for k_, v_ in prm.items(): setattr(p, k_, v_)
        tau_o, _, st_o = orc.step_batch(kind, orc.model(b["model"]), p, b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], nthreads=cores)
        tau_l, _, st_l = old.step_batch(kind, old.model(b["model"]), old.params(kind, **prm), b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], nthreads=cores)
        tau_l = tau_l.astype(np.float64)
        ok = (st == 0) & (st_o == 0) & (st_l == 0)
        mism_h += int(((st == 0) != (st_l == 0)).sum()); mism_o += int(((st_o == 0) != (st_l == 0)).sum())
        a, c, d = rel(tau[:, ok], tau_l[:, ok]), rel(tau_o[:, ok], tau_l[:, ok]), rel(tau[:, ok], tau_o[:, ok])
        rh.append(a); ro.append(c); rho.append(d)
        m = d > 1e-6
        dis += int(m.sum()); closer += int((a[m] < c[m]).sum()); tot += n
    rh, ro, rho = np.concatenate(rh), np.concatenate(ro), np.concatenate(rho)
    print("| %s cfg %d %s | %d | %.2e, %s | %.2e, %s | %.2e, %s | %d of %d | %d / %d |" % (
        kind, cfg, str(prm) if prm else "", tot, rh.max(), cnt(rh), ro.max(), cnt(ro), rho.max(), cnt(rho), closer, dis, mism_h, mism_o), flush=True)
