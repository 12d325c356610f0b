#!/usr/bin/env python3
"""Soak (GPU box): many fresh random batches of every law through the default kernel against the oracle (all host
threads).  Prints one line per law: instances, worst relative torque error, status mismatches, non-finite outputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle_py as orc
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cores = len(os.sched_getaffinity(0))
cases = [("mptc", 3, MPTCController, {}), ("mptc", 5, MPTCController, {}), ("mptc", 4, MPTCController, {}), ("id", 2, IDController, {}),
         ("id", 3, IDController, {}), ("pc", 3, PCController, {}), ("clf", 3, CLFController, {}), ("mptc", 3, MPTCController, {"tau_max": 12.0})]
for kind, cfg, cls, prm in cases:
    worst, mism, nonfin, tot, t0 = 0.0, 0, 0, 0, time.time()
    for s in range(seeds):
        b = workloads.make_batch(cfg, n=n, seed=50000 + 97 * s + cfg)
        ctrl = cls(model=b["model"], max_batch=n, device=0, params=prm or None)
        up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
        tau, met, st = ctrl.step(*(up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")))
        ctrl.sync(); tau = tau.cpu().numpy(); st = st.cpu().numpy(); met = met.cpu().numpy(); ctrl.close()
        p = orc.params(kind)
        for k_, v_ in prm.items(): setattr(p, k_, v_)
        tau_o, met_o, st_o = orc.step_batch(kind, orc.model(b["model"]), p, b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], nthreads=cores)
        ok = (st == 0) & (st_o == 0)
        mism += int(((st == 0) != (st_o == 0)).sum()); nonfin += int((~np.isfinite(tau)).sum() + (~np.isfinite(met)).sum())
        r = np.abs(tau[:, ok] - tau_o[:, ok]).max(0) / np.maximum(np.abs(tau_o[:, ok]).max(0), 1e-3)
        worst = max(worst, float(r.max())); tot += n
    print("%-5s cfg %d %-16s %8d instances: worst rel torque err %.2e, status mismatches %d, non-finite %d  (%.1f s)" % (
        kind, cfg, str(prm) if prm else "", tot, worst, mism, nonfin, time.time() - t0), flush=True)
