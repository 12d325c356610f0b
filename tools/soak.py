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
         ("id", 3, IDController, {}), ("pc", 3, PCController, {}), ("clf", 3, CLFController, {}), ("mptc", 3, MPTCController, {"tau_max": 12.0}),
         # round 3: the rewritten generic loop (one-reflection drops) on the drop-heavy combinations as well
         ("mptc", 2, MPTCController, {}), ("pc", 2, PCController, {}), ("clf", 2, CLFController, {}), ("id", 4, IDController, {}),
         ("id", 2, IDController, {"tau_max": 12.0}), ("clf", 3, CLFController, {"tau_max": 12.0}), ("pc", 3, PCController, {"tau_max": 10.0})]
if len(sys.argv) > 3:
    cases = [c for i, c in enumerate(cases) if str(i) in sys.argv[3].split(",")]
for kind, cfg, cls, prm in cases:
    worst, mism, nonfin, tot, t0 = 0.0, 0, 0, 0, time.time()
    over = np.zeros(3, np.int64)      # instances above 1e-6 / 1e-5 / 1e-4
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
        over += np.array([(r > 1e-6).sum(), (r > 1e-5).sum(), (r > 1e-4).sum()])
    print("%-5s cfg %d %-16s %8d instances: worst rel torque err %.2e (above 1e-6 / 1e-5 / 1e-4: %d / %d / %d), status mismatches %d, non-finite %d  (%.1f s)" % (
        kind, cfg, str(prm) if prm else "", tot, worst, over[0], over[1], over[2], mism, nonfin, time.time() - t0), flush=True)
