#!/usr/bin/env python3
"""Round 4 (GPU box): the instances of a deep soak line on which HIP path and double-precision oracle differ by more than a threshold, each
adjudicated by the oracle compiled in extended precision (only those instances go through the slow extended build).
    python3 tools/lab/r04/adjudicate.py pc 2 16384 1024 1e-6"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from oracle import oracle_py as orc, oracle_ld as old
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads
kind, cfg, n, seeds, thr = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
cls = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[kind]
cores = len(os.sched_getaffinity(0))
rel = lambda a, ref: np.abs(a - ref).max(0) / np.maximum(np.abs(ref).max(0), 1e-3)
ctrl = cls(model="mini_cheetah", max_batch=n, device=0)
rows = []
for s in range(seeds):
    b = workloads.make_batch(cfg, n=n, seed=50000 + 97 * s + cfg)
    up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
    tau, met, st = ctrl.step(*(up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale"))); ctrl.sync()
    tau = tau.cpu().numpy(); st = st.cpu().numpy()
    tau_o, _, st_o = orc.step_batch(kind, orc.model(b["model"]), orc.params(kind), b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], nthreads=cores)
    r = rel(tau, tau_o)
    idx = np.where((r > thr) & (st == 0) & (st_o == 0))[0]
    if idx.size == 0: continue
    sl = lambda x: None if x is None else (x[:, idx] if x.ndim == 2 else x[idx])
    tau_l, _, st_l = old.step_batch(kind, old.model(b["model"]), old.params(kind), sl(b["q"]), sl(b["v"]), sl(b["targets"]), sl(b["mask"]), sl(b["mu"]), sl(b["mass_scale"]))
    tau_l = tau_l.astype(np.float64)
    rh, ro = rel(tau[:, idx], tau_l), rel(tau_o[:, idx], tau_l)
    for j, i in enumerate(idx): rows.append((s, int(i), float(r[i]), float(rh[j]), float(ro[j])))
ctrl.close()
rows.sort(key=lambda t: -t[2])
print("%s cfg %d: %d instances of %d with |HIP - oracle| > %.0e; against the extended-precision oracle:" % (kind, cfg, len(rows), n * seeds, thr))
print("  HIP path worst %.2e (above 1e-6: %d) | double oracle worst %.2e (above 1e-6: %d) | HIP closer on %d of %d" % (
    max(t[3] for t in rows), sum(t[3] > 1e-6 for t in rows), max(t[4] for t in rows), sum(t[4] > 1e-6 for t in rows), sum(t[3] < t[4] for t in rows), len(rows)))
for t in rows[:12]: print("  seed %4d robot %5d: HIP vs oracle %.2e | HIP vs extended %.2e | oracle vs extended %.2e" % t)
