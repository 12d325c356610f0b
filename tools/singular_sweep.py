#!/usr/bin/env python3
"""Singular envelope of the reduced formulation (GPU box): knee angle -> 0 (|det J_leg| -> 0) and pitch -> +-pi/2
(the RPY chart's own singularity), GPU kernel vs the dense oracle.   python3 tools/singular_sweep.py > profiles/r03/singular_envelope.md
Status 3 ("ill-conditioned": MPTC / PC with |sin(knee)| < 1e-4, torques written) counts as solved-and-reported: the last column
must read 0 on every row -- no instance may come back with status 0 on both sides and torques that differ by more than 1e-4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, PCController, workloads
from oracle import oracle_py as orc

def run(kind, q, v, tg, mk):
    n = q.shape[1]
    ctrl = {"id": IDController, "mptc": MPTCController, "pc": PCController}[kind](max_batch=n, device=0)
    up = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda:0")
    tau, met, st = ctrl.step(up(q), up(v), up(tg), up(mk)); ctrl.sync()
    out = tau.cpu().numpy(), st.cpu().numpy(); ctrl.close()
    tau_o, _, st_o = orc.step_batch(kind, orc.model("mini_cheetah"), orc.params(kind), q, v, tg, mk)
    return out[0], out[1], tau_o, st_o

def rel(tau, tau_o):
    return np.abs(tau - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)

n = 64
print("# Singular envelope: GPU kernel vs dense oracle (Mini Cheetah, %d random states per cell, BASELINE config 3 distribution)\n" % n)
print("## Knee angle -> 0 on ONE leg (straight leg: the 3x3 foot Jacobian of that leg loses rank; |det J| ~ 0.04 sin(knee))\n")
print("| law | leg role | knee [rad] | oracle status 0 / 3 | GPU status 0 / 3 | status equal | max rel torque err (both status 0) | median | status 0 on both and err > 1e-4 |")
print("|---|---|---|---|---|---|---|---|---|")
for kind in ("id", "mptc", "pc"):
    for role in ("contact", "swing"):
        for e in (1e-1, 1e-2, 1e-3, 2e-4, 1e-4, 5e-5, 1e-5, 1e-6, 1e-7, 1e-8, 0.0):
            b = workloads.make_batch(3, n=n, seed=77)
            q = b["q"].copy(); mk = b["mask"].copy()
            q[7 + 2] = e                                  # LF knee
            if role == "contact": mk |= 1
            else: mk &= 0b1110
            mk[mk == 0] = 0b0110
            tau, st, tau_o, st_o = run(kind, q, b["v"], b["targets"], mk)
            ok = (st == 0) & (st_o == 0)
            r = rel(tau[:, ok], tau_o[:, ok]) if ok.any() else np.array([np.nan])
            print("| %s | %s | %.0e | %d / %d | %d / %d | %d/%d | %.1e | %.1e | %d |" % (kind.upper(), role, e, (st_o == 0).sum(), (st_o == 3).sum(),
                  (st == 0).sum(), (st == 3).sum(), (st == st_o).sum(), n, np.nanmax(r), np.nanmedian(r), int((r > 1e-4).sum()) if ok.any() else 0))
print("\n## Pitch -> pi/2 (the reference's own RPY chart is singular there: inverse_dynamics_controller.py:163-166,192)\n")
print("| law | pi/2 - pitch [rad] | oracle ok | GPU ok | status equal | max rel torque err (both ok) | median |")
print("|---|---|---|---|---|---|---|")
for kind in ("id", "mptc"):
    for e in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8):
        b = workloads.make_batch(3, n=n, seed=78)
        q = b["q"].copy()
        rng = np.random.default_rng(5)
        rpy = np.stack([rng.uniform(-0.2, 0.2, n), np.full(n, np.pi / 2 - e), rng.uniform(-0.2, 0.2, n)])
        q[0:4] = workloads.rpy_to_quat(rpy)
        tg = b["targets"].copy(); tg[9:12] = rpy + rng.normal(0, 0.02, (3, n))    # rpy target near the state
        tau, st, tau_o, st_o = run(kind, q, b["v"], tg, b["mask"])
        ok = (st == 0) & (st_o == 0)
        r = rel(tau[:, ok], tau_o[:, ok]) if ok.any() else np.array([np.nan])
        print("| %s | %.0e | %d/%d | %d/%d | %d/%d | %.1e | %.1e |" % (kind.upper(), e, (st_o == 0).sum(), n, (st == 0).sum(), n, (st == st_o).sum(), n,
              np.nanmax(r), np.nanmedian(r)))
