"""Batch counterpart of the reference's entry script (simulate.py): the same "common parameters" -- planning method,
control method, sim_time, dt -- and the same log (`output_metrics` = [V, err, res, Vdot], simulate.py:142,184-215), for
N robot instances at once on one MI355X, closed loop on the device.

    python -m quadruped_drake_amd.simulate --control ID --planner basic --scenario raise_foot --n 4096 --sim-time 2

What replaces what: the initial state is simulate.py:171-179; `--planner basic` serves one of the reference's scenarios
(planners/simple.py), `--planner towr --messages FILE` a recorded stream of 549-byte `trunk_state` messages
(planners/towr.py; TOWR itself is out of scope); the controller is the fused tick; Drake's MultibodyPlant(time_step=dt)
and contact solver (simulate.py:38-64) are replaced by the rigid-contact forward step of `wbc_rollout` (DESIGN.md
section 9: stance feet are held by the QP's contact rows, nothing slips or lifts on its own) -- a harness for the
controllers, not a physics engine.  No visualiser, no LCM."""
import argparse
import json
import sys

import numpy as np

CONTROL = ("ID", "MPTC", "PC", "CLF")


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--control", default="ID", choices=CONTROL, help="control_method (simulate.py:13)")
    ap.add_argument("--planner", default="basic", choices=("basic", "towr"), help="planning_method (simulate.py:12)")
    ap.add_argument("--scenario", default="standing", help="basic planner: standing | orientation | raise_foot | edge")
    ap.add_argument("--messages", default=None, help="towr planner: file of concatenated 549-byte trunk_state messages")
    ap.add_argument("--sim-time", type=float, default=6.0, help="sim_time (simulate.py:20)")
    ap.add_argument("--dt", type=float, default=None, help="dt (simulate.py:21: 5e-3; MPTC / PC default to 1e-3, DESIGN.md section 9)")
    ap.add_argument("--n", type=int, default=1, help="robot instances")
    ap.add_argument("--perturb", type=float, default=0.0, help="uniform joint-angle perturbation of instances 1.. (rad)")
    ap.add_argument("--model", default="mini_cheetah")
    ap.add_argument("--log-every", type=int, default=10, help="ticks between two samples of the metrics log")
    ap.add_argument("--log", default=None, help="write the log (t, V, err, res, Vdot of every instance) to this .npz")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    if a.dt is None:
        a.dt = 5e-3 if a.control in ("ID", "CLF") else 1e-3
    if a.planner == "towr" and not a.messages:
        ap.error("--planner towr needs --messages FILE (a recorded trunk_state stream)")
    if a.n < 1 or a.sim_time <= 0 or a.dt <= 0 or a.log_every < 1:
        ap.error("--n, --sim-time, --dt and --log-every must be positive")
    return a


def run(a):
    """-> dict(t [S], metrics [S, 4, n], status_nonzero, ticks, q [19, n], v [18, n], ticks_per_second)"""
    import time as _time
    import torch
    from . import CLFController, IDController, MPTCController, PCController, workloads
    from .planners import TowrTrunkPlanner, scenario_trajectory
    cls = {"ID": IDController, "MPTC": MPTCController, "PC": PCController, "CLF": CLFController}[a.control]
    dev = "cuda:%d" % a.device
    if a.planner == "basic":
        traj = scenario_trajectory(a.scenario, a.sim_time, a.dt, model=a.model, device=a.device)
    else:
        raw = open(a.messages, "rb").read()
        if len(raw) % 549:
            raise ValueError("%s: not a whole number of 549-byte trunk_state messages" % a.messages)
        traj = TowrTrunkPlanner([raw[i:i + 549] for i in range(0, len(raw), 549)], model=a.model).trajectory(a.device)
    q0, v0 = workloads.nominal_state(a.model, a.n)                      # simulate.py:171-179
    if a.perturb > 0 and a.n > 1:
        q0[7:, 1:] += np.random.default_rng(a.seed).uniform(-a.perturb, a.perturb, (12, a.n - 1))
    ctrl = cls(model=a.model, max_batch=a.n, device=a.device)
    q, v = torch.tensor(q0, device=dev), torch.tensor(v0, device=dev)
    time = torch.zeros(a.n, dtype=torch.float64, device=dev)
    steps = int(round(a.sim_time / a.dt))
    ts, mets = [], []
    ctrl.stats(reset=True)
    ctrl.sync(); t0 = _time.perf_counter()
    done = 0
    while done < steps:
        k = min(a.log_every, steps - done)
        tau, met, st, tg, mk = ctrl.rollout(traj, k, a.dt, q, v, time)
        done += k
        ts.append(done * a.dt); mets.append(met.clone())
    ctrl.sync(); wall = _time.perf_counter() - t0
    s = ctrl.stats()
    out = dict(t=np.array(ts), metrics=torch.stack(mets).cpu().numpy(), status_nonzero=int(s["status_nonzero"]),
               ticks=int(s["ticks"]), q=q.cpu().numpy(), v=v.cpu().numpy(), ticks_per_second=s["ticks"] / wall,
               iters_mean=s["iters_sum"] / max(s["ticks"], 1.0))
    ctrl.close(); traj.close()
    return out


def main(argv=None):
    a = parse(argv)
    r = run(a)
    if a.log:
        np.savez_compressed(a.log, t=r["t"], V=r["metrics"][:, 0], err=r["metrics"][:, 1], res=r["metrics"][:, 2],
                            Vdot=r["metrics"][:, 3], q=r["q"], v=r["v"])
    m = r["metrics"]
    print(json.dumps({"control": a.control, "planner": a.planner, "scenario": a.scenario if a.planner == "basic" else a.messages,
                      "n": a.n, "sim_time": a.sim_time, "dt": a.dt, "ticks": r["ticks"], "status_nonzero": r["status_nonzero"],
                      "ticks_per_second": r["ticks_per_second"], "active_set_iterations_mean": r["iters_mean"],
                      "final": {"V": float(m[-1, 0].mean()), "err": float(m[-1, 1].mean()), "Vdot": float(m[-1, 3].mean())},
                      "body_height_final": [float(r["q"][6].min()), float(r["q"][6].max())]}))
    return 0 if r["status_nonzero"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
