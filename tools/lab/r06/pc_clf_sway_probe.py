import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from quadruped_drake_amd import PCController, CLFController, workloads
from quadruped_drake_amd.trajectory import TrunkTrajectory
import test_rollout as tr
n = 512
for kind, cls, dt in (("pc", PCController, 1e-3), ("clf", CLFController, 5e-3)):
    for amp in (0.02, 0.03, 0.036, 0.042):
        for steps in (20, 40, 80):
            ts, tg, masks = tr._sway_trajectory(dt, amp=amp)
            traj = TrunkTrajectory(ts, tg, masks, wait_time=0.0, device=0)
            q0, v0 = workloads.nominal_state("mini_cheetah", n)
            rng = np.random.default_rng(5)
            q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
            t0 = rng.uniform(0.0, 0.5, n)
            c = cls(max_batch=n, device=0)
            q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0"); t = torch.tensor(t0, device="cuda:0")
            tau, met, st, _, _ = c.rollout(traj, steps, dt, q, v, t); c.sync()
            s = c.stats()
            print(kind, "amp", amp, "steps", steps, "iters/tick %.2f" % (s["iters_sum"] / s["ticks"]), "status!=0 ticks", int(s["status_nonzero"]), "final st", np.bincount(st.cpu().numpy(), minlength=4).tolist())
            c.close(); traj.close()
