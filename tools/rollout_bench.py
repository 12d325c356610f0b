#!/usr/bin/env python3
"""Closed-loop rollout rate on the device (GPU box): lookup -> tick -> forward step -> time += dt, N robots."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, workloads
from quadruped_drake_amd.trajectory import TrunkTrajectory
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
for cls, dt in ((MPTCController, 1e-3), (IDController, 5e-3)):
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(0)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
    st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
    traj = TrunkTrajectory(np.zeros(0), np.zeros((0, 54)), np.zeros(0, np.uint8), wait_time=1e9, device=0,
                           standing_targets=st_t, standing_mask=0b1111)
    ctrl = cls(max_batch=n, device=0)
    q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0")
    t = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    ctrl.rollout(traj, 20, dt, q, v, t); ctrl.sync()
    t0 = time.perf_counter()
    ctrl.rollout(traj, steps, dt, q, v, t); ctrl.sync()
    el = time.perf_counter() - t0
    s = ctrl.stats()
    print("%s N=%d: %.1f us per closed-loop step, %.1f M ticks/s, status_nonzero=%d, mean iterations %.2f" % (cls.__name__, n, el / steps * 1e6, n * steps / el / 1e6, s["status_nonzero"], s["iters_sum"] / max(s["ticks"], 1)))
    ctrl.close()


# ---- trot: a synthetic stored trajectory with alternating diagonal-pair contacts (what a TOWR trot streams), robots
# at staggered phases, MPTC, dt = 1 ms
K = 4000
ts = np.arange(K) * 1e-3
st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
tg = np.tile(st_t, (K, 1))
tg[:, 0] += 0.01 * np.sin(2 * np.pi * ts / 0.3); tg[:, 3] = 0.01 * 2 * np.pi / 0.3 * np.cos(2 * np.pi * ts / 0.3)
masks = np.where((np.arange(K) // 150) % 2 == 0, 0b1001, 0b0110).astype(np.uint8)
for f in range(4):
    sw = ((masks >> f) & 1) == 0
    tg[sw, 18 + 9 * f + 2] += 0.02
traj = TrunkTrajectory(ts, tg, masks, wait_time=0.0, device=0, standing_targets=st_t, standing_mask=0b1111)
q0, v0 = workloads.nominal_state("mini_cheetah", n)
rng = np.random.default_rng(1)
q0[7:] += rng.uniform(-0.03, 0.03, (12, n))
for variant in ("hex",):
    ctrl = MPTCController(max_batch=n, device=0)
    q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0")
    t = torch.tensor(rng.uniform(0.0, 0.6, n), device="cuda:0")
    ctrl.rollout(traj, 20, 1e-3, q, v, t); ctrl.sync(); ctrl.stats(reset=True)
    t0 = time.perf_counter()
    tau, met, st, tgo, mk = ctrl.rollout(traj, steps, 1e-3, q, v, t); ctrl.sync()
    el = time.perf_counter() - t0
    s = ctrl.stats()
    print("trot MPTC N=%d (%s: %s): %.1f us per closed-loop step, %.1f M ticks/s, status_nonzero=%d, mean iterations %.2f, masks now %s" % (
        n, variant, "one persistent launch" if variant == "hex" else "four launches per step", el / steps * 1e6, n * steps / el / 1e6,
        s["status_nonzero"], s["iters_sum"] / s["ticks"], sorted(set(mk.cpu().numpy().tolist()))))
    ctrl.close()
