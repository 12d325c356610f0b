#!/usr/bin/env python3
"""Analysis (GPU box), VERDICT r5 task 5 step A: would a warm-started active set pay inside the persistent rollout?  The reference cold-starts every tick
(inverse_dynamics_controller.py:200: a fresh MathematicalProgram); wbc_hex_rollout_kernel keeps a robot's state on chip between ticks, so last tick's active
set would be free to keep -- IF closed loops keep rows active.  Measured here, per chunk of ticks: active-set iterations per tick and microseconds per
closed-loop step, for
  edge       the reference's EdgeTest scenario (planners/simple.py:110-115: body target 0.63 m to the side, friction rows active), from the nominal state
  stand2     BASELINE config 2's states (4-contact stands far outside their friction pyramids) released under the standing targets
  trot       the synthetic trot of tools/rollout_bench.py
    python3 tools/lab/r06/warm_probe.py [n]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, workloads, planners
from quadruped_drake_amd.trajectory import TrunkTrajectory
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"


def sway_trajectory(dt, duration=4.0, amp=0.05, hz=2.0, mu_note=""):
    """Body target swaying sideways so hard that the feet need more tangential force than friction gives at the peaks (a = amp (2 pi hz)^2 = 7.9 m/s^2 against
    mu g = 6.9): a closed loop that sits on its friction limits for part of every period and stays solvable for ever."""
    ts = np.arange(int(round(duration / dt)) + 1) * dt
    tg = workloads.standing_targets("mini_cheetah", ts.size)
    w = 2 * np.pi * hz
    tg[1] += amp * np.sin(w * ts); tg[4] = amp * w * np.cos(w * ts); tg[7] = -amp * w * w * np.sin(w * ts)
    return TrunkTrajectory(ts, np.ascontiguousarray(tg.T), np.full(ts.size, 0b1111, np.uint8), wait_time=0.0, device=0)


def run(name, cls, dt, traj, q0, v0, t0, chunks, warm=False):
    ctrl = cls(max_batch=n, device=0)
    ctrl.set_warm_start(warm)
    name = name + (" WARM" if warm else " cold")
    q = torch.tensor(q0, device=dev); v = torch.tensor(v0, device=dev); t = torch.tensor(t0, device=dev)
    ctrl.rollout(traj, 1, dt, q.clone(), v.clone(), t.clone()); ctrl.sync()     # code warm, state untouched
    done = 0
    rows = []
    for steps in chunks:
        ctrl.stats(reset=True)
        a = time.perf_counter()
        ctrl.rollout(traj, steps, dt, q, v, t); ctrl.sync()
        el = time.perf_counter() - a
        s = ctrl.stats()
        rows.append((done, done + steps, s["iters_sum"] / s["ticks"], el / steps * 1e6, int(s["status_nonzero"])))
        done += steps
    ctrl.close()
    print("%s (%s, dt %.0e, N = %d)" % (name, cls.__name__, dt, n))
    for a, b, it, us, bad in rows:
        print("   ticks %4d - %4d: %5.2f iterations per tick, %6.1f us per closed-loop step, status != 0 on %d ticks" % (a, b, it, us, bad))


chunks = [1, 1, 2, 4, 8, 16, 32, 64, 128, 256]
for cls, dt in ((IDController, 5e-3), (MPTCController, 1e-3)):
  for warm in (False, True):
    # edge: every robot from (a jittered) nominal state under the EdgeTest targets
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(5)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
    traj = planners.scenario_trajectory("edge", 4.0, dt)
    run("edge", cls, dt, traj, q0, v0, np.zeros(n), chunks[:8], warm)
    traj.close()
    # stand2: config 2's states, held by the standing targets (their own targets are one-tick snapshots, not a trajectory)
    b = workloads.make_batch(2, n=n)
    traj = planners.scenario_trajectory("standing", 4.0, dt)
    run("stand2", cls, dt, traj, b["q"], b["v"], np.zeros(n), chunks[:8], warm)
    traj.close()
    # sway: saturated part of every period, solvable for ever
    traj = sway_trajectory(dt)
    run("sway", cls, dt, traj, q0, v0, rng.uniform(0.0, 0.5, n), [64, 64, 128, 256], warm)
    traj.close()
