import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_drake_amd import MPTCController, workloads
from quadruped_drake_amd import stats as wstats
b = workloads.make_batch(3, n=4096); n = 4096
ctrl = MPTCController(model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
out = (torch.empty((12, n), dtype=torch.float64, device="cuda:0"), torch.empty((4, n), dtype=torch.float64, device="cuda:0"), torch.empty((n,), dtype=torch.int32, device="cuda:0"))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0: ctrl.time_steps(100, *args, out=out)
def T(f, reps=200):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps * 1e6
print("time_steps(1)            %.1f us" % T(lambda: ctrl.time_steps(1, *args, out=out)))
print("time_steps(20)           %.1f us  (20 x kernel = %.1f)" % (T(lambda: ctrl.time_steps(20, *args, out=out)), 20 * 25.6))
print("ctrl.stats()             %.1f us" % T(lambda: ctrl.stats()))
print("all_gather_stats(local)  %.1f us" % T(lambda: wstats.all_gather_stats({"ticks": 1.0, "status_nonzero": 0.0, "iters_sum": 0.0, "tau_abs_sum": 0.0, "err_sum": 0.0, "tau_abs_max": 0.0, "mask_count": [0.0] * 16})))
print("torch.cuda.synchronize   %.1f us" % T(lambda: torch.cuda.synchronize()))
print("ctrl.step (launch only)  %.1f us" % T(lambda: ctrl.step(*args, out=out)))
