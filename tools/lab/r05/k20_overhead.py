#!/usr/bin/env python3
"""Analysis (GPU box): where do the ~60 us of fixed cost inside bench.py's timed region go at the driver's K = 20?  The region is replayed with a host time stamp after
every call (median of 40 repetitions), once as bench.py does it and with each candidate trimmed.
    python3 tools/lab/r05/k20_overhead.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_drake_amd import MPTCController, workloads
n = 4096; K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
b = workloads.make_batch(3, n=n)
ctrl = MPTCController(model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
out = (torch.empty((12, n), dtype=torch.float64, device="cuda:0"), torch.empty((4, n), dtype=torch.float64, device="cuda:0"), torch.empty((n,), dtype=torch.int32, device="cuda:0"))
bound = ctrl.bind(*args, out=out)
sync = torch.cuda.synchronize
t = time.perf_counter()
while time.perf_counter() - t < 1.0: ctrl.time_steps(100, *args, out=out)
rows = []
for rep in range(40):
    for _ in range(5): bound.step()
    ctrl.stats_reset()
    sync(); sync()
    t0 = time.perf_counter()
    bound.time_steps(K, wait=False); t1 = time.perf_counter()
    st = ctrl.stats(); t2 = time.perf_counter()
    ms = bound.time_steps_result(); t3 = time.perf_counter()
    sync(); t4 = time.perf_counter()
    sync(); t5 = time.perf_counter()
    rows.append([(t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t5 - t4) * 1e6, (t5 - t0) * 1e6, ms * 1e3 * K])
r = np.median(np.array(rows), axis=0)
print("K = %d: enqueue %.1f us | stats() (reduce kernel + the one wait) %.1f | events %.1f | sync %.1f | sync %.1f | TOTAL %.1f us; device time of the K launches (events) %.1f us -> fixed cost %.1f us" % (
    K, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[5] - r[6]))
print("   => host-side: the first kernel starts ~%.1f us after t0 if the device time ends when stats() returns minus the reduce kernel" % (r[0] + r[1] - r[6]))
