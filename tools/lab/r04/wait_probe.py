#!/usr/bin/env python3
"""Round 4 (GPU box): what the wait at the end of a tick costs the host.  WBC_HIP_LIB=<build> python3 tools/lab/r04/wait_probe.py
Build variants: the product (hipStreamSynchronize) against a build that polls an event (measured: slower, not adopted).
(a) BASELINE configs[0] on the product path: one robot, ID law, every tick waited for (the LeafSystem adapter's host-pointer handle, and a
device-pointer handle with bound buffers); (b) the statistics read-out; (c) the bench line's timed region at K = 20 (K launches queued,
ONE wait inside wbc_stats_get)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, workloads
q0, v0 = workloads.nominal_state("mini_cheetah", 1); tg0 = workloads.standing_targets("mini_cheetah", 1); mk0 = np.array([0b1111], dtype=np.uint8)
c = IDController(max_batch=1, device=0, host_ptrs=True)
for _ in range(200): c.step(q0, v0, tg0, mk0); c.sync()
t0 = time.perf_counter()
for _ in range(2000): c.step(q0, v0, tg0, mk0); c.sync()
print("one robot, host-pointer handle, step + sync: %.1f us per tick" % ((time.perf_counter() - t0) / 2000 * 1e6)); c.close()
dev = torch.device("cuda", 0)
c = IDController(max_batch=1, device=0); up = lambda x: torch.tensor(x, device=dev)
bound = c.bind(up(q0), up(v0), up(tg0), up(mk0))
for _ in range(200): bound.step(); c.sync()
t0 = time.perf_counter()
for _ in range(2000): bound.step(); c.sync()
print("one robot, device-pointer handle bound once, step + sync: %.1f us per tick" % ((time.perf_counter() - t0) / 2000 * 1e6)); c.close()
n = 4096; b = workloads.make_batch(3, n=n)
ctrl = MPTCController(model=b["model"], max_batch=n, device=0)
args = [None if b[k] is None else torch.tensor(b[k], device=dev) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
bound = ctrl.bind(*args)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0: bound.time_steps(100)
def T(f, reps=300):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps * 1e6
print("ctrl.stats() on an idle stream: %.1f us" % T(lambda: ctrl.stats()))
def region(K):
    torch.cuda.synchronize(); t = time.perf_counter()
    bound.time_steps(K, wait=False); ctrl.stats(); ms = bound.time_steps_result()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e6, ms * 1e3 * K
for K in (20, 200):
    r = np.array([region(K) for _ in range(200)])
    print("timed region at K = %d: wall %.1f us, %d launches by the events %.1f us -> fixed cost %.1f us (median of 200)" % (K, np.median(r[:, 0]), K, np.median(r[:, 1]), np.median(r[:, 0] - r[:, 1])))
