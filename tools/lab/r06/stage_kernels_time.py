#!/usr/bin/env python3
"""Analysis (GPU box): device time of the launch-per-stage kernels either side of the tick -- the target lookup over a 5001-sample table and the forward
step -- for the library in WBC_HIP_LIB (HIP events around 200 back-to-back launches after a warm-up; N = 4096 and 32768):
    python3 tools/lab/r06/stage_kernels_time.py -- a.so b.so"""
import os, subprocess, sys
if "--" in sys.argv:
    libs = sys.argv[sys.argv.index("--") + 1:]
    for rnd in (1, 2):
        for lib in libs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, WBC_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
            for l in (r.stdout.strip().splitlines() or [r.stderr[-300:]]):
                print("round %d  %-36s %s" % (rnd, lib, l), flush=True)
    sys.exit(0)
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_drake_amd import IDController, workloads
from quadruped_drake_amd.trajectory import TrunkTrajectory
K = 5001
ts = np.arange(K) * 1e-3
st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
tg = np.tile(st_t, (K, 1)); tg[:, 0] += 0.01 * np.sin(ts)
traj = TrunkTrajectory(ts, tg, np.full(K, 0xF, np.uint8), wait_time=1.0, device=0, standing_targets=st_t, standing_mask=0xF)
for n in (4096, 32768):
    rng = np.random.default_rng(3)
    t = torch.tensor(rng.uniform(0.0, 6.5, n), device="cuda:0")
    out = (torch.empty((54, n), dtype=torch.float64, device="cuda:0"), torch.empty((n,), dtype=torch.uint8, device="cuda:0"))
    ctrl = IDController(max_batch=n, device=0)
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0"); vd = torch.zeros((18, n), dtype=torch.float64, device="cuda:0")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(300): traj.lookup(t, out=out)
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    ev[0].record(s)
    for _ in range(200): traj.lookup(t, out=out)
    ev[1].record(s)
    for _ in range(100): ctrl.integrate(q, v, vd, 1e-3)
    ctrl.sync(); torch.cuda.synchronize()
    ev[2].record(s)
    for _ in range(200): ctrl.integrate(q, v, vd, 1e-3)
    ev[3].record(s); ctrl.sync(); torch.cuda.synchronize()
    print("N = %5d: lookup %.2f us per launch, integrate %.2f us per launch" % (n, ev[0].elapsed_time(ev[1]) * 1e3 / 200, ev[2].elapsed_time(ev[3]) * 1e3 / 200))
    ctrl.close()
