#!/usr/bin/env python3
"""Analysis (GPU box): launch time of one law on one seeded batch for the library in WBC_HIP_LIB, interleaved rounds:
    python3 tools/lab/r05/law_time.py clf:3:4096 pc:3:4096 ... -- a.so b.so"""
import os, subprocess, sys
if "--" in sys.argv:
    k = sys.argv.index("--"); cases, libs = sys.argv[1:k], sys.argv[k + 1:]
    for rnd in (1, 2):
        for case in cases:
            for lib in libs:
                env = dict(os.environ, WBC_HIP_LIB=os.path.abspath(lib))
                r = subprocess.run([sys.executable, os.path.abspath(__file__), case], env=env, capture_output=True, text=True)
                print("round %d  %-40s %s" % (rnd, lib, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
    sys.exit(0)
sys.path.insert(0, os.getcwd())
import time, numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads
kind, cfg, n = sys.argv[1].split(":"); cfg = int(cfg); n = int(n)
cls = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[kind]
b = workloads.make_batch(cfg, n=n)
c = cls(model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
out = (torch.empty((12, n), dtype=torch.float64, device="cuda:0"), torch.empty((4, n), dtype=torch.float64, device="cuda:0"), torch.empty((n,), dtype=torch.int32, device="cuda:0"))
t0 = time.time()
while time.time() - t0 < 1.0: c.time_steps(100, *args, out=out)
c.stats(reset=True)
ms, _ = c.time_steps(300, *args, out=out)
st = c.stats()
print("%s cfg %d n %d: %.2f us per launch, %.3f iterations per tick, status != 0: %d, sum|tau| %.9e" % (kind, cfg, n, ms * 1e3, st["iters_sum"] / st["ticks"], st["status_nonzero"], st["tau_abs_sum"] / 300))
