"""diagnostic: what the iteration spread costs at N = 4096 (variants interleaved, after a clock ramp)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quadruped_drake_amd import MPTCController, workloads
b = workloads.make_batch(3, n=4096)
n = 4096
def variant(src):
    bb = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    if src is not None:
        for k in ("q", "v", "targets"): bb[k][:] = bb[k][:, src:src + 1]
        bb["mask"][:] = bb["mask"][src]
    return bb
ctrl = MPTCController(model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
out = (torch.empty((12, n), dtype=torch.float64, device="cuda:0"), torch.empty((4, n), dtype=torch.float64, device="cuda:0"), torch.empty((n,), dtype=torch.int32, device="cuda:0"))
V = {}
for name, src in (("real batch", None), ("all = robot 0", 0), ("all = robot 124 (drop)", 124), ("all = robot 5", 5), ("all = robot 9", 9)):
    bb = variant(src); V[name] = [up(bb[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
t0 = time.time()
while time.time() - t0 < 1.0:
    ctrl.time_steps(200, *V["real batch"], out=out)
for rep in range(3):
    for name, args in V.items():
        ctrl.stats(reset=True)
        ms, _ = ctrl.time_steps(300, *args, out=out)
        st = ctrl.stats()
        print("%-24s %6.2f us  iters/tick %.2f" % (name, ms * 1e3, st["iters_sum"] / st["ticks"]), flush=True)
