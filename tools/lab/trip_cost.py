"""diagnostic (GPU box): time per launch at N = 4096 with every robot replaced by ONE robot of the batch (no lock-step
divergence), for robots with different active-set histories -> cost per add / per drop.
usage: trip_cost.py kind:cfg [--lib x.so] robots..."""
import os, sys, time, argparse
sys.path.insert(0, os.getcwd())
ap = argparse.ArgumentParser(); ap.add_argument("case"); ap.add_argument("--lib", default=None); ap.add_argument("robots", nargs="*", type=int)
a = ap.parse_args()
if a.lib: os.environ["WBC_HIP_LIB"] = os.path.abspath(a.lib)
import numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads
kind, cfg = a.case.split(":"); cfg = int(cfg)
CLS = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}
n = 4096
b = workloads.make_batch(cfg, n=n)
def variant(src):
    bb = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    if src is not None:
        for k in ("q", "v", "targets"): bb[k][:] = bb[k][:, src:src + 1]
        bb["mask"][:] = bb["mask"][src]
    return bb
ctrl = CLS[kind](model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
out = (torch.empty((12, n), dtype=torch.float64, device="cuda:0"), torch.empty((4, n), dtype=torch.float64, device="cuda:0"), torch.empty((n,), dtype=torch.int32, device="cuda:0"))
V = {"real": [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]}
for r in a.robots:
    bb = variant(r); V["robot %d" % r] = [up(bb[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
t0 = time.time()
while time.time() - t0 < 1.0:
    ctrl.time_steps(200, *V["real"], out=out)
res = {}
for rep in range(2):
    for name, args in V.items():
        ctrl.stats(reset=True)
        ms, _ = ctrl.time_steps(200, *args, out=out)
        st = ctrl.stats()
        res.setdefault(name, []).append(ms * 1e3)
        it = st["iters_sum"] / st["ticks"]
        res[name + "_it"] = it
for name in V:
    print("%-12s %7.2f us  iters %.2f" % (name, min(res[name]), res[name + "_it"]), flush=True)
