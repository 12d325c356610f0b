#!/usr/bin/env python3
"""Quick kernel timing on the GPU box: python3 tools/qt.py [--lib path.so] [cases...]   case = kind:cfg:n
Prints HIP-event us per launch, iterations per tick, status count and the worst relative torque error of the
first 128 instances against the oracle (the oracle is only the checker here)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("cases", nargs="*", default=["mptc:3:4096", "mptc:5:32768", "id:2:4096", "id:2:32768", "pc:3:4096", "clf:3:4096", "mptc:4:4096"])
a = ap.parse_args()
if a.lib:
    os.environ["WBC_HIP_LIB"] = os.path.abspath(a.lib)
import numpy as np, torch
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads
from oracle import oracle_py as orc
CLS = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}
for case in a.cases:
    kind, cfg, n = case.split(":"); cfg = int(cfg); n = int(n)
    b = workloads.make_batch(cfg, n=n)
    ctrl = CLS[kind](model=b["model"], max_batch=n, device=0)
    up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
    args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
    out = (torch.empty((12, n), dtype=torch.float64, device="cuda:0"), torch.empty((4, n), dtype=torch.float64, device="cuda:0"),
           torch.empty((n,), dtype=torch.int32, device="cuda:0"))
    for _ in range(10):
        ctrl.step(*args, out=out)
    ctrl.sync(); ctrl.stats(reset=True)
    ms, _ = ctrl.time_steps(a.steps, *args, out=out)
    st = ctrl.stats()
    k = min(128, n)
    sl = lambda x: None if x is None else (x[:, :k] if x.ndim == 2 else x[:k])
    tau_o, _, st_o = orc.step_batch(kind, orc.model(b["model"]), orc.params(kind), sl(b["q"]), sl(b["v"]), sl(b["targets"]), sl(b["mask"]),
                                    sl(b["mu"]), sl(b["mass_scale"]))
    tau = out[0][:, :k].cpu().numpy()
    rel = (np.abs(tau - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)).max()
    info = ctrl.kernel_info()
    print("%-5s cfg%d n=%-6d %8.2f us  %7.2f Mticks/s  iters/tick %.2f  bad %d  rel_err %.1e  regs %d scratch %d" % (
        kind, cfg, n, ms * 1e3, n / ms / 1e3, st["iters_sum"] / max(st["ticks"], 1), int((out[2] != 0).sum()), rel,
        info["num_regs"], info["scratch_bytes_per_lane"]), flush=True)
    ctrl.close()
