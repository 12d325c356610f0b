#!/usr/bin/env python3
"""Diagnostic (-DWBC_STAMPS -DWBC_STAMPS_GI build): shader cycles per section of the active set's generic trips."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
os.environ["WBC_HIP_LIB"] = os.path.abspath(sys.argv[1])
import torch
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads, _lib
kind, cfg = sys.argv[2].split(":"); cfg = int(cfg); n = 4096
src = int(sys.argv[3]) if len(sys.argv) > 3 else None
b = workloads.make_batch(cfg, n=n)
if src is not None:
    for k in ("q", "v", "targets"): b[k][:] = b[k][:, src:src + 1]
    b["mask"][:] = b["mask"][src]
ctrl = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[kind](model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
for _ in range(200): ctrl.step(*args)
ctrl.sync(); ctrl.stats(reset=True)
ctrl.step(*args); ctrl.sync()
st = ctrl.stats()
nb = n // 4
buf = np.zeros((nb, 16), dtype=np.uint64)
L = _lib.lib(); L.wbc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.wbc_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
acc = buf[:, 10:15].astype(np.float64)
tot = acc.sum(1)
print("iters/tick %.2f; active-set phase (stamp 15 -> 6 n/a); generic sections, mean cycles per wavefront per tick:" % (st["iters_sum"] / st["ticks"]))
for i, nm in enumerate(["pick", "fetch", "step", "drop vector", "reflection"]):
    print("  %-12s mean %8.0f  max %8.0f" % (nm, acc[:, i].mean(), acc[:, i].max()))
print("  total        mean %8.0f  max %8.0f  (%.2f us at 2.4 GHz)" % (tot.mean(), tot.max(), tot.max() / 2400))
print("  slot 15 (generic trips counted by the build, if it counts them): mean %.2f max %d" % (buf[:, 15].mean(), buf[:, 15].max()))
