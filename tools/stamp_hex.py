#!/usr/bin/env python3
"""Diagnostic (stamp build of the 16-lane kernel, -DWBC_STAMPS): shader-clock stamps of the prologue."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from quadruped_drake_amd import MPTCController, workloads, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = workloads.make_batch(3, n=n)
ctrl = MPTCController(model=b["model"], max_batch=n, device=0); ctrl.set_variant("hex")
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
for _ in range(5): ctrl.step(*args)
ctrl.sync()
nb = (n + 3) // 4
buf = np.zeros((nb, 16), dtype=np.uint64)
L = _lib.lib(); L.wbc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.wbc_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
t = buf[:, :6].astype(np.int64)
d = np.diff(t, axis=1)
names = ["kernarg + address setup + load issue", "mask/mu loads issued -> loads landed, LDS writes", "barrier", "tick", "stats tail"]
for i, nm in enumerate(names): print("%-52s median %7d  p90 %7d cycles" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 90)))
tot = t[:, 5] - t[:, 0]
print("wave lifetime first->last stamp: p10 %d p50 %d p90 %d p99 %d max %d cycles" % tuple(np.percentile(tot, [10, 50, 90, 99, 100])))
print("first stamp spread across blocks (launch skew): p10 %d p50 %d p90 %d max %d cycles" % tuple(np.percentile(t[:, 0] - t[:, 0].min(), [10, 50, 90, 100])))
print("last stamp - earliest first stamp: %d cycles" % (t[:, 5].max() - t[:, 0].min()))
