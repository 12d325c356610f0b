#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the quad kernel (stamp build, -DWBC_STAMPS).
Never quote this build's run time: the stamps fence the scheduler.  Read the SHARES."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from quadruped_drake_amd import MPTCController, IDController, workloads, _lib  # noqa

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b = workloads.make_batch(cfg, n=n)
cls = IDController if b["kind"] == "id" else MPTCController
ctrl = cls(model=b["model"], max_batch=n, device=0)
ctrl.set_variant("quad")
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
for _ in range(5):
    ctrl.step(*args)
ctrl.sync()
nb = (n + 15) // 16
buf = np.zeros((nb, 16), dtype=np.uint64)
L = _lib.lib()
L.wbc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.wbc_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
d = np.diff(buf[:, :15].astype(np.int64), axis=1)
names = ["inputs+base", "sincos+FK", "RNEA+CRBA", "Jl/Pm/xi/2xRNEA", "X,Y,stage", "Gs,kv (qsums)", "solve6 -> B", "diag init",
         "MPTC assembly / ID rows", "L1 append(s)", "Tc + L2 append", "gather R, z, J rows", "active set", "outputs", "metrics+end"]
tot = (buf[:, 14] - buf[:, 0]).astype(np.int64)
print("blocks", nb, "total cycles median %d  p10 %d  p90 %d" % (np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
for i, nm in enumerate(names[:14]):
    print("%-26s median %7d cycles  %5.1f %%" % (nm, np.median(d[:, i]), 100 * np.median(d[:, i]) / np.median(tot)))
