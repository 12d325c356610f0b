#!/usr/bin/env python3
"""Analysis (GPU box, -DWBC_STAMPS -DWBC_STAMPS_GI build via WBC_HIP_LIB): for the wavefronts of config 3 that run the generic loop although the host
emulation of each robot alone stays on the fast path -- is it one robot (device arithmetic differs from the host's) or the combination (lock step)?"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from quadruped_drake_amd import MPTCController, workloads, _lib
b = workloads.make_batch(3, n=4096)
L = _lib.lib(); L.wbc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
up = lambda x: None if x is None else torch.tensor(np.ascontiguousarray(x), device="cuda:0")


def run(idx):
    n = len(idx)
    ctrl = MPTCController(model=b["model"], max_batch=n, device=0)
    args = [up(b[k][..., idx]) for k in ("q", "v", "targets", "mask")]
    ctrl.step(*args); ctrl.sync(); ctrl.stats(reset=True)
    tau, met, st = ctrl.step(*args); ctrl.sync()
    s = ctrl.stats()
    nb = (n + 3) // 4
    buf = np.zeros((nb, 16), dtype=np.uint64)
    assert L.wbc_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
    ctrl.close()
    return int(s["iters_sum"]), buf[:, 10:16].astype(np.int64).sum(1).tolist(), st.cpu().numpy().tolist()


bidx = np.arange(1024); eff = ((bidx >> 5) << 5) + ((bidx & 7) << 2) + ((bidx >> 3) & 3)
for w in [int(x) for x in sys.argv[1:]] or [638, 231, 977, 31]:
    r = [4 * int(eff[w]) + k for k in range(4)]
    print("block %d robots %s" % (w, r))
    print("   all four together: iters_sum %d generic-section cycles %s status %s" % run(r))
    for k in range(4):
        print("   robot %d alone (the wavefront's other slots replicate it): iters_sum %d generic %s status %s" % ((r[k],) + run([r[k]])))
    for a in range(4):
        for c in range(a + 1, 4):
            it, g, st = run([r[a], r[c]])
            print("   pair (%d, %d): iters_sum %d generic %s" % (r[a], r[c], it, g))
