#!/usr/bin/env python3
"""Analysis (GPU box, -DWBC_STAMPS -DWBC_STAMPS_GI build via WBC_HIP_LIB): which wavefronts of config 3 run the active set's GENERIC loop on the
device (non-zero accumulated section cycles), against what the host emulation of each of their robots ALONE does."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import host_tick as ht
from oracle import oracle_py as orc
from quadruped_drake_amd import MPTCController, workloads, _lib
n = 4096
seed = int(sys.argv[1]) if len(sys.argv) > 1 else None
b = workloads.make_batch(3, n=n, seed=seed)
ctrl = MPTCController(model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
for _ in range(20): ctrl.step(*args)
ctrl.sync()
nb = n // 4
buf = np.zeros((nb, 16), dtype=np.uint64)
L = _lib.lib(); L.wbc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.wbc_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
sec = buf[:, 10:16].astype(np.int64)           # pick, fetch, step, drop vector, reflection (accumulated over the generic trips)
life = buf[:, 5].astype(np.int64) - buf[:, 0].astype(np.int64)
gen = sec.sum(1) > 0
print("wavefronts that ran the generic loop: %d of %d" % (gen.sum(), nb))
t = orc.load_model_json(b["model"])
hl = ht.lib()
stats = np.zeros(3, np.int32)
bidx = np.arange(nb); eff = ((bidx >> 5) << 5) + ((bidx & 7) << 2) + ((bidx >> 3) & 3)
for w in np.argsort(-life)[:16]:
    rows = []
    for i in range(4 * eff[w], 4 * eff[w] + 4):
        sl = slice(i, i + 1)
        hl.host_gi_stats(stats.ctypes.data_as(C.POINTER(C.c_int)), 1)
        r = ht.run("mptc", t["flat"], b["q"][:, sl], b["v"][:, sl], b["targets"][:, sl], b["mask"][sl], hexv=True)
        hl.host_gi_stats(stats.ctypes.data_as(C.POINTER(C.c_int)), 1)
        rows.append((int(r[3][0]), int(stats[0]), int(stats[1]), int(stats[2])))
    print("block %4d lifetime %6d generic sections %s | robots (iters, fast, generic, drops): %s" % (w, life[w], sec[w].tolist(), rows))
