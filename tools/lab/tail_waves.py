#!/usr/bin/env python3
"""Analysis (GPU box, -DWBC_STAMPS build via WBC_HIP_LIB): active-set cycles of every wavefront of config 3 against the iterations of its
four robots (host emulation of the same kernel): which wavefronts make the launch's tail."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import host_tick as ht
from oracle import oracle_py as orc
from quadruped_drake_amd import MPTCController, workloads, _lib
n = 4096
b = workloads.make_batch(3, n=n)
ctrl = MPTCController(model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
for _ in range(5): ctrl.step(*args)
ctrl.sync()
nb = n // 4
buf = np.zeros((nb, 16), dtype=np.uint64)
L = _lib.lib(); L.wbc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.wbc_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
gi = (buf[:, 6].astype(np.int64) - buf[:, 15].astype(np.int64))
tot = (buf[:, 5].astype(np.int64) - buf[:, 0].astype(np.int64))
t = orc.load_model_json(b["model"])
hl = ht.lib()
it = np.zeros(n, np.int32); drops = np.zeros(n, np.int32); gen = np.zeros(n, np.int32)
stats = np.zeros(3, np.int32)
for i in range(n):
    sl = slice(i, i + 1)
    hl.host_gi_stats(stats.ctypes.data_as(C.POINTER(C.c_int)), 1)
    r = ht.run("mptc", t["flat"], b["q"][:, sl], b["v"][:, sl], b["targets"][:, sl], b["mask"][sl], None if b["mu"] is None else b["mu"][sl], None if b["mass_scale"] is None else b["mass_scale"][sl], hexv=True)
    hl.host_gi_stats(stats.ctypes.data_as(C.POINTER(C.c_int)), 1)
    it[i] = r[3][0]; gen[i] = stats[1]; drops[i] = stats[2]
# stamps are indexed by the hardware blockIdx, robots by the XCD-aware remap of it (wbc_kernels.hip: hex_effective_block)
bidx = np.arange(nb); eff = ((bidx >> 5) << 5) + ((bidx & 7) << 2) + ((bidx >> 3) & 3)
wi = it.reshape(-1, 4)[eff]; wd = drops.reshape(-1, 4)[eff]; wg = gen.reshape(-1, 4)[eff]
print("active-set cycles: p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(gi, [50, 90, 99, 100])))
for k in range(0, 8):
    sel = wi.max(1) == k
    if sel.any():
        nd = wd[sel].sum(1) > 0
        print("wave max iters %d: %4d waves, cycles median %6d max %6d | with a drop: %3d waves, median %6d" % (
            k, sel.sum(), np.median(gi[sel]), gi[sel].max(), nd.sum(), np.median(gi[sel][nd]) if nd.any() else 0))
order = np.argsort(-gi)[:12]
print("slowest waves: cycles, iters of the four robots, drops of the four")
for w in order: print("  ", gi[w], wi[w].tolist(), wd[w].tolist(), "generic trips", wg[w].tolist(), "lifetime", tot[w])
print("lifetime p50 %d p99 %d max %d" % tuple(np.percentile(tot, [50, 99, 100])))
