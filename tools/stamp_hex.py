#!/usr/bin/env python3
"""Diagnostic (stamp build of the 16-lane kernel, -DWBC_STAMPS): shader-clock stamps of the prologue."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 4096
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
kind = sys.argv[sys.argv.index("--kind") + 1] if "--kind" in sys.argv else "mptc"      # the stamp build must hold that law (-DWBC_DEV_ONLY=<kind id>)
cfg = int(sys.argv[sys.argv.index("--config") + 1]) if "--config" in sys.argv else 3
b = workloads.make_batch(cfg, n=n)
ctrl = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[kind](model=b["model"], max_batch=n, device=0); ctrl.set_variant("hex")
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
# phase stamps inside the tick: 3 (start) -> 10 state -> 11 leg -> 12 G_b + solve -> 13 level-1 rows -> 14 append -> 15 J rows -> 6 active set -> 4 outputs
ph = buf[:, [3, 10, 11, 12, 13, 14, 15, 6, 4]].astype(np.int64)
dp = np.diff(ph, axis=1)
for i, nm in enumerate(["state", "leg", "G_b + solve", "level-1 rows (+ level-2 rows, park)", "30-row append", "J rows, z", "active set", "outputs"]):
    print("  phase %-38s median %6d  p90 %6d  max %6d cycles" % (nm, np.median(dp[:, i]), np.percentile(dp[:, i], 90), dp[:, i].max()))
if json_out:   # median cycles per phase under the names of tools/lab/isa_mix.py
    import json
    cyc = {"loads issue": float(np.median(d[:, 0])), "mask/mu + lds writes": float(np.median(d[:, 1])), "barrier": float(np.median(d[:, 2])),
           "stats tail": float(np.median(d[:, 4]))}
    for i, nm in enumerate(["state", "leg", "G_b + solve", "rows", "append", "J rows", "active set (all paths)", "outputs"]):
        cyc[nm] = float(np.median(dp[:, i]))
    cyc["_active_set_max"] = float(dp[:, 6].max()); cyc["_tick_median"] = float(np.median(d[:, 3])); cyc["_n"] = n
    json.dump(cyc, open(json_out, "w"), indent=1)
tot = t[:, 5] - t[:, 0]
print("wave lifetime first->last stamp: p10 %d p50 %d p90 %d p99 %d max %d cycles" % tuple(np.percentile(tot, [10, 50, 90, 99, 100])))
print("first stamp spread across blocks (launch skew): p10 %d p50 %d p90 %d max %d cycles" % tuple(np.percentile(t[:, 0] - t[:, 0].min(), [10, 50, 90, 100])))
print("last stamp - earliest first stamp: %d cycles" % (t[:, 5].max() - t[:, 0].min()))

# placement: how many wavefronts share a (xcc, se, cu, simd) and when the late ones start
hw = buf[:, 8].astype(np.int64); xcc = buf[:, 9].astype(np.int64) & 0xF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
u, cnt = np.unique(key, return_counts=True)
print("distinct SIMDs used: %d of %d wavefronts; wavefronts per SIMD histogram: %s" % (u.size, nb, np.bincount(cnt).tolist()))
cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
u2, c2 = np.unique(cukey, return_counts=True)
print("distinct CUs used: %d; wavefronts per CU histogram: %s" % (u2.size, np.bincount(c2).tolist()))
t0 = t[:, 0] - t[:, 0].min(); t5 = t[:, 5] - t[:, 0].min()
print("starts after the earliest finish (second-round wavefronts): %d" % int((t0 > t5.min()).sum()))
print("start percentiles p50 %d p90 %d p99 %d max %d ; end percentiles p1 %d p50 %d p90 %d p99 %d max %d" % (
    tuple(np.percentile(t0, [50, 90, 99, 100])) + tuple(np.percentile(t5, [1, 50, 90, 99, 100]))))
late = np.argsort(-t5)[:8]
print("latest finishers: block, start, end, lifetime, xcc, se, cu, simd")
for i in late: print("  ", i, t0[i], t5[i], tot[i], xcc[i], se[i], cu[i], simd[i])
