#!/usr/bin/env python3
"""Analysis (GPU box, -DWBC_STAMPS build via WBC_HIP_LIB): is the launch's tail the SAME wavefronts every launch?  R launches of config 3
at N = 4096, phase stamps of every launch: per wavefront the active-set cycles and the lifetime; which wavefronts are the slowest, how
often, and in which phase a slow wavefront loses its time against the median."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from quadruped_drake_amd import MPTCController, IDController, workloads, _lib
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = 4096
b = workloads.make_batch(cfg, n=n)
ctrl = (IDController if b["kind"] == "id" else MPTCController)(model=b["model"], max_batch=n, device=0)
up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
for _ in range(200): ctrl.step(*args)
ctrl.sync()
nb = n // 4
L = _lib.lib(); L.wbc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
order = [3, 10, 11, 12, 13, 14, 15, 6, 4]
names = ["state", "leg", "G_b", "rows", "append", "J rows", "active set", "outputs"]
allph = []; life = []
for r in range(R):
    for _ in range(3): ctrl.step(*args)
    ctrl.sync()
    buf = np.zeros((nb, 16), dtype=np.uint64)
    assert L.wbc_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
    ph = np.diff(buf[:, order].astype(np.int64), axis=1)
    allph.append(ph); life.append(buf[:, 5].astype(np.int64) - buf[:, 0].astype(np.int64))
allph = np.array(allph); life = np.array(life)          # [R, nb, 8], [R, nb]
med = np.median(allph, axis=(0, 1))
print("median cycles per phase:", dict(zip(names, med.astype(int))))
print("lifetime per launch: max %s" % life.max(1).tolist())
slow = life.argmax(1)
print("slowest wavefront of each launch:", slow.tolist())
top = np.argsort(-np.median(life, axis=0))[:12]
print("wavefronts with the largest MEDIAN lifetime over %d launches: block, median, min, max lifetime; median active-set cycles" % R)
for w in top:
    print("  %4d  %6d %6d %6d   gi %6d" % (w, np.median(life[:, w]), life[:, w].min(), life[:, w].max(), np.median(allph[:, w, 6])))
# where does a launch's slowest wavefront lose its time against the median wavefront?
exc = np.array([allph[r, slow[r]] - med for r in range(R)])
print("excess of the slowest wavefront over the median, per phase (median over launches):", dict(zip(names, np.median(exc, axis=0).astype(int))))
print("   the same, maximum over launches:", dict(zip(names, exc.max(0).astype(int))))
# random or deterministic: correlation of per-wavefront active-set cycles between launches
g = allph[:, :, 6].astype(float)
c = np.corrcoef(g)
print("correlation of per-wavefront active-set cycles between launches: median %.3f" % np.median(c[np.triu_indices(R, 1)]))
dev = g - np.median(g, axis=0)
print("per-wavefront deviation from its own median active-set time: p50 %.0f p99 %.0f max %.0f cycles; wavefronts x launches above +3000: %d of %d" % (
    np.percentile(np.abs(dev), 50), np.percentile(np.abs(dev), 99), dev.max(), int((dev > 3000).sum()), dev.size))
for nm, k in (("append", 4), ("leg", 1), ("rows", 3)):
    d2 = allph[:, :, k].astype(float); d2 = d2 - np.median(d2, axis=0)
    print("   %s: deviation p99 %.0f max %.0f; above +3000: %d" % (nm, np.percentile(np.abs(d2), 99), d2.max(), int((d2 > 3000).sum())))
