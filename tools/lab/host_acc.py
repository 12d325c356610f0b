"""Analysis: host emulation of the 16-lane kernel against the oracle on a config (error distribution, iterations)."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_tick as ht
from quadruped_drake_amd import workloads
from oracle import oracle_py as orc
cfg = int(sys.argv[1]); n = int(sys.argv[2]); kind = sys.argv[3]; seed = int(sys.argv[4]) if len(sys.argv) > 4 else None
prm = {}
if len(sys.argv) > 5 and sys.argv[5]: prm = {"tau_max": float(sys.argv[5])}
b = workloads.make_batch(cfg, n=n, seed=seed)
t = orc.load_model_json(b["model"])
p = orc.params(kind)
pp = None
for k, v in prm.items(): setattr(p, k, v)
if prm:
    pp = np.array([getattr(p, f) for f in ("Kp_body_p", "Kd_body_p", "Kp_body_rpy", "Kd_body_rpy", "Kp_foot", "Kd_foot", "w_body", "w_foot", "mu", "Kd_contact", "tau_max", "eps2")])
tau_o, met_o, st_o = orc.step_batch(kind, orc.model(b["model"]), p, b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], nthreads=8)
st3 = np.zeros(3, np.int32)
ht.lib().host_gi_stats(st3.ctypes.data_as(C.POINTER(C.c_int)), 1)
tau, met, st, it = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], params12=pp, hexv=True)
ht.lib().host_gi_stats(st3.ctypes.data_as(C.POINTER(C.c_int)), 1)
ok = (st == 0) & (st_o == 0)
r = np.abs(tau[:, ok] - tau_o[:, ok]).max(0) / np.maximum(np.abs(tau_o[:, ok]).max(0), 1e-3)
print("%s cfg %d n %d: status mismatches %d, rel err median %.2e p99 %.2e max %.2e | iters mean %.2f max %d wave4 mean %.2f | fast/generic/drops %s" % (
    kind, cfg, n, ((st == 0) != (st_o == 0)).sum(), np.median(r), np.percentile(r, 99), r.max(), it.mean(), it.max(),
    it[: n // 4 * 4].reshape(-1, 4).max(1).mean(), st3))
idx = np.argsort(-r)[:8]
print("worst robots:", [(int(np.where(ok)[0][i]), "%.1e" % r[i], int(it[np.where(ok)[0][i]])) for i in idx])
if len(sys.argv) > 6:
    np.save(sys.argv[6], r)
