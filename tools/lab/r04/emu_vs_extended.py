"""Round 4 analysis: host emulation of the kernel (tools/libhost_tick.so, or WBC_HOST_LIB=<other build>) against the oracle compiled in
extended precision, case list kind:cfg[:tau_max],...   python3 emu_vs_extended.py id:2,mptc:2,pc:2 8192 70000"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import host_tick as ht
if os.environ.get("WBC_HOST_LIB"): ht._LIB = C.CDLL(os.environ["WBC_HOST_LIB"])
from quadruped_drake_amd import workloads
from oracle import oracle_py as orc, oracle_ld as old
rel=lambda a,ref: np.abs(a-ref).max(0)/np.maximum(np.abs(ref).max(0),1e-3)
cases = sys.argv[1].split(','); n = int(sys.argv[2]); seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
F = ("Kp_body_p", "Kd_body_p", "Kp_body_rpy", "Kd_body_rpy", "Kp_foot", "Kd_foot", "w_body", "w_foot", "mu", "Kd_contact", "tau_max", "tiebreak_eps2")
for c in cases:
    parts = c.split(':'); kind = parts[0]; cfg = int(parts[1]); tmax = float(parts[2]) if len(parts) > 2 else None
    b = workloads.make_batch(cfg, n=n, seed=seed0 + cfg); t = orc.load_model_json(b["model"])
    p = orc.params(kind); pl = old.params(kind); pp = None
    if tmax is not None:
        p.tau_max = tmax; pl = old.params(kind, tau_max=tmax)
        pp = np.array([getattr(p, f) for f in F])
    tau, met, st, it = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], params12=pp, hexv=True)
    tauL, _, stL = old.step_batch(kind, old.model(b["model"]), pl, b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"]); tauL = tauL.astype(float)
    tauO, _, stO = orc.step_batch(kind, orc.model(b["model"]), p, b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"])
    ok = (st == 0) & (stL == 0)
    r = rel(tau, tauL)[ok]; ro = rel(tauO, tauL)[ok]
    print("%-5s cfg %d %s n %d seed %d: emu vs ext max %.2e p99.9 %.2e >1e-7 %d >1e-6 %d >1e-5 %d | oracle vs ext max %.2e | status mismatch %d | iters mean %.2f" % (
        kind, cfg, "" if tmax is None else "tau_max %.0f" % tmax, n, seed0, r.max(), np.percentile(r, 99.9), (r > 1e-7).sum(), (r > 1e-6).sum(), (r > 1e-5).sum(), ro.max(), (st != stL).sum(), it.mean()), flush=True)
