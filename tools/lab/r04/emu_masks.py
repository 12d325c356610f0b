"""Round 4 analysis: host emulation of the kernel on saturated states with RANDOM contact masks (0-4 feet down), per contact count, against the\noracle compiled in extended precision.   python3 emu_masks.py mptc 2 4096 12   (profiles/r04/masks_sweep_host.txt)"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import host_tick as ht
if os.environ.get("WBC_HOST_LIB"): ht._LIB = C.CDLL(os.environ["WBC_HOST_LIB"])
from quadruped_drake_amd import workloads
from oracle import oracle_py as orc, oracle_ld as old
rel=lambda a,ref: np.abs(a-ref).max(0)/np.maximum(np.abs(ref).max(0),1e-3)
kind = sys.argv[1]; cfg = int(sys.argv[2]); n = int(sys.argv[3]); seed = int(sys.argv[4])
b = workloads.make_batch(cfg, n=n, seed=seed); t = orc.load_model_json(b["model"])
rng = np.random.default_rng(seed)
lo = 1 if kind in ("mptc", "pc") else 0
mask = rng.integers(lo, 16, n).astype(np.uint8)
tau, met, st, it = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], mask, b["mu"], b["mass_scale"], hexv=True)
tauL, _, stL = old.step_batch(kind, old.model(b["model"]), old.params(kind), b["q"], b["v"], b["targets"], mask, b["mu"], b["mass_scale"]); tauL = tauL.astype(float)
tauO, _, stO = orc.step_batch(kind, orc.model(b["model"]), orc.params(kind), b["q"], b["v"], b["targets"], mask, b["mu"], b["mass_scale"])
nc = np.array([bin(m).count("1") for m in mask])
print("%s cfg %d states, random masks, n %d: status mismatches vs ext %d (oracle %d)" % (kind, cfg, n, (st != stL).sum(), (stO != stL).sum()))
for c in range(5):
    ok = (st == 0) & (stL == 0) & (nc == c)
    if ok.sum() == 0: continue
    r = rel(tau, tauL)[ok]; ro = rel(tauO, tauL)[ok]
    print("   nc=%d: %5d robots  emu vs ext max %.2e (>1e-7: %d, >1e-6: %d) | oracle vs ext max %.2e | iters mean %.1f" % (c, ok.sum(), r.max(), (r > 1e-7).sum(), (r > 1e-6).sum(), ro.max(), it[ok].mean()))
