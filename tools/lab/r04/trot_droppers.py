"""Round 4 analysis: the robots of a trot batch that drop a row, with and without the evaluation after a drop (a -DWBC_NO_DROP_REFINE build of
the host emulation at /tmp/lab/libhost_norefine.so), against the extended-precision oracle: within 1e-11 either way."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import host_tick as ht
from quadruped_drake_amd import workloads
from oracle import oracle_py as orc, oracle_ld as old
rel=lambda a,ref: np.abs(a-ref).max(0)/np.maximum(np.abs(ref).max(0),1e-3)
kind, cfg, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
b = workloads.make_batch(cfg, n=n, seed=seed); t = orc.load_model_json(b["model"])
tauA, _, stA, itA = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], hexv=True)
ht._LIB = C.CDLL('/tmp/lab/libhost_norefine.so')
tauB, _, stB, itB = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], hexv=True)
diff = np.abs(tauA - tauB).max(0) > 0
idx = np.where(diff)[0]
sl = lambda x: None if x is None else (x[:, idx] if x.ndim == 2 else x[idx])
tauL, _, stL = old.step_batch(kind, old.model(b["model"]), old.params(kind), sl(b["q"]), sl(b["v"]), sl(b["targets"]), sl(b["mask"]), sl(b["mu"]), sl(b["mass_scale"])); tauL = tauL.astype(float)
rA = rel(tauA[:, idx], tauL); rB = rel(tauB[:, idx], tauL)
print("%s cfg %d seed %d: %d of %d robots dropped a row; err vs extended with refinement: max %.2e median %.2e | without: max %.2e median %.2e" % (kind, cfg, seed, len(idx), n, rA.max(), np.median(rA), rB.max(), np.median(rB)), flush=True)
