#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs and the ORACLE's outputs for small batches of every
BASELINE config.  The reference itself (pydrake + OSQP) cannot be imported or built in this
container, so these vectors come from the pinned CPU restatement (oracle/), not from Drake:
"parity unpinned" at the Drake boundary."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle_py as orc  # noqa
from quadruped_drake_amd import workloads  # noqa

CASES = [("cfg2_id", 2, "id", 48), ("cfg3_mptc", 3, "mptc", 48), ("cfg3_id", 3, "id", 32),
         ("cfg4_anymal_mptc", 4, "mptc", 48), ("cfg5_rand_mptc", 5, "mptc", 48), ("cfg3_pc", 3, "pc", 64),
         ("cfg2_pc", 2, "pc", 32), ("cfg3_clf", 3, "clf", 48), ("cfg2_clf", 2, "clf", 32)]

for name, cfg, kind, n in CASES:
    b = workloads.make_batch(cfg, n=n)
    m = orc.model(b["model"]); p = orc.params(kind)
    tau, met, st = orc.step_batch(kind, m, p, b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), model=b["model"], kind=kind, q=b["q"], v=b["v"],
                        targets=b["targets"], mask=b["mask"],
                        mu=np.zeros(0) if b["mu"] is None else b["mu"],
                        mass_scale=np.zeros(0) if b["mass_scale"] is None else b["mass_scale"],
                        tau=tau, metrics=met, status=st)
    print(name, "n", n, "status", np.bincount(st))

# all 16 contact masks on one state set, both laws
b = workloads.make_batch(3, n=16)
m = orc.model("mini_cheetah")
for kind in ("id", "mptc", "pc", "clf"):
    mk = np.arange(16, dtype=np.uint8)
    tau, met, st = orc.step_batch(kind, m, orc.params(kind), b["q"], b["v"], b["targets"], mk)
    np.savez_compressed(os.path.join(HERE, "masks16_%s.npz" % kind), model="mini_cheetah", kind=kind, q=b["q"],
                        v=b["v"], targets=b["targets"], mask=mk, mu=np.zeros(0), mass_scale=np.zeros(0), tau=tau,
                        metrics=met, status=st)
    print("masks16", kind, np.bincount(st))
