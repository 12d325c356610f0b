#!/usr/bin/env python3
"""Golden wire vectors for trunk_state_t, produced HERE with the reference's own generated encoder
(/root/reference/lcm_types/trunklcm/trunk_state_t.py -- it only needs `struct`).  Outputs are data:
  trunk_state_msgs.bin       K x 549 bytes, a synthetic trot-like trajectory + edge values
  trunk_state_expected.npz   the field values that were encoded
Nothing of the reference's source is stored."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/lcm_types")
from trunklcm.trunk_state_t import trunk_state_t  # noqa: E402  (reference code, imported not copied)

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20261002)
K = 64
msgs, exp = [], {k: [] for k in ("timestamp", "finished", "body", "feet", "contact", "f")}
for k in range(K):
    m = trunk_state_t()
    m.timestamp = 0.001 * k if k < 60 else [0.059, 0.0625, 0.0625, 1e9][k - 60]   # duplicates + huge value
    m.finished = (k == K - 1)
    body = rng.normal(0, 1, (6, 3)); feet = rng.normal(0, 1, (12, 3)); f = rng.normal(0, 50, (4, 3))
    if k == 7:
        body[0] = [np.pi, -0.0, 1e-300]; feet[3] = [1e300, -1e-308, 5e-324]
    m.base_p, m.base_pd, m.base_pdd, m.base_rpy, m.base_rpyd, m.base_rpydd = [list(r) for r in body]
    (m.lf_p, m.rf_p, m.lh_p, m.rh_p, m.lf_pd, m.rf_pd, m.lh_pd, m.rh_pd,
     m.lf_pdd, m.rf_pdd, m.lh_pdd, m.rh_pdd) = [list(r) for r in feet]
    ct = [(k // 8) % 2 == 0, (k // 8) % 2 == 1, (k // 8) % 2 == 1, (k // 8) % 2 == 0] if k % 16 else [True] * 4
    m.lf_contact, m.rf_contact, m.lh_contact, m.rh_contact = ct
    m.lf_f, m.rf_f, m.lh_f, m.rh_f = [list(r) for r in f]
    b = m.encode()
    assert len(b) == 549
    msgs.append(b)
    exp["timestamp"].append(m.timestamp); exp["finished"].append(m.finished); exp["body"].append(body)
    exp["feet"].append(feet); exp["contact"].append(ct); exp["f"].append(f)
with open(os.path.join(HERE, "trunk_state_msgs.bin"), "wb") as fh:
    fh.write(b"".join(msgs))
np.savez_compressed(os.path.join(HERE, "trunk_state_expected.npz"), **{k: np.array(v) for k, v in exp.items()})
print("wrote", K, "messages")
