#!/usr/bin/env python3
"""Golden wire vectors for robot_state_control_lcmt, produced HERE with the reference's own generated codec
(/root/reference/lcm_types/cheetahlcm/robot_state_control_lcmt.py -- it only needs `struct`).  Outputs are data:
  robot_state_msgs.bin       K x 204 bytes: "robot_current_state" messages (random states + edge values)
  robot_state_expected.npz   what the reference's decoder returns for them (float32 values), and for the
                             "robot_control_input" direction: actuator-order torques u, the plant's numbering
                             (order, act_joint) and the message bytes the reference would publish --
                             tau = (S'u)[-12:] (controllers/basic_controller.py:308-314), q = v = 0
Nothing of the reference's source is stored."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/lcm_types")
from cheetahlcm.robot_state_control_lcmt import robot_state_control_lcmt  # noqa: E402  (reference code, imported not copied)

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("GOLDEN_OUT", HERE)
rng = np.random.default_rng(20261003)
K = 48
msgs, dq, dv, dt = [], [], [], []
for k in range(K):
    m = robot_state_control_lcmt()
    q = rng.normal(0, 1, 19); v = rng.normal(0, 2, 18); tau = rng.normal(0, 10, 12)
    if k == 5:
        q[:6] = [0.0, -0.0, 1e-45, -1e-39, 3.4e38, np.pi]          # zero signs, float32 denormals, near FLT_MAX
        v[:3] = [1.0 + 2.0 ** -24, 1.0 + 3 * 2.0 ** -24, 1.0 + 2.0 ** -23]   # ties of the double -> float rounding
    m.q, m.v, m.tau = list(q), list(v), list(tau)
    b = m.encode()
    assert len(b) == 204
    d = robot_state_control_lcmt.decode(b)
    msgs.append(b); dq.append(d.q); dv.append(d.v); dt.append(d.tau)
with open(os.path.join(OUT, "robot_state_msgs.bin"), "wb") as fh:
    fh.write(b"".join(msgs))

# the outgoing direction on a plant that numbers joints breadth-first and actuators at random
order = [4 * (j % 3) + j // 3 for j in range(12)]          # canonical joint j sits at plant index order[j]
act = [int(x) for x in np.random.default_rng(2).permutation(12)]   # actuator k drives canonical joint act[k]
B = np.zeros((18, 12))
for k, j in enumerate(act):
    B[6 + order[j], k] = 1.0                                # MakeActuationMatrix of such a plant
S = B.T
U = rng.normal(0, 8, (12, 16))
ctl = []
for i in range(U.shape[1]):
    m = robot_state_control_lcmt()
    m.tau = (S.T @ U[:, i])[-12:]                           # basic_controller.py:311
    ctl.append(m.encode())
np.savez_compressed(os.path.join(OUT, "robot_state_expected.npz"), q=np.array(dq, dtype=np.float64), v=np.array(dv, dtype=np.float64),
                    tau=np.array(dt, dtype=np.float64), u=U, order=np.array(order), act_joint=np.array(act),
                    control_msgs=np.frombuffer(b"".join(ctl), dtype=np.uint8))
print("wrote", K, "state messages and", U.shape[1], "control messages")
