"""CPU ORACLE (test infrastructure) for the callers of the path: numpy/struct restatements of
  lcm_types/trunklcm/trunk_state_t.py:83-121  (decode of the 549-byte big-endian message)
  planners/towr.py:92-148                     (nearest-timestamp lookup and dict unpacking)
Pinned by fixtures produced with the reference's OWN encoder (tests/golden/make_trunk_state_golden.py).
"""
import struct

import numpy as np

FINGERPRINT = struct.pack(">Q", ((0xbd03c56c9649d0b6 << 1) & 0xffffffffffffffff) + (0xbd03c56c9649d0b6 >> 63))


def decode(buf):
    """trunk_state_t.py:83-121"""
    if buf[:8] != FINGERPRINT:
        raise ValueError("Decode error")
    off = 8
    ts, fin = struct.unpack_from(">db", buf, off); off += 9
    v = []
    for _ in range(18):
        v.append(struct.unpack_from(">3d", buf, off)); off += 24
    ct = struct.unpack_from(">bbbb", buf, off); off += 4
    f = []
    for _ in range(4):
        f.append(struct.unpack_from(">3d", buf, off)); off += 24
    assert off == 549
    names = ["base_p", "base_pd", "base_pdd", "base_rpy", "base_rpyd", "base_rpydd", "lf_p", "rf_p", "lh_p", "rh_p",
             "lf_pd", "rf_pd", "lh_pd", "rh_pd", "lf_pdd", "rf_pdd", "lh_pdd", "rh_pdd"]
    d = {n: np.array(x) for n, x in zip(names, v)}
    d.update(timestamp=ts, finished=bool(fin), contact=[bool(c) for c in ct], foot_f=np.array(f))
    return d


def to_targets(d):
    """planners/towr.py:111-148 -> 54 rows + mask (order of include/wbc.h)."""
    t = np.concatenate([d["base_p"], d["base_pd"], d["base_pdd"], d["base_rpy"], d["base_rpyd"], d["base_rpydd"]] +
                       [np.concatenate([d[f + "_p"], d[f + "_pd"], d[f + "_pdd"]]) for f in ("lf", "rf", "lh", "rh")])
    mask = sum(1 << i for i, c in enumerate(d["contact"]) if c)
    return t, mask


def lookup(time, timestamps, table, masks, standing, standing_mask, wait_time):
    """planners/towr.py:92-106 per instance: t < wait_time -> standing; else argmin |ts - (t - wait)|."""
    time = np.asarray(time, float)
    ts = np.asarray(timestamps, float)
    out = np.zeros((54, time.size)); mk = np.zeros(time.size, np.uint8)
    for i, t in enumerate(time):
        if t < wait_time or ts.size == 0:
            out[:, i] = standing; mk[i] = standing_mask
        else:
            k = int(np.abs(ts - (t - wait_time)).argmin())
            out[:, i] = table[k]; mk[i] = masks[k]
    return out, mk


def integrate(q, v, vd, dt):
    """Semi-implicit Euler in the reference's coordinates (include/wbc.h wbc_integrate): numpy restatement.
    q [19, N], v [18, N], vd [18, N] -> (q+, v+)."""
    q = np.array(q, float); v = np.array(v, float) + dt * np.asarray(vd, float)
    w = v[0:3]
    wn = np.sqrt((w * w).sum(0))
    ang = 0.5 * wn * dt
    sc = np.where(wn > 0, np.sin(ang) / np.where(wn > 0, wn, 1.0), 0.0)
    dw = np.where(wn > 0, np.cos(ang), 1.0); dx, dy, dz = sc * w[0], sc * w[1], sc * w[2]
    w1, x1, y1, z1 = q[0].copy(), q[1].copy(), q[2].copy(), q[3].copy()
    qq = np.stack([dw * w1 - dx * x1 - dy * y1 - dz * z1, dw * x1 + dx * w1 + dy * z1 - dz * y1,
                   dw * y1 - dx * z1 + dy * w1 + dz * x1, dw * z1 + dx * y1 - dy * x1 + dz * w1])
    q[0:4] = qq / np.sqrt((qq * qq).sum(0))
    q[4:7] += dt * v[3:6]
    q[7:] += dt * v[6:]
    return q, v


# ---- robot_state_control_lcmt (lcm_types/cheetahlcm/robot_state_control_lcmt.py:28-79), the use_lcm path of
# controllers/basic_controller.py:79-87,289-314
RS_FINGERPRINT = struct.pack(">Q", ((0xbe14089c923ad667 << 1) & 0xffffffffffffffff) + (0xbe14089c923ad667 >> 63))


def robot_state_decode(buf):
    """-> (q[19], v[18], tau[12]) as the float32 values on the wire, widened to float64 (np.asarray(msg.q), :85-86)"""
    if buf[:8] != RS_FINGERPRINT:
        raise ValueError("Decode error")
    x = np.array(struct.unpack_from(">49f", buf, 8), dtype=np.float64)
    return x[:19], x[19:37], x[37:]


def robot_control_message(u, order, act_joint):
    """basic_controller.py:308-314: msg.tau = (S'u)[-12:] with S' = MakeActuationMatrix() of a plant whose canonical joint j
    sits at index order[j] and whose actuator k drives canonical joint act_joint[k]; q, v of the message stay zero."""
    t = np.zeros(12)
    for k in range(12):
        t[order[act_joint[k]]] = u[k]
    return RS_FINGERPRINT + struct.pack(">37f", *([0.0] * 37)) + struct.pack(">12f", *t)


def pd_control_law(q, v, q_nom=None, order=None, act_joint=None, kp=30.0, kd=1.5, u_max=150.0):
    """controllers/basic_controller.py:322-352 for a batch: q[19, N], v[18, N] in the plant's numbering -> u[12, N] in actuator
    order.  tau = -Kp q_err - Kd v on the joint rows (where MapQDotToVelocity is the identity), u = clip(S tau, +-u_max)."""
    q = np.asarray(q, float); v = np.asarray(v, float)
    qn = np.array([1.0, 0, 0, 0, 0, 0, 0.3] + [0.0, -0.8, 1.6] * 4) if q_nom is None else np.asarray(q_nom, float)
    order = list(range(12)) if order is None else list(order)
    act = list(range(12)) if act_joint is None else list(act_joint)
    tau_j = -(kp * (q[7:] - qn[7:, None])) - kd * v[6:]                 # plant joint order
    u = np.stack([tau_j[order[act[k]]] for k in range(12)])
    return np.clip(u, -u_max, u_max)
