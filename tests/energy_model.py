"""Independent numpy derivation of the rigid-body terms (tests only).

Nothing here shares code or recursion structure with oracle/wbc_oracle.c or the HIP kernels:
it builds plain forward kinematics from the model JSON, one 6x18 body Jacobian per body, and
then uses Kane / d'Alembert projection:

    M      = sum_b  Jw_b' Ic_b Jw_b + m_b Jc_b' Jc_b          (== d2T/dv2 of T = 1/2 sum ...)
    tau_g  = sum_b  m_b g Jc_b' e_z                           (== dU/dq N, reference sign)
    Cv     = sum_b  Jw_b' (Ic_b al_b + w_b x Ic_b w_b) + m_b Jc_b' a_b   with vdot = 0,

where the body accelerations (al_b, a_b) at vdot = 0 are obtained by *numerically*
differentiating the body twists J_b(q(t)) v along the exact flow qdot = N(q) v (central
difference).  This is valid for Drake's quasi-velocities v = [w_WB(world), v_WBo(world), qd]
(the textbook Lagrange formula Mdot v - 1/2 grad(v'Mv) is NOT, because w is non-holonomic).
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    with open(os.path.join(HERE, "..", "quadruped_drake_amd", "models", name + ".json")) as f:
        return json.load(f)


def quat_R(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rodrigues(a, th):
    a = np.asarray(a, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def skew(r):
    return np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])


def I6_to_mat(I):
    return np.array([[I[0], I[3], I[4]], [I[3], I[1], I[5]], [I[4], I[5], I[2]]])


def bodies(model, q):
    """List of dicts (R, p origin, mass, com_world, Ic_world, Jw 3x18, Jc 3x18) + feet (p, J)."""
    q = np.asarray(q, float)
    Rb = quat_R(q[:4]); pb = q[4:7]
    out = []
    feet = []

    def add(R, p, mass, com, I6, Jw, Jo):
        c = R @ np.asarray(com)
        Io = R @ I6_to_mat(I6) @ R.T
        Ic = Io - mass * (c @ c * np.eye(3) - np.outer(c, c))
        Jc = Jo - skew(c) @ Jw  # v_c = v_o + w x c
        out.append(dict(R=R, p=p, m=mass, c=p + c, Ic=Ic, Jw=Jw, Jc=Jc))

    Jw0 = np.zeros((3, 18)); Jw0[:, 0:3] = np.eye(3)
    Jo0 = np.zeros((3, 18)); Jo0[:, 3:6] = np.eye(3)
    b = model["base"]
    add(Rb, pb, b["mass"], b["com"], b["I"], Jw0, Jo0)
    for l, leg in enumerate(model["legs"]):
        R, p, Jw, Jo = Rb, pb, Jw0, Jo0
        for k, L in enumerate(leg["links"]):
            off = R @ np.asarray(L["off"])
            a_w = R @ np.asarray(L["axis"])
            # origin of the child moves with the parent: v_o' = v_o + w x off
            Jo = Jo - skew(off) @ Jw
            p = p + off
            Jw = Jw.copy()
            Jw[:, 6 + 3 * l + k] = a_w
            R = R @ rodrigues(L["axis"], q[7 + 3 * l + k])
            add(R, p, L["mass"], L["com"], L["I"], Jw, Jo)
        d = R @ np.asarray(leg["foot_off"])
        feet.append(dict(p=p + d, J=Jo - skew(d) @ Jw))
    return out, feet


def mass_matrix(model, q):
    bs, _ = bodies(model, q)
    M = np.zeros((18, 18))
    for b in bs:
        M += b["Jw"].T @ b["Ic"] @ b["Jw"] + b["m"] * b["Jc"].T @ b["Jc"]
    return M


def gravity_term(model, q):
    bs, _ = bodies(model, q)
    g = model["gravity"]
    t = np.zeros(18)
    for b in bs:
        t += b["m"] * g * b["Jc"][2, :]
    return t


def potential(model, q):
    bs, _ = bodies(model, q)
    return sum(b["m"] * model["gravity"] * b["c"][2] for b in bs)


def flow(q, v, h):
    """Exact integral of qdot = N(q) v over time h for constant v."""
    q = np.asarray(q, float).copy()
    w = np.asarray(v[:3], float)
    ang = np.linalg.norm(w) * h
    if ang != 0:
        ax = w / np.linalg.norm(w)
        dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    else:
        dq = np.array([1.0, 0, 0, 0])
    w0, x0, y0, z0 = dq
    w1, x1, y1, z1 = q[:4]
    # world-frame angular velocity: q(t+h) = dq (x) q
    q[:4] = [w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1,
             w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
             w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1,
             w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1]
    q[4:7] += h * np.asarray(v[3:6])
    q[7:] += h * np.asarray(v[6:])
    return q


def bias_term(model, q, v, h=1e-5):
    """C(q,v)v by Kane projection with numerically differentiated body twists."""
    v = np.asarray(v, float)
    bs0, _ = bodies(model, q)
    bsp, _ = bodies(model, flow(q, v, h))
    bsm, _ = bodies(model, flow(q, v, -h))
    out = np.zeros(18)
    for b0, bp, bm in zip(bs0, bsp, bsm):
        w = b0["Jw"] @ v
        al = (bp["Jw"] @ v - bm["Jw"] @ v) / (2 * h)
        ac = (bp["Jc"] @ v - bm["Jc"] @ v) / (2 * h)
        out += b0["Jw"].T @ (b0["Ic"] @ al + np.cross(w, b0["Ic"] @ w)) + b0["m"] * b0["Jc"].T @ ac
    return out


def foot_jacobian_dot_fd(model, q, v, foot, h=1e-5):
    _, fp = bodies(model, flow(q, v, h))
    _, fm = bodies(model, flow(q, v, -h))
    return (fp[foot]["J"] - fm[foot]["J"]) / (2 * h)


def rpy_from_R(R):
    return np.array([np.arctan2(R[2, 1], R[2, 2]),
                     np.arctan2(-R[2, 0], np.hypot(R[0, 0], R[1, 0])),
                     np.arctan2(R[1, 0], R[0, 0])])


def rpy_E(rpy):
    r, p, y = rpy
    return np.array([[np.cos(p) * np.cos(y), -np.sin(y), 0],
                     [np.cos(p) * np.sin(y), np.cos(y), 0],
                     [-np.sin(p), 0, 1]])
