"""ctypes view of oracle/libwbc_oracle.so -- CPU ORACLE, test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (quadruped_drake_amd) never does.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class OrcLink(C.Structure):
    _fields_ = [("off", C.c_double * 3), ("axis", C.c_double * 3), ("mass", C.c_double),
                ("com", C.c_double * 3), ("I", C.c_double * 6)]


class OrcModel(C.Structure):
    _fields_ = [("base_mass", C.c_double), ("base_com", C.c_double * 3), ("base_I", C.c_double * 6),
                ("link", (OrcLink * 3) * 4), ("foot_off", (C.c_double * 3) * 4),
                ("gravity", C.c_double), ("act_perm", C.c_int * 12)]


class OrcParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("Kp_body_p", "Kd_body_p", "Kp_body_rpy", "Kd_body_rpy", "Kp_foot", "Kd_foot",
                 "w_body", "w_foot", "mu", "Kd_contact", "tau_max", "tiebreak_eps2")]


class OrcQP(C.Structure):
    _fields_ = [("n", C.c_int), ("nc", C.c_int), ("me", C.c_int), ("mi", C.c_int), ("mls", C.c_int),
                ("Q", C.c_double * (43 * 43)), ("c", C.c_double * 43),
                ("Aeq", C.c_double * (30 * 43)), ("beq", C.c_double * 30),
                ("Ain", C.c_double * (42 * 43)), ("bin", C.c_double * 42),
                ("Als", C.c_double * (18 * 43)), ("bls", C.c_double * 18),
                ("x", C.c_double * 43), ("iters", C.c_int), ("status", C.c_int),
                ("primal_res", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "libwbc_oracle.so")
    src = os.path.join(_HERE, "wbc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_qp_solve.restype = C.c_int
        _LIB.orc_id_control_law.restype = C.c_int
        _LIB.orc_mptc_control_law.restype = C.c_int
        _LIB.orc_pc_control_law.restype = C.c_int
        _LIB.orc_clf_control_law.restype = C.c_int
        _LIB.orc_step_batch.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(c_double_p)


def load_model_json(name):
    path = os.path.join(_HERE, "..", "quadruped_drake_amd", "models", name + ".json")
    with open(path) as f:
        return json.load(f)


def model(name_or_table):
    t = load_model_json(name_or_table) if isinstance(name_or_table, str) else name_or_table
    flat = np.ascontiguousarray(t["flat"], dtype=np.float64)
    assert flat.size == 215
    m = OrcModel()
    lib().orc_model_from_flat(_p(flat), C.byref(m))
    for i, a in enumerate(t.get("act_perm", range(12))):
        m.act_perm[i] = a
    return m


def model_scaled(name_or_table, mass_scale):
    """The per-instance trunk mass / inertia scale of BASELINE config 5, as orc_step_batch applies it."""
    m = model(name_or_table)
    m.base_mass *= mass_scale
    for k in range(6):
        m.base_I[k] *= mass_scale
    return m


def kind_index(kind):
    if isinstance(kind, str):
        return {"id": 0, "mptc": 1, "pc": 2, "clf": 3}[kind.lower()]
    return int(kind)


def params(kind):
    p = OrcParams()
    if kind_index(kind) in (0, 3):       # CLF inherits from IDController (clf_controller.py:3)
        lib().orc_params_id_default(C.byref(p))
    else:
        lib().orc_params_mptc_default(C.byref(p))
    return p


def calc_dynamics(m, q, v):
    q = np.ascontiguousarray(q, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    M = np.zeros((18, 18)); Cv = np.zeros(18); tg = np.zeros(18)
    lib().orc_calc_dynamics(C.byref(m), _p(q), _p(v), _p(M), _p(Cv), _p(tg))
    return M, Cv, tg


def inverse_dynamics(m, q, v, vd, with_gravity=True):
    q, v, vd = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, v, vd))
    tau = np.zeros(18)
    lib().orc_inverse_dynamics(C.byref(m), _p(q), _p(v), _p(vd), int(with_gravity), _p(tau))
    return tau


def coriolis_matrix(m, q, v):
    q = np.ascontiguousarray(q, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
    Cm = np.zeros((18, 18))
    lib().orc_coriolis_matrix(C.byref(m), _p(q), _p(v), _p(Cm))
    return Cm


def foot_quantities(m, q, v, foot):
    q = np.ascontiguousarray(q, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
    p = np.zeros(3); J = np.zeros((3, 18)); Jdv = np.zeros(3)
    lib().orc_foot_quantities(C.byref(m), _p(q), _p(v), int(foot), _p(p), _p(J), _p(Jdv))
    return p, J, Jdv


def foot_jacobian_dot(m, q, v, foot):
    q = np.ascontiguousarray(q, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
    Jd = np.zeros((3, 18))
    lib().orc_foot_jacobian_dot(C.byref(m), _p(q), _p(v), int(foot), _p(Jd))
    return Jd


def body_quantities(m, q, v):
    q = np.ascontiguousarray(q, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
    R = np.zeros((3, 3)); p = np.zeros(3); J = np.zeros((6, 18)); Jdv = np.zeros(6)
    lib().orc_body_quantities(C.byref(m), _p(q), _p(v), _p(R), _p(p), _p(J), _p(Jdv))
    return R, p, J, Jdv


def rpy_from_R(R):
    R = np.ascontiguousarray(R, dtype=np.float64)
    rpy = np.zeros(3)
    lib().orc_rpy_from_R(_p(R), _p(rpy))
    return rpy


def control_law(kind, m, p, q, v, targets, contact, want_qp=False):
    """One tick.  Returns (tau[12], metrics[4], status[, qp dict])."""
    q, v, targets = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, v, targets))
    ct = (C.c_int * 4)(*[int(bool(c)) for c in contact])
    tau = np.zeros(12); met = np.zeros(4)
    qp = OrcQP() if want_qp else None
    fn = [lib().orc_id_control_law, lib().orc_mptc_control_law, lib().orc_pc_control_law,
          lib().orc_clf_control_law][kind_index(kind)]
    st = fn(C.byref(m), C.byref(p), _p(q), _p(v), _p(targets), ct, _p(tau), _p(met),
            C.byref(qp) if want_qp else None)
    if not want_qp:
        return tau, met, st
    n = qp.n
    d = dict(n=n, nc=qp.nc, me=qp.me, mi=qp.mi, mls=qp.mls, iters=qp.iters, status=qp.status,
             primal_res=qp.primal_res,
             Q=np.array(qp.Q[:n * n]).reshape(n, n), c=np.array(qp.c[:n]),
             Aeq=np.array(qp.Aeq[:qp.me * n]).reshape(qp.me, n), beq=np.array(qp.beq[:qp.me]),
             Ain=np.array(qp.Ain[:qp.mi * n]).reshape(qp.mi, n), bin=np.array(qp.bin[:qp.mi]),
             Als=np.array(qp.Als[:qp.mls * n]).reshape(qp.mls, n), bls=np.array(qp.bls[:qp.mls]),
             x=np.array(qp.x[:n]))
    return tau, met, st, d


def qp_solve(Als, bls, eps2, dreg, Aeq, beq, Ain, bin_):
    Als, bls, dreg, Aeq, beq, Ain, bin_ = (np.ascontiguousarray(a, dtype=np.float64)
                                           for a in (Als, bls, dreg, Aeq, beq, Ain, bin_))
    n = dreg.size
    x = np.zeros(n); it = C.c_int(0); res = C.c_double(0)
    st = lib().orc_qp_solve(n, bls.size, _p(Als), _p(bls), C.c_double(eps2), _p(dreg), beq.size, _p(Aeq),
                            _p(beq), bin_.size, _p(Ain), _p(bin_), _p(x), C.byref(it), C.byref(res))
    return x, st, it.value, res.value


def step_batch(kind, m, p, q, v, targets, mask, mu=None, mass_scale=None, nthreads=0):
    """SoA batch: q[19,N], v[18,N], targets[54,N], mask[N] uint8 -> tau[12,N], metrics[4,N], status[N]."""
    q, v, targets = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, v, targets))
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    n = q.shape[1]
    tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, dtype=np.int32)
    mu_a = np.ascontiguousarray(mu, dtype=np.float64) if mu is not None else None
    ms_a = np.ascontiguousarray(mass_scale, dtype=np.float64) if mass_scale is not None else None
    mu_p = _p(mu_a) if mu_a is not None else None
    ms_p = _p(ms_a) if ms_a is not None else None
    kind_i = kind_index(kind)
    lib().orc_step_batch(C.byref(m), C.byref(p), kind_i, n, n, _p(q), _p(v), _p(targets),
                         mask.ctypes.data_as(C.POINTER(C.c_ubyte)), mu_p, ms_p, _p(tau), _p(met),
                         st.ctypes.data_as(c_int_p), int(nthreads))
    return tau, met, st


def bench_batch(kind, m, p, q, v, targets, mask, mu=None, mass_scale=None, nthreads=0, reps=1):
    """`reps` passes over the batch inside one OpenMP region; returns tau of the last pass."""
    q, v, targets = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, v, targets))
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    n = q.shape[1]
    tau = np.zeros((12, n)); st = np.zeros(n, dtype=np.int32)
    mu_a = np.ascontiguousarray(mu, dtype=np.float64) if mu is not None else None
    ms_a = np.ascontiguousarray(mass_scale, dtype=np.float64) if mass_scale is not None else None
    kind_i = kind_index(kind)
    lib().orc_bench_batch(C.byref(m), C.byref(p), kind_i, n, n, _p(q), _p(v), _p(targets),
                          mask.ctypes.data_as(C.POINTER(C.c_ubyte)), _p(mu_a) if mu_a is not None else None,
                          _p(ms_a) if ms_a is not None else None, _p(tau), st.ctypes.data_as(c_int_p),
                          int(nthreads), int(reps))
    return tau, st
