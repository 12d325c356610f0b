"""ctypes view of oracle/ld/libwbc_oracle_ld.so -- the CPU oracle compiled in x87 extended precision (long double, 64-bit
mantissa): TEST INFRASTRUCTURE ONLY, like oracle_py.  Same source, same algorithm; used to adjudicate disagreements between the
HIP path and the double-precision oracle (tools/lab/truth.py).  x86-64 only (numpy's longdouble must be the C long double)."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import oracle_py as _o

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
LD = np.longdouble
assert np.dtype(LD).itemsize == C.sizeof(C.c_longdouble) == 16 and np.finfo(LD).nmant == 63, "needs x87 extended precision"
_ldp = C.POINTER(C.c_longdouble)


def _widen(t):
    """ctypes type with every c_double replaced by c_longdouble (arrays and structures, recursively)."""
    if t is C.c_double:
        return C.c_longdouble
    if isinstance(t, type) and issubclass(t, C.Array):
        return _widen(t._type_) * t._length_
    if isinstance(t, type) and issubclass(t, C.Structure):
        return type(t.__name__ + "LD", (C.Structure,), {"_fields_": [(n, _widen(ft)) for n, ft in t._fields_]})
    return t


OrcModelLD = _widen(_o.OrcModel)
OrcParamsLD = _widen(_o.OrcParams)


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", _HERE, "-s", "ld"], stderr=subprocess.DEVNULL)
        _LIB = C.CDLL(os.path.join(_HERE, "ld", "libwbc_oracle_ld.so"))
        _LIB.orc_step_batch.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(_ldp)


def model(name_or_table):
    t = _o.load_model_json(name_or_table) if isinstance(name_or_table, str) else name_or_table
    flat = np.ascontiguousarray(t["flat"], dtype=LD)          # the doubles of the model table, exactly
    m = OrcModelLD()
    lib().orc_model_from_flat(_p(flat), C.byref(m))
    for i, a in enumerate(t.get("act_perm", range(12))):
        m.act_perm[i] = a
    return m


def params(kind, **over):
    p = OrcParamsLD()
    (lib().orc_params_id_default if _o.kind_index(kind) in (0, 3) else lib().orc_params_mptc_default)(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def step_batch(kind, m, p, q, v, targets, mask, mu=None, mass_scale=None, nthreads=0):
    """oracle_py.step_batch in extended precision: float64 inputs are widened exactly, outputs come back as longdouble."""
    q, v, targets = (np.ascontiguousarray(a, dtype=LD) for a in (q, v, targets))
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    n = q.shape[1]
    tau = np.zeros((12, n), LD); met = np.zeros((4, n), LD); st = np.zeros(n, dtype=np.int32)
    mu_a = None if mu is None else np.ascontiguousarray(mu, dtype=LD)
    ms_a = None if mass_scale is None else np.ascontiguousarray(mass_scale, dtype=LD)
    lib().orc_step_batch(C.byref(m), C.byref(p), _o.kind_index(kind), n, n, _p(q), _p(v), _p(targets),
                         mask.ctypes.data_as(C.POINTER(C.c_ubyte)), None if mu_a is None else _p(mu_a),
                         None if ms_a is None else _p(ms_a), _p(tau), _p(met), st.ctypes.data_as(C.POINTER(C.c_int)), int(nthreads))
    return tau, met, st
