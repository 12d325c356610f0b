"""ctypes view of tools/libhost_tick.so: the kernel math instantiated on the host (tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_ROOT, "tools", "libhost_tick.so")
        srcs = [os.path.join(_ROOT, "tools", "host_tick.cpp"), os.path.join(_ROOT, "tools", "wbc_scalar_tick.hpp"),
                os.path.join(_ROOT, "quadruped_drake_amd", "csrc", "wbc_tick.hpp"),
                os.path.join(_ROOT, "quadruped_drake_amd", "csrc", "wbc_model.hpp"),
                os.path.join(_ROOT, "quadruped_drake_amd", "csrc", "wbc_hex.hpp"),
                os.path.join(_ROOT, "quadruped_drake_amd", "csrc", "wbc_traj_dev.hpp")]
        if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-ffp-contract=off",
                                   "-o", so, srcs[0]])
        _LIB = C.CDLL(so)
    return _LIB


def build_variant(tmp_path, name, flags, compiler="g++"):
    """A host instantiation of the kernel headers with the CLOSED alternatives available again (the -DWBC_... switches that round 6 removed from
    the product source: tools/lab/patches/closed_switches.patch re-introduces them): the sources are copied to `tmp_path`, patched there and
    compiled with `flags`.  Without flags the patched tree is the product's code, so a comparison against the default build is a comparison
    against what ships.  Returns the loaded library."""
    import shutil
    tree = os.path.join(str(tmp_path), "tree_" + name)
    for sub in ("quadruped_drake_amd/csrc", "tools", "include"):
        os.makedirs(os.path.join(tree, sub), exist_ok=True)
    for f in os.listdir(os.path.join(_ROOT, "quadruped_drake_amd", "csrc")):
        shutil.copy(os.path.join(_ROOT, "quadruped_drake_amd", "csrc", f), os.path.join(tree, "quadruped_drake_amd", "csrc", f))
    for f in ("host_tick.cpp", "wbc_scalar_tick.hpp"):
        shutil.copy(os.path.join(_ROOT, "tools", f), os.path.join(tree, "tools", f))
    for f in os.listdir(os.path.join(_ROOT, "include")):
        shutil.copy(os.path.join(_ROOT, "include", f), os.path.join(tree, "include", f))
    subprocess.check_call(["patch", "-p1", "-s", "--no-backup-if-mismatch", "-i", os.path.join(_ROOT, "tools", "lab", "patches", "closed_switches.patch")], cwd=tree)
    so = os.path.join(str(tmp_path), name)
    # HOST_TICK_HEX_ONLY: the 16-lane emulation alone (host_hex_batch is all a variant is asked) -- half the compile time of the full tool
    subprocess.check_call([compiler, "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-ffp-contract=off", "-DHOST_TICK_HEX_ONLY"] + list(flags) +
                          ["-o", so, os.path.join(tree, "tools", "host_tick.cpp")])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def run(kind, flat, q, v, targets, mask, mu=None, mass_scale=None, params12=None, q_perm=None, act_perm=None,
        want_vdot=False, hexv=False):
    q, v, targets = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, v, targets))
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    n = q.shape[1]
    tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
    mu = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
    ms = None if mass_scale is None else np.ascontiguousarray(mass_scale, dtype=np.float64)
    pp = None if params12 is None else np.ascontiguousarray(params12, dtype=np.float64)
    qp = None if q_perm is None else np.ascontiguousarray(q_perm, dtype=np.int32)
    ap = None if act_perm is None else np.ascontiguousarray(act_perm, dtype=np.int32)
    k = {"id": 0, "mptc": 1, "pc": 2, "clf": 3}[kind.lower()] if isinstance(kind, str) else int(kind)
    vdot = np.zeros((18, n)) if want_vdot else None
    lib().host_set_vdot_sink.argtypes = [C.c_void_p]
    lib().host_set_vdot_sink(vdot.ctypes.data_as(C.c_void_p) if want_vdot else None)
    fn = lib().host_hex_batch if hexv else lib().host_tick_batch
    rc = fn(k, _p(flat), _p(pp), qp.ctypes.data_as(_ip) if qp is not None else None,
                               ap.ctypes.data_as(_ip) if ap is not None else None, n, n, _p(q), _p(v),
                               _p(targets), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), _p(mu), _p(ms),
                               _p(tau), _p(met), st.ctypes.data_as(_ip), it.ctypes.data_as(_ip))
    assert rc == 0
    lib().host_set_vdot_sink(None)
    if want_vdot:
        return tau, met, st, it, vdot
    return tau, met, st, it


def count(kind, flat, q, v, targets, mask, mu=None, mass_scale=None):
    q, v, targets = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, v, targets))
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    flat = np.ascontiguousarray(flat, dtype=np.float64)
    n = q.shape[1]
    out = np.zeros(6)
    mu = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
    ms = None if mass_scale is None else np.ascontiguousarray(mass_scale, dtype=np.float64)
    k = {"id": 0, "mptc": 1, "pc": 2, "clf": 3}[kind.lower()] if isinstance(kind, str) else int(kind)
    rc = lib().host_tick_count(k, _p(flat), None, n, n, _p(q), _p(v), _p(targets),
                               mask.ctypes.data_as(C.POINTER(C.c_ubyte)), _p(mu), _p(ms), _p(out))
    assert rc == 0
    return dict(zip(["add", "mul", "div", "sqrt", "trig", "cmp"], out))
