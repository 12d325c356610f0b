"""Round 4 analysis: accuracy of QR + row-wise triangular inversion in float64 for different pivot orders of the 12 reduced variables, on the
kernel's own rows (dump_qr_inputs.py), measured through the long-double projection on the true active set.   python3 pivot_order_stats.py id 2 2048"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/lab'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from quadruped_drake_amd import workloads
from oracle import oracle_py as orc
import drop_lab, gi_lab
LD = np.longdouble
L = C.CDLL('/tmp/lab/libdumpA.so'); dp = C.POINTER(C.c_double)
kind = sys.argv[1]; cfg = int(sys.argv[2]); n = int(sys.argv[3]); k = {"id": 0, "mptc": 1}[kind]; seed = int(sys.argv[4]) if len(sys.argv) > 4 else 50000 + cfg
P1 = 6 if kind == "id" else 18
b = workloads.make_batch(cfg, n=n, seed=seed); t = orc.load_model_json(b["model"])
q, v, tg = (np.ascontiguousarray(b[x]) for x in ("q", "v", "targets")); flat = np.ascontiguousarray(t["flat"]); mask = np.ascontiguousarray(b["mask"])
buf = np.zeros((n, 16, 16)); qbuf = np.zeros((n, 16, 64))
tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
L.host_gi_dump.argtypes = [C.c_void_p]; L.host_qr_dump.argtypes = [C.c_void_p]
L.host_gi_dump(buf.ctypes.data_as(C.c_void_p)); L.host_qr_dump(qbuf.ctypes.data_as(C.c_void_p))
mu_a = np.ascontiguousarray(b["mu"]) if b.get("mu") is not None else None
ms_a = np.ascontiguousarray(b["mass_scale"]) if b.get("mass_scale") is not None else None
rc = L.host_hex_batch(k, flat.ctypes.data_as(dp), None, None, None, n, n, q.ctypes.data_as(dp), v.ctypes.data_as(dp), tg.ctypes.data_as(dp), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), mu_a.ctypes.data_as(dp) if mu_a is not None else None, ms_a.ctypes.data_as(dp) if ms_a is not None else None, tau.ctypes.data_as(dp), met.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)))
L.host_gi_dump(None); L.host_qr_dump(None)
lanes = [4 * (i // 3) + i % 3 for i in range(12)]
mu_n = buf[:, 0, 14]; inv_s = buf[:, :, 15].max(1); ct = buf[:, ::4, 15] > 0
def hhqr(A, T):
    A = A.astype(T).copy(); m, nn = A.shape; ncol = nn - 1
    for c in range(ncol):
        x = A[c:, c].copy(); nrm = np.sqrt(x @ x)
        if nrm == 0: continue
        alpha = -nrm if x[0] > 0 else nrm
        vv = x.copy(); vv[0] -= alpha; beta = 2 / (vv @ vv)
        A[c:, :] -= np.outer(vv, (vv @ A[c:, :]) * beta)
    return A[:ncol, :ncol], A[:ncol, ncol]
def trinv(R, T):
    nn = R.shape[0]; X = np.zeros((nn, nn), T)
    for j in range(nn):
        e = np.zeros(nn, T); e[j] = 1
        for i in range(nn - 1, -1, -1): X[i, j] = (e[i] - R[i, i + 1:] @ X[i + 1:, j]) / R[i, i]
    return X
def inv_rowwise(R):
    n_ = R.shape[0]; X = np.zeros((n_, n_))
    for rr in range(n_):
        for c in range(rr, n_):
            X[rr, c] = ((1.0 if c == rr else 0.0) - (X[rr, :c] @ R[:c, c])) / R[c, c]
    return X
orders = {"natural": list(range(12)), "z0z1z2x0y0y2": [2, 5, 8, 0, 1, 7, 3, 4, 6, 9, 10, 11]}
if len(sys.argv) > 5:
    for o in sys.argv[5:]: orders[o] = [int(x) for x in o.split('.')]
res = {kk: [] for kk in list(orders) + ["natural colwise"]}
for i in range(n):
    Rc = qbuf[i, :, :12]; Ac = qbuf[i, :, 16:16 + P1 + 12]
    A = np.zeros((12 + P1 + 12, 13))
    for c, ln in enumerate(lanes): A[:12, c] = Rc[ln]; A[12:, c] = Ac[ln]
    A[:12, 12] = Rc[3]; A[12:, 12] = Ac[3]
    N = drop_lab.normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
    Rl, yl = hhqr(A, LD); Jl = trinv(Rl, LD)
    D = (Jl.astype(float).T @ N.T).T
    Aset, adds, drops, ok = gi_lab.gi(D, yl.astype(float), elig)
    if not ok or not Aset: continue
    def proj(Jx, z0x, NN):
        A_ = NN[Aset].astype(LD); Hinv = Jx @ Jx.T; S = A_ @ Hinv @ A_.T; rhs = A_ @ z0x
        lam = np.linalg.solve(S.astype(float), rhs.astype(float)).astype(LD)
        for _ in range(4): lam = lam + np.linalg.solve(S.astype(float), (rhs - S @ lam).astype(float)).astype(LD)
        return (z0x - Hinv @ (A_.T @ lam)).astype(float)
    zt = proj(Jl, Jl @ yl, N); sc = 1 + np.abs(zt).max()
    for name, order in orders.items():
        Ap = A[:, order + [12]]; top = Ap[:12]; rows = [int(np.argmax(np.abs(top[:, c]))) if np.abs(top[:, c]).max() > 0 else -1 for c in range(12)]
        # rows with zero diag (MPTC swing diag rows are zero): keep a zero row
        newtop = np.zeros((12, 13))
        for c in range(12):
            if rows[c] >= 0: newtop[c] = top[rows[c]]
        Ap = np.vstack([newtop, Ap[12:]])
        Rp, yp = hhqr(Ap, np.float64); Jp = inv_rowwise(Rp)
        zz = proj(Jp.astype(LD), Jp.astype(LD) @ yp.astype(LD), N[:, order])
        res[name].append(np.abs(zz - zt[order]).max() / sc)
        if name == "natural":
            Jc = trinv(Rp, np.float64); zz = proj(Jc.astype(LD), Jc.astype(LD) @ yp.astype(LD), N)
            res["natural colwise"].append(np.abs(zz - zt).max() / sc)
for kk, vv in res.items():
    vv = np.array(vv); print("%-22s n=%d median %.1e p99 %.1e max %.1e  >1e-8: %d >1e-7: %d >1e-6: %d" % (kk, len(vv), np.median(vv), np.percentile(vv, 99), vv.max(), (vv > 1e-8).sum(), (vv > 1e-7).sum(), (vv > 1e-6).sum()))
