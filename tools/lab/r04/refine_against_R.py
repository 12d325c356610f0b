"""Round 4 analysis: iterative refinement of the final z against the accurate triangular factor R through the active set's (defective) slot
structure -- built, measured, NOT adopted: it does not contract (profiles/r04/accuracy.md).   python3 refine_against_R.py id 2048"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/lab'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from quadruped_drake_amd import workloads
from oracle import oracle_py as orc
import drop_lab, gi_lab, refine_lab as rl
LD = np.longdouble
L = C.CDLL('/tmp/lab/libdumpA.so'); dp = C.POINTER(C.c_double)
kind = sys.argv[1]; cfg = 2; n = int(sys.argv[2]); k = {"id": 0, "mptc": 1}[kind]; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 50000 + cfg
P1 = 6 if kind == "id" else 18
b = workloads.make_batch(cfg, n=n, seed=seed); t = orc.load_model_json(b["model"])
q, v, tg = (np.ascontiguousarray(b[x]) for x in ("q", "v", "targets")); flat = np.ascontiguousarray(t["flat"]); mask = np.ascontiguousarray(b["mask"])
buf = np.zeros((n, 16, 16)); qbuf = np.zeros((n, 16, 64))
tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
L.host_gi_dump.argtypes = [C.c_void_p]; L.host_qr_dump.argtypes = [C.c_void_p]
L.host_gi_dump(buf.ctypes.data_as(C.c_void_p)); L.host_qr_dump(qbuf.ctypes.data_as(C.c_void_p))
rc = L.host_hex_batch(k, flat.ctypes.data_as(dp), None, None, None, n, n, q.ctypes.data_as(dp), v.ctypes.data_as(dp), tg.ctypes.data_as(dp), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), None, None, tau.ctypes.data_as(dp), met.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)))
L.host_gi_dump(None); L.host_qr_dump(None)
lanes = [4 * (i // 3) + i % 3 for i in range(12)]
J = buf[:, lanes, :12]; z0 = buf[:, lanes, 13]; mu_n = buf[:, 0, 14]; inv_s = buf[:, :, 15].max(1); ct = buf[:, ::4, 15] > 0
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
res = {kk: [] for kk in ("accumulated", "term2", "opt5 on accumulated", "opt5 on term2", "opt5 x2")}
for i in range(n):
    Rc = qbuf[i, :, :12]; Ac = qbuf[i, :, 16:16 + P1 + 12]
    A = np.zeros((12 + P1 + 12, 13))
    for c, ln in enumerate(lanes): A[:12, c] = Rc[ln]; A[12:, c] = Ac[ln]
    A[:12, 12] = Rc[3]; A[12:, 12] = Ac[3]
    Rk = np.zeros((12, 12))
    for c, ln in enumerate(lanes): Rk[:, c] = qbuf[i, ln, 48:60]
    yk = qbuf[i, 3, 48:60].copy(); Rk = np.triu(Rk)
    N = drop_lab.normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
    s = rl.solve(J[i], z0[i], yk, N, elig)
    if s is None or not s['act'].any(): continue
    Aset = list(np.where(s['act'])[0])
    # truth: ld QR of A, ld projection on the same active set
    Rl, yl = hhqr(A, LD); Jl = trinv(Rl, LD)
    A_ = N[Aset].astype(LD); Hinv = Jl @ Jl.T; S = A_ @ Hinv @ A_.T; rhs = A_ @ (Jl @ yl)
    lam = np.linalg.solve(S.astype(float), rhs.astype(float)).astype(LD)
    for _ in range(4): lam = lam + np.linalg.solve(S.astype(float), (rhs - S @ lam).astype(float)).astype(LD)
    zt = (Jl @ yl - Hinv @ (A_.T @ lam)).astype(float); sc = 1 + np.abs(zt).max(); e = lambda z: np.abs(z - zt).max() / sc
    def opt5(z):
        rho = yk - Rk @ z
        g = Rk.T @ rho
        tt = s['Jr'].T @ g; qq = s['q']
        return z + s['Jr'][:, qq:] @ tt[qq:]
    z2 = rl.refine(s, N, 2)
    res["accumulated"].append(e(s['z'])); res["term2"].append(e(z2)); res["opt5 on accumulated"].append(e(opt5(s['z']))); res["opt5 on term2"].append(e(opt5(z2))); res["opt5 x2"].append(e(opt5(opt5(s['z']))))
for kk, vv in res.items():
    vv = np.array(vv); print("%-22s n=%d median %.1e p99 %.1e max %.1e  >1e-8: %d >1e-7: %d >1e-6: %d" % (kk, len(vv), np.median(vv), np.percentile(vv, 99), vv.max(), (vv > 1e-8).sum(), (vv > 1e-7).sum(), (vv > 1e-6).sum()))
