"""Round 4 analysis: where does the QR stage of the kernel lose accuracy on a given robot?  R (fine), the explicit inverse J = R^-1 by row-wise
forward substitution (not fine when the eps-sized pivots are interleaved with level-1 sized ones), column-wise inversion, graded pivot order.
   python3 qr_check.py 831,523"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools/lab')
from quadruped_drake_amd import workloads
from oracle import oracle_py as orc, oracle_ld as old
import drop_lab, gi_lab
LD = np.longdouble
L = C.CDLL('/tmp/lab/libdumpA.so'); dp = C.POINTER(C.c_double)
kind = 'id'; cfg = 2; n = 2048; k = 0
b = workloads.make_batch(cfg, n=n, seed=50000 + cfg); t = orc.load_model_json(b["model"])
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
    """Householder QR of [A|b] in dtype T, natural column order, returns R (12x12), y (12)"""
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
        for i in range(nn - 1, -1, -1):
            X[i, j] = (e[i] - R[i, i + 1:] @ X[i + 1:, j]) / R[i, i]
    return X
m_ = orc.model(b["model"]); p_ = orc.params(kind)
for i in [int(x) for x in sys.argv[1].split(',')]:
    Rc = qbuf[i, :, :12]; Ac = qbuf[i, :, 16:16 + 18]
    A = np.zeros((30, 13))
    # Rcol on lane(c): column c of initial R (diag): R0[k][c] = Rcol_lane(c)[k]
    for c, ln in enumerate(lanes): A[:12, c] = Rc[ln]; A[12:, c] = Ac[ln]
    A[:12, 12] = Rc[3]; A[12:, 12] = Ac[3]
    sv = np.linalg.svd(A[:, :12], compute_uv=False)
    Rl, yl = hhqr(A, LD); Jl = trinv(Rl, LD); z0l = Jl @ yl
    Rd, yd = hhqr(A, np.float64); Jd = trinv(Rd, np.float64); z0d = Jd @ yd
    print("robot %d: singular values of A: %s" % (i, " ".join("%.1e" % s for s in sv)))
    print("   diag R (ld): %s" % " ".join("%.1e" % abs(float(x)) for x in np.diag(Rl)))
    print("   z0: kernel vs ld-QR %.2e ; numpy-double-QR vs ld %.2e" % (np.abs(z0[i] - z0l.astype(float)).max() / (1 + np.abs(z0[i]).max()), np.abs(z0d - z0l.astype(float)).max() / (1 + np.abs(z0d).max())))
    # H^-1 comparison
    Hk = J[i] @ J[i].T; Hl = (Jl @ Jl.T).astype(float)
    print("   H^-1 = JJ': kernel vs ld rel(max-norm) %.2e" % (np.abs(Hk - Hl).max() / np.abs(Hl).max()))
    # solve the constrained problem with ld J,z0 (active set from clean GI), compare with oracle f
    N = drop_lab.normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
    Jlf = Jl.astype(float); z0lf = z0l.astype(float)
    D = (Jlf.T @ N.T).T; y0 = yl.astype(float)  # note J here is R^-1 (un-rotated) => y0 = y
    Aset, adds, drops, ok = gi_lab.gi(D, y0, elig)
    def ref_ld(Jx, z0x, Aset):
        A_ = N[Aset].astype(LD); Hinv = Jx @ Jx.T; S = A_ @ Hinv @ A_.T; rhs = A_ @ z0x
        lam = np.linalg.solve(S.astype(float), rhs.astype(float)).astype(LD)
        for _ in range(4): lam = lam + np.linalg.solve(S.astype(float), (rhs - S @ lam).astype(float)).astype(LD)
        return (z0x - Hinv @ (A_.T @ lam)).astype(float)
    ze_l = ref_ld(Jl, z0l, Aset)
    ze_k = ref_ld(J[i].astype(LD), z0[i].astype(LD), Aset)
    tau1, met1, st1, qp = orc.control_law(kind, m_, p_, b["q"][:, i], b["v"][:, i], b["targets"][:, i], [1, 1, 1, 1], want_qp=True)
    f = qp["x"][30:42]
    sc = 1 + np.abs(f).max()
    print("   constrained z: (ld QR of kernel's A) vs oracle f %.2e ; (kernel J,z0) vs oracle f %.2e ; active %s" % (np.abs(ze_l - f).max() / sc, np.abs(ze_k - f).max() / sc, sorted(Aset)))
print("==== which step loses: QR (double) or the triangular inversion (double) or z0 = J y")
for i in [int(x) for x in sys.argv[1].split(',')]:
    Rc = qbuf[i, :, :12]; Ac = qbuf[i, :, 16:16 + 18]
    A = np.zeros((30, 13))
    for c, ln in enumerate(lanes): A[:12, c] = Rc[ln]; A[12:, c] = Ac[ln]
    A[:12, 12] = Rc[3]; A[12:, 12] = Ac[3]
    Rl, yl = hhqr(A, LD); Rd, yd = hhqr(A, np.float64)
    N = drop_lab.normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
    Jl = trinv(Rl, LD)
    D = (Jl.astype(float).T @ N.T).T
    Aset, adds, drops, ok = gi_lab.gi(D, yl.astype(float), elig)
    def ref_ld(Jx, z0x):
        A_ = N[Aset].astype(LD); Hinv = Jx @ Jx.T; S = A_ @ Hinv @ A_.T; rhs = A_ @ z0x
        lam = np.linalg.solve(S.astype(float), rhs.astype(float)).astype(LD)
        for _ in range(4): lam = lam + np.linalg.solve(S.astype(float), (rhs - S @ lam).astype(float)).astype(LD)
        return (z0x - Hinv @ (A_.T @ lam)).astype(float)
    zt = ref_ld(Jl, Jl @ yl); sc = 1 + np.abs(zt).max()
    e = lambda z: np.abs(z - zt).max() / sc
    Jd_from_Rd_ld = trinv(Rd.astype(LD), LD)
    print("robot %d" % i)
    print("   R double, y double; inversion + z0 + projection in ld:       %.2e" % e(ref_ld(Jd_from_Rd_ld, Jd_from_Rd_ld @ yd.astype(LD))))
    Jd_from_Rl = trinv(Rl.astype(float), np.float64).astype(LD)
    print("   R ld; inversion in double; z0 = J y in ld; projection ld:     %.2e" % e(ref_ld(Jd_from_Rl, Jd_from_Rl @ yl)))
    Jd_ = trinv(Rd, np.float64)
    print("   R double; inversion double; z0 double; projection ld:         %.2e" % e(ref_ld(Jd_.astype(LD), (Jd_ @ yd).astype(LD))))
    print("   kernel J, z0; projection ld:                                  %.2e" % e(ref_ld(J[i].astype(LD), z0[i].astype(LD))))
    # alternative: projection formulated with R (no explicit inverse): z = Z w, min |R Z w - y|
    A_ = N[Aset]; U, S_, Vt = np.linalg.svd(A_); Z = Vt[len(Aset):].T
    w = np.linalg.lstsq(Rd @ Z, yd, rcond=None)[0]
    print("   nullspace LS with R double (no inverse), all double:          %.2e   cond(RZ) %.1e" % (e(Z @ w), np.linalg.cond(Rd @ Z)))
print("==== kernel's R after the append vs ld")
for i in [int(x) for x in sys.argv[1].split(',')]:
    Rc = qbuf[i, :, :12]; Ac = qbuf[i, :, 16:16 + 18]
    A = np.zeros((30, 13))
    for c, ln in enumerate(lanes): A[:12, c] = Rc[ln]; A[12:, c] = Ac[ln]
    A[:12, 12] = Rc[3]; A[12:, 12] = Ac[3]
    Rl, yl = hhqr(A, LD); Rd, yd = hhqr(A, np.float64)
    Rk = np.zeros((12, 12)); 
    for c, ln in enumerate(lanes): Rk[:, c] = qbuf[i, ln, 48:60]
    yk = qbuf[i, 3, 48:60].copy()
    Rk = np.triu(Rk)
    # fix row signs to match ld
    sg = np.sign(np.diag(Rk)) * np.sign(np.diag(Rl).astype(float)); Rk = Rk * sg[:, None]; yk = yk * sg
    sgd = np.sign(np.diag(Rd)) * np.sign(np.diag(Rl).astype(float)); Rd2 = Rd * sgd[:, None]
    rowscale = np.abs(Rl.astype(float)).max(1)
    print("robot %d: row-wise rel err of R (max over row / row max): kernel %s" % (i, " ".join("%.0e" % x for x in (np.abs(Rk - Rl.astype(float)).max(1) / rowscale))))
    print("                                                        numpy  %s" % " ".join("%.0e" % x for x in (np.abs(Rd2 - Rl.astype(float)).max(1) / rowscale)))
    N = drop_lab.normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
    Jl = trinv(Rl, LD); D = (Jl.astype(float).T @ N.T).T
    Aset, adds, drops, ok = gi_lab.gi(D, yl.astype(float), elig)
    def ref_ld(Jx, z0x):
        A_ = N[Aset].astype(LD); Hinv = Jx @ Jx.T; S = A_ @ Hinv @ A_.T; rhs = A_ @ z0x
        lam = np.linalg.solve(S.astype(float), rhs.astype(float)).astype(LD)
        for _ in range(4): lam = lam + np.linalg.solve(S.astype(float), (rhs - S @ lam).astype(float)).astype(LD)
        return (z0x - Hinv @ (A_.T @ lam)).astype(float)
    zt = ref_ld(Jl, Jl @ yl); sc = 1 + np.abs(zt).max(); e = lambda z: np.abs(z - zt).max() / sc
    Jk_ld = trinv(Rk.astype(LD), LD)
    print("   kernel R, y -> inversion, z0, projection in ld: %.2e" % e(ref_ld(Jk_ld, Jk_ld @ yk.astype(LD))))
    print("   kernel R -> ld inversion; kernel z0:            %.2e" % e(ref_ld(Jk_ld, z0[i].astype(LD))))
    print("   kernel J; z0 = J y_kernel in ld:                %.2e" % e(ref_ld(J[i].astype(LD), J[i].astype(LD) @ yk.astype(LD) * 1)))
    print("   kernel J vs ld-inverse of kernel R: rel err per column (col max): %s" % " ".join("%.0e" % x for x in (np.abs(J[i] * sg[None, :] - Jk_ld.astype(float)).max(0) / np.abs(Jk_ld.astype(float)).max(0))))
print("==== row-wise vs column-wise triangular inversion of the kernel's R (double), projection in ld")
def inv_rowwise(R, T=np.float64):
    n_ = R.shape[0]; X = np.zeros((n_, n_), T); R = R.astype(T)
    for rr in range(n_):
        for c in range(rr, n_):
            s = T(1.0 if c == rr else 0.0) - (X[rr, :c] @ R[:c, c])
            X[rr, c] = s / R[c, c]
    return X
from itertools import permutations
for i in [int(x) for x in sys.argv[1].split(',')]:
    Rc = qbuf[i, :, :12]; Ac = qbuf[i, :, 16:16 + 18]
    A = np.zeros((30, 13))
    for c, ln in enumerate(lanes): A[:12, c] = Rc[ln]; A[12:, c] = Ac[ln]
    A[:12, 12] = Rc[3]; A[12:, 12] = Ac[3]
    Rl, yl = hhqr(A, LD)
    Rk = np.zeros((12, 12))
    for c, ln in enumerate(lanes): Rk[:, c] = qbuf[i, ln, 48:60]
    yk = qbuf[i, 3, 48:60].copy(); Rk = np.triu(Rk)
    N = drop_lab.normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
    Jl = trinv(Rl, LD); D = (Jl.astype(float).T @ N.T).T
    Aset, adds, drops, ok = gi_lab.gi(D, yl.astype(float), elig)
    def ref_ld(Jx, z0x, NN=N):
        A_ = NN[Aset].astype(LD); Hinv = Jx @ Jx.T; S = A_ @ Hinv @ A_.T; rhs = A_ @ z0x
        lam = np.linalg.solve(S.astype(float), rhs.astype(float)).astype(LD)
        for _ in range(4): lam = lam + np.linalg.solve(S.astype(float), (rhs - S @ lam).astype(float)).astype(LD)
        return (z0x - Hinv @ (A_.T @ lam)).astype(float)
    zt = ref_ld(Jl, Jl @ yl); sc = 1 + np.abs(zt).max(); e = lambda z: np.abs(z - zt).max() / sc
    Jr_ = inv_rowwise(Rk); Jc_ = trinv(Rk, np.float64)
    print("robot %d: row-wise %.2e | column-wise %.2e | kernel J %.2e | row-wise in ld %.2e" % (i, e(ref_ld(Jr_.astype(LD), Jr_.astype(LD) @ yk.astype(LD))), e(ref_ld(Jc_.astype(LD), Jc_.astype(LD) @ yk.astype(LD))), e(ref_ld(J[i].astype(LD), J[i].astype(LD) @ yk.astype(LD))), e(ref_ld(inv_rowwise(Rk, LD), inv_rowwise(Rk, LD) @ yk.astype(LD)))))
    # graded order experiment: QR (double) with a permuted column order, row-wise inversion
    for name, order in (("natural", list(range(12))), ("z0 z1 z2 x0 y0 y2 | rest", [2, 5, 8, 0, 1, 7, 3, 4, 6, 9, 10, 11])):
        Ap = A[:, order + [12]]
        # initial diag rows must stay upper triangular: permuting columns of a diagonal block = permuted diag; re-sort those rows
        top = Ap[:12]; rows = [int(np.argmax(np.abs(top[:, c]))) for c in range(12)]; Ap = np.vstack([top[rows], Ap[12:]])
        Rp, yp = hhqr(Ap, np.float64); Rpl, ypl = hhqr(Ap, LD)
        Jp = inv_rowwise(Rp)
        Np = N[:, order]
        ztp = zt[order]
        zz = ref_ld(Jp.astype(LD), Jp.astype(LD) @ yp.astype(LD), Np)
        print("   order %-28s diag R: %s  -> err %.2e" % (name, " ".join("%.0e" % abs(x) for x in np.diag(Rp)), np.abs(zz - ztp).max() / sc))
