"""Analysis (not product code): active-set strategies for the friction-cone projection of the reduced QP, in numpy.
Problem per robot:  min 1/2 |y - y0|^2  s.t. D_h . y >= 0   (y = J^-1 z, D_h = J' n_h, 16 rows, <= 3 independent per leg)."""
import sys, numpy as np
np.set_printoptions(linewidth=200, precision=4, suppress=True)

def load(path):
    d = np.load(path)
    J, z0, mu_n, inv_s, ct = d["J"], d["z0"], d["mu_n"], d["inv_s"], d["ct"]
    n = J.shape[0]
    N = np.zeros((n, 16, 12))
    for leg in range(4):
        for r in range(4):
            h = 4 * leg + r
            sg = np.where(r & 1, inv_s, -inv_s)
            N[:, h, 3 * leg + (r >> 1)] = sg
            N[:, h, 3 * leg + 2] = mu_n
            N[:, h] *= ct[:, leg, None]
    D = np.einsum("nij,nhi->nhj", J, N)      # D[h] = J' n_h
    y0 = np.linalg.solve(J, z0[..., None])[..., 0]
    return D, y0, ct, d["iters"]

def eqp(D, y0, A):
    """projection of y0 on {D_A y = 0}; returns y, lam (y = y0 + D_A' lam), s = D y"""
    if len(A) == 0:
        return y0.copy(), np.zeros(0), D @ y0
    DA = D[A]
    lam = np.linalg.solve(DA @ DA.T, -(DA @ y0))
    y = y0 + DA.T @ lam
    return y, lam, D @ y

def gi(D, y0, elig, tol=1e-11):
    """Goldfarb-Idnani, most violated row first.  returns (A, adds, drops)"""
    A = []; adds = drops = 0
    y = y0.copy(); lam = np.zeros(0)
    for it in range(200):
        s = D @ y
        cand = [h for h in range(16) if elig[h] and h not in A]
        if not cand: break
        p = min(cand, key=lambda h: s[h])
        if s[p] > -tol * (1 + abs(y0).max()): break
        u = np.append(lam, 0.0)
        while True:
            # step direction for adding p with A active
            if A:
                DA = D[A]
                G = DA @ DA.T
                r = np.linalg.solve(G, DA @ D[p])
                zdir = D[p] - DA.T @ r
            else:
                r = np.zeros(0); zdir = D[p].copy()
            zz = zdir @ zdir
            t2 = -(D[p] @ y) / zz if zz > 1e-18 * (D[p] @ D[p]) else np.inf
            t1 = np.inf; jd = -1
            for j in range(len(A)):
                if r[j] > 0 and u[j] / r[j] < t1: t1 = u[j] / r[j]; jd = j
            t = min(t1, t2)
            if not np.isfinite(t): return A, adds, drops, False
            if np.isfinite(t2): y = y + t * zdir
            u[:-1] -= t * r; u[-1] += t
            if t2 <= t1:
                A.append(p); lam = u.copy(); adds += 1
                break
            drops += 1
            A.pop(jd); u = np.delete(u, jd)
    return A, adds, drops, True

def bpp(D, y0, elig, tol=1e-11, maxr=50, cap=3, order="viol", per_leg=None):
    """block exchange: add violated rows (<= cap per leg in total, most violated first, optionally <= per_leg new per round),
    drop all rows with negative multipliers.  returns rounds, adds, drops, ok"""
    A = []; adds = drops = 0
    sc = 1 + abs(y0).max()
    hist = []
    for rnd in range(maxr):
        y, lam, s = eqp(D, y0, A)
        neg = [A[j] for j in range(len(A)) if lam[j] < -tol * sc]
        viol = [h for h in range(16) if elig[h] and h not in A and s[h] < -tol * sc]
        if not neg and not viol:
            return rnd, adds, drops, True, A
        key = (tuple(sorted(A)))
        for h in neg: A.remove(h); drops += 1
        viol.sort(key=lambda h: s[h])
        newleg = [0] * 4
        for h in viol:
            leg = h // 4
            if sum(1 for a in A if a // 4 == leg) >= cap: continue
            if per_leg is not None and newleg[leg] >= per_leg: continue
            A.append(h); adds += 1; newleg[leg] += 1
        hist.append(key)
        if hist.count(key) > 2:
            return rnd, adds, drops, False, A
    return maxr, adds, drops, False, A

if __name__ == "__main__":
    D, y0, ct, iters = load(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    elig = np.repeat(ct, 4, axis=1)
    res = {"gi": [], "bpp": [], "bpp1": [], "bpp2": []}
    fin = []
    for i in range(n):
        A, a, d, ok = gi(D[i], y0[i], elig[i]); res["gi"].append((a + d, a, d, ok)); fin.append(len(A))
        r, a, d, ok, A2 = bpp(D[i], y0[i], elig[i]); res["bpp"].append((r, a, d, ok))
        if ok and sorted(A2) != sorted(A):
            y1 = eqp(D[i], y0[i], A)[0]; y2 = eqp(D[i], y0[i], A2)[0]
            if abs(y1 - y2).max() > 1e-7 * (1 + abs(y1).max()): print("MISMATCH", i, sorted(A), sorted(A2), abs(y1 - y2).max())
        r, a, d, ok, _ = bpp(D[i], y0[i], elig[i], per_leg=1); res["bpp1"].append((r, a, d, ok))
        r, a, d, ok, _ = bpp(D[i], y0[i], elig[i], per_leg=2); res["bpp2"].append((r, a, d, ok))
    print("final active mean %.2f max %d; kernel iters mean %.2f" % (np.mean(fin), max(fin), iters[:n].mean()))
    for k, v in res.items():
        v = np.array(v, float)
        print("%-5s rounds/iters mean %.2f p99 %.0f max %.0f | adds mean %.2f max %.0f | drops mean %.2f max %.0f | fail %d" % (
            k, v[:, 0].mean(), np.percentile(v[:, 0], 99), v[:, 0].max(), v[:, 1].mean(), v[:, 1].max(), v[:, 2].mean(), v[:, 2].max(), (v[:, 3] == 0).sum()))

def gi_from(D, y0, elig, A0, tol=1e-11, pick="viol"):
    """GI continued from a dual-feasible active set A0 (lam >= 0).  Returns A, adds, drops"""
    A = list(A0); adds = drops = 0
    sc = 1 + abs(y0).max()
    y, lam, s = eqp(D, y0, A)
    for it in range(200):
        s = D @ y
        cand = [h for h in range(16) if elig[h] and h not in A and sum(1 for a in A if a // 4 == h // 4) < 4]
        if not cand: break
        if pick == "viol": p = min(cand, key=lambda h: s[h])
        if s[p] > -tol * sc: break
        u = np.append(lam, 0.0)
        while True:
            if A:
                DA = D[A]
                r = np.linalg.lstsq(DA.T, D[p], rcond=None)[0]
                zdir = D[p] - DA.T @ r
            else:
                r = np.zeros(0); zdir = D[p].copy()
            zz = zdir @ zdir
            t2 = -(D[p] @ y) / zz if zz > 1e-18 * (D[p] @ D[p]) else np.inf
            t1 = np.inf; jd = -1
            for j in range(len(A)):
                if r[j] > 0 and u[j] / r[j] < t1: t1 = u[j] / r[j]; jd = j
            t = min(t1, t2)
            if not np.isfinite(t): return A, adds, drops, False
            if np.isfinite(t2): y = y + t * zdir
            u[:-1] -= t * r; u[-1] += t
            if t2 <= t1:
                A.append(p); lam = u.copy(); adds += 1
                break
            drops += 1
            A.pop(jd); u = np.delete(u, jd)
    return A, adds, drops, True

def warm(D, y0, elig, tol=1e-11, cap=3):
    """block-add the rows violated at y0, purge negative multipliers (all at once, repeated), continue with GI"""
    sc = 1 + abs(y0).max()
    s = D @ y0
    viol = sorted([h for h in range(16) if elig[h] and s[h] < -tol * sc], key=lambda h: s[h])
    A = []
    for h in viol:
        if sum(1 for a in A if a // 4 == h // 4) < cap: A.append(h)
    n0 = len(A); purged = 0; rounds = 0
    while True:
        y, lam, s2 = eqp(D, y0, A)
        neg = [A[j] for j in range(len(A)) if lam[j] < -tol * sc]
        if not neg: break
        rounds += 1
        for h in neg: A.remove(h); purged += 1
    A2, a, d, ok = gi_from(D, y0, elig, A)
    return n0, purged, rounds, a, d, ok, A2

if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "warm":
    D, y0, ct, iters = load(sys.argv[1])
    elig = np.repeat(ct, 4, axis=1)
    rows = []
    for i in range(n):
        A, a, d, ok = gi(D[i], y0[i], elig[i])
        n0, purged, rounds, a2, d2, ok2, A2 = warm(D[i], y0[i], elig[i])
        y1 = eqp(D[i], y0[i], A)[0]; y2 = eqp(D[i], y0[i], A2)[0]
        bad = abs(y1 - y2).max() > 1e-7 * (1 + abs(y1).max())
        rows.append((n0, purged, rounds, a2, d2, ok2, bad, len(A), len(set(A) & set(A2))))
    v = np.array(rows, float)
    print("warm: |A0| mean %.2f  purged mean %.2f (rounds %.2f)  then GI adds %.2f drops %.2f | fails %d mismatches %d | final %.2f" % (
        v[:, 0].mean(), v[:, 1].mean(), v[:, 2].mean(), v[:, 3].mean(), v[:, 4].mean(), (v[:, 5] == 0).sum(), v[:, 6].sum(), v[:, 7].mean()))
    tot = v[:, 0] + v[:, 1] + v[:, 3] + v[:, 4]
    print("total updates mean %.2f p99 %.0f max %.0f" % (tot.mean(), np.percentile(tot, 99), tot.max()))

def gi2(D, y0, elig, tol=1e-11, pick="viol"):
    """GI with selectable pick rule; incremental QR not modelled (dense solves)."""
    A = []; adds = drops = 0
    sc = 1 + abs(y0).max()
    y = y0.copy(); lam = np.zeros(0)
    dn = np.sqrt((D * D).sum(1)) + 1e-300
    for it in range(300):
        s = D @ y
        cand = [h for h in range(16) if elig[h] and h not in A and s[h] < -tol * sc]
        if not cand: break
        if A:
            DA = D[A]
            Q, _ = np.linalg.qr(DA.T)
            Dfree = D - (D @ Q) @ Q.T
        else:
            Dfree = D
        fn = np.sqrt((Dfree * Dfree).sum(1)) + 1e-300
        if pick == "viol": p = min(cand, key=lambda h: s[h])
        elif pick == "norm": p = min(cand, key=lambda h: s[h] / dn[h])
        elif pick == "gain": p = min(cand, key=lambda h: -(s[h] / fn[h]) ** 2)
        elif pick == "dist": p = min(cand, key=lambda h: s[h] / fn[h])
        elif pick == "least": p = min(cand)
        elif pick == "fewleg": p = min(cand, key=lambda h: (sum(1 for a in A if a // 4 == h // 4), s[h]))
        u = np.append(lam, 0.0)
        while True:
            if A:
                DA = D[A]
                r = np.linalg.lstsq(DA.T, D[p], rcond=None)[0]
                zdir = D[p] - DA.T @ r
            else:
                r = np.zeros(0); zdir = D[p].copy()
            zz = zdir @ zdir
            t2 = -(D[p] @ y) / zz if zz > 1e-18 * (D[p] @ D[p]) else np.inf
            t1 = np.inf; jd = -1
            for j in range(len(A)):
                if r[j] > 0 and u[j] / r[j] < t1: t1 = u[j] / r[j]; jd = j
            t = min(t1, t2)
            if not np.isfinite(t): return A, adds, drops, False
            if np.isfinite(t2): y = y + t * zdir
            u[:-1] -= t * r; u[-1] += t
            if t2 <= t1:
                A.append(p); lam = u.copy(); adds += 1
                break
            drops += 1
            A.pop(jd); u = np.delete(u, jd)
    return A, adds, drops, True

if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "pick":
    D, y0, ct, iters = load(sys.argv[1])
    elig = np.repeat(ct, 4, axis=1)
    for pick in ("viol", "norm", "gain", "dist", "least", "fewleg"):
        v = []
        for i in range(n):
            A, a, d, ok = gi2(D[i], y0[i], elig[i], pick=pick)
            v.append((a + d, a, d, ok))
        v = np.array(v, float)
        w4 = v[: (n // 4) * 4, 0].reshape(-1, 4).max(1)
        print("%-7s iters mean %.2f p99 %.0f max %.0f | adds %.2f drops %.2f | fail %d | wave(4) mean %.2f max %.0f" % (
            pick, v[:, 0].mean(), np.percentile(v[:, 0], 99), v[:, 0].max(), v[:, 1].mean(), v[:, 2].mean(), (v[:, 3] == 0).sum(), w4.mean(), w4.max()))

def gi_trace(D, y0, elig, tol=1e-11):
    A = []; tr = []
    sc = 1 + abs(y0).max()
    y = y0.copy(); lam = np.zeros(0)
    for it in range(300):
        s = D @ y
        cand = [h for h in range(16) if elig[h] and h not in A and s[h] < -tol * sc]
        if not cand: break
        p = min(cand, key=lambda h: s[h])
        u = np.append(lam, 0.0)
        while True:
            if A:
                DA = D[A]
                r = np.linalg.lstsq(DA.T, D[p], rcond=None)[0]
                zdir = D[p] - DA.T @ r
            else:
                r = np.zeros(0); zdir = D[p].copy()
            zz = zdir @ zdir
            t2 = -(D[p] @ y) / zz if zz > 1e-18 * (D[p] @ D[p]) else np.inf
            t1 = np.inf; jd = -1
            for j in range(len(A)):
                if r[j] > 0 and u[j] / r[j] < t1: t1 = u[j] / r[j]; jd = j
            t = min(t1, t2)
            if np.isfinite(t2): y = y + t * zdir
            u[:-1] -= t * r; u[-1] += t
            if t2 <= t1:
                A.append(p); lam = u.copy(); tr.append("+%d.%d" % (p // 4, p % 4))
                break
            tr.append("-%d.%d" % (A[jd] // 4, A[jd] % 4))
            A.pop(jd); u = np.delete(u, jd)
    return tr, A

if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "trace":
    D, y0, ct, iters = load(sys.argv[1])
    elig = np.repeat(ct, 4, axis=1)
    J = np.load(sys.argv[1])["J"]
    for i in range(n):
        tr, A = gi_trace(D[i], y0[i], elig[i])
        s0 = D[i] @ y0[i]
        z0 = J[i] @ y0[i]
        print(i, " ".join(tr), "| final", sorted("%d.%d" % (a // 4, a % 4) for a in A))
        print("    z0", z0.reshape(4, 3))
