"""Analysis: numerical quality of the active set's DROP variants, emulated in numpy float64 on dumped problems
(reference solution: equality-constrained projection with the final active set in longdouble)."""
import sys, numpy as np
np.set_printoptions(linewidth=200, precision=3)

def load(path):
    d = np.load(path)
    return d["J"], d["z0"], d["mu_n"], d["inv_s"], d["ct"]

def normals(mu_n, inv_s, ct):
    N = np.zeros((16, 12))
    for leg in range(4):
        if not ct[leg]: continue
        for r in range(4):
            h = 4 * leg + r
            N[h, 3 * leg + (r >> 1)] = inv_s if (r & 1) else -inv_s
            N[h, 3 * leg + 2] = mu_n
    return N

def solve(J, z0, N, elig, drop="hw", refine=False, exactW=False, log=None):
    Jr = J.copy(); Dh = N @ J          # Dh[h] = J' n_h
    W = np.zeros((16, 12)); u = np.zeros(16); act = np.zeros(16, bool)
    z = z0.copy(); sh = N @ z
    q = 0; tol = 1e-13 * (1 + abs(z).max())
    p = -1; need_pick = True; up = 0.0; sp = 0.0
    ndrop = 0
    order = []          # active rows by position (givens variant)
    for trip in range(300):
        if need_pick:
            cand = [h for h in range(16) if elig[h] and not act[h]]
            if not cand: break
            p = min(cand, key=lambda h: sh[h]); sp = sh[p]
            if not (sp < -tol): break
            up = 0.0; need_pick = False
        d = Dh[p].copy(); dm = d.copy(); dm[:q] = 0
        d2n = dm @ dm
        zd = Jr @ dm; sd = Dh @ dm
        r = np.where(act, W @ d, 0.0)
        t1 = np.inf; hd = -1
        for h in range(16):
            if act[h] and r[h] > 0 and u[h] / r[h] < t1: t1 = u[h] / r[h]; hd = h
        dependent = not (d2n > 1e-22 * (Dh[p] @ Dh[p])) or q == 12
        t2 = -sp / d2n if d2n > 0 else np.inf
        if dependent and hd < 0: return z, act, -1, ndrop
        full = (not dependent) and (hd < 0 or not (t1 < t2))
        t = t2 if full else t1
        u -= t * r; up += t
        tz = 0.0 if dependent else t
        z = z + tz * zd; sh = sh + tz * sd; sp = sp + tz * d2n
        if full:
            x = dm; tq = q
        else:
            ndrop += 1
            w = W[hd].copy()
            if refine:
                rho = np.where(act & (np.arange(16) != hd), Dh @ w, 0.0)
                w = w - rho @ W
            if drop == "givens":
                ld = order.index(hd); order.pop(ld)
                W[hd] = 0; u[hd] = 0; act[hd] = False
                for j in range(ld, q - 1):
                    row = order[j]                       # the row now at position j has entries at slots j, j+1
                    a, b = Dh[row, j], Dh[row, j + 1]
                    hh = np.hypot(a, b); c, sn = a / hh, b / hh
                    for X in (Jr, Dh, W):
                        xj, xj1 = X[:, j].copy(), X[:, j + 1].copy()
                        X[:, j] = c * xj + sn * xj1; X[:, j + 1] = c * xj1 - sn * xj
                q -= 1
                W[:, q] = 0
                continue
            if drop == "exact":
                others = [h for h in range(16) if act[h] and h != hd]
                M = Dh[others][:, :q].T                      # q x (q-1)
                Q, _ = np.linalg.qr(M, mode="complete")
                G = np.eye(12); G[:q, :q] = Q.T              # new slots = Q' old slots: remaining images -> first q-1 slots
                Jr = Jr @ G.T; Dh = Dh @ G.T
                q -= 1
                Mr = Dh[others][:, :q].T
                Wn = np.linalg.inv(Mr)
                W[:] = 0
                for i, h in enumerate(others): W[h, :q] = Wn[i]
                act[hd] = False; u[hd] = 0
                continue
            q -= 1
            x = w; tq = q
        n2 = x @ x; xq = x[tq]; nrm = np.sqrt(n2)
        alpha = -nrm if xq > 0 else nrm
        beta = 1.0 / (nrm * (nrm + abs(xq)))
        hv = x.copy(); hv[tq] -= alpha
        Jr = Jr - np.outer((Jr @ hv) * beta, hv)
        Dh = Dh - np.outer((Dh @ hv) * beta, hv)
        if full:
            W[:, q] = np.where(act, -r / alpha, 0.0); W[p, q] = 1 / alpha
            u[p] = up; act[p] = True; q += 1; need_pick = True; order.append(p)
        else:
            W = W - np.outer((W @ hv) * beta, hv)
            W[:, tq] = 0; W[hd] = 0; u[hd] = 0; act[hd] = False
            if exactW:
                others = [h for h in range(16) if act[h]]
                Mr = Dh[others][:, :q].T
                Wn = np.linalg.inv(Mr)
                for i, h in enumerate(others): W[h, :q] = Wn[i]
    return z, act, trip, ndrop

def reference(J, z0, N, act):
    """projection with the given active set in longdouble"""
    L = np.longdouble
    Jl = J.astype(L); A = N[act].astype(L)
    Hinv = Jl @ Jl.T
    S = A @ Hinv @ A.T
    lam = np.linalg.solve(S.astype(np.float64), (A @ z0.astype(L)).astype(np.float64)).astype(L)
    # refine the solve in longdouble
    for _ in range(3):
        res = A @ z0.astype(L) - S @ lam
        lam = lam + np.linalg.solve(S.astype(np.float64), res.astype(np.float64)).astype(L)
    return (z0.astype(L) - Hinv @ (A.T @ lam)).astype(np.float64)

if __name__ == "__main__":
    J, z0, mu_n, inv_s, ct = load(sys.argv[1])
    robots = [int(x) for x in sys.argv[2].split(",")]
    if robots == [-1]:
        # statistics over many robots
        import collections
        agg = collections.defaultdict(list)
        for i in range(int(sys.argv[3])):
            N = normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
            for name, kw in (("householder-W", dict(drop="hw")), ("hw+refine", dict(drop="hw", refine=True)), ("givens", dict(drop="givens")), ("exact drop", dict(drop="exact"))):
                z, act, trips, nd = solve(J[i], z0[i], N, elig, **kw)
                if nd == 0: continue
                zr = reference(J[i], z0[i], N, act)
                agg[name].append(abs(z - zr).max() / (1 + abs(zr).max()))
        for k, v in agg.items():
            v = np.array(v); print("%-16s n=%d  median %.1e  p90 %.1e  p99 %.1e  max %.1e" % (k, len(v), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
        sys.exit(0)
    for i in robots:
        N = normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
        out = {}
        for name, kw in (("householder-W", dict(drop="hw")), ("hw+refine", dict(drop="hw", refine=True)), ("hw+exactW", dict(drop="hw", exactW=True)),
                         ("hw+refine+exactW", dict(drop="hw", refine=True, exactW=True)), ("givens", dict(drop="givens")), ("exact drop", dict(drop="exact"))):
            z, act, trips, nd = solve(J[i], z0[i], N, elig, **kw)
            zr = reference(J[i], z0[i], N, act)
            out[name] = (abs(z - zr).max() / (1 + abs(zr).max()), trips, nd, int(act.sum()))
        print(i, {k: ("%.1e" % v[0], v[1], v[2], v[3]) for k, v in out.items()})
