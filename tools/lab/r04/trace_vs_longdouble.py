"""Round 4 analysis: the emulated active set in float64 and, following the same decisions, in long double: after which trip does which
tracked quantity (z, constraint values, multipliers, step length, W) part from the truth?  On saturated stands: at the SECOND of two
consecutive drops (profiles/r04/accuracy.md).   python3 trace_vs_longdouble.py id 88,286   (needs /tmp/gi_cfg2_id.npz from ../gi_dump.py)"""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools/lab')
import drop_lab as dl
kind = sys.argv[1]; robots = [int(x) for x in sys.argv[2].split(',')]
J, z0, mu_n, inv_s, ct = dl.load('/tmp/gi_cfg2_%s.npz' % kind)
def solve_gen(J, z0, N, elig, T, script=None):
    """generator version; T = dtype; script: list of (p, full, hd) decisions to follow (from another run)"""
    J = J.astype(T); z0 = z0.astype(T); N = N.astype(T)
    Jr = J.copy(); Dh = N @ J
    W = np.zeros((16, 12), T); u = np.zeros(16, T); act = np.zeros(16, bool)
    z = z0.copy(); sh = N @ z
    q = 0; tol = T(1e-13) * (1 + abs(z).max())
    p = -1; need_pick = True; up = T(0); sp = T(0)
    hist = []
    for trip in range(300):
        if need_pick:
            cand = [h for h in range(16) if elig[h] and not act[h]]
            if not cand: break
            p = min(cand, key=lambda h: sh[h]); 
            if script is not None:
                if trip >= len(script): break
                p = script[trip][0]
            sp = sh[p]
            if script is None and not (sp < -tol): break
            up = T(0); need_pick = False
        d = Dh[p].copy(); dm = d.copy(); dm[:q] = 0
        d2n = dm @ dm
        zd = Jr @ dm; sd = Dh @ dm
        r = np.where(act, W @ d, T(0))
        t1 = np.inf; hd = -1
        for h in range(16):
            if act[h] and r[h] > 0 and u[h] / r[h] < t1: t1 = u[h] / r[h]; hd = h
        dnp = Dh[p] @ Dh[p]
        t2 = -sp / d2n if d2n > 0 else np.inf
        full = (hd < 0 or not (t1 < t2))
        if script is not None:
            full = script[trip][1]; 
            if not full: hd = script[trip][2]; t1 = u[hd] / r[hd]
        t = t2 if full else t1
        u = u - t * r; up = up + t
        z = z + t * zd; sh = sh + t * sd; sp = sp + t * d2n
        if full:
            x = dm; tq = q
        else:
            w = W[hd].copy(); q -= 1; x = w; tq = q
        n2 = x @ x; xq = x[tq]; nrm = np.sqrt(n2)
        alpha = -nrm if xq > 0 else nrm
        beta = 1 / (nrm * (nrm + abs(xq)))
        hv = x.copy(); hv[tq] -= alpha
        Jr = Jr - np.outer((Jr @ hv) * beta, hv)
        Dh = Dh - np.outer((Dh @ hv) * beta, hv)
        if full:
            W[:, q] = np.where(act, -r / alpha, T(0)); W[p, q] = 1 / alpha
            u[p] = up; act[p] = True; q += 1; need_pick = True
        else:
            W = W - np.outer((W @ hv) * beta, hv)
            W[:, tq] = 0; W[hd] = 0; u[hd] = 0; act[hd] = False
        hist.append(dict(trip=trip, p=p, full=full, hd=hd, q=q, t=t, t1=t1, t2=t2, z=z.copy(), sh=sh.copy(), u=u.copy(), sp=sp, d2n=d2n, r=r.copy(),
                         Jr=Jr.copy(), Dh=Dh.copy(), W=W.copy(), act=act.copy()))
    return hist
L = np.longdouble
for i in robots:
    N = dl.normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
    hd_ = solve_gen(J[i], z0[i], N, elig, np.float64)
    script = [(h['p'], h['full'], h['hd']) for h in hd_]
    hl_ = solve_gen(J[i], z0[i], N, elig, L, script)
    print("robot", i, "trips", len(hd_), len(hl_))
    for a, b in zip(hd_, hl_):
        rel = lambda x, y: float(abs(x - y).max() / (abs(y).max() + 1e-300))
        # orthogonality defect of active images in free slots, double run
        act = a['act']; q = a['q']
        defect = abs(a['Dh'][act][:, q:]).max() / abs(a['Dh'][act]).max() if act.any() else 0
        trueD = (N @ a['Jr'].astype(L))   # true images of the rotated J
        dD = float(abs(trueD - a['Dh']).max() / abs(trueD).max())
        print("  trip %2d %-4s p=%2d hd=%2d q=%2d t %.3e | dz %.1e  dsh %.1e  du %.1e dt %.1e  dW %.1e | free-slot defect %.1e  Dh vs J'n %.1e" % (
            a['trip'], 'add' if a['full'] else 'drop', a['p'], a['hd'], q, a['t'], rel(a['z'], b['z']), rel(a['sh'], b['sh']), rel(a['u'], b['u']),
            abs(float(a['t'] - b['t'])) / abs(float(b['t'])), rel(a['W'], b['W']), defect, dD))
