"""Round 4 analysis (CPU, numpy): the active set of the 16-lane kernel emulated on dumped problems (tools/lab/gi_dump.py 2 2048 <kind> /tmp/gi_cfg2_<kind>.npz),
with the rotated right-hand side carried along, and the candidate remedies for the loss on saturated stands:
  fresh_z  z = J[:, q:] y[q:] evaluated at the end;  term1  + feasibility correction;  term2  + the first-order correction for what the
  active images left in the free slots (the one adopted: csrc/wbc_hex.hpp, hex_gi);  both.   python3 refine_lab.py id 1024   (profiles/r04/accuracy.md)"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import drop_lab as dl
L = np.longdouble
def solve(J, z0, y0, N, elig):
    Jr = J.copy(); Dh = N @ J; y = y0.copy()
    W = np.zeros((16, 12)); u = np.zeros(16); act = np.zeros(16, bool)
    z = z0.copy(); sh = N @ z
    q = 0; tol = 1e-13 * (1 + abs(z).max())
    p = -1; need_pick = True; up = 0.0; sp = 0.0; nd = 0
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
        if dependent and hd < 0: return None
        full = (not dependent) and (hd < 0 or not (t1 < t2))
        t = t2 if full else t1
        u -= t * r; up += t
        tz = 0.0 if dependent else t
        z = z + tz * zd; sh = sh + tz * sd; sp = sp + tz * d2n
        if full: x = dm; tq = q
        else:
            w = W[hd].copy(); q -= 1; x = w; tq = q; nd += 1
        n2 = x @ x; xq = x[tq]; nrm = np.sqrt(n2)
        alpha = -nrm if xq > 0 else nrm
        beta = 1.0 / (nrm * (nrm + abs(xq)))
        hv = x.copy(); hv[tq] -= alpha
        Jr = Jr - np.outer((Jr @ hv) * beta, hv)
        Dh = Dh - np.outer((Dh @ hv) * beta, hv)
        y = y - hv * ((hv @ y) * beta)
        if full:
            W[:, q] = np.where(act, -r / alpha, 0.0); W[p, q] = 1 / alpha
            u[p] = up; act[p] = True; q += 1; need_pick = True
        else:
            W = W - np.outer((W @ hv) * beta, hv)
            W[:, tq] = 0; W[hd] = 0; u[hd] = 0; act[hd] = False
    return dict(z=z, act=act, Jr=Jr, Dh=Dh, W=W, y=y, q=q, nd=nd, u=u)
def refine(s, N, terms=3, from_z=None):
    q = s['q']; act = s['act']; Jr, Dh, W, y = s['Jr'], s['Dh'], s['W'], s['y']
    zf = Jr[:, q:] @ y[q:] if from_z is None else from_z
    e = np.where(act, N @ zf, 0.0)                 # fresh constraint values of the active rows
    nu = np.where(act, W @ y, 0.0)                 # W slots >= q are zero
    c = np.zeros(12)
    if terms & 1: c[:q] += (W.T @ e)[:q]
    if terms & 2: c[q:] += (Dh.T @ nu)[q:]
    return zf - Jr @ c
if __name__ == '__main__':
    kind = sys.argv[1]; n = int(sys.argv[2])
    J, z0, mu_n, inv_s, ct = dl.load('/tmp/gi_cfg2_%s.npz' % kind)
    res = {k: [] for k in ("base", "fresh_z", "term1", "term2", "both", "both_on_z", "both x2")}
    for i in range(n):
        N = dl.normals(mu_n[i], inv_s[i], ct[i]); elig = np.repeat(ct[i], 4)
        y0 = np.linalg.solve(J[i], z0[i])
        for _ in range(3):
            r = (z0[i].astype(L) - J[i].astype(L) @ y0.astype(L)).astype(float); y0 = y0 + np.linalg.solve(J[i], r)
        s = solve(J[i], z0[i], y0, N, elig)
        if s is None or s['act'].sum() == 0: continue
        zr = dl.reference(J[i], z0[i], N, s['act'])
        sc = 1 + abs(zr).max(); E = lambda z: abs(z - zr).max() / sc
        res["base"].append(E(s['z'])); res["fresh_z"].append(E(s['Jr'][:, s['q']:] @ s['y'][s['q']:]))
        res["term1"].append(E(refine(s, N, 1))); res["term2"].append(E(refine(s, N, 2))); res["both"].append(E(refine(s, N, 3)))
        res["both_on_z"].append(E(refine(s, N, 3, from_z=s['z'])))
    for k, v in res.items():
        if not v: continue
        v = np.array(v); print("%-10s n=%d median %.1e p90 %.1e p99 %.1e max %.1e  >1e-8: %d >1e-7: %d >1e-6: %d" % (k, len(v), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max(), (v > 1e-8).sum(), (v > 1e-7).sum(), (v > 1e-6).sum()))
