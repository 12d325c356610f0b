"""Analysis (not product code), round 5: the APEX rule.  The four friction rows of a foot satisfy n_0 + n_1 = n_2 + n_3 (= 2 mu_n e_z): with three of
them active the fourth is linearly dependent and its value is IDENTICALLY zero -- in floating point a rounding remainder of either sign.  When
the remainder is negative beyond the tolerance the active set picks the row, finds it dependent, drops one of the three and adds the fourth: two trips
(on the device: two GENERIC trips and the fast path lost for the whole wavefront) that swap one description of the apex for another.
Rule: a row whose three leg-mates are active is not a candidate.  Numpy Goldfarb-Idnani (most violated row; tolerance as the kernel's, in z units)
on the problems dumped from the host emulation, with and without the rule.

    python tools/lab/r05/apex_rule.py /tmp/gi_cfg2_id_4096.npz [n] [noise]
`noise`: relative perturbation applied to the constraint values before every pick (models the device's rounding, which differs from numpy's)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gi_lab import load, eqp   # noqa: E402


def gi(D, y0, elig, tol_abs, apex, rng=None, noise=0.0):
    A = []; adds = drops = 0; apex_hits = 0
    y = y0.copy(); lam = np.zeros(0)
    for it in range(300):
        s = D @ y
        if noise:
            s = s + noise * np.abs(D).dot(np.abs(y)) * rng.standard_normal(16)     # rounding-level noise of the evaluated constraint values
        cand = []
        for h in range(16):
            if not elig[h] or h in A:
                continue
            mates = sum(1 for a in A if a // 4 == h // 4)
            if mates == 3:
                if s[h] < -tol_abs:
                    apex_hits += 1
                if apex:
                    continue
            cand.append(h)
        if not cand:
            break
        p = min(cand, key=lambda h: s[h])
        if s[p] > -tol_abs:
            break
        u = np.append(lam, 0.0)
        while True:
            if A:
                r = np.linalg.lstsq(D[A].T, D[p], rcond=None)[0]
                zdir = D[p] - D[A].T @ r
            else:
                r = np.zeros(0); zdir = D[p].copy()
            zz = zdir @ zdir
            t2 = -(D[p] @ y) / zz if zz > 1e-22 * (D[p] @ D[p]) else np.inf
            t1 = np.inf; jd = -1
            for j in range(len(A)):
                if r[j] > 0 and u[j] / r[j] < t1:
                    t1 = u[j] / r[j]; jd = j
            t = min(t1, t2)
            if not np.isfinite(t):
                return A, adds, drops, apex_hits, False
            if np.isfinite(t2):
                y = y + t * zdir
            u[:-1] -= t * r; u[-1] += t
            if t2 <= t1:
                A.append(p); lam = u.copy(); adds += 1
                break
            drops += 1
            A.pop(jd); u = np.delete(u, jd)
    return A, adds, drops, apex_hits, True


if __name__ == "__main__":
    path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    D, y0, ct, iters = load(path)
    z0 = np.load(path)["z0"]
    elig = np.repeat(ct, 4, axis=1)
    n = min(n, D.shape[0])
    rng = np.random.default_rng(0)
    res = {False: [], True: []}
    mism = 0
    for i in range(n):
        tol = 1e-13 * (1.0 + np.abs(z0[i]).max())
        out = {}
        for apex in (False, True):
            A, a, d, hits, ok = gi(D[i], y0[i], elig[i], tol, apex, rng, noise)
            res[apex].append((a + d, a, d, hits, ok, len(A)))
            out[apex] = eqp(D[i], y0[i], A)[0]
        if np.abs(out[False] - out[True]).max() > 1e-7 * (1 + np.abs(out[False]).max()):
            mism += 1
    for apex in (False, True):
        v = np.array(res[apex], float)
        m = (n // 4) * 4
        w4 = v[:m, 0].reshape(-1, 4).max(1)
        print("%-12s trips mean %.2f p99 %.0f max %.0f | adds %.2f drops %.2f | noise-violated apex rows seen %d (robots %d) | fail %d | lock step (4): mean %.2f max %.0f | final rows %.2f" % (
            "apex rule" if apex else "today", v[:, 0].mean(), np.percentile(v[:, 0], 99), v[:, 0].max(), v[:, 1].mean(), v[:, 2].mean(), v[:, 3].sum(),
            (v[:, 3] > 0).sum(), (v[:, 4] == 0).sum(), w4.mean(), w4.max(), v[:, 5].mean()))
    print("solutions that differ between the two (1e-7): %d of %d; kernel (host emulation) iterations mean %.2f max %d" % (mism, n, iters[:n].mean(), iters[:n].max()))
