"""Analysis (not product code), round 5: can the active set AVOID the few genuine drops of a trot batch?  Goldfarb-Idnani may add ANY violated row.  Rule tried:
when the picked row's full step is blocked (an active multiplier would go negative: a partial step and a drop), take the next-best violated row whose full step is
NOT blocked instead; only when every violated row is blocked, do the partial step.  Numpy on the dumped problems of config 3 (greatest-dual-gain pick).
    python tools/lab/r05/avoid_blocked.py /tmp/gi_cfg3_mptc_4096.npz [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gi_lab import load, eqp   # noqa: E402
from double_add import gains   # noqa: E402


def step_info(D, y, A, lam, p):
    u = np.append(lam, 0.0)
    if A:
        r = np.linalg.lstsq(D[A].T, D[p], rcond=None)[0]
        zdir = D[p] - D[A].T @ r
    else:
        r = np.zeros(0); zdir = D[p].copy()
    zz = zdir @ zdir
    t2 = -(D[p] @ y) / zz if zz > 1e-18 * (D[p] @ D[p]) else np.inf
    t1 = np.inf; jd = -1
    for j in range(len(A)):
        if r[j] > 0 and u[j] / r[j] < t1:
            t1 = u[j] / r[j]; jd = j
    return u, r, zdir, t1, t2, jd


def run(D, y0, elig, avoid, tol=1e-13):
    sc = 1 + abs(y0).max()
    A = []; y = y0.copy(); lam = np.zeros(0)
    adds = drops = refetch = 0
    for it in range(200):
        g, s, _ = gains(D, y, A, elig, tol, sc)
        g = {h: v for h, v in g.items() if sum(1 for a in A if a // 4 == h // 4) < 3}      # apex rule
        if not g:
            break
        order = sorted(g, key=lambda h: (-round(g[h], 9 - int(np.floor(np.log10(abs(g[h]) + 1e-300)))), h))
        p = order[0]
        if avoid:
            for cand in order:
                u, r, zdir, t1, t2, jd = step_info(D, y, A, lam, cand)
                if t2 <= t1:
                    p = cand
                    break
                refetch += 1
        u, r, zdir, t1, t2, jd = step_info(D, y, A, lam, p)
        while True:
            t = min(t1, t2)
            if not np.isfinite(t):
                return A, adds, drops, refetch, False
            if np.isfinite(t2):
                y = y + t * zdir
            u[:-1] -= t * r; u[-1] += t
            if t2 <= t1:
                A.append(p); lam = u.copy(); adds += 1
                break
            drops += 1
            A.pop(jd); u = np.delete(u, jd)
            lam_ = u[:-1]
            uu, r, zdir, t1, t2, jd = step_info(D, y, A, lam_, p)
            u = np.append(lam_, u[-1])
    return A, adds, drops, refetch, True


if __name__ == "__main__":
    D, y0, ct, iters = load(sys.argv[1])
    n = min(int(sys.argv[2]) if len(sys.argv) > 2 else 4096, D.shape[0])
    elig = np.repeat(ct, 4, axis=1)
    ref = None
    for avoid in (False, True):
        v = []; sols = []
        for i in range(n):
            A, a, d, rf, ok = run(D[i], y0[i], elig[i], avoid)
            v.append((a + d, a, d, rf, ok)); sols.append(eqp(D[i], y0[i], A)[0])
        v = np.array(v, float)
        m = n // 4 * 4
        w4 = v[:m, 0].reshape(-1, 4).max(1)
        mism = 0 if ref is None else sum(np.abs(a - b).max() > 1e-7 * (1 + np.abs(a).max()) for a, b in zip(ref, sols))
        ref = ref or sols
        print("%-14s trips mean %.3f max %d | adds %.3f drops %d (robots %d) | extra fetches %d | fail %d | lock step (4) mean %.2f max %d | differing solutions %d" % (
            "avoid blocked" if avoid else "today", v[:, 0].mean(), v[:, 0].max(), v[:, 1].mean(), v[:, 2].sum(), (v[:, 2] > 0).sum(), v[:, 3].sum(), (v[:, 4] == 0).sum(),
            w4.mean(), w4.max(), mism))
