"""Analysis (not product code), round 5: the structural TIE on the way into an apex.  With one x-side row (a) and one y-side row (b) of a foot active, the
two remaining rows of that foot have EXACTLY the same value (2 mu_n f_z) and the same free part (both are 2 mu_n e_z modulo the active images): most
violated row and greatest dual gain both tie, rounding decides.  Either choice leads to the apex, but the multipliers of the final representation differ:
choosing the row whose pair-mate is a leaves b's multiplier reduced by the new row's -- if that goes negative the step is blocked: a drop and a re-add.
Tie-break rules on the dumped problems (numpy Goldfarb-Idnani, most violated row, apex rule on):
  noise   rounding decides (today)                 mate-small   complete the pair whose active mate has the SMALLER multiplier
  mate-big  ... the LARGER multiplier              first        lowest row index
  exact     the candidate whose step is not blocked, if either (what a swap after the blocked trip would achieve)
    python tools/lab/r05/apex_tie.py /tmp/gi_cfg2_id_4096.npz [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gi_lab import load, eqp   # noqa: E402


def gi(D, y0, elig, tol_abs, rule, rng):
    A = []; adds = drops = ties = 0
    y = y0.copy(); lam = np.zeros(0)
    for it in range(300):
        s = D @ y
        cand = []
        for h in range(16):
            if not elig[h] or h in A:
                continue
            if sum(1 for a in A if a // 4 == h // 4) == 3:
                continue                                         # apex rule
            cand.append(h)
        if not cand:
            break
        key = {h: s[h] for h in cand}
        # structural ties: leg with exactly one x-side and one y-side row active
        for leg in range(4):
            act = [a for a in A if a // 4 == leg]
            if len(act) == 2 and (act[0] % 4) // 2 != (act[1] % 4) // 2:
                rest = [h for h in range(4 * leg, 4 * leg + 4) if h not in act]
                if all(h in key for h in rest):
                    ties += 1
                    m = 0.5 * (key[rest[0]] + key[rest[1]])
                    u = {a: lam[A.index(a)] for a in act}
                    mate = {h: next(a for a in act if (a % 4) // 2 == (h % 4) // 2) for h in rest}    # the active row on the same axis
                    if rule == "noise":
                        pref = rest[int(rng.integers(2))]
                    elif rule == "mate-small":
                        pref = min(rest, key=lambda h: u[mate[h]])
                    elif rule == "mate-big":
                        pref = max(rest, key=lambda h: u[mate[h]])
                    elif rule == "exact":
                        # the candidate whose step is FULL, if either is (same z and t2 for both, r differs by e_b - e_a: apex_rule.md section 6)
                        def full(h):
                            r = np.linalg.lstsq(D[A].T, D[h], rcond=None)[0]
                            zd = D[h] - D[A].T @ r; zz = zd @ zd
                            if zz <= 1e-22 * (D[h] @ D[h]):
                                return False
                            t2 = -(D[h] @ y) / zz
                            return all(not (r[j] > 0 and lam[j] / r[j] < t2) for j in range(len(A)))
                        ok = [h for h in rest if full(h)]
                        pref = ok[0] if ok else min(rest)
                    else:
                        pref = min(rest)
                    for h in rest:
                        key[h] = m * (1.0 + 1e-9) if h == pref else m * (1.0 - 1e-9)      # values are negative when violated: the preferred one is "more violated"
        p = min(cand, key=lambda h: key[h])
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
                return A, adds, drops, ties, False
            if np.isfinite(t2):
                y = y + t * zdir
            u[:-1] -= t * r; u[-1] += t
            if t2 <= t1:
                A.append(p); lam = u.copy(); adds += 1
                break
            drops += 1
            A.pop(jd); u = np.delete(u, jd)
    return A, adds, drops, ties, True


if __name__ == "__main__":
    path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    D, y0, ct, iters = load(path)
    z0 = np.load(path)["z0"]
    elig = np.repeat(ct, 4, axis=1)
    n = min(n, D.shape[0])
    ref = None
    for rule in ("noise", "mate-small", "mate-big", "first", "exact"):
        rng = np.random.default_rng(0)
        v = []; sols = []
        for i in range(n):
            tol = 1e-13 * (1.0 + np.abs(z0[i]).max())
            A, a, d, ties, ok = gi(D[i], y0[i], elig[i], tol, rule, rng)
            v.append((a + d, a, d, ties, ok)); sols.append(eqp(D[i], y0[i], A)[0])
        v = np.array(v, float)
        m = (n // 4) * 4
        w4 = v[:m, 0].reshape(-1, 4).max(1)
        mism = 0 if ref is None else sum(np.abs(a - b).max() > 1e-7 * (1 + np.abs(a).max()) for a, b in zip(ref, sols))
        ref = ref or sols
        print("%-10s trips mean %.2f p99 %.0f max %.0f | adds %.2f drops %.2f | ties met %d | fail %d | lock step (4): mean %.2f max %.0f | differing solutions %d" % (
            rule, v[:, 0].mean(), np.percentile(v[:, 0], 99), v[:, 0].max(), v[:, 1].mean(), v[:, 2].mean(), v[:, 3].sum(), (v[:, 4] == 0).sum(), w4.mean(), w4.max(), mism))
