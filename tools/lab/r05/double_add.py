"""Analysis (not product code), round 5, review item 5: a SPECULATIVE DOUBLE ADD on the active set's fast path (headline MPTC trot, config 3).
The launch at N = 4096 ends with the ~31 robots that need 5 - 6 adds, 0.73 us per fast trip.  Idea: fetch the images of the TWO best rows in
one crossbar round trip and run both reflections back to back; afterwards verify that the sequential algorithm would have done exactly that
(bit-identity): after the first full step the second row must be THE next pick (greatest dual gain among the violated rows), sit on another leg,
and both steps must be full steps (no blocking multiplier).  Otherwise the pair is abandoned and the ordinary single trip runs.

Numpy on the problems dumped from the host emulation (tools/lab/gi_dump.py 3 4096 mptc): the fraction of consecutive add pairs that verify.
Go / no-go: build only if >= 60 % verify AND the tail wavefronts lose >= 1 us.

    python tools/lab/r05/double_add.py /tmp/gi_cfg3_mptc_4096.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gi_lab import load   # noqa: E402


ANY_LEG = os.environ.get("DOUBLE_ADD_ANY_LEG") == "1"


def gains(D, y, A, elig, tol, sc):
    """dual gain s^2 / |D_h restricted to the free space|^2 of every violated inactive row (the kernel's GAIN pick)"""
    s = D @ y
    if A:
        Q, _ = np.linalg.qr(D[A].T)
        Df = D - (D @ Q) @ Q.T
    else:
        Df = D
    fn2 = (Df * Df).sum(1)
    g = {}
    for h in range(16):
        if elig[h] and h not in A and s[h] < -tol * sc:
            g[h] = (s[h] ** 2) / fn2[h] if fn2[h] > 1e-22 * (D[h] @ D[h]) else 1e290
    return g, s, Df


def run(D, y0, elig, tol=1e-13):
    """sequential GI with the gain pick; returns the list of trips [(kind, row, full)] and, for every add, the runner-up at pick time"""
    sc = 1 + abs(y0).max()
    A = []; y = y0.copy(); lam = np.zeros(0)
    trips = []
    for it in range(100):
        g, s, Df = gains(D, y, A, elig, tol, sc)
        if not g:
            break
        order = sorted(g, key=lambda h: -g[h])
        p = order[0]
        second = next((h for h in order[1:] if ANY_LEG or h // 4 != p // 4), None)     # best row on ANOTHER leg (ANY_LEG: the runner-up, wherever it sits)
        u = np.append(lam, 0.0)
        first = True
        while True:
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
            t = min(t1, t2)
            if not np.isfinite(t):
                return trips, False
            if np.isfinite(t2):
                y = y + t * zdir
            u[:-1] -= t * r; u[-1] += t
            if t2 <= t1:
                A.append(p); lam = u.copy()
                trips.append(("add", p, first, second))
                break
            trips.append(("drop", A[jd], False, None))
            A.pop(jd); u = np.delete(u, jd); first = False
    return trips, True


if __name__ == "__main__":
    D, y0, ct, iters = load(sys.argv[1])
    n = min(int(sys.argv[2]) if len(sys.argv) > 2 else 4096, D.shape[0])
    elig = np.repeat(ct, 4, axis=1)
    pairs = ok = 0
    why = {"no runner-up on another leg": 0, "runner-up is not the next pick": 0, "a step of the pair is partial": 0}
    per_robot = []
    for i in range(n):
        tr, good = run(D[i], y0[i], elig[i])
        adds = len([t for t in tr if t[0] == "add"])
        # walk the sequence as the speculative kernel would: at an add with a successor, try the pair
        k = 0; trips_spec = 0
        while k < len(tr):
            if k + 1 < len(tr) and tr[k][0] == "add":
                pairs += 1
                a, b = tr[k], tr[k + 1]
                if a[3] is None:
                    why["no runner-up on another leg"] += 1
                elif not (b[0] == "add" and b[1] == a[3]):
                    why["runner-up is not the next pick"] += 1
                elif not (a[2] and b[2]):
                    why["a step of the pair is partial"] += 1
                else:
                    ok += 1; trips_spec += 1; k += 2; continue
            trips_spec += 1; k += 1
        per_robot.append((len(tr), trips_spec))
    pr = np.array(per_robot)
    print("%d robots, trips per robot mean %.2f max %d (kernel: mean %.2f max %d)" % (n, pr[:, 0].mean(), pr[:, 0].max(), iters[:n].mean(), iters[:n].max()))
    print("consecutive (add, next trip) pairs tried: %d, verified: %d = %.1f %%" % (pairs, ok, 100.0 * ok / max(pairs, 1)))
    for k, v in why.items():
        print("   failed because %-34s %5d (%.1f %%)" % (k + ":", v, 100.0 * v / max(pairs, 1)))
    tail = pr[pr[:, 0] >= 5]
    print("robots with >= 5 trips (the launch's tail): %d; their trips %.2f -> %.2f crossbar round trips with verified pairs" % (
        len(tail), tail[:, 0].mean(), tail[:, 1].mean()))
    m = n // 4 * 4
    w0 = pr[:m, 0].reshape(-1, 4).max(1); w1 = pr[:m, 1].reshape(-1, 4).max(1)
    print("lock step over 4 robots: trips per wavefront mean %.2f -> %.2f, max %d -> %d" % (w0.mean(), w1.mean(), w0.max(), w1.max()))
