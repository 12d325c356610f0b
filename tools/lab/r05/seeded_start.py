"""Analysis (not product code), round 5, review item 2(b): a SEEDED start for the active set of the ID 4-contact stand (BASELINE config 2).
Append, in one straight chain without pick / blocking logic, every row violated at the unconstrained minimiser (one per leg x axis), drop
while a multiplier is negative, then ordinary Goldfarb-Idnani from that S-pair.  Numpy on the problems dumped from the host emulation of the
kernel (tools/lab/gi_dump.py).  Go / no-go: mean COST-WEIGHTED trips <= 0.8 x today's and the worst robot <= 22.

    python tools/lab/gi_dump.py 2 4096 id /tmp/gi_cfg2_id_4096.npz
    python tools/lab/r05/seeded_start.py /tmp/gi_cfg2_id_4096.npz [n]
Cost model (stamps of round 3 / 4, cycles on the lone wavefront): ordinary append trip 2200-2400, drop trip 2700, a chained append without
pick, blocking ratio and step logic ~900 -> weights add 1.0, drop 1.15, seeded add 0.41."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gi_lab import load, eqp, gi, gi_from   # noqa: E402

W_ADD, W_DROP, W_SEED = 1.0, 1.15, 0.41


def seeds(D, y0, elig, tol, rule):
    sc = 1 + abs(y0).max()
    s = D @ y0
    A = []
    for leg in range(4):
        for ax in range(2):                      # rows 2*ax, 2*ax+1 of a leg: the two sides of one pyramid axis
            pair = [4 * leg + 2 * ax, 4 * leg + 2 * ax + 1]
            v = [h for h in pair if elig[h] and s[h] < -tol * sc]
            if not v:
                continue
            if rule == "axis":                   # at most one per leg x axis: the more violated side
                A.append(min(v, key=lambda h: s[h]))
            elif rule == "all":
                A += v
    if rule == "leg":                            # at most one per leg: its most violated row
        A = []
        for leg in range(4):
            v = [h for h in range(4 * leg, 4 * leg + 4) if elig[h] and s[h] < -tol * sc]
            if v:
                A.append(min(v, key=lambda h: s[h]))
    return sorted(A, key=lambda h: s[h])


def purge_one_at_a_time(D, y0, A, tol):
    sc = 1 + abs(y0).max()
    A = list(A); n = 0
    while A:
        y, lam, _ = eqp(D, y0, A)
        j = int(np.argmin(lam))
        if lam[j] >= -tol * sc:
            break
        A.pop(j); n += 1
    return A, n


def run(path, n, rule):
    D, y0, ct, iters = load(path)
    elig = np.repeat(ct, 4, axis=1)
    n = min(n, D.shape[0])
    rows = []
    for i in range(n):
        A, a, d, ok = gi(D[i], y0[i], elig[i])
        A0 = seeds(D[i], y0[i], elig[i], 1e-11, rule)
        # rank-deficient seed sets (four rows of one leg) are cut to independent ones
        keep = []
        for h in A0:
            if np.linalg.matrix_rank(D[i][keep + [h]]) == len(keep) + 1:
                keep.append(h)
        A1, purged = purge_one_at_a_time(D[i], y0[i], keep, 1e-11)
        A2, a2, d2, ok2 = gi_from(D[i], y0[i], elig[i], A1)
        y1 = eqp(D[i], y0[i], A)[0]; y2 = eqp(D[i], y0[i], A2)[0]
        bad = abs(y1 - y2).max() > 1e-7 * (1 + abs(y1).max())
        rows.append((a, d, len(keep), purged, a2, d2, ok2, bad, len(set(A1) & set(A)), len(A)))
    v = np.array(rows, float)
    c_now = W_ADD * v[:, 0] + W_DROP * v[:, 1]
    c_seed = W_SEED * v[:, 2] + W_DROP * v[:, 3] + W_ADD * v[:, 4] + W_DROP * v[:, 5]
    t_now = v[:, 0] + v[:, 1]; t_seed = v[:, 2] + v[:, 3] + v[:, 4] + v[:, 5]
    m = (n // 4) * 4
    wave = lambda c: c[:m].reshape(-1, 4).max(1)
    print("rule %-5s n %d | today: adds %.2f drops %.2f = %.2f trips (max %d) cost %.2f | seeded: %.2f chained adds, %.2f purges, then %.2f adds "
          "%.2f drops = %.2f updates (max %d) cost %.2f | ratio %.3f | seeds kept in the final set %.2f of %.2f | fails %d mismatches %d" % (
              rule, n, v[:, 0].mean(), v[:, 1].mean(), t_now.mean(), t_now.max(), c_now.mean(), v[:, 2].mean(), v[:, 3].mean(), v[:, 4].mean(),
              v[:, 5].mean(), t_seed.mean(), t_seed.max(), c_seed.mean(), c_seed.mean() / c_now.mean(), v[:, 8].mean(), v[:, 9].mean(),
              (v[:, 6] == 0).sum(), v[:, 7].sum()))
    print("      lock-step over 4 robots (cost of a wavefront = its slowest robot): today mean %.2f max %.2f | seeded mean %.2f max %.2f | ratio %.3f, worst %.3f" % (
        wave(c_now).mean(), wave(c_now).max(), wave(c_seed).mean(), wave(c_seed).max(), wave(c_seed).mean() / wave(c_now).mean(),
        wave(c_seed).max() / wave(c_now).max()))


if __name__ == "__main__":
    path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    for rule in ("axis", "all", "leg"):
        run(path, n, rule)
