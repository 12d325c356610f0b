"""Analysis (not product code), round 5, review item 2(a): how far could a drop-capable chain of compile-time-length trips carry a wavefront?
The chain (tools/lab/patches/fast_path_drop_trips.patch) runs a trip with the list length q as a compile-time constant, which needs every LIVE
robot of the wavefront at the SAME q.  From the add / drop sequence of every robot (numpy Goldfarb-Idnani on the problems dumped from the host
emulation, tools/lab/gi_dump.py) the four robots of each wavefront are replayed in lock step: a trip is "chain-capable" while all live robots
have agreed on q at every trip so far (the chain cannot be re-entered: after the generic loop took over, the registers hold its layout).

    python tools/lab/r05/chain_agreement.py /tmp/gi_cfg2_id_4096.npz [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gi_lab import load, gi_trace   # noqa: E402

path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
D, y0, ct, iters = load(path)
elig = np.repeat(ct, 4, axis=1)
n = min(n, D.shape[0]) // 4 * 4
seqs = []
for i in range(n):
    tr, A = gi_trace(D[i], y0[i], elig[i])
    seqs.append([+1 if t[0] == "+" else -1 for t in tr])
tot = today = chain = agree_any = 0
first_exit_today, first_exit_chain, lens = [], [], []
for w in range(0, n, 4):
    s4 = seqs[w:w + 4]
    T = max(len(s) for s in s4)
    q = [0, 0, 0, 0]
    ok_today = ok_chain = True
    ft = fc = T
    for t in range(T):
        live = [r for r in range(4) if t < len(s4[r])]
        anydrop = any(s4[r][t] < 0 for r in live)
        same_before = len({q[r] for r in live}) == 1
        if ok_today and (anydrop or not same_before):
            ok_today = False; ft = t
        if ok_chain and not same_before:
            ok_chain = False; fc = t
        tot += 1
        today += ok_today
        chain += ok_chain
        agree_any += same_before
        for r in live:
            q[r] += s4[r][t]
    first_exit_today.append(ft); first_exit_chain.append(fc); lens.append(T)
lens = np.array(lens)
print("%d wavefronts, lock-step trips per wavefront mean %.2f max %d" % (n // 4, lens.mean(), lens.max()))
print("trips on the fast path today (no drop yet, all live robots at the same q):   %.1f %%  (exit after %.2f trips on average)" % (100.0 * today / tot, np.mean(first_exit_today)))
print("trips a drop-capable chain could run (all live robots at the same q so far): %.1f %%  (exit after %.2f trips on average)" % (100.0 * chain / tot, np.mean(first_exit_chain)))
print("trips at which the live robots happen to agree on q at all (re-entry allowed): %.1f %%" % (100.0 * agree_any / tot))
worst = np.argsort(-lens)[:max(1, len(lens) // 100)]
print("the slowest 1 %% of the wavefronts (the launch's tail at N = 4096): %.1f trips, chain-capable %.1f of them, fast today %.1f" % (
    lens[worst].mean(), np.mean([first_exit_chain[i] for i in worst]), np.mean([first_exit_today[i] for i in worst])))
# ---- re-entrant uniform trips: all live robots at the same q AND doing the same kind of step (all add / all drop) in this trip
uni = uni_add = uni_drop = 0
tail_uni = tail_tot = 0
for w in range(0, n, 4):
    s4 = seqs[w:w + 4]
    T = max(len(s) for s in s4)
    q = [0, 0, 0, 0]
    wu = 0
    for t in range(T):
        live = [r for r in range(4) if t < len(s4[r])]
        same_q = len({q[r] for r in live}) == 1
        ops = {s4[r][t] for r in live}
        if same_q and len(ops) == 1:
            uni += 1; wu += 1
            if ops == {1}: uni_add += 1
            else: uni_drop += 1
        for r in live:
            q[r] += s4[r][t]
    if T >= np.percentile(lens, 99):
        tail_uni += wu; tail_tot += T
print("re-entrant uniform trips (same q, same kind of step for every live robot): %.1f %% of the lock-step trips (adds %.1f %%, drops %.1f %%); in the slowest 1 %% of the wavefronts: %.1f %%" % (
    100.0 * uni / tot, 100.0 * uni_add / tot, 100.0 * uni_drop / tot, 100.0 * tail_uni / max(tail_tot, 1)))
