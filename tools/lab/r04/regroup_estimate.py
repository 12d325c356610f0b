"""Round 4 analysis (CPU): what dealing robots to wavefronts by a difficulty proxy could save of the ID stand's lock-step trips.
Input: tools/lab/gi_dump.py 2 2048 id /tmp/gi_cfg2_id.npz.   python3 regroup_estimate.py   (profiles/r04/id_lockstep.md)"""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools/lab')
import drop_lab as dl
d = np.load('/tmp/gi_cfg2_id.npz')
it = d["iters"].astype(int); n = it.size // 16 * 16
it = it[:n]
J, z0, mu_n, inv_s, ct = d["J"][:n], d["z0"][:n], d["mu_n"][:n], d["inv_s"][:n], d["ct"][:n]
viol = np.zeros(n, int); worst = np.zeros(n)
for i in range(n):
    N = dl.normals(mu_n[i], inv_s[i], ct[i]); s = N @ z0[i]
    viol[i] = (s < -1e-9).sum(); worst[i] = -s.min()
def lockstep(order):   # trips per wavefront = max over its four robots
    return it[order].reshape(-1, 4).max(1)
base = lockstep(np.arange(n))
print("robots %d: iterations per robot mean %.2f; lock-step trips per wavefront (consecutive robots): mean %.2f, max %d" % (n, it.mean(), base.mean(), base.max()))
def regroup(key, block=16):
    order = np.arange(n).reshape(-1, block)
    out = []
    for blk in order:
        out.append(blk[np.argsort(key[blk], kind="stable")])
    return np.concatenate(out)
for name, key in (("true iteration count (upper bound, not available in advance)", it), ("rows violated at z0", viol), ("worst violation at z0", worst),
                  ("violated rows, then worst violation", viol + worst / (1 + worst.max()))):
    for block in (16, 64):
        t = lockstep(regroup(key.astype(float), block))
        print("  regrouped inside %2d-robot workgroups by %-62s mean %.2f (%.1f %% fewer lock-step trips)" % (block, name + ":", t.mean(), 100 * (1 - t.mean() / base.mean())))
print("correlation of the iteration count with rows violated at z0: %.2f, with the worst violation: %.2f" % (np.corrcoef(it, viol)[0, 1], np.corrcoef(it, worst)[0, 1]))
