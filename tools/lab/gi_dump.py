"""Analysis helper (not product code): dump the active set's inputs of a config through the host emulation of the
16-lane kernel -> npz with J[n,12,12] (J J' = H^-1), z0[n,12], mu_n[n], inv_s[n], ct[n,4], iters[n], tau[12,n]"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_tick as ht
from quadruped_drake_amd import workloads
from oracle import oracle_py as orc


def dump(cfg, n, kind=None, seed=None):
    b = workloads.make_batch(cfg, n=n, seed=seed)
    kind = kind or b["kind"]
    t = orc.load_model_json(b["model"])
    buf = np.zeros((n, 16, 16))
    L = ht.lib()
    L.host_gi_dump.argtypes = [C.c_void_p]
    L.host_gi_dump(buf.ctypes.data_as(C.c_void_p))
    st = np.zeros(3, np.int32)
    L.host_gi_stats(st.ctypes.data_as(C.POINTER(C.c_int)), 1)
    tau, met, status, it = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], hexv=True)
    L.host_gi_stats(st.ctypes.data_as(C.POINTER(C.c_int)), 1)
    L.host_gi_dump(None)
    lanes = [4 * (k // 3) + k % 3 for k in range(12)]
    J = buf[:, lanes, :12]
    z0 = buf[:, lanes, 13]
    mu_n = buf[:, 0, 14]
    inv_s = buf[:, :, 15].max(1)
    ct = buf[:, ::4, 15] > 0
    return dict(J=J, z0=z0, mu_n=mu_n, inv_s=inv_s, ct=ct, iters=it, tau=tau, status=status, stats=st)


if __name__ == "__main__":
    cfg = int(sys.argv[1]); n = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else None
    d = dump(cfg, n, kind)
    out = sys.argv[4] if len(sys.argv) > 4 else "/tmp/gi_cfg%d_%s_%d.npz" % (cfg, kind or "def", n)
    np.savez(out, **d)
    it = d["iters"]
    print(out, "iters mean %.2f max %d" % (it.mean(), it.max()), "fast/generic/drops", d["stats"], "hist", np.bincount(it))
