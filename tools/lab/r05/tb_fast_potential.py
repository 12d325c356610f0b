#!/usr/bin/env python3
"""Analysis (host emulation): would a fast path help the torque-box kernels?  A fast path for them would carry the torque rows' images along and hand over to the
generic loop at the first trip in which ANY robot of the wavefront picks a torque row, meets a blocked step or has a drop.  Per robot the host records the first trip
that picked a torque row and the first trip with a drop; in lock step of four the wavefront leaves at the minimum.
Needs the two analysis hooks of tools/lab/patches/gi_trace.patch (host builds only; kept out of csrc/ so that the kernel-source identity of the committed counters stands):
    git apply tools/lab/patches/gi_trace.patch && python3 tools/lab/r05/tb_fast_potential.py; git checkout quadruped_drake_amd/csrc/wbc_hex.hpp tools/host_tick.cpp"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import host_tick as ht
from oracle import oracle_py as orc
from quadruped_drake_amd import workloads
L = ht.lib()
for kind, cfg, n, tmax in (("mptc", 3, 512, 12.0), ("mptc", 3, 512, 8.0), ("id", 3, 512, 12.0), ("mptc", 2, 256, 12.0)):
    b = workloads.make_batch(cfg, n=n)
    t = orc.load_model_json(b["model"]); p = orc.params(kind); p.tau_max = tmax
    pp = np.array([p.Kp_body_p, p.Kd_body_p, p.Kp_body_rpy, p.Kd_body_rpy, p.Kp_foot, p.Kd_foot, p.w_body, p.w_foot, p.mu, p.Kd_contact, p.tau_max, p.tiebreak_eps2])
    tr = np.full((n, 2), -1, np.int32)
    L.host_gi_trace.argtypes = [C.c_void_p]; L.host_gi_trace(tr.ctypes.data_as(C.c_void_p))
    tau, met, st, it = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"], params12=pp, hexv=True)
    L.host_gi_trace(None)
    big = 10 ** 6
    first = np.where(tr < 0, big, tr).min(1)                  # the robot's first trip the fast path cannot do (torque row or drop)
    leave = first.reshape(-1, 4).min(1)                       # the wavefront leaves there
    trips = it.reshape(-1, 4).max(1)                          # lock-step trips of the wavefront
    fast = np.minimum(leave, trips)
    print("%s cfg %d tau_max %.0f: trips per robot %.2f, robots with a torque row picked %.0f %%, with a drop %.0f %% | lock step of 4: %.2f trips per wavefront, of which a fast path could take %.2f (%.0f %%); wavefronts that never leave it %.0f %%" % (
        kind, cfg, tmax, it.mean(), 100 * (tr[:, 0] >= 0).mean(), 100 * (tr[:, 1] >= 0).mean(), trips.mean(), fast.mean(), 100 * fast.sum() / trips.sum(), 100 * (leave >= trips).mean()))
