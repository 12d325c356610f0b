#!/usr/bin/env python3
"""Analysis (GPU box, -DWBC_STAMPS -DWBC_STAMPS_GI build via WBC_HIP_LIB): is the tail finding of profiles/r05/apex_rule.md a property of ONE seeded batch?  For several seeds of
BASELINE config 3 (N = 4096): how many wavefronts run the generic loop, the slowest wavefront's lifetime, the slowest fast-path-only wavefront, iterations per tick.
    WBC_HIP_LIB=build_variants/stamps_gi_mptc.so python3 tools/lab/r05/tail_seeds.py [seed ...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT)
import torch
from quadruped_drake_amd import MPTCController, workloads, _lib
seeds = [int(x) for x in sys.argv[1:]] or [1002, 7, 8, 9, 10, 11, 12, 13]
n = 4096; nb = n // 4
L = _lib.lib(); L.wbc_debug_stamps.argtypes = [C.c_void_p, C.c_int]
for seed in seeds:
    b = workloads.make_batch(3, n=n, seed=seed)
    ctrl = MPTCController(model=b["model"], max_batch=n, device=0)
    up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
    args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
    zero = np.zeros((nb, 16), dtype=np.uint64)
    for _ in range(50): ctrl.step(*args)
    ctrl.sync(); ctrl.stats(reset=True)
    lifes = []; gens = []
    for rep in range(8):
        ctrl.step(*args); ctrl.sync()
        buf = np.zeros((nb, 16), dtype=np.uint64)
        assert L.wbc_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
        lifes.append(buf[:, 5].astype(np.int64) - buf[:, 0].astype(np.int64))
    st = ctrl.stats()
    # generic-loop wavefronts: the accumulated section slots are only written by wavefronts that ran the loop; clear them and launch once more
    ctrl.close()
    life = np.median(np.array(lifes), axis=0)
    print("seed %5d: iterations per tick %.3f | slowest wavefront (median of 8 launches) %6d cycles, p99 %6d, median %6d" % (
        seed, st["iters_sum"] / st["ticks"], life.max(), np.percentile(life, 99), np.median(life)), flush=True)
