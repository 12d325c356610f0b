"""Batch sharding and the one collective of the path.

The robot-instance batch shards embarrassingly: rank g owns instances [g*N/W, (g+1)*N/W) and no
instance reads another's state, so there is no data-path collective.  The only exchange is one
all-gather of a 22-double statistics vector at the end of a rollout (the wbc_stats_pack layout of include/wbc.h, reduced
by wbc_stats_reduce: sums and one maximum; a C caller does the same with its own ncclAllGather)
-- RCCL over xGMI on the GPU box (backend "nccl"), gloo in the CPU tests.  At < 200 B the message
is latency-bound; link bandwidth is irrelevant.
"""
import numpy as np



def shard_range(n_total, rank, world):
    """Contiguous split; the first n_total % world ranks get one extra instance."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch, rank, world):
    lo, hi = shard_range(batch["n"], rank, world)
    out = dict(batch)
    out["n"] = hi - lo
    for k in ("q", "v", "targets"):
        out[k] = np.ascontiguousarray(batch[k][:, lo:hi])
    for k in ("mask", "mu", "mass_scale"):
        out[k] = None if batch[k] is None else np.ascontiguousarray(batch[k][lo:hi])
    return out


FIELDS = ["ticks", "status_nonzero", "iters_sum", "tau_abs_sum", "tau_abs_max", "err_sum"]   # wbc_stats, declaration order (include/wbc.h)
NSTAT = 22


def to_vector(stats):
    """The wbc_stats_pack vector (WBC_NSTAT doubles in the field order of wbc_stats) of a statistics dict."""
    return np.array([stats[k] for k in FIELDS] + list(stats["mask_count"]), dtype=np.float64)


def from_vector(vec):
    out = {k: float(vec[i]) for i, k in enumerate(FIELDS)}
    out["mask_count"] = [float(x) for x in vec[len(FIELDS):NSTAT]]
    return out


MAX_FIELD = FIELDS.index("tau_abs_max")


def reduce_vectors_numpy(gathered):
    """The same fold in numpy: every field summed, tau_abs_max the maximum."""
    g = np.ascontiguousarray(gathered, dtype=np.float64).reshape(-1, NSTAT)
    acc = g[0].copy()
    for r in range(1, g.shape[0]):      # rank order, like wbc_stats_reduce (the sums then round identically)
        acc += g[r]
    acc[MAX_FIELD] = g[:, MAX_FIELD].max()
    return from_vector(acc)


def reduce_vectors(gathered):
    """wbc_stats_reduce of the C ABI on `world` gathered vectors ([world, 22]): the reduction a C caller with its own
    ncclAllGather runs -- the Python mirror goes through the same function.  This is 22 additions on the host and no part of
    the hot path: a host-only rank without the HIP extension (gloo dry runs, CPU tests before build()) folds in numpy instead
    (tests/test_abi_cpu.py holds the two equal)."""
    import ctypes as C
    from . import _lib
    g = np.ascontiguousarray(gathered, dtype=np.float64).reshape(-1, NSTAT)
    try:
        L = _lib.lib()
    except (_lib.WbcError, OSError):
        return reduce_vectors_numpy(g)
    out = _lib.WbcStats()
    _lib.check(L.wbc_stats_reduce(g.ctypes.data_as(_lib.c_double_p), int(g.shape[0]), C.byref(out)))
    d = {k: float(getattr(out, k)) for k in FIELDS}
    d["mask_count"] = [float(x) for x in out.mask_count]
    return d


def all_gather_stats(stats, device=None):
    """ONE collective: gather every rank's 22-double vector (wbc_stats_pack layout), reduce with wbc_stats_reduce.
    Returns (reduced dict, per-rank list of dicts, world size seen by the process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(stats), [dict(stats)], 1
    # an initialised group of ONE rank still goes through the collective (bench.py --force-pg: the RCCL path on a one-GPU box)
    t = torch.tensor(to_vector(stats), dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    a = torch.stack(parts).cpu().numpy()
    return reduce_vectors(a), [from_vector(a[r]) for r in range(a.shape[0])], dist.get_world_size()


def all_reduce_stats(stats, device=None):
    """Reduce a wbc_stats dict over the default process group (no-op without one)."""
    return all_gather_stats(stats, device)[0]
