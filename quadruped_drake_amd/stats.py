"""Batch sharding and the one collective of the path.

The robot-instance batch shards embarrassingly: rank g owns instances [g*N/W, (g+1)*N/W) and no
instance reads another's state, so there is no data-path collective.  The only exchange is one
all-gather of a 22-double statistics vector at the end of a rollout (reduced locally: SUM fields and one MAX field)
-- RCCL over xGMI on the GPU box (backend "nccl"), gloo in the CPU tests.  At < 200 B the message
is latency-bound; link bandwidth is irrelevant.
"""
import numpy as np

SUM_FIELDS = ["ticks", "status_nonzero", "iters_sum", "tau_abs_sum", "err_sum"]
MAX_FIELDS = ["tau_abs_max"]


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


def pack(stats):
    s = np.array([stats[k] for k in SUM_FIELDS] + list(stats["mask_count"]), dtype=np.float64)
    m = np.array([stats[k] for k in MAX_FIELDS], dtype=np.float64)
    return s, m


def unpack(s, m):
    out = {k: float(s[i]) for i, k in enumerate(SUM_FIELDS)}
    out["mask_count"] = [float(x) for x in s[len(SUM_FIELDS):]]
    out.update({k: float(m[i]) for i, k in enumerate(MAX_FIELDS)})
    return out


def all_gather_stats(stats, device=None):
    """ONE collective: gather every rank's 22-double vector, reduce (sum / max) locally.
    Returns (reduced dict, per-rank list of dicts, world size seen by the process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(stats), [dict(stats)], 1
    # an initialised group of ONE rank still goes through the collective (bench.py --force-pg: the RCCL path on a one-GPU box)
    s, m = pack(stats)
    t = torch.tensor(np.concatenate([s, m]), dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    a = torch.stack(parts).cpu().numpy()
    per_rank = [unpack(a[r, :len(s)], a[r, len(s):]) for r in range(a.shape[0])]
    return unpack(a[:, :len(s)].sum(0), a[:, len(s):].max(0)), per_rank, dist.get_world_size()


def all_reduce_stats(stats, device=None):
    """Reduce a wbc_stats dict over the default process group (no-op without one)."""
    return all_gather_stats(stats, device)[0]
