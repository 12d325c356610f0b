"""The N>1 path on CPU: contiguous batch shard + the one statistics all-reduce, world_size 2, gloo."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_batch():
    from quadruped_drake_amd import stats as ws
    for n, w in ((4096, 8), (10, 3), (7, 8), (32768, 8)):
        r = [ws.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import oracle_py as orc
    from quadruped_drake_amd import stats as ws, workloads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = workloads.make_batch(5, n=50)
    sh = ws.shard_batch(batch, rank, world)
    # each rank computes its shard's statistics (with the oracle here: no GPU in this container)
    m = orc.model(sh["model"]); p = orc.params(sh["kind"])
    tau, met, st = orc.step_batch(sh["kind"], m, p, sh["q"], sh["v"], sh["targets"], sh["mask"], sh["mu"], sh["mass_scale"])
    local = dict(ticks=float(sh["n"]), status_nonzero=float((st != 0).sum()), iters_sum=0.0,
                 tau_abs_sum=float(np.abs(tau).sum()), tau_abs_max=float(np.abs(tau).max()),
                 err_sum=float(met[1].sum()), mask_count=[float((sh["mask"] == k).sum()) for k in range(16)])
    red = ws.all_reduce_stats(local)
    q.put((rank, sh["n"], red))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_shard_and_stats_reduce():
    sys.path.insert(0, ROOT)
    from oracle import oracle_py as orc
    from quadruped_drake_amd import workloads
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    batch = workloads.make_batch(5, n=50)
    tau, met, st = orc.step_batch("mptc", orc.model("mini_cheetah"), orc.params("mptc"), batch["q"], batch["v"],
                                  batch["targets"], batch["mask"], batch["mu"], batch["mass_scale"])
    assert sorted(r[1] for r in res) == [25, 25]
    for _, _, red in res:
        assert red["ticks"] == 50.0
        assert abs(red["tau_abs_sum"] - np.abs(tau).sum()) < 1e-9 * np.abs(tau).sum()
        assert red["tau_abs_max"] == np.abs(tau).max()
        assert abs(red["err_sum"] - met[1].sum()) < 1e-12 + 1e-12 * abs(met[1].sum())
        assert red["mask_count"] == [float((batch["mask"] == k).sum()) for k in range(16)]
