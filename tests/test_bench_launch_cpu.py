"""`python bench.py --gpus N` without a launcher: the parent starts N fresh rank processes itself.
Exercised here on CPU through the `--backend gloo` test switch (no GPU: launch + shard + statistics exchange only)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_self_launch_world2_gloo_dry_run():
    r = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--per-gpu", "64"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE JSON line, relayed by the parent
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["per_rank_ticks"] == [64.0 * 3, 64.0 * 3]       # contiguous shard: 64 instances per rank, 3 steps
    assert d["rollout_stats"]["ticks"] == 2 * 64 * 3
    assert d["scaling"] == "weak" and d["config"]["parallelism"] == "batch-shard x2"


def test_self_launch_propagates_a_failing_rank():
    # WORLD_SIZE is checked against --gpus inside every rank: a child that fails must fail the parent quickly
    r = _run(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--config", "7"], timeout=120)   # config 7 does not exist
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_rank_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    r = _run(["--gpus", "1", "--steps", "1"], timeout=120)
    assert r.returncode != 0 and "no GPU" in (r.stderr + r.stdout)


import pytest


@pytest.mark.gpu
def test_self_launch_two_ranks_compute_on_the_gpu():
    """The self-launched N > 1 path with the REAL kernel: two rank processes, contiguous shards of BASELINE config 5's batch,
    one statistics all-gather -- both ranks share GPU 0 here (gloo test switch; RCCL needs one device per rank)."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "5", "--warmup", "2", "--ramp-seconds", "0", "--per-gpu", "512"],
             timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "dry_run" not in d and d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["per_rank_ticks"] == [512.0 * 5, 512.0 * 5] and d["rollout_stats"]["ticks"] == 2 * 512 * 5
    assert d["status_nonzero"] == 0 and d["value"] > 0 and d["config"]["domain_randomised"] is True
