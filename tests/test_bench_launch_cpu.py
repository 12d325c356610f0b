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
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "1", "--per-gpu", "64"])   # --dry-run: the same on a box WITH a GPU
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


def test_one_rank_process_group_gloo_dry_run():
    """--force-pg: ONE rank still initialises the group and sends its statistics through the collective (the RCCL path's CPU twin)."""
    r = _run(["--gpus", "1", "--force-pg", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "1", "--per-gpu", "64"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["dry_run"] is True and d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["per_rank_ticks"] == [64.0 * 3]


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
    r = _run(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "5", "--warmup", "2", "--ramp-seconds", "0", "--per-gpu", "512",
              "--cpu-seconds", "0.5"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert "dry_run" not in d and d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["per_rank_ticks"] == [512.0 * 5, 512.0 * 5] and d["rollout_stats"]["ticks"] == 2 * 512 * 5
    assert d["status_nonzero"] == 0 and d["value"] > 0 and d["config"]["domain_randomised"] is True
    # the N > 1 line carries its own parity and CPU baseline: every rank checked 64 instances of ITS shard against the oracle
    assert len(d["per_rank_torque_rel_err"]) == 2 and all(0.0 < e < 1e-6 for e in d["per_rank_torque_rel_err"])
    assert d["torque_rel_err_vs_cpu_ref"] == max(d["per_rank_torque_rel_err"])
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1


@pytest.mark.gpu
def test_rccl_one_rank_group_runs_the_statistics_collective():
    """RCCL itself (backend "nccl" on ROCm) before the multi-GPU node does: a ONE-rank group initialised exactly like bench.py's
    ranks (device_id=, 127.0.0.1 rendezvous, HSA_ENABLE_IPC_MODE_LEGACY=0) runs stats.all_gather_stats on DEVICE tensors."""
    code = r'''
import os, socket, sys
sys.path.insert(0, %r)
with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from quadruped_drake_amd import stats
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
local = dict(ticks=4096.0, status_nonzero=3.0, iters_sum=8000.0, tau_abs_sum=1.5e5, tau_abs_max=41.5, err_sum=2.0,
             mask_count=[float(k) for k in range(16)])
red, per_rank, seen = stats.all_gather_stats(local, device=torch.device("cuda", 0))
assert seen == 1 and len(per_rank) == 1 and red == local and per_rank[0] == local, (red, local)
t = torch.ones(8, dtype=torch.float64, device="cuda:0"); dist.all_reduce(t); torch.cuda.synchronize()
assert float(t.sum()) == 8.0
dist.barrier(); dist.destroy_process_group()
print("rccl ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.gpu
def test_bench_one_gpu_through_the_process_group_branch():
    """bench.py --gpus 1 --force-pg: the N > 1 code path (RCCL init with device_id=, the device-side statistics gather, the
    per-rank timing gather) on the one GPU of this box; the line carries ranks_seen / per_rank_* for N = 1."""
    r = _run(["--gpus", "1", "--force-pg", "--steps", "10", "--warmup", "2", "--ramp-seconds", "0.2", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["process_group"] == "nccl" and d["n_gpus"] == 1 and d["ranks_seen"] == 1
    assert d["per_rank_ticks"] == [4096.0 * 10] and len(d["per_rank_kernel_ms"]) == 1 and d["per_rank_kernel_ms"][0] > 0
    assert d["status_nonzero"] == 0 and d["value"] > 0 and d["value_cold"] > 0
    # HBM bytes per launch are replayed from the committed counter file -- only when it was collected on THIS kernel build
    import bench
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
        ent = json.load(f)["mptc_cfg3_n4096_hex"]
    same_build = ent.get("kernel_src_sha16") == bench.kernel_src_sha16()
    assert (d["roofline"]["traffic"] == ent["bytes_per_launch"]) if same_build else (d["roofline"]["traffic"] is None)
    # the timed region's own accounting (round 5): four parts by the host clock, the device time of the K launches inside the total, the line's
    # ms_per_step = total / K, the kernel-only rate from the HIP events
    reg = d["region_us"]
    parts = reg["queue_K_launches"] + reg["statistics_reduce_and_the_one_wait"] + reg["gather"] + reg["closing_bracket"]
    assert abs(parts - reg["total"]) < 1.0 and reg["device_time_of_the_K_launches"] < reg["total"]
    assert abs(d["ms_per_step"] * 1e3 * 10 - reg["total"]) < 1.0
    assert abs(d["value_kernel_only"] - 4096.0 / (d["roofline"]["kernel_ms"] * 1e-3)) < 1e-6 * d["value_kernel_only"] and d["value_kernel_only"] > d["value"]


def _torchrun(nproc, args, timeout=600):
    """The driver's own launch line for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ..."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_the_drivers_torchrun_line_world2_gloo_dry_run():
    """bench.py under the launcher the driver uses (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run, not from bench.py's own
    self-launch): two ranks, contiguous shards, one statistics exchange, ONE JSON line from rank 0."""
    r = _torchrun(2, ["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "1", "--per-gpu", "64"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["per_rank_ticks"] == [64.0 * 3, 64.0 * 3]


@pytest.mark.gpu
def test_the_drivers_torchrun_line_two_real_ranks_sharing_the_gpu():
    """The same launcher with the REAL kernels: two ranks on GPU 0 (gloo stands in for RCCL, which needs a device per rank), the config-5 shards of the
    8-GPU hand-over kit -- the line must match profiles/scale8_expected.json like a self-launched run does."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale8_expected as s8
    r = _torchrun(2, ["--gpus", "2", "--config", "5", "--per-gpu", "4096", "--backend", "gloo", "--share-gpu", "--steps", "10", "--warmup", "2",
                      "--ramp-seconds", "0", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        f.write(r.stdout)
    ok, rows = s8.check([f.name])
    os.unlink(f.name)
    assert ok, rows
