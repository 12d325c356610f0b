"""The 8-GPU hand-over kit (tools/scale8.sh, tools/scale8_expected.py, profiles/scale8_expected.json): SURVEY 8e's sharded path has never met a
node with more than one GPU, so what such a run must print is computed where it CAN be computed -- every rank's window of the sharded batch stepped
on one GPU -- and committed.  CPU: the file is well formed and the checker accepts / rejects what it should.  GPU: the file is what this build
produces (bit-exact torque checksums of all 1 + 2 + 4 + 8 windows), and a real one-GPU run of both hosts passes the checker."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scale8_expected as s8  # noqa: E402


def test_expected_file_is_well_formed():
    exp = json.load(open(s8.EXPECTED))
    assert exp["config"] == 5 and exp["per_gpu"] == 4096 and sorted(exp["worlds"]) == ["1", "2", "4", "8"]
    seen = set()
    for w, d in exp["worlds"].items():
        w = int(w)
        assert d["instances"] == 4096 * w and len(d["per_rank_tau_fnv1a64"]) == w == len(d["per_rank"])
        assert all(len(h) == 16 and int(h, 16) >= 0 for h in d["per_rank_tau_fnv1a64"])
        assert d["per_launch"]["ticks"] == 4096 * w and d["per_launch"]["status_nonzero"] == 0
        assert sum(d["per_launch"]["mask_count"]) == 4096 * w
        assert all(r["ticks"] == 4096 and r["status_nonzero_outputs"] == 0 for r in d["per_rank"])
        seen.update(d["per_rank_tau_fnv1a64"])
    assert len(seen) == 15          # fifteen different windows of four different batches: fifteen different checksums


def _line(world, exp, steps=200, **over):
    e = exp["worlds"][str(world)]
    pl = e["per_launch"]
    d = {"n_gpus": world, "ranks_seen": world, "steps": steps, "value": 1.0e8 * world, "per_rank_kernel_ms": [0.0229] * world,
         "per_rank_tau_fnv1a64": list(e["per_rank_tau_fnv1a64"]), "per_rank_ticks": [steps * r["ticks"] for r in e["per_rank"]],
         "kernel_src_sha16": exp["kernel_src_sha16"],
         "rollout_stats": {"ticks": steps * pl["ticks"], "status_nonzero": 0, "iters_sum": steps * pl["iters_sum"], "tau_abs_max": pl["tau_abs_max"]}}
    d.update(over)
    return d


def test_checker_accepts_the_prediction_and_names_what_differs(tmp_path):
    exp = json.load(open(s8.EXPECTED))
    good = tmp_path / "bench_8.json"; good.write_text("some log line\n" + json.dumps(_line(8, exp)) + "\n")
    ok, rows = s8.check([str(good)])
    assert ok and "OK" in rows[0][1]
    # one rank stepped the wrong window
    h = list(exp["worlds"]["8"]["per_rank_tau_fnv1a64"]); h[5] = h[4]
    bad = tmp_path / "bad.json"; bad.write_text(json.dumps(_line(8, exp, per_rank_tau_fnv1a64=h)))
    ok, rows = s8.check([str(bad)])
    assert not ok and "rank(s) [5]" in rows[0][1]
    # a rank missing from the gather
    line = _line(4, exp, ranks_seen=3)
    line["rollout_stats"]["ticks"] -= 200 * 4096
    miss = tmp_path / "miss.json"; miss.write_text(json.dumps(line))
    ok, rows = s8.check([str(miss)])
    assert not ok and "ranks_seen" in rows[0][1] and "ticks" in rows[0][1]
    # a run of another kernel build says so instead of reporting sixteen wrong checksums as a sharding bug
    other = tmp_path / "other.json"; other.write_text(json.dumps(_line(2, exp, kernel_src_sha16="0" * 16)))
    ok, rows = s8.check([str(other)])
    assert not ok and "regenerate" in rows[0][1]


def test_kit_script_is_executable_and_self_contained():
    p = os.path.join(ROOT, "tools", "scale8.sh")
    assert os.access(p, os.X_OK)
    src = open(p).read()
    assert "/root/reference" not in src and "bench.py --gpus $n" in src and "examples/wbc_host" in src and "scale8_expected.py --check" in src


@pytest.mark.gpu
def test_expected_file_is_what_this_build_computes():
    """All fifteen windows on one GPU against the committed file.  A kernel change moves the torque bits: regenerate the file
    (python3 tools/scale8_expected.py --write on a GPU box) in the same commit."""
    import bench
    exp = json.load(open(s8.EXPECTED))
    assert exp["kernel_src_sha16"] == bench.kernel_src_sha16(), "profiles/scale8_expected.json was computed on other kernel sources: tools/scale8_expected.py --write"
    got = s8.compute()
    for w in ("1", "2", "4", "8"):
        assert got[w]["per_rank_tau_fnv1a64"] == exp["worlds"][w]["per_rank_tau_fnv1a64"], w
        for f in ("ticks", "status_nonzero", "iters_sum", "tau_abs_max", "mask_count"):
            assert got[w]["per_launch"][f] == exp["worlds"][w]["per_launch"][f], (w, f)


@pytest.mark.gpu
def test_one_gpu_runs_of_both_hosts_pass_the_checker(tmp_path):
    """The kit's N = 1 leg for real: bench.py and examples/wbc_host on the config-5 shard of one GPU, through the checker."""
    from quadruped_drake_amd import workloads
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    b = tmp_path / "bench_1.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "5", "--per-gpu", "4096", "--steps", "50", "--warmup", "5",
                        "--no-cpu-baseline", "--ramp-seconds", "0.2"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    b.write_text(r.stdout)
    files = [str(b)]
    exe = os.path.join(ROOT, "examples", "wbc_host")
    if os.path.exists(exe):
        path = workloads.dump_batch(str(tmp_path / "cfg5.bin"), workloads.make_batch(5, n=4096))
        r = subprocess.run([exe, "--batch", path, "--gpus", "1", "--steps", "50", "--warmup", "5", "--ramp-seconds", "0.2"], capture_output=True, text=True,
                           env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        h = tmp_path / "host_1.json"; h.write_text(r.stdout)
        files.append(str(h))
    ok, rows = s8.check(files)
    assert ok, rows


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_real_multi_rank_runs_sharing_one_gpu_match_the_predictions(world, tmp_path):
    """The nearest thing to an N-GPU run this pool allows: `bench.py --gpus N` self-launches N REAL rank processes -- each generates its own window of the
    sharded config-5 batch, steps it with the real kernels, joins the ONE statistics all-gather -- all on GPU 0 (`--share-gpu --backend gloo`: RCCL refuses
    two ranks on one device, so the collective is gloo's; sharding, windows, launcher, statistics and the MAX over ranks are the N-GPU run's own).  Every
    rank's torque checksum and the gathered statistics must be the ones profiles/scale8_expected.json predicts for that N."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", "5", "--per-gpu", "4096", "--backend", "gloo", "--share-gpu",
                        "--steps", "10", "--warmup", "2", "--ramp-seconds", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    f = tmp_path / ("bench_%d.json" % world); f.write_text(r.stdout)
    ok, rows = s8.check([str(f)])
    assert ok, rows
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == world == d["ranks_seen"] and len(set(d["per_rank_tau_fnv1a64"])) == world and d["status_nonzero"] == 0
