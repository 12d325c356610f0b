#!/usr/bin/env python3
"""8-GPU hand-over kit, part 1: what an N-GPU run of the sharded path MUST print, computed on ONE GPU.

north_star: "the robot-instance batch shards embarrassingly across the 8 GPUs of one node with RCCL over xGMI only to gather end-of-rollout
statistics".  Sharding is contiguous windows of one seeded batch (quadruped_drake_amd/stats.py: shard_range; workloads.make_batch(window=...)), the
ticks are deterministic, so the torque bits of every rank's shard -- and every exact field of the gathered statistics -- are known before a
multi-GPU node is ever touched: this tool steps each of the 1 + 2 + 4 + 8 windows of BASELINE configs[4] (4096 instances per GPU, weak scaling:
the batch of an N-GPU run has 4096 N instances) on GPU 0 and records the FNV-1a 64 checksum of its [12][n] torques, as bench.py and
examples/wbc_host.cpp print them per rank (`per_rank_tau_fnv1a64`), plus the per-launch statistics.

    python3 tools/scale8_expected.py --write            (GPU box)  -> profiles/scale8_expected.json
    python3 tools/scale8_expected.py --check a.json ...             compares JSON lines of bench.py / wbc_host runs with the file (no GPU needed)

The file carries the kernel-source identity it was computed on (bench.kernel_src_sha16); tests/test_scale8_kit.py recomputes it on the GPU box."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXPECTED = os.path.join(ROOT, "profiles", "scale8_expected.json")
WORLDS = (1, 2, 4, 8)
PER_GPU = 4096
CONFIG = 5


def compute(worlds=WORLDS, per_gpu=PER_GPU, device=0):
    """{world: {"per_rank_tau_fnv1a64": [...], "per_rank": [per-launch statistics of each shard], "per_launch": their fold}} on one GPU."""
    import numpy as np
    import torch
    import bench
    from quadruped_drake_amd import MPTCController, workloads
    from quadruped_drake_amd import stats as wstats
    dev = "cuda:%d" % device
    out = {}
    ctrl = MPTCController(model="mini_cheetah", max_batch=per_gpu + 1, device=device)
    for world in worlds:
        n_total = per_gpu * world
        hashes, per_rank = [], []
        for rank in range(world):
            sh = workloads.make_batch(CONFIG, n=n_total, window=wstats.shard_range(n_total, rank, world))
            up = lambda x: torch.tensor(np.ascontiguousarray(x), device=dev)
            ctrl.stats(reset=True)
            tau, met, st = ctrl.step(up(sh["q"]), up(sh["v"]), up(sh["targets"]), up(sh["mask"]), up(sh["mu"]), up(sh["mass_scale"]))
            ctrl.sync()
            s = ctrl.stats()
            hashes.append("%016x" % bench.fnv1a64(tau.cpu().numpy().tobytes()))
            per_rank.append({"ticks": s["ticks"], "status_nonzero": s["status_nonzero"], "iters_sum": s["iters_sum"], "tau_abs_max": s["tau_abs_max"],
                             "mask_count": s["mask_count"], "status_nonzero_outputs": int((st != 0).sum())})
        fold = {"ticks": sum(r["ticks"] for r in per_rank), "status_nonzero": sum(r["status_nonzero"] for r in per_rank),
                "iters_sum": sum(r["iters_sum"] for r in per_rank), "tau_abs_max": max(r["tau_abs_max"] for r in per_rank),
                "mask_count": [sum(r["mask_count"][k] for r in per_rank) for k in range(16)]}
        out[str(world)] = {"instances": n_total, "per_rank_tau_fnv1a64": hashes, "per_rank": per_rank, "per_launch": fold}
    ctrl.close()
    return out


def expected_doc(worlds_doc):
    import bench
    from quadruped_drake_amd import workloads
    return {"what": "per-rank torque checksums (FNV-1a 64 over the [12][n] doubles of the last launch) and exact per-launch statistics of the sharded "
                    "BASELINE configs[4] batch, computed on ONE GPU window by window: an N-GPU run of bench.py / examples/wbc_host must print these",
            "config": CONFIG, "per_gpu": PER_GPU, "seed": workloads.make_batch(CONFIG, n=8, window=(0, 8))["seed"], "kind": "mptc",
            "kernel_src_sha16": bench.kernel_src_sha16(),
            "statistics_note": "per launch; a run of K timed steps reports K times ticks / status_nonzero / iters_sum / mask_count and the same tau_abs_max "
                               "(tau_abs_sum and err_sum are floating-point sums whose last bits depend on the order of the atomics: not compared)",
            "worlds": worlds_doc}


def check(paths, expected_path=EXPECTED):
    """Compare JSON lines (the LAST line starting with '{' of each file) with the expected file; returns (ok, rows)."""
    exp = json.load(open(expected_path))
    rows, ok_all = [], True
    for path in paths:
        lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
        if not lines:
            rows.append((path, "no JSON line", False)); ok_all = False
            continue
        d = json.loads(lines[-1])
        world = int(d["n_gpus"])
        e = exp["worlds"].get(str(world))
        host = "wbc_host" if "host" in d else "bench.py"
        if e is None:
            rows.append((path, "no expectation for %d GPUs" % world, False)); ok_all = False
            continue
        problems = []
        if d.get("ranks_seen") != world:
            problems.append("ranks_seen %r != %d" % (d.get("ranks_seen"), world))
        if d["per_rank_tau_fnv1a64"] != e["per_rank_tau_fnv1a64"]:
            bad = [r for r, (a, b) in enumerate(zip(d["per_rank_tau_fnv1a64"], e["per_rank_tau_fnv1a64"])) if a != b]
            problems.append("torque checksum differs on rank(s) %s" % bad)
        k = int(d["steps"])
        rs, pl = d["rollout_stats"], e["per_launch"]
        for f in ("ticks", "status_nonzero", "iters_sum"):
            if float(rs[f]) != k * pl[f]:
                problems.append("%s %r != %d x %r" % (f, rs[f], k, pl[f]))
        if float(rs["tau_abs_max"]) != pl["tau_abs_max"]:
            problems.append("tau_abs_max %r != %r" % (rs["tau_abs_max"], pl["tau_abs_max"]))
        if "mask_count" in rs and [float(x) for x in rs["mask_count"]] != [k * x for x in pl["mask_count"]]:
            problems.append("mask_count differs")
        if [float(x) for x in d["per_rank_ticks"]] != [k * r["ticks"] for r in e["per_rank"]]:
            problems.append("per_rank_ticks %r" % d["per_rank_ticks"])
        if d.get("kernel_src_sha16") not in (None, exp["kernel_src_sha16"]):
            problems.append("kernel sources %s, expectations computed on %s: regenerate with --write" % (d["kernel_src_sha16"], exp["kernel_src_sha16"]))
        kms = d.get("per_rank_kernel_ms") or [d.get("kernel_ms")]
        ag = (d.get("rccl") or {}).get("allgather_us")
        rows.append((path, "%-8s N=%d  %.4g ticks/s  kernel_ms/rank %s%s  %s" % (
            host, world, d["value"], ["%.4f" % x for x in kms], ("  all-gather %.0f us" % ag) if ag is not None else "",
            "OK: shard checksums and statistics as predicted" if not problems else "MISMATCH: " + "; ".join(problems)), not problems))
        ok_all = ok_all and not problems
    return ok_all, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="GPU box: compute and write profiles/scale8_expected.json")
    ap.add_argument("--out", default=EXPECTED)
    ap.add_argument("--check", nargs="*", help="JSON outputs of bench.py / examples/wbc_host runs to compare with the file")
    a = ap.parse_args()
    if a.write:
        doc = expected_doc(compute())
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")
        print("wrote %s (kernel sources %s): %s" % (a.out, doc["kernel_src_sha16"], {w: d["per_rank_tau_fnv1a64"][:2] for w, d in doc["worlds"].items()}))
    if a.check is not None:
        ok, rows = check(a.check, a.out)
        for path, text, good in rows:
            print("%s  %s" % (os.path.basename(path), text))
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
