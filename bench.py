#!/usr/bin/env python3
"""Headline benchmark: whole-body-QP control ticks/s (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (one fused HIP launch = one control tick for every robot of
the rank's shard) over a synthetic randomised-state batch that is resident in HBM before the
timed region starts.  Per-GPU work is fixed at 4096 Mini-Cheetah instances (weak scaling):
  N = 1 : BASELINE.json configs[2] -- 4096 Mini Cheetah, trot contact modes, MPTC controller;
  N > 1 : BASELINE.json configs[4] pattern -- 4096*N instances with per-instance mu / mass scale,
          sharded contiguously, no data-path collective; one RCCL all-reduce of the end-of-rollout
          statistics vector closes the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Counted with the operation-counting scalar (tools/host_tick.cpp) on the full config; frozen in
# BASELINE.md section 4.  add + mul + div + sqrt each count 1 (an FMA is 2); trig/compare are not counted.
FLOPS_PER_TICK = {("mptc", 3): 37629.0, ("mptc", 5): 37703.0, ("id", 2): 35667.0, ("mptc", 4): 36085.0}
BYTES_PER_TICK = {False: 864.0, True: 880.0}     # SURVEY.md section 8(d); True = with mu and mass scale
PEAK_FP64_VALU_TFLOPS = 78.6                      # MI355X vector FP64 (spec); FP64 MFMA peak is the same figure
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--per-gpu", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--config", type=int, default=0, help="override workload config (2,3,4,5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", default="auto", choices=["auto", "lane", "quad", "hex"], help="kernel variant (A/B runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work (core-seconds) for the baseline sample")
    return ap.parse_args()


def cpu_baseline(batch, seconds):
    """Oracle (CPU restatement of the Drake+OSQP path, kind "port") on this box's host cores."""
    from oracle import oracle_py as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    m = orc.model(batch["model"]); p = orc.params(batch["kind"])
    n0 = min(512, batch["n"])
    sl = lambda a, n: None if a is None else (a[:, :n] if a.ndim == 2 else a[:n])
    args = lambda n: (sl(batch["q"], n), sl(batch["v"], n), sl(batch["targets"], n), sl(batch["mask"], n),
                      sl(batch["mu"], n), sl(batch["mass_scale"], n))
    t = time.perf_counter()
    orc.step_batch(batch["kind"], m, p, *args(n0), nthreads=1)
    rate1 = n0 / (time.perf_counter() - t)
    n = batch["n"]
    reps = max(1, int(round(rate1 * seconds / n)))
    t = time.perf_counter()
    orc.bench_batch(batch["kind"], m, p, *args(n), nthreads=cores, reps=reps)   # one OpenMP region
    dt = time.perf_counter() - t
    return {"value": n * reps / dt, "unit": "ticks/s", "cores": cores, "kind": "port",
            "single_core_ticks_per_s": rate1,
            "sample": "the same %d-instance batch x %d passes (one OpenMP region, dynamic schedule), C oracle = dense "
                      "restatement of the Drake+OSQP tick" % (n, reps)}


def closed_loop(shard, device, steps=300):
    """Closed-loop rate of the same law (wbc_rollout: lookup -> tick -> forward step, one persistent launch):
    nominal standing states with small perturbations, standing targets, dt = 1 ms (MPTC) / 5 ms (ID)."""
    import numpy as np
    import torch
    from quadruped_drake_amd import IDController, MPTCController, workloads
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    n = shard["n"]
    cls, dt = (IDController, 5e-3) if shard["kind"] == "id" else (MPTCController, 1e-3)
    q0, v0 = workloads.nominal_state(shard["model"], n)
    rng = np.random.default_rng(0)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
    st_t = workloads.standing_targets(shard["model"], 1)[:, 0]
    traj = TrunkTrajectory(np.zeros(0), np.zeros((0, 54)), np.zeros(0, np.uint8), wait_time=1e9, device=device,
                           standing_targets=st_t, standing_mask=0b1111, model=shard["model"])
    ctrl = cls(model=shard["model"], max_batch=n, device=device)
    dev = torch.device("cuda", device)
    q = torch.tensor(q0, device=dev); v = torch.tensor(v0, device=dev); t = torch.zeros(n, dtype=torch.float64, device=dev)
    ctrl.rollout(traj, 20, dt, q, v, t); ctrl.sync()
    t0 = time.perf_counter()
    ctrl.rollout(traj, steps, dt, q, v, t); ctrl.sync()
    el = time.perf_counter() - t0
    bad = ctrl.stats()["status_nonzero"]
    ctrl.close()
    return {"ticks_per_s": n * steps / el, "us_per_step": el / steps * 1e6, "steps": steps, "dt": dt, "status_nonzero": bad,
            "scenario": "%d x %s standing, %s, targets from the stored-trajectory lookup, semi-implicit Euler forward step; "
                        "one persistent launch for the whole rollout" % (n, shard["model"], shard["kind"].upper())}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from quadruped_drake_amd import IDController, MPTCController, workloads
    from quadruped_drake_amd import stats as wstats

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif a.gpus > 1:
        raise SystemExit("--gpus %d needs the torch.distributed.run launcher (one rank per GPU)" % a.gpus)
    assert world == a.gpus, "WORLD_SIZE must equal --gpus"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = a.config or (3 if world == 1 else 5)
    n_total = a.per_gpu * world
    batch = workloads.make_batch(cfg, n=n_total)
    shard = wstats.shard_batch(batch, rank, world)
    n = shard["n"]
    cls = IDController if shard["kind"] == "id" else MPTCController
    ctrl = cls(model=shard["model"], max_batch=n, device=local)
    ctrl.set_variant(a.variant)
    up = lambda x: None if x is None else torch.tensor(x, device=dev)
    q, v, tg, mask, mu, ms = (up(shard[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale"))
    out = (torch.empty((12, n), dtype=torch.float64, device=dev), torch.empty((4, n), dtype=torch.float64, device=dev),
           torch.empty((n,), dtype=torch.int32, device=dev))

    for _ in range(a.warmup):
        ctrl.step(q, v, tg, mask, mu, ms, out=out)
    wstats.all_reduce_stats(ctrl.stats(), device=dev)            # warm the statistics exchange (RCCL channel set-up) as well
    ctrl.stats(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # exactly K steps; HIP events on the launch stream bracket the same K launches
    ms_per_launch, _ = ctrl.time_steps(a.steps, q, v, tg, mask, mu, ms, out=out)
    st = wstats.all_reduce_stats(ctrl.stats(), device=dev)      # end-of-rollout statistics (RCCL when world > 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt, ms_per_launch], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, ms_per_launch = float(tt[0]), float(tt[1])

    if rank == 0:
        status = out[2].cpu().numpy()
        used = ctrl.variant_for(n)            # "auto" resolves by batch size (include/wbc.h)
        key = (shard["kind"], cfg)
        flops = FLOPS_PER_TICK.get(key, 37629.0)
        bpt = BYTES_PER_TICK[mu is not None]
        sec = ms_per_launch * 1e-3
        achieved = flops * n / sec / 1e12
        # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (separate runs; committed under profiles/): null when no measurement matches this build
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                tr = json.load(f)
            ent = tr.get("%s_cfg%d_n%d_%s" % (shard["kind"], cfg, n, used))
            if ent:
                traffic = ent["bytes_per_launch"]
        except (OSError, ValueError):
            pass
        line = {
            "metric": "whole-body-QP control ticks/s at N=4096 Mini Cheetah",
            "value": n_total * a.steps / dt, "unit": "ticks/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %d x %s, %s controller, %s" % (
                cfg - 1, n_total, shard["model"], shard["kind"].upper(),
                "trot contact masks" if cfg != 2 else "4-contact stand"),
                "instances_per_gpu": n, "seed": shard["seed"],
                "domain_randomised": mu is not None, "parallelism": "batch-shard x%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
                         "kernel": "%s<%s>" % ({"lane": "wbc_tick_kernel", "hex": "wbc_hex_kernel", "quad": "wbc_quad_kernel"}[used], shard["kind"].upper()),
                         "kernel_ms": ms_per_launch, "flops_per_tick": flops,
                         "hbm": {"achieved_GBs": bpt * n / sec / 1e9, "peak_GBs": PEAK_HBM_GBS,
                                 "frac": bpt * n / sec / 1e9 / PEAK_HBM_GBS, "bytes_per_tick": bpt},
                         "bound_detail": "FP64 compute: priced against the dense FP64 peak (78.6 TFLOP/s, the same figure for the "
                                         "FP64 MFMA and the FP64 vector ALU); the kernel issues vector FMAs, not MFMA (DESIGN.md section 5)",
                         "note": "HBM does not bind this path (SURVEY 8d): 864 algorithmic bytes per 37.6 kflop tick; HBM fraction stated beside it"},
            "status_nonzero": int((status != 0).sum()),
            "rollout_stats": {k: st[k] for k in ("ticks", "status_nonzero", "iters_sum", "tau_abs_max")},
            "kernel_info": ctrl.kernel_info(),
        }
        if world == 1 and not a.no_cpu_baseline:
            line["closed_loop"] = closed_loop(shard, local)          # informational, outside the timed region
            line["cpu_baseline"] = cpu_baseline(batch, a.cpu_seconds)
            tau_gpu = out[0][:, :256].cpu().numpy()
            from oracle import oracle_py as orc
            tau_o, _, _ = orc.step_batch(shard["kind"], orc.model(shard["model"]), orc.params(shard["kind"]),
                                         batch["q"][:, :256], batch["v"][:, :256], batch["targets"][:, :256],
                                         batch["mask"][:256], None if batch["mu"] is None else batch["mu"][:256],
                                         None if batch["mass_scale"] is None else batch["mass_scale"][:256])
            rel = np.abs(tau_gpu - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)
            line["torque_rel_err_vs_cpu_ref"] = float(rel.max())
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
