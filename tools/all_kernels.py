#!/usr/bin/env python3
"""Every product kernel under rocprofv3 (GPU box).  One workload per profiler run -- the laws, robots and options share
kernel TEMPLATES, so only separate runs keep their launch times apart:

    cd /tmp; export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d $OUT/<workload> -o run --output-format csv -- python3 $ROOT/tools/all_kernels.py <workload> $OUT
    python3 tools/all_kernels.py --report $OUT > $OUT/all_kernels.md            (tools/profile_round.sh does both)

Each run ramps the clock for 1 s, then launches its kernel 300 times (the persistent rollout kernel: 3 launches of 300
closed-loop steps) and leaves <workload>.json (registers, LDS, scratch, instances, counted flops per tick) beside the
profiler's kernel_stats.csv.  The report recomputes the FP64 fraction from the profiler's AVERAGE duration:
counted flops per tick x instances / average ns / 78.6 TFLOP/s."""
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 78.6e12
# counted flops per tick (operation-counting scalar, tools/host_tick.cpp): the four frozen figures of BASELINE.md section 4
# plus PC / CLF / ID-on-trot-states counted the same way on 512 instances of config 3
FLOPS = {"mptc3": 37629.0, "id2": 35667.0, "anymal4": 36085.0, "rand5": 37703.0, "rand5_32768": 37703.0, "tb_mptc3": 37629.0,
         "pc3": 38223.0, "clf3": 27811.0, "id3": 25218.0, "rollout_mptc": 37629.0, "rollout_id": 35667.0}
WORKLOADS = ["mptc3", "id2", "pc3", "clf3", "id3", "anymal4", "rand5", "rand5_32768", "tb_mptc3", "rollout_mptc", "rollout_id",
             "lookup", "integrate"]


def run(name, outdir):
    import numpy as np
    import torch
    from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController, workloads
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    dev = "cuda:0"
    up = lambda x: None if x is None else torch.tensor(x, device=dev)
    info = {"workload": name}
    spec = {"mptc3": (MPTCController, 3, 4096, {}), "id2": (IDController, 2, 4096, {}), "pc3": (PCController, 3, 4096, {}),
            "clf3": (CLFController, 3, 4096, {}), "id3": (IDController, 3, 4096, {}), "anymal4": (MPTCController, 4, 4096, {}),
            "rand5": (MPTCController, 5, 4096, {}), "rand5_32768": (MPTCController, 5, 32768, {}),
            "tb_mptc3": (MPTCController, 3, 4096, {"tau_max": 12.0})}
    if name in spec:
        cls, cfg, n, prm = spec[name]
        b = workloads.make_batch(cfg, n=n)
        c = cls(model=b["model"], max_batch=n, device=0, params=prm or None)
        args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
        out = (torch.empty((12, n), dtype=torch.float64, device=dev), torch.empty((4, n), dtype=torch.float64, device=dev),
               torch.empty((n,), dtype=torch.int32, device=dev))
        t0 = time.time()
        while time.time() - t0 < 1.0:
            c.time_steps(100, *args, out=out)
        c.stats(reset=True)
        ms, _ = c.time_steps(300, *args, out=out)
        st = c.stats()
        info.update(n=n, cfg=cfg, law=cls.__name__, hip_event_us=ms * 1e3, iters_per_tick=st["iters_sum"] / st["ticks"],
                    status_nonzero=st["status_nonzero"], kernel_info=c.kernel_info(), match="wbc_hex_kernel")
        c.close()
    elif name.startswith("rollout"):
        n, steps = 4096, 300
        cls, dt = (MPTCController, 1e-3) if name.endswith("mptc") else (IDController, 5e-3)
        q0, v0 = workloads.nominal_state("mini_cheetah", n)
        rng = np.random.default_rng(0)
        q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
        st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
        traj = TrunkTrajectory(np.zeros(0), np.zeros((0, 54)), np.zeros(0, np.uint8), wait_time=1e9, device=0,
                               standing_targets=st_t, standing_mask=0b1111)
        c = cls(max_batch=n, device=0)
        q, v, t = up(q0), up(v0), torch.zeros(n, dtype=torch.float64, device=dev)
        for _ in range(4):
            c.rollout(traj, steps, dt, q, v, t)
        c.sync()
        info.update(n=n, law=cls.__name__, steps_per_launch=steps, kernel_info=c.kernel_info(rollout=True), match="wbc_hex_rollout_kernel")
        c.close()
    elif name == "lookup":
        n, K = 4096, 5001
        ts = np.arange(K) * 1e-3
        tg = np.random.default_rng(0).normal(size=(K, 54))
        traj = TrunkTrajectory(ts, tg, np.full(K, 9, np.uint8), wait_time=1.0, device=0)
        t = up(np.random.default_rng(1).uniform(0.0, 6.0, n))
        for _ in range(300):
            traj.lookup(t)
        torch.cuda.synchronize()
        info.update(n=n, samples=K, match="traj_lookup_kernel", bytes_per_instance=54 * 8 + 8 + 1)
    elif name == "integrate":
        n = 4096
        b = workloads.make_batch(3, n=n)
        c = MPTCController(max_batch=n, device=0)
        q, v, vd = up(b["q"]), up(b["v"]), torch.zeros((18, n), dtype=torch.float64, device=dev)
        for _ in range(300):
            c.integrate(q, v, vd, 1e-3)
        c.sync()
        info.update(n=n, match="wbc_integrate_kernel", bytes_per_instance=(19 + 18) * 8 * 2 + 18 * 8)
        c.close()
    else:
        raise SystemExit("unknown workload " + name)
    os.makedirs(outdir, exist_ok=True)
    json.dump(info, open(os.path.join(outdir, name + ".json"), "w"), indent=1)


def report(outdir):
    import csv
    rnd = "".join(c for c in os.path.basename(os.path.dirname(os.path.abspath(outdir))) if c.isdigit()) or "?"   # gpurun_out/r04/all -> 4
    print("# Round %s: every product kernel under `rocprofv3 --kernel-trace --stats` (MI355X, one profiler run per workload, "
          "`tools/profile_round.sh`)\n" % rnd.lstrip("0"))
    print("Average duration = the profiler's `AverageNs` over the run's launches (1 s of clock-ramp launches + 300 timed ones); fraction = "
          "counted flops per tick x instances / average duration / 78.6 TFLOP/s (FP64 vector peak).  The HIP-event column is the last 300 "
          "launches timed by `wbc_time_steps` inside the same PROFILED process: launch-to-launch time, i.e. the kernel plus the gap the "
          "profiler's per-dispatch interception leaves between two launches (an upper bound: 0 - 10 % above the kernel trace depending "
          "on the box; outside the profiler `bench.py`'s HIP events and the trace agree within 1 %, `hex_bench.json` / `kernel_stats.csv`).  "
          "Registers / LDS / scratch: of the kernel in the row (`wbc_kernel_info`, `wbc_rollout_kernel_info`).\n")
    print("| workload | kernel | N | calls | rocprofv3 avg µs | min µs | HIP-event µs | iters/tick | VGPR+AGPR | LDS B | scratch B/lane | flops/tick | TFLOP/s | frac of FP64 peak |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name in WORKLOADS:
        jf = os.path.join(outdir, name + ".json")
        files = glob.glob(os.path.join(outdir, name, "**", "*kernel_stats.csv"), recursive=True)
        if not os.path.exists(jf) or not files:
            print("| %s | (no data) |" % name)
            continue
        info = json.load(open(jf))
        rows = [r for r in csv.DictReader(open(files[0])) if info["match"] in r["Name"]]
        if not rows:
            print("| %s | (kernel %s not in the trace) |" % (name, info["match"]))
            continue
        r = max(rows, key=lambda r: int(r["Calls"]))
        avg, mn = float(r["AverageNs"]), float(r["MinNs"])
        import re
        mm = re.search(r"((?:wbc_|traj_)\w+(?:<[^>]*>)?)", r["Name"])
        kname = mm.group(1) if mm else r["Name"][:40]
        ki = info.get("kernel_info", {})
        n = info["n"]
        per = avg / info.get("steps_per_launch", 1)            # the persistent kernel: per closed-loop step
        fl = FLOPS.get(name)
        tf = fl * n / (per * 1e-9) / 1e12 if fl else None
        extra = " (per step of %d)" % info["steps_per_launch"] if "steps_per_launch" in info else ""
        print("| %s | `%s` | %d | %s | %.2f%s | %.2f | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            name, kname, n, r["Calls"], per / 1e3, extra, mn / 1e3 / info.get("steps_per_launch", 1),
            "%.2f" % info["hip_event_us"] if "hip_event_us" in info else "—",
            "%.2f" % info["iters_per_tick"] if "iters_per_tick" in info else "—",
            ki.get("num_regs", "—"), ki.get("lds_bytes", ki.get("static_lds", "—")), ki.get("scratch_bytes_per_lane", "—"),
            "%.0f" % fl if fl else "%d B moved / instance" % info.get("bytes_per_instance", 0),
            "%.2f" % tf if tf else "%.1f GB/s" % (info.get("bytes_per_instance", 0) * n / per),
            "%.1f %%" % (100 * tf * 1e12 / PEAK) if tf else "%.2f %% of HBM" % (100 * info.get("bytes_per_instance", 0) * n / per / 8000.0)))
    print("\nWorkloads: mptc3 = BASELINE configs[2] (headline); id2 = configs[1] at N = 4096; pc3 / clf3 / id3 = the other laws on the "
          "configs[2] states; anymal4 = configs[3]; rand5 = the per-GPU shard of configs[4] (per-instance mu / mass scale), rand5_32768 = all "
          "of it on one GPU; tb_mptc3 = torque box |tau| <= 12 N m; rollout_* = the persistent closed-loop kernel (standing, 300 steps per "
          "launch); lookup = device-side nearest-sample search + 54-double gather over 5001 samples; integrate = the forward step.")


if __name__ == "__main__":
    if sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run(sys.argv[1], sys.argv[2])
