#!/usr/bin/env python3
"""Headline benchmark: whole-body-QP control ticks/s (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (one fused HIP launch = one control tick for every robot of
the rank's shard) over a synthetic randomised-state batch that is resident in HBM before the
timed region starts.  Per-GPU work is fixed at 4096 Mini-Cheetah instances (weak scaling):
  N = 1 : BASELINE.json configs[2] -- 4096 Mini Cheetah, trot contact modes, MPTC controller;
  N > 1 : BASELINE.json configs[4] pattern -- 4096*N instances with per-instance mu / mass scale,
          sharded contiguously, no data-path collective; one RCCL all-gather of the end-of-rollout
          statistics vector closes the timed region.
Launching: under torch.distributed.run the rank environment is used as given.  Plain
`python bench.py --gpus N` (no WORLD_SIZE in the environment) SELF-LAUNCHES: the parent -- which never
imports torch or touches a GPU -- starts N fresh child processes of this file (one rank per GPU,
127.0.0.1 rendezvous), relays rank 0's JSON line and exits non-zero if any child does.
`--backend gloo` is the CPU test switch: without a GPU it runs launch + shard + statistics exchange
only (`"dry_run": true`, no kernel, no throughput claim) -- tests/test_bench_launch_cpu.py.
Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import gc
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Counted with the operation-counting scalar (tools/host_tick.cpp) on the full config; frozen in
# BASELINE.md section 4.  add + mul + div + sqrt each count 1 (an FMA is 2); trig/compare are not counted.
FLOPS_PER_TICK = {("mptc", 3): 37629.0, ("mptc", 5): 37703.0, ("id", 2): 35667.0, ("mptc", 4): 36085.0}
BYTES_PER_TICK = {False: 864.0, True: 880.0}     # SURVEY.md section 8(d); True = with mu and mass scale
PEAK_FP64_VALU_TFLOPS = 78.6                      # MI355X vector FP64 (spec); FP64 MFMA peak is the same figure
PEAK_HBM_GBS = 8000.0
PMC_ROUND = "r06"                                  # profiles/<round>/hex_pmc.json: the committed counter pass roofline.issued restates


# what the identity of a kernel build covers: an explicit list (a stray backup file under csrc/ does not change it)
KERNEL_SOURCES = ["quadruped_drake_amd/csrc/hipcc_flags.txt", "quadruped_drake_amd/csrc/wbc_hex.hpp", "quadruped_drake_amd/csrc/wbc_kernels.hip",
                  "quadruped_drake_amd/csrc/wbc_model.hpp", "quadruped_drake_amd/csrc/wbc_tick.hpp", "quadruped_drake_amd/csrc/wbc_traj.hip",
                  "quadruped_drake_amd/csrc/wbc_traj_dev.hpp", "quadruped_drake_amd/csrc/wbc_device_guard.hpp", "include/wbc.h", "include/wbc_extras.h"]


def kernel_src_sha16():
    """Identity of the kernel sources and their code-generation flags (KERNEL_SOURCES: the files of csrc/, hipcc_flags.txt included, + the
    C ABI headers): committed counter files carry it, and a counter file that was collected on another build is not mixed with
    this build's launch time (roofline.issued becomes null instead)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def fnv1a64(data):
    """FNV-1a, 64 bit, over a bytes object (examples/wbc_host.cpp: fnv1a64)."""
    h = 1469598103934665603
    for c in data:
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--per-gpu", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--config", type=int, default=0, help="override workload config (2,3,4,5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", default="auto", choices=["auto", "hex"], help="kernel variant (one product family)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work (seconds) for the baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo = CPU test switch (dry run without a GPU)")
    ap.add_argument("--dry-run", action="store_true", help="launch + shard + statistics exchange only, no kernel")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test switch (with --backend gloo): every rank computes on GPU 0 -- exercises the self-launched N > 1 path on a one-GPU box")
    ap.add_argument("--launch-timeout", type=float, default=1800.0, help="self-launch: seconds before the children are stopped")
    ap.add_argument("--ramp-seconds", type=float, default=1.0,
                    help="untimed launches before the W warm-up steps until the GPU holds its sustained clock (DVFS ramp)")
    ap.add_argument("--force-pg", action="store_true",
                    help="initialise the process group (RCCL with --backend nccl) even for ONE rank: the collective path of the "
                         "multi-GPU run exercised on a one-GPU box")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ self-launch
def self_launch(a):
    """Parent of a plain `python bench.py --gpus N`: N fresh rank processes, no GPU touched here."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0f = tempfile.TemporaryFile(mode="w+")   # rank 0's stdout: a file, not a pipe -- library chatter (RCCL / NCCL_DEBUG, HIP
    for r in range(a.gpus):                     # notices) of any size can never block rank 0 in write() while the parent polls
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0f if r == 0 else sys.stderr))
    deadline = time.time() + a.launch_timeout
    rc = 0
    # poll every child: one rank dying must not leave the others waiting at the rendezvous until the timeout
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
            rc = 124 if time.time() > deadline else 1
            break
        time.sleep(0.1)
    for r, p in enumerate(procs):
        if p.poll() is None:            # stop exactly the processes started here
            p.kill()
    for r, p in enumerate(procs):
        p.wait()
        if p.returncode != 0:
            sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, p.returncode))
            rc = rc or (p.returncode if p.returncode > 0 else 1)
    out0f.seek(0)
    out0 = out0f.read()
    out0f.close()
    sys.stdout.write(out0)
    sys.stdout.flush()
    if rc == 0 and not any(l.startswith("{") for l in out0.splitlines()):
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_limits():
    """Host CPU resources as this process sees them: affinity, cgroup quota (cpu.max), usable threads."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    raw = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                raw = f.read().strip()
            if path.endswith("cpu.max"):
                q, per = raw.split()
                quota = None if q == "max" else float(q) / float(per)
            else:
                q = float(raw)
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = float(f.read().strip())
                quota = None if q <= 0 else q / per
            break
        except (OSError, ValueError):
            continue
    usable = aff if quota is None else max(1, min(aff, int(quota + 0.999)))
    return {"nproc": os.cpu_count(), "affinity": aff, "cgroup_cpu_max": raw, "cgroup_quota_cpus": quota, "usable": usable}


def cpu_baseline(batch, seconds):
    """Oracle (CPU restatement of the Drake+OSQP path, kind "port") on this box's host cores."""
    import numpy as np
    from oracle import oracle_py as orc
    from quadruped_drake_amd import workloads
    lim = cpu_limits()
    cores = lim["usable"]
    m = orc.model(batch["model"]); p = orc.params(batch["kind"])
    n = batch["n"]
    sl = lambda a, k: None if a is None else (a[:, :k] if a.ndim == 2 else a[:k])
    args = lambda k: (sl(batch["q"], k), sl(batch["v"], k), sl(batch["targets"], k), sl(batch["mask"], k),
                      sl(batch["mu"], k), sl(batch["mass_scale"], k))
    n0 = min(512, n)
    orc.bench_batch(batch["kind"], m, p, *args(64), nthreads=1, reps=1)      # page the library in
    t = time.perf_counter()
    orc.bench_batch(batch["kind"], m, p, *args(n0), nthreads=1, reps=1)
    rate1 = n0 / (time.perf_counter() - t)

    def timed(threads, ticks):
        reps = max(1, int(round(ticks / n)))
        t0 = time.perf_counter()
        orc.bench_batch(batch["kind"], m, p, *args(n), nthreads=threads, reps=reps)
        return n * reps / (time.perf_counter() - t0), reps

    # thread-scaling curve (about 1.5 s each): shows whether the lease really has the cores its affinity mask lists
    curve = {}
    for th in sorted({1, 2, 4, 8, 64, cores, lim["affinity"]}):
        if th > lim["affinity"]:
            continue
        r, _ = timed(th, rate1 * 1.5 * min(th, 8))
        curve[str(th)] = r
    best_threads = max(curve, key=lambda k: curve[k])
    # the reported baseline: every usable core, ~`seconds` of wall time
    use = int(best_threads) if curve[best_threads] > 1.15 * curve.get(str(cores), 0.0) else cores
    rate, reps = timed(use, curve[str(use)] * seconds if str(use) in curve else rate1 * cores * seconds)
    # BASELINE config 1 restated: N = 1, ID law, q0 of simulate.py:171-176, standing targets,
    # 1200 sequential ticks (6 s at dt = 5 ms, simulate.py:20-22) on ONE core
    q0, v0 = workloads.nominal_state("mini_cheetah", 1)
    tg0 = workloads.standing_targets("mini_cheetah", 1)
    mk0 = np.array([0b1111], dtype=np.uint8)
    mid = orc.model("mini_cheetah"); pid = orc.params("id")
    orc.bench_batch("id", mid, pid, q0, v0, tg0, mk0, nthreads=1, reps=10)
    t0 = time.perf_counter()
    orc.bench_batch("id", mid, pid, q0, v0, tg0, mk0, nthreads=1, reps=1200)
    c1 = time.perf_counter() - t0
    return {"value": rate, "unit": "ticks/s", "cores": min(use, cores), "threads": use, "kind": "port",
            "reduced": cpu_baseline_reduced(batch, use, max(2.0, seconds / 4.0)),
            "single_core_ticks_per_s": rate1, "thread_scaling_ticks_per_s": curve, "host": lim,
            "build": "gcc -O3 -march=x86-64-v3 -fopenmp -ffp-contract=off (oracle/Makefile)",
            "config1": {"workload": "BASELINE configs[0] restated: 1 Mini Cheetah, ID law, q0 of simulate.py:171-176, "
                                    "standing targets, 1200 sequential ticks, 1 core", "seconds": c1,
                        "ticks_per_s": 1200.0 / c1, "realtime_factor_at_200Hz": (1200.0 / c1) / 200.0},
            "sample": "the same %d-instance batch x %d passes on %d threads (one OpenMP region, dynamic schedule); C oracle = "
                      "dense restatement of the Drake+OSQP tick, NOT Drake+OSQP" % (n, reps, use)}


def cpu_baseline_reduced(batch, threads, seconds):
    """CONTEXT beside the literal port: the REDUCED algorithm the kernels run (12-variable QP, QR + Goldfarb-Idnani; the scalar one-robot host
    instantiation of tools/wbc_scalar_tick.hpp through tools/libhost_tick.so), same batch, same thread count.  4 - 5 x fewer flops than the dense
    restatement of the Drake + OSQP tick: the honest CPU figure for the algorithm the GPU runs -- never the thing measured or shipped, never the target."""
    import ctypes as C
    import numpy as np
    so = os.path.join(ROOT, "tools", "libhost_tick.so")
    try:
        L = C.CDLL(so)
        fn = L.host_tick_bench
    except (OSError, AttributeError) as e:
        return {"value": None, "why": "tools/libhost_tick.so not built (%s)" % e}
    from oracle import oracle_py as orc
    n = batch["n"]
    dp = C.POINTER(C.c_double)
    flat = np.ascontiguousarray(orc.load_model_json(batch["model"])["flat"], dtype=np.float64)
    q, v, tg = (np.ascontiguousarray(batch[k], dtype=np.float64) for k in ("q", "v", "targets"))
    mask = np.ascontiguousarray(batch["mask"], dtype=np.uint8)
    mu = None if batch["mu"] is None else np.ascontiguousarray(batch["mu"], dtype=np.float64)
    ms = None if batch["mass_scale"] is None else np.ascontiguousarray(batch["mass_scale"], dtype=np.float64)
    tau = np.zeros((12, n)); st = np.zeros(n, np.int32)
    kind = {"id": 0, "mptc": 1, "pc": 2, "clf": 3}[batch["kind"]]
    P = lambda a: None if a is None else a.ctypes.data_as(dp)

    def timed(th, reps):
        t0 = time.perf_counter()
        rc = fn(kind, P(flat), n, n, P(q), P(v), P(tg), mask.ctypes.data_as(C.POINTER(C.c_ubyte)), P(mu), P(ms), P(tau), st.ctypes.data_as(C.POINTER(C.c_int)),
                int(th), int(reps))
        if rc != 0:
            raise RuntimeError("host_tick_bench rc %d" % rc)
        return n * reps / (time.perf_counter() - t0)

    r1 = timed(1, 1)
    rate1 = timed(1, max(1, int(round(r1 * 1.0 / n))))
    reps = max(1, int(round(rate1 * threads * seconds / n)))
    rate = timed(threads, reps)
    return {"value": rate, "unit": "ticks/s", "threads": int(threads), "single_core_ticks_per_s": rate1, "kind": "port-reduced",
            "status_nonzero": int((st != 0).sum()),
            "build": "g++ -O2 -ffp-contract=off (tools/host_tick.cpp, built by build())",
            "sample": "the same %d-instance batch x %d passes on %d std::threads (chunks of 8 from an atomic counter); the kernels' reduced 12-variable "
                      "algorithm as a scalar one-robot host instantiation -- context for the GPU figure, not a baseline to beat" % (n, reps, threads)}


def closed_loop(shard, device, steps=300):
    """Closed-loop rate of the same law (wbc_rollout: lookup -> tick -> forward step, one persistent launch) on a stored TROT
    trajectory (alternating diagonal contact pairs as a TOWR trot streams them, robots at staggered phases; MPTC dt = 1 ms,
    ID dt = 5 ms), and -- `standing` -- on nominal standing states with standing targets (the easiest workload: the friction
    rows stay inactive)."""
    import numpy as np
    import torch
    from quadruped_drake_amd import IDController, MPTCController, workloads
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    n = shard["n"]
    cls, dt = (IDController, 5e-3) if shard["kind"] == "id" else (MPTCController, 1e-3)
    dev = torch.device("cuda", device)
    st_t = workloads.standing_targets(shard["model"], 1)[:, 0]

    def run(traj, q0, v0, t0v):
        ctrl = cls(model=shard["model"], max_batch=n, device=device)
        q = torch.tensor(q0, device=dev); v = torch.tensor(v0, device=dev); t = torch.tensor(t0v, device=dev)
        ctrl.rollout(traj, 20, dt, q, v, t); ctrl.sync()
        ctrl.stats(reset=True)
        t0 = time.perf_counter()
        ctrl.rollout(traj, steps, dt, q, v, t); ctrl.sync()
        el = time.perf_counter() - t0
        s = ctrl.stats()
        ctrl.close()
        return {"ticks_per_s": n * steps / el, "us_per_step": el / steps * 1e6, "steps": steps, "dt": dt,
                "status_nonzero": s["status_nonzero"], "iters_per_tick": s["iters_sum"] / max(1.0, s["ticks"])}

    # trot: 4 s of samples at 1 kHz, contact pair switching every 150 ms, swing feet 2 cm up, body swaying 1 cm
    K = 4000
    ts = np.arange(K) * 1e-3
    tg = np.tile(st_t, (K, 1))
    tg[:, 0] += 0.01 * np.sin(2 * np.pi * ts / 0.3); tg[:, 3] = 0.01 * 2 * np.pi / 0.3 * np.cos(2 * np.pi * ts / 0.3)
    masks = np.where((np.arange(K) // 150) % 2 == 0, 0b1001, 0b0110).astype(np.uint8)
    for f in range(4):
        sw = ((masks >> f) & 1) == 0
        tg[sw, 18 + 9 * f + 2] += 0.02
    rng = np.random.default_rng(1)
    q0, v0 = workloads.nominal_state(shard["model"], n)
    q0[7:] += rng.uniform(-0.03, 0.03, (12, n))
    trot = TrunkTrajectory(ts, tg, masks, wait_time=0.0, device=device, standing_targets=st_t, standing_mask=0b1111, model=shard["model"])
    out = run(trot, q0, v0, rng.uniform(0.0, 0.6, n))
    out["scenario"] = ("%d x %s, %s, stored trot trajectory (diagonal pairs switching every 150 ms, staggered phases), targets from the "
                       "device-side lookup, semi-implicit Euler forward step; one persistent launch for the whole rollout" % (
                           n, shard["model"], shard["kind"].upper()))
    rng = np.random.default_rng(0)
    q0, v0 = workloads.nominal_state(shard["model"], n)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
    stand = TrunkTrajectory(np.zeros(0), np.zeros((0, 54)), np.zeros(0, np.uint8), wait_time=1e9, device=device,
                            standing_targets=st_t, standing_mask=0b1111, model=shard["model"])
    out["standing"] = run(stand, q0, v0, np.zeros(n))
    # a closed loop that sits on its friction limits (body target swaying sideways at 7.9 m/s^2 against mu g = 6.9), cold and with the warm-started
    # active set of wbc_set_warm_start (profiles/r06/warm_start.md): 100 ticks each from the same start -- informational like the rest of this object
    dur = 0.4
    tss = np.arange(int(round(dur / dt)) + 1) * dt
    tgs = workloads.standing_targets(shard["model"], tss.size)
    w = 2 * np.pi * 2.0
    tgs[1] += 0.05 * np.sin(w * tss); tgs[4] = 0.05 * w * np.cos(w * tss); tgs[7] = -0.05 * w * w * np.sin(w * tss)
    sway = TrunkTrajectory(tss, np.ascontiguousarray(tgs.T), np.full(tss.size, 0b1111, np.uint8), wait_time=0.0, device=device, model=shard["model"])
    t0s = np.random.default_rng(5).uniform(0.0, 0.2, n)
    sat = {}
    for name, warm in (("cold", False), ("warm_start", True)):
        ctrl = cls(model=shard["model"], max_batch=n, device=device)
        ctrl.set_warm_start(warm)
        q = torch.tensor(q0, device=dev); v = torch.tensor(v0, device=dev); t = torch.tensor(t0s, device=dev)
        ctrl.rollout(sway, 1, dt, q.clone(), v.clone(), t.clone()); ctrl.sync()
        ctrl.stats(reset=True)
        ta = time.perf_counter()
        ctrl.rollout(sway, 100, dt, q, v, t); ctrl.sync()
        el = time.perf_counter() - ta
        s = ctrl.stats()
        ctrl.close()
        sat[name] = {"us_per_step": el / 100 * 1e6, "iters_per_tick": s["iters_sum"] / max(1.0, s["ticks"]), "status_nonzero": s["status_nonzero"]}
    sat["scenario"] = "sideways sway at 2 Hz, 5 cm (saturated part of every period), 100 closed-loop ticks, dt %g; warm_start = wbc_set_warm_start(1), off by default" % dt
    out["saturated"] = sat
    return out


def config1_gpu(device, ticks=1200):
    """BASELINE configs[0] on the PRODUCT path: ONE Mini Cheetah, ID law, q0 of simulate.py:171-176 with the standing targets
    of planners/simple.py:45-85, 1200 sequential ticks (6 s at dt = 5 ms, simulate.py:20-22), each tick waited for like the
    simulator's control loop does:  (i) the host-pointer handle make_leaf_system uses (H2D copies + launch + D2H + wait per
    tick);  (ii) a device-pointer handle with the buffers bound once (launch + wait per tick)."""
    import numpy as np
    import torch
    from quadruped_drake_amd import IDController, workloads
    q0, v0 = workloads.nominal_state("mini_cheetah", 1)
    tg0 = workloads.standing_targets("mini_cheetah", 1)
    mk0 = np.array([0b1111], dtype=np.uint8)
    res = {"workload": "BASELINE configs[0] on the product path: 1 Mini Cheetah, ID law, q0 of simulate.py:171-176, standing targets, "
                       "%d sequential ticks, every tick waited for" % ticks}
    c = IDController(max_batch=1, device=device, host_ptrs=True)
    for _ in range(50):
        tau, met, st = c.step(q0, v0, tg0, mk0); c.sync()
    t0 = time.perf_counter()
    for _ in range(ticks):
        tau, met, st = c.step(q0, v0, tg0, mk0); c.sync()
    el = time.perf_counter() - t0
    c.close()
    res["host_pointer_handle"] = {"us_per_tick": el / ticks * 1e6, "ticks_per_s": ticks / el, "realtime_factor_at_200Hz": ticks / el / 200.0,
                                  "status": int(st[0]), "what": "wbc_step on a WBC_HOST_PTRS handle + wbc_sync (the LeafSystem adapter's path)"}
    dev = torch.device("cuda", device)
    c = IDController(max_batch=1, device=device)
    up = lambda x: torch.tensor(x, device=dev)
    out = (torch.empty((12, 1), dtype=torch.float64, device=dev), torch.empty((4, 1), dtype=torch.float64, device=dev),
           torch.empty((1,), dtype=torch.int32, device=dev))
    bound = c.bind(up(q0), up(v0), up(tg0), up(mk0), out=out)
    for _ in range(50):
        bound.step(); c.sync()
    t0 = time.perf_counter()
    for _ in range(ticks):
        bound.step(); c.sync()
    el = time.perf_counter() - t0
    res["device_pointer_handle_bound"] = {"us_per_tick": el / ticks * 1e6, "ticks_per_s": ticks / el,
                                          "realtime_factor_at_200Hz": ticks / el / 200.0, "status": int(out[2][0]),
                                          "what": "bind() once, then wbc_step + wbc_sync per tick on device-resident buffers"}
    c.close()
    return res


def large_batch(device, steps=40):
    """The single-GPU size of BASELINE configs[4] (N = 32768, per-instance mu / mass scale) on this one GPU:
    the large-batch roofline fraction, driver-observed (informational; `value` stays the N = 4096 line)."""
    import torch
    from quadruped_drake_amd import MPTCController, workloads
    n = 32768
    b = workloads.make_batch(5, n=n)
    dev = torch.device("cuda", device)
    ctrl = MPTCController(model=b["model"], max_batch=n, device=device)
    up = lambda x: torch.tensor(x, device=dev)
    args = [up(b[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale")]
    out = (torch.empty((12, n), dtype=torch.float64, device=dev), torch.empty((4, n), dtype=torch.float64, device=dev),
           torch.empty((n,), dtype=torch.int32, device=dev))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:          # the clock ramp of the main line, for this batch size
        ctrl.time_steps(20, *args, out=out)
    ms, _ = ctrl.time_steps(steps, *args, out=out)
    bad = int((out[2] != 0).sum())
    ctrl.close()
    fl = FLOPS_PER_TICK[("mptc", 5)]
    ach = fl * n / (ms * 1e-3) / 1e12
    return {"workload": "32768 x mini_cheetah, MPTC, trot masks, per-instance mu / mass scale, ONE GPU", "steps": steps,
            "kernel_ms": ms, "ticks_per_s": n / (ms * 1e-3), "achieved_TFLOPs": ach, "frac": ach / PEAK_FP64_VALU_TFLOPS,
            "status_nonzero": bad}


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(a):
    import numpy as np
    import torch
    import torch.distributed as dist
    from quadruped_drake_amd import workloads
    from quadruped_drake_amd import stats as wstats

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.share_gpu:
        assert a.backend == "gloo", "--share-gpu is a gloo test switch (RCCL refuses two ranks on one device)"
        local = 0
    assert world == a.gpus, "WORLD_SIZE must equal --gpus"
    have_gpu = torch.cuda.is_available()
    dry = a.dry_run or (a.backend == "gloo" and not have_gpu)
    if not dry and not have_gpu:
        raise SystemExit("bench.py: no GPU visible -- the hot path has no CPU fallback (use --backend gloo for the dry-run test switch)")
    use_pg = world > 1 or a.force_pg
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:      # --force-pg without a launcher: a one-rank group on a free local port
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if a.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))     # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local) if (have_gpu and not dry) else torch.device("cpu")
    if dev.type == "cuda":
        torch.cuda.set_device(local)
    cdev = dev if a.backend == "nccl" else torch.device("cpu")       # where the collective's tensors live

    cfg = a.config or (3 if world == 1 else 5)
    n_total = a.per_gpu * world
    # every rank keeps only its own contiguous window of the batch (same seeded stream as the whole batch: workloads.make_batch)
    shard = workloads.make_batch(cfg, n=n_total, window=wstats.shard_range(n_total, rank, world))
    batch = shard if world == 1 else None
    n = shard["n"]

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if use_pg:
            dist.barrier()

    if dry:
        # launch + shard + ONE statistics exchange, no kernel: what the CPU test can exercise
        local_stats = dict(ticks=float(n * a.steps), status_nonzero=0.0, iters_sum=0.0, tau_abs_sum=0.0, tau_abs_max=0.0,
                           err_sum=0.0, mask_count=[float((shard["mask"] == k).sum()) * a.steps for k in range(16)])
        barrier()
        t0 = time.perf_counter()
        st, per_rank, seen = wstats.all_gather_stats(local_stats, device=cdev)
        barrier()
        dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({
                "metric": "whole-body-QP control ticks/s at N=4096 Mini Cheetah", "value": 0.0, "unit": "ticks/s",
                "dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "backend": a.backend, "ranks_seen": seen, "per_rank_ticks": [r["ticks"] for r in per_rank],
                "config": {"workload": "DRY RUN (no GPU): launch, shard and statistics exchange of BASELINE configs[%d]" % (cfg - 1),
                           "instances_per_gpu": n, "parallelism": "batch-shard x%d" % world},
                "rollout_stats": {k: st[k] for k in ("ticks", "status_nonzero", "iters_sum", "tau_abs_max")},
                "exchange_seconds": dt}))
        if use_pg:
            dist.barrier()
            dist.destroy_process_group()
        return

    from quadruped_drake_amd import IDController, MPTCController
    cls = IDController if shard["kind"] == "id" else MPTCController
    ctrl = cls(model=shard["model"], max_batch=n, device=local)
    ctrl.set_variant(a.variant)
    up = lambda x: None if x is None else torch.tensor(x, device=dev)
    q, v, tg, mask, mu, ms = (up(shard[k]) for k in ("q", "v", "targets", "mask", "mu", "mass_scale"))
    out = (torch.empty((12, n), dtype=torch.float64, device=dev), torch.empty((4, n), dtype=torch.float64, device=dev),
           torch.empty((n,), dtype=torch.int32, device=dev))

    # cold figure first: W warm-up steps, then K timed steps BEFORE any clock ramp (what a process that ticks only now and
    # then would see); reported as value_cold beside value
    bound = ctrl.bind(q, v, tg, mask, mu, ms, out=out)          # tensors validated once: the loops below are the C ABI only
    for _ in range(a.warmup):
        bound.step()
    sync(); barrier(); sync()
    t0c = time.perf_counter()
    bound.time_steps(a.steps, wait=False)
    ctrl.stats()
    ms_cold = bound.time_steps_result()
    sync(); barrier(); sync()
    dt_cold = time.perf_counter() - t0c

    def rollout(steps):
        """K launches + the two events queued, the end-of-rollout statistics (reduction kernel, the ONE host wait, the gather: RCCL when world > 1),
        the closing bracket.  Host time stamps (rank-local): start, launches queued, statistics here, gathered, bracket closed.  Nothing else is inside:
        the events are read and the dictionaries built after the bracket has closed."""
        ts = [time.perf_counter()]
        bound.time_steps(steps, wait=False)        # exactly `steps` launches; HIP events on the launch stream bracket the same launches
        ts.append(time.perf_counter())
        local_st = ctrl.stats()
        ts.append(time.perf_counter())
        red = wstats.all_gather_stats(local_st, device=cdev) if use_pg else None
        ts.append(time.perf_counter())
        sync(); barrier(); sync()
        ts.append(time.perf_counter())
        ms_launch = bound.time_steps_result()      # the events completed before the statistics did: no wait here
        if red is None:
            red = (dict(local_st), [dict(local_st)], 1)
        return ts, ms_launch, red

    # A REHEARSAL of the timed region -- the same calls in the same order, W steps -- BEFORE the clock ramp, so that the region itself meets warm code
    # paths (interpreter, ctypes thunks, the runtime's wait path) and a connected statistics exchange (RCCL channel set-up): with the driver's K = 20 a
    # region entered for the first time spent 67 - 78 us outside its launches, a rehearsed one 35 - 40 (`region_us` below says where).  Before the
    # ramp, not after it: the GPU's clock governor answers a few hundred idle microseconds after a burst with a lower clock (tools/lab/r05/k20_probe.py:
    # 23.2 -> 25.5 us per launch), which a K = 20 region never recovers from; between the ramp and t0 lie only the W warm-up steps, an asynchronous
    # statistics reset and the bracket's own waits.
    rollout(max(a.warmup, 1))
    # clock ramp: the GPU raises its clock over the first ~1 s of sustained load (measured: 29.4 -> 26.7 us per launch,
    # profiles/r02/tail_experiment.md); steady state is what a control loop sees, so the ramp is not part of the W + K steps
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < a.ramp_seconds:
        bound.time_steps(100)
    for _ in range(a.warmup):
        bound.step()
    ctrl.stats_reset()                                            # queued, not waited for: the statistics cover exactly the K timed steps
    gc_was = gc.isenabled()
    gc.disable()                                                  # no collector pause inside a 0.5 ms region (what timeit does)
    sync(); barrier(); sync()
    ts, ms_per_launch, (st, per_rank, seen) = rollout(a.steps)
    if gc_was:
        gc.enable()
    t0 = ts[0]
    dt = ts[4] - t0
    # where this rank's region went (host clock, microseconds): queueing the K launches | the statistics reduction and the ONE wait | the gather | the closing bracket
    region_us = [round((ts[i + 1] - ts[i]) * 1e6, 1) for i in range(4)]
    per_rank_kernel_ms = [ms_per_launch]
    # parity on EVERY rank: 64 instances spread over this rank's shard against the oracle (the checker), status of the whole shard
    par_rel, par_bad = 0.0, float((out[2] != 0).sum())        # par_bad: the DEVICE's count over the whole shard, nothing else
    par_obad, par_mis = 0.0, 0.0                                # oracle's non-zero statuses on the sample / sample instances whose status differs
    if not a.no_cpu_baseline:
        from oracle import oracle_py as orc
        idx = np.unique(np.linspace(0, n - 1, min(64, n)).astype(int))
        sl = lambda x: None if x is None else (x[:, idx] if x.ndim == 2 else x[idx])
        tau_o, _, st_o = orc.step_batch(shard["kind"], orc.model(shard["model"]), orc.params(shard["kind"]), sl(shard["q"]), sl(shard["v"]),
                                        sl(shard["targets"]), sl(shard["mask"]), sl(shard["mu"]), sl(shard["mass_scale"]))
        tau_g = out[0][:, torch.tensor(idx, device=dev)].cpu().numpy()
        par_rel = float((np.abs(tau_g - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)).max())
        st_g = out[2][torch.tensor(idx, device=dev)].cpu().numpy()
        par_obad = float((st_o != 0).sum())
        par_mis = float((st_g != st_o).sum())
    # checksum of this rank's torques of the last launch: FNV-1a 64 over the [12][n] doubles -- what examples/wbc_host.cpp prints for
    # the same shard (two hosts, one library: the same bits)
    tau_bytes = out[0].cpu().numpy().tobytes()
    tau_hash = fnv1a64(tau_bytes)
    per_rank_rel, bad_total, obad_total, mis_total = [par_rel], par_bad, par_obad, par_mis
    per_rank_hash = ["%016x" % tau_hash]
    if use_pg:
        tt = torch.tensor([dt, ms_per_launch, dt_cold, ms_cold, par_rel, par_bad, par_obad, par_mis, float(tau_hash >> 32), float(tau_hash & 0xffffffff)],
                          dtype=torch.float64, device=cdev)
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        allr = torch.stack(parts).cpu().numpy()
        per_rank_kernel_ms = [float(x) for x in allr[:, 1]]
        per_rank_rel = [float(x) for x in allr[:, 4]]
        bad_total, obad_total, mis_total = (float(allr[:, c].sum()) for c in (5, 6, 7))
        per_rank_hash = ["%016x" % ((int(allr[r, 8]) << 32) | int(allr[r, 9])) for r in range(world)]
        dt, ms_per_launch, dt_cold, ms_cold = (float(x) for x in allr[:, :4].max(0))     # MAX over ranks
    # per-launch distribution (outside the timed region): one HIP event between every two launches
    each, _ = ctrl.time_steps_each(a.steps, q, v, tg, mask, mu, ms, out=out)

    if rank == 0:
        status = out[2].cpu().numpy()
        used = ctrl.variant_for(n)
        key = (shard["kind"], cfg)
        flops = FLOPS_PER_TICK.get(key, 37629.0)
        bpt = BYTES_PER_TICK[mu is not None]
        sec = ms_per_launch * 1e-3
        achieved = flops * n / sec / 1e12
        # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (separate runs; committed under profiles/): null when no measurement matches this build
        traffic, traffic_source = None, "none: profiles/hbm_traffic.json holds no PMC pass of this kernel build"
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                tr = json.load(f)
            ent = tr.get("%s_cfg%d_n%d_%s" % (shard["kind"], cfg, n, used))
            if ent and ent.get("kernel_src_sha16") == kernel_src_sha16():     # counters of another kernel build are not mixed in
                traffic = ent["bytes_per_launch"]
                # REPLAYED, not measured in this run: the builder's committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs of this
                # command on another box), admitted only for the very kernel sources this process runs
                traffic_source = "replayed: profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, %s; kernel sources sha16 %s)" % (
                    ent.get("source", "committed by the builder"), ent["kernel_src_sha16"])
        except (OSError, ValueError):
            pass
        # the committed profiler summary of this kernel (rocprofv3 --kernel-trace --stats, profiles/<round>/kernel_stats.csv) was collected on ANOTHER box of
        # the pool: its average launch time gives `frac_profile_box`, stated beside `frac` (this run, HIP events) so that the two cannot be confused
        frac_profile_box, profile_box_us = None, None
        try:
            import csv
            with open(os.path.join(ROOT, "profiles", PMC_ROUND, "kernel_stats.csv")) as f:
                for row in csv.reader(f):
                    if row and "wbc_hex_kernel<%d, false>" % {"id": 0, "mptc": 1, "pc": 2, "clf": 3}[shard["kind"]] in row[0]:
                        profile_box_us = float(row[3]) / 1e3                 # AverageNs
                        break
            if profile_box_us and cfg == 3 and n == 4096:
                frac_profile_box = flops * n / (profile_box_us * 1e-6) / 1e12 / PEAK_FP64_VALU_TFLOPS
        except (OSError, ValueError, IndexError):
            pass
        # what the issue slots did (same separate-pass PMC file, committed): executed FP64 flops include the arithmetic the
        # 16-lane mapping replicates on sub-lanes, so they say how busy the ALU was, not how much of it was useful
        issued = None
        try:
            if shard["kind"] == "mptc" and cfg == 3 and n == 4096:
                with open(os.path.join(ROOT, "profiles", PMC_ROUND, "hex_pmc.json")) as f:
                    pj = json.load(f)
                if pj.get("kernel_src_sha16") != kernel_src_sha16():
                    raise KeyError("counter file is from another kernel build")
                pc = pj["counters"]
                g = lambda k: pc[k]["mean_per_launch"]
                ex = 64.0 * (2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64"))
                issued = {"executed_fp64_flops_per_launch": ex, "executed_TFLOPs": ex / sec / 1e12,
                          "executed_frac_of_peak": ex / sec / 1e12 / PEAK_FP64_VALU_TFLOPS,
                          "redundancy": ex / (flops * n), "valu_insts_per_wavefront": g("SQ_INSTS_VALU") / g("SQ_WAVES"),
                          "mfma_insts": g("SQ_INSTS_MFMA"),
                          "source": "profiles/" + PMC_ROUND + "/hex_pmc.json (rocprofv3 --pmc, separate passes of this command; same kernel "
                                    "sources: sha16 %s)" % pj["kernel_src_sha16"]}
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "whole-body-QP control ticks/s at N=4096 Mini Cheetah",
            "value": n_total * a.steps / dt, "unit": "ticks/s",
            # the same K launches without the region's fixed cost (stream start-up, statistics reduction and gather, one wake-up: ~30 us,
            # which weigh 1.5 us per step at the driver's K = 20 and 0.15 at K = 200): instances / slowest rank's HIP-event launch time
            "value_kernel_only": n_total / (ms_per_launch * 1e-3),
            "value_cold": n_total * a.steps / dt_cold, "kernel_ms_cold": ms_cold,
            "region_us": {"queue_K_launches": region_us[0], "statistics_reduce_and_the_one_wait": region_us[1], "gather": region_us[2],
                          "closing_bracket": region_us[3], "total": round(dt * 1e6, 1), "device_time_of_the_K_launches": round(ms_per_launch * 1e3 * a.steps, 1),
                          "note": "rank 0's host clock inside the timed region"},
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "backend": a.backend, "ramp_seconds": a.ramp_seconds,
            "ranks_seen": seen, "per_rank_ticks": [r["ticks"] for r in per_rank], "per_rank_kernel_ms": per_rank_kernel_ms,
            "process_group": (a.backend if use_pg else None), "kernel_src_sha16": kernel_src_sha16(),
            "per_rank_tau_fnv1a64": per_rank_hash,
            "config": {"workload": "BASELINE configs[%d]: %d x %s, %s controller, %s" % (
                cfg - 1, n_total, shard["model"], shard["kind"].upper(),
                "trot contact masks" if cfg != 2 else "4-contact stand"),
                "instances_per_gpu": n, "seed": shard["seed"],
                "domain_randomised": mu is not None, "parallelism": "batch-shard x%d" % world},
            "roofline": {"bound": "fp64-valu", "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "frac_profile_box": frac_profile_box, "profile_box_kernel_us": profile_box_us,
                         "frac_note": "frac = THIS run (HIP events around the K timed launches on this box); frac_profile_box = the committed rocprofv3 summary "
                                      "profiles/%s/kernel_stats.csv, another box of the pool (boxes differ by up to 5 %%)" % PMC_ROUND,
                         "kernel": "wbc_hex_kernel<%d, false>" % {"id": 0, "mptc": 1, "pc": 2, "clf": 3}[shard["kind"]],   # the name rocprofv3 prints
                         "kernel_law": shard["kind"].upper(),
                         "kernel_ms": ms_per_launch, "flops_per_tick": flops, "issued": issued,
                         "kernel_ms_dist": {"median": float(np.median(each)), "p10": float(np.percentile(each, 10)),
                                            "p90": float(np.percentile(each, 90)), "launches": int(each.size),
                                            "how": "one HIP event between every two launches, after the timed region"},
                         "hbm": {"achieved_GBs": bpt * n / sec / 1e9, "peak_GBs": PEAK_HBM_GBS,
                                 "frac": bpt * n / sec / 1e9 / PEAK_HBM_GBS, "bytes_per_tick": bpt},
                         "bound_detail": "FP64 vector ALU: counted flops per tick x ticks / HIP-event launch time, priced against "
                                         "the dense FP64 peak (78.6 TFLOP/s, the same figure for FP64 MFMA and FP64 VALU); the "
                                         "kernel issues v_fma_f64, not MFMA (profiles/r02/micro_mfma_f64.md)",
                         "note": "HBM does not bind this path (SURVEY 8d): 864 algorithmic bytes per 37.6 kflop tick; HBM fraction stated beside it"},
            "status_nonzero": int(bad_total),                     # instances with status != 0 on the device: every rank's whole shard, summed
            "oracle_status_nonzero": (int(obad_total) if not a.no_cpu_baseline else None),    # the oracle's, on the parity samples
            "status_mismatch": (int(mis_total) if not a.no_cpu_baseline else None),           # sample instances where device and oracle disagree
            "torque_rel_err_vs_cpu_ref": (max(per_rank_rel) if not a.no_cpu_baseline else None),
            "per_rank_torque_rel_err": (per_rank_rel if not a.no_cpu_baseline else None),
            "parity_sample": "64 instances spread over EVERY rank's shard against the oracle on that rank's host; maximum over ranks",
            "rollout_stats": {k: st[k] for k in ("ticks", "status_nonzero", "iters_sum", "tau_abs_max")},
            "iters_per_tick": st["iters_sum"] / max(1.0, st["ticks"]),
            "kernel_info": ctrl.kernel_info(),
        }
        if world > 1 and not a.no_cpu_baseline:
            # the N > 1 line carries the CPU baseline as well: rank 0's host cores on rank 0's shard (the other ranks wait at the
            # closing barrier; outside the timed region)
            line["cpu_baseline"] = cpu_baseline(shard, a.cpu_seconds)
        if world == 1 and not a.no_cpu_baseline:
            line["n32768"] = large_batch(local)                       # informational, outside the timed region
            line["closed_loop"] = closed_loop(shard, local)           # informational, outside the timed region
            line["config1_gpu"] = config1_gpu(local)                  # BASELINE configs[0] on the product path (informational)
            line["cpu_baseline"] = cpu_baseline(batch, a.cpu_seconds)
            # parity beside the number: full torque vector (tier ii, 256 leading instances on top of the per-rank sample) and the
            # solver-independent accelerations (tier i)
            from oracle import oracle_py as orc
            k = 256
            tau_gpu = out[0][:, :k].cpu().numpy()
            sl = lambda x: None if x is None else (x[:, :k] if x.ndim == 2 else x[:k])
            tau_o, _, _ = orc.step_batch(shard["kind"], orc.model(shard["model"]), orc.params(shard["kind"]),
                                         sl(batch["q"]), sl(batch["v"]), sl(batch["targets"]), sl(batch["mask"]),
                                         sl(batch["mu"]), sl(batch["mass_scale"]))
            rel = np.abs(tau_gpu - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)
            line["torque_rel_err_vs_cpu_ref"] = max(line["torque_rel_err_vs_cpu_ref"], float(rel.max()))
            vd = torch.zeros((18, n), dtype=torch.float64, device=dev)
            ctrl.set_vdot_output(vd)
            ctrl.step(q, v, tg, mask, mu, ms, out=out); ctrl.sync()
            ctrl.set_vdot_output(None)
            vdg = vd[:, :32].cpu().numpy()
            worst = 0.0
            mo = orc.model(shard["model"])
            for i in range(32):
                p = orc.params(shard["kind"])
                if batch["mu"] is not None:
                    p.mu = float(batch["mu"][i])
                mi = mo if batch["mass_scale"] is None else orc.model_scaled(shard["model"], float(batch["mass_scale"][i]))
                ct = [(int(batch["mask"][i]) >> j) & 1 for j in range(4)]
                _, _, _, qp = orc.control_law(shard["kind"], mi, p, batch["q"][:, i], batch["v"][:, i], batch["targets"][:, i],
                                              ct, want_qp=True)
                worst = max(worst, float(np.abs(vdg[:, i] - qp["x"][:18]).max() / (1.0 + np.abs(qp["x"][:18]).max())))
            line["vdot_rel_err_vs_cpu_ref_qp"] = worst          # tier (i): solver-independent part of the solution
        print(json.dumps(line))
    ctrl.close()
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    run_rank(a)


if __name__ == "__main__":
    main()
