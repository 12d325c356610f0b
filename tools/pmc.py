#!/usr/bin/env python3
"""PMC collection for the hot kernel (GPU box only; run through gpurun).

    python3 tools/pmc.py --out gpurun_out/pmc_x.json --groups "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_IFETCH SQ_INSTS_VALU" \
        -- --variant hex --per-gpu 4096 --steps 20 --warmup 3

One rocprofv3 pass per counter group (counters of a group share one pass; --pmc is never combined
with the sys/hip traces), the program under the profiler is `python3 bench.py <args> --no-cpu-baseline`.
Writes {counter: mean value per launch of the wbc_* kernel with the most launches}.
"""
import argparse, csv, glob, json, os, subprocess, sys, tempfile

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--groups", nargs="+", required=True)
ap.add_argument("rest", nargs=argparse.REMAINDER)
a = ap.parse_args()
rest = [x for x in a.rest if x != "--"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench as _bench   # kernel_src_sha16(): the identity of the kernel sources these counters belong to (no torch import)
res = {"bench_args": rest, "kernel_src_sha16": _bench.kernel_src_sha16(), "counters": {}}
env = dict(os.environ, TMPDIR="/tmp")
for gi, grp in enumerate(a.groups):
    d = tempfile.mkdtemp(prefix="pmc%d_" % gi, dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + grp.split() + ["-d", d, "-o", "run", "--output-format", "csv", "--",
           sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + rest
    p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        res["counters"][grp] = "no output: " + p.stderr[-300:]
        continue
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if "wbc_" not in row["Kernel_Name"]:
                continue
            k = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
            s = acc.setdefault(k, [0.0, 0])
            s[0] += float(row["Counter_Value"]); s[1] += 1
            res.setdefault("dispatch", {"grid": row["Grid_Size"], "wg": row["Workgroup_Size"], "lds": row["LDS_Block_Size"],
                                        "scratch": row["Scratch_Size"], "vgpr": row["VGPR_Count"], "agpr": row["Accum_VGPR_Count"]})
    launches = {}
    for (kn, cn), (s, c) in acc.items():
        launches[kn] = max(launches.get(kn, 0), c)
    if not launches:                               # a pass that produced no wbc_* rows (profiler refused the counters, ...)
        res.setdefault("empty_passes", []).append(grp)
        continue
    hot = max(launches, key=launches.get)          # the hot kernel = the wbc_* kernel with the most launches
    for (kn, cn), (s, c) in acc.items():
        if kn == hot:
            res["counters"][cn] = {"mean_per_launch": s / c, "launches": c, "kernel": kn}
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
json.dump(res, open(a.out, "w"), indent=1)
print(json.dumps({k: (v["mean_per_launch"] if isinstance(v, dict) else v) for k, v in res["counters"].items()}, indent=1))
