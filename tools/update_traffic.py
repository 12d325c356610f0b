#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the PMC passes of tools/profile_round.sh (gpurun_out/<tag>/pmc*.json, rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE, separate passes):  python3 tools/update_traffic.py r04
Every entry carries the identity of the kernel sources the counters were collected on (bench.kernel_src_sha16 at collection time):
bench.py replays an entry only on that build."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", tag)
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
tr = json.load(open(path))
rnd = int("".join(c for c in tag if c.isdigit()) or 0)
for key, fn, what in (("mptc_cfg3_n4096_hex", "pmc.json", "BASELINE configs[2], N = 4096"),
                      ("mptc_cfg5_n32768_hex", "pmc_n32768.json", "N = 32768, per-instance mu / mass scale"),
                      ("mptc_cfg5_n4096_hex", "pmc_cfg5_n4096.json", "the 4096-instance shard of configs[4] (N > 1 bench lines)"),
                      ("id_cfg2_n4096_hex", "pmc_id.json", "BASELINE configs[1] at N = 4096, ID law")):
    f = os.path.join(src, fn)
    if not os.path.exists(f):
        print("missing", f); continue
    pj = json.load(open(f)); c = pj["counters"]
    if not (isinstance(c.get("FETCH_SIZE"), dict) and isinstance(c.get("WRITE_SIZE"), dict)):
        print("no traffic counters in", f); continue
    fe, wr = c["FETCH_SIZE"]["mean_per_launch"], c["WRITE_SIZE"]["mean_per_launch"]
    tr[key] = {"round": rnd, "kernel_src_sha16": pj["kernel_src_sha16"],
               "build": "round-%d kernel, %s (profiles/%s/%s)" % (rnd, what, tag, "hex_" + fn if fn == "pmc.json" else fn),
               "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1), "bytes_per_launch": int(round((2.0 * fe + wr) * 1024))}
    print(key, tr[key])
json.dump(tr, open(path, "w"), indent=1)
