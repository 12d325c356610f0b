"""Analysis (not product code): WHERE the non-FP64 instructions of a tick kernel live.  Static instruction mix per phase of a
-DWBC_STAMPS build (the phase stamps are s_memtime instructions in the emitted assembly), optionally multiplied out with the
measured cycles per phase of the same build (tools/stamp_hex.py --json on the GPU box).

    python tools/lab/isa_mix.py KIND [--cycles stamps.json] [--md out.md] [extra hipcc flags]

Classes: f64 = v_fma / v_fmac / v_mul / v_add _f64 without DPP; f64dpp = the fused v_fmac_f64_dpp broadcast-FMA; other f64 = min / max /
rcp / rsq / cmp / ldexp ... _f64; cndmask = v_cndmask_b32 (a double select is two); v_mov = plain v_mov_b32 / b64; dppmov = v_mov_*_dpp
(quad_perm, row_ror, row_newbcast moves); agpr = v_accvgpr_read / write (the allocator's spill space); bperm = ds_bpermute_b32;
lds = every other ds_*; salu = s_* except s_nop / s_waitcnt; wait = s_waitcnt + s_nop; vmem; other = remaining VALU (integer, compares ...)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["pre-stamp0", "loads issue", "mask/mu + lds writes", "barrier", "state", "leg", "G_b + solve", "rows", "append", "J rows",
         "active set (all paths)", "outputs", "stats tail"]
ORDER = ["f64", "f64dpp", "f64other", "cndmask", "v_mov", "dppmov", "agpr", "bperm", "lds", "salu", "wait", "vmem", "other"]


def cls(s):
    op = s.split()[0]
    dpp = ("row_" in s) or ("quad_perm" in s) or op.endswith("_dpp")
    if op.startswith("v_fmac_f64") and dpp:
        return "f64dpp"
    if re.match(r"v_(fma|fmac|mul|add)_f64", op):
        return "f64"
    if op.startswith("v_mov") and dpp:
        return "dppmov"
    if op.endswith("_f64") or "_f64_" in op:
        return "f64other"
    if "cndmask" in op:
        return "cndmask"
    if "accvgpr" in op:
        return "agpr"
    if op.startswith("ds_bpermute"):
        return "bperm"
    if op.startswith("ds_"):
        return "lds"
    if op in ("s_nop", "s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_mov"):
        return "v_mov"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def phases(kind, extra):
    flags = open(os.path.join(ROOT, "quadruped_drake_amd", "csrc", "hipcc_flags.txt")).read().split()
    os.makedirs("/tmp/asm", exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DWBC_STAMPS", "-DWBC_DEV_ONLY=" + kind, "-S", "--cuda-device-only", "-o", "/tmp/asm/mix.s",
                           os.path.join(ROOT, "quadruped_drake_amd", "csrc", "wbc_kernels.hip")] + extra, stderr=subprocess.DEVNULL)
    txt = open("/tmp/asm/mix.s").read().split("\n")
    key = "wbc_hex_kernelILi%sELb0E" % kind
    start = [i for i, l in enumerate(txt) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l][0]
    end = [i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end")][0]
    seg = collections.Counter(); segs = []
    for l in txt[start + 1:end]:
        s = l.split(";")[0].strip()
        if not s or s.endswith(":") or s.startswith("."):
            continue
        if s.startswith("s_memtime"):
            segs.append(seg); seg = collections.Counter(); continue
        seg[cls(s)] += 1
    segs.append(seg)
    return segs


def main():
    a = sys.argv[1:]
    kind = a.pop(0)
    cyc = md = None
    if "--cycles" in a:
        i = a.index("--cycles"); cyc = json.load(open(a[i + 1])); del a[i:i + 2]
    if "--md" in a:
        i = a.index("--md"); md = a[i + 1]; del a[i:i + 2]
    segs = phases(kind, a)
    rows = []
    tot = collections.Counter()
    hdr = "| phase | instr | " + " | ".join(ORDER) + " | non-FP64 VALU | cycles (median) | cycles / instr |"
    rows.append(hdr); rows.append("|" + "---|" * (len(ORDER) + 5))
    for i, sg in enumerate(segs):
        n = sum(sg.values()); tot.update(sg)
        nm = NAMES[i] if i < len(NAMES) else "seg%d" % i
        nonf = sum(sg[k] for k in ("cndmask", "v_mov", "dppmov", "agpr", "other", "f64other"))
        c = cyc.get(nm) if cyc else None
        rows.append("| %s | %d | %s | %d | %s | %s |" % (nm, n, " | ".join(str(sg[k]) for k in ORDER), nonf,
                                                       "%d" % c if c else "", "%.1f" % (c / n) if c and n else ""))
    n = sum(tot.values())
    nonf = sum(tot[k] for k in ("cndmask", "v_mov", "dppmov", "agpr", "other", "f64other"))
    rows.append("| **total** | %d | %s | %d | | |" % (n, " | ".join(str(tot[k]) for k in ORDER), nonf))
    out = "\n".join(rows)
    print(out)
    if md:
        open(md, "w").write(out + "\n")


if __name__ == "__main__":
    main()
