"""Analysis: static instruction count and mix between consecutive phase stamps (s_memtime) of a -DWBC_STAMPS build.
usage: phase_count.py KIND [extra hipcc flags]"""
import collections, re, subprocess, sys
kind = sys.argv[1]
FLAGS = open("/root/repo/quadruped_drake_amd/csrc/hipcc_flags.txt").read().split()
subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [
                       "-DWBC_STAMPS", "-DWBC_DEV_ONLY=" + kind, "-S", "--cuda-device-only", "-o", "/tmp/asm/ph.s",
                       "/root/repo/quadruped_drake_amd/csrc/wbc_kernels.hip"] + sys.argv[2:], stderr=subprocess.DEVNULL)
txt = open("/tmp/asm/ph.s").read().split("\n")
key = "wbc_hex_kernelILi%sELb0E" % kind
start = [i for i, l in enumerate(txt) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l][0]
end = [i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end")][0]
def cls(s):
    op = s.split()[0]
    if re.match(r"v_(fma|fmac|mul|add)_f64", op): return "f64"
    if "dpp" in s: return "dpp"
    if "cndmask" in op: return "cndmask"
    if "accvgpr" in op: return "agpr"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "scalar"
    if op.startswith("v_mov"): return "v_mov"
    return "valu_other"
seg = collections.Counter(); segs = []
for l in txt[start + 1:end]:
    s = l.split(";")[0].strip()
    if not s or s.endswith(":") or s.startswith("."): continue
    if s.startswith("s_memtime"):
        segs.append(seg); seg = collections.Counter(); continue
    seg[cls(s)] += 1
segs.append(seg)
names = ["pre-stamp0", "loads issue", "mask/mu + lds writes", "barrier", "state", "leg", "G_b + solve", "rows", "append", "J rows", "active set (all paths)", "outputs", "stats tail"]
tot = collections.Counter()
for i, sg in enumerate(segs):
    n = sum(sg.values()); tot.update(sg)
    print("%-26s %5d  %s" % (names[i] if i < len(names) else "seg%d" % i, n, dict(sg.most_common())))
print("%-26s %5d  %s" % ("total", sum(tot.values()), dict(tot.most_common())))
