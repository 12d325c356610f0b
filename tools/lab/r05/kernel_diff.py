#!/usr/bin/env python3
"""Analysis: which kernels of two builds of the library differ in their INSTRUCTIONS (addresses and branch targets ignored)?  Carves the gfx950 code
object out of each .so, disassembles it with llvm-objdump and compares kernel by kernel -- the check that a change meant for some laws left the
others' code alone (a timing difference on an instruction-identical kernel is code placement, profiles/r05/code_placement.md).
    python3 tools/lab/r05/kernel_diff.py build_variants/a.so build_variants/b.so"""
import re, struct, subprocess, sys, tempfile, os
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernels(so):
    b = open(so, "rb").read()
    for m in re.finditer(b"\x7fELF\x02\x01\x01", b):
        o = m.start()
        if struct.unpack_from("<H", b, o + 18)[0] == 224:      # EM_AMDGPU
            shoff = struct.unpack_from("<Q", b, o + 40)[0]; shentsize, shnum = struct.unpack_from("<HH", b, o + 58)
            co = b[o:o + shoff + shentsize * shnum]; break
    else:
        raise SystemExit("no gfx code object in " + so)
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(co)
    out = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
    os.unlink(f.name)
    d, cur = {}, None
    for l in out.splitlines():
        l = re.sub(r"\s*// [0-9A-Fa-f]+:.*$", "", l.strip())
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if m:
            cur = m.group(1); d[cur] = []; continue
        if cur is not None and l:
            d[cur].append(re.sub(r"<.*?>", "", re.sub(r"\b0x[0-9a-f]+\b|\b[0-9]+\b(?=\s*$)", "N", l)))
    return d


if __name__ == "__main__":
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    def dem(k):
        t = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        m = re.search(r"(\w+(<[^>]*>)?)\(", t.replace("(anonymous namespace)::", ""))
        return m.group(1) if m else t[:80]
    for k in sorted(set(a) | set(b)):
        na, nb = len(a.get(k, [])), len(b.get(k, []))
        print("%-9s %6d %6d  %s" % ("same" if a.get(k) == b.get(k) else "DIFFERENT", na, nb, dem(k)))
