#!/usr/bin/env python3
"""Build-time lint over the device assembly (hipcc -save-temps): the gfx9 hazard "VALU writes a VGPR -> a DPP instruction
reads that VGPR as its DPP source: 2 wait states" must hold around every DPP read.

Why a lint: the fused broadcast-FMA (v_fmac_f64_dpp ... row_newbcast, csrc/wbc_kernels.hip HexDev::fma_bc / dot_bc /
rows3_bc) is emitted through inline asm, where the compiler's hazard recogniser does not see the DPP read; the rule is
kept by construction (dpp_fence() + volatile asm ordering), and a compiler or flag change that slipped a copy
(v_mov, v_accvgpr_read) between a fence and the asm would break it silently -- only GPU parity tests would notice.

    python3 tools/dpp_lint.py build/wbc_kernels-hip-amdgcn-amd-amdhsa-gfx950.s     (exit code 1 on a violation)

Model: every instruction is one wait state, `s_nop N` is N + 1.  For each instruction carrying a DPP control the two
wait states before it (straight-line layout order, labels crossed: the fall-through predecessor is checked, a branch
predecessor cannot be) must not contain a VALU instruction whose destination overlaps the DPP source register(s)."""
import re
import sys

DPP = re.compile(r"\b(quad_perm:|row_shl:|row_shr:|row_ror:|row_newbcast:|row_bcast:|row_share:|row_xmask:|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror)")
REG = re.compile(r"^(v|a)(?:(\d+)|\[(\d+):(\d+)\])$")


def regs(tok):
    m = REG.match(tok.strip())
    if not m or m.group(1) != "v":
        return set()
    if m.group(2) is not None:
        return {int(m.group(2))}
    return set(range(int(m.group(3)), int(m.group(4)) + 1))


def operands(line):
    body = line.split(None, 1)
    if len(body) < 2:
        return []
    ops = [o.strip() for o in body[1].split(",")]
    return [o.split()[0] if o else o for o in ops]     # strip modifiers that follow the last operand


def lint(path):
    bad = []
    n_dpp = n_fused = 0
    hist = []     # (wait_states, dest_regs, text) of the preceding instructions of the current function
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            continue
        if line.endswith(":"):
            if line.startswith("_Z") or line.startswith("__"):
                hist = []          # a new function
            continue
        op = line.split()[0]
        ops = operands(line)
        if DPP.search(line):
            n_dpp += 1
            n_fused += op.startswith("v_fmac_f64_dpp")
            # DPP source = src0: operand 1 for mov / 2-operand ALU forms (dst, src0, ...)
            src = regs(ops[1]) if len(ops) > 1 else set()
            ws = 0
            for w, dst, txt in reversed(hist):
                if ws >= 2:
                    break
                if dst & src:
                    bad.append((ln, line, txt))
                    break
                ws += w
        if op == "s_nop":
            w = int(ops[0], 0) + 1 if ops else 1
            hist.append((w, set(), line))
        else:
            dst = regs(ops[0]) if (op.startswith("v_") and ops and not op.startswith("v_cmp")) else set()
            hist.append((1, dst, line))
        if len(hist) > 8:
            hist = hist[-8:]
    return bad, n_dpp, n_fused


def scratch(path):
    """Second check on the same file: no kernel of the product may spill to scratch (the code object's metadata:
    .private_segment_fixed_size per kernel).  The persistent rollout kernels did for two rounds -- machine LICM hoisted the
    tick's literal constants out of the step loop and the allocator spilled them -- while the resource report only looked at
    the tick kernels."""
    txt = open(path).read()
    out = []
    for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size", txt, re.S):
        nm = re.search(r"\.name:\s+(\S+)", blk)
        ps = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        if nm and ps:
            out.append((nm.group(1), int(ps.group(1))))
    return out


if __name__ == "__main__":
    bad, n_dpp, n_fused = lint(sys.argv[1])
    print("dpp_lint: %d DPP instructions (%d fused v_fmac_f64_dpp), %d hazard violations" % (n_dpp, n_fused, len(bad)))
    for ln, line, prev in bad[:20]:
        print("  line %d: %s\n      <- written by: %s" % (ln, line, prev))
    ks = scratch(sys.argv[1])
    spilled = [(n, b) for n, b in ks if b > 0]
    if ks:
        print("scratch: %d kernels, %d with a private segment%s" % (len(ks), len(spilled), "".join("\n  %s: %d B/lane" % x for x in spilled)))
    sys.exit(1 if (bad or spilled) else 0)
