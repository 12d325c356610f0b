"""Round 4 analysis helper: builds /tmp/lab/libdumpA.so, a copy of the host emulation (tools/host_tick.cpp) patched to dump the inputs
and the result of the QR append of every robot (initial diagonal, the 18 / 30 appended rows, the final factor) -- what qr_check.py,
pivot_order_stats.py and refine_against_R.py read."""
import os; os.makedirs("/tmp/lab", exist_ok=True)
import shutil, subprocess, tempfile, os, sys, ctypes as C, numpy as np
_ROOT = "/root/repo"
_T = tempfile.mkdtemp(prefix="dumpA_")
shutil.copytree(_ROOT + "/quadruped_drake_amd/csrc", _T + "/quadruped_drake_amd/csrc"); shutil.copytree(_ROOT + "/include", _T + "/include")
os.makedirs(_T + "/tools"); shutil.copy(_ROOT + "/tools/host_tick.cpp", _T + "/tools/"); shutil.copy(_ROOT + "/tools/wbc_scalar_tick.hpp", _T + "/tools/")
s = open(_T + "/quadruped_drake_amd/csrc/wbc_hex.hpp").read()
old = "  hex_qr_append<Q, P1 + NZ, NV>(qo, Rcol, Acol);\n  WBC_STAMP(14);"
assert old in s
s = s.replace(old, old + "\n  if (g_qr_dump) { double* o = g_qr_dump + h * 64; for (int k = 0; k < NV; k++) o[48 + k] = Rcol[k]; }", 1)
s = s.replace("#ifndef WBC_NO_SWING_COMPACT\n  // Task-space laws: the swing rows", "  if (g_qr_dump) { double* o = g_qr_dump + h * 64; for (int k = 0; k < NV; k++) o[k] = Rcol[k]; for (int k = 0; k < P1 + NZ; k++) o[16 + k] = Acol[k]; }\n#ifndef WBC_NO_SWING_COMPACT\n  // Task-space laws: the swing rows", 1)
s = s.replace("extern double* g_gi_dump;", "extern double* g_gi_dump; extern double* g_qr_dump;")
open(_T + "/quadruped_drake_amd/csrc/wbc_hex.hpp", "w").write(s)
s = open(_T + "/tools/host_tick.cpp").read()
s = s.replace("double* g_gi_dump = nullptr;", "double* g_gi_dump = nullptr; double* g_qr_dump = nullptr; static double* g_qr_dump_base = nullptr;\nextern \"C\" void host_qr_dump(double* buf) { g_qr_dump_base = buf; }")
s = s.replace("    g_gi_dump = g_gi_dump_base ? g_gi_dump_base + (size_t)i * 256 : nullptr;", "    g_gi_dump = g_gi_dump_base ? g_gi_dump_base + (size_t)i * 256 : nullptr;\n    g_qr_dump = g_qr_dump_base ? g_qr_dump_base + (size_t)i * 1024 : nullptr;")
open(_T + "/tools/host_tick.cpp", "w").write(s)
subprocess.check_call(["g++", "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-ffp-contract=off", "-o", "/tmp/lab/libdumpA.so", _T + "/tools/host_tick.cpp"])
print("built /tmp/lab/libdumpA.so")
