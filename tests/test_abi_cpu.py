"""CPU-side checks of the C-ABI library and the host logic (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from quadruped_drake_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    l = _lib.lib()
    import glob
    hdr = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))
    declared = sorted(set(re.findall(r"\b(wbc_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(_lib.SYMBOLS), (declared, sorted(_lib.SYMBOLS))
    for s in declared:
        assert hasattr(l, s), s
    assert l.wbc_version() >= 100


def test_the_boundary_header_is_the_controller_interface_only():
    """include/wbc.h = the reference's controller interface (+ the widened rows of SURVEY 8f); the frozen out-of-scope exports
    (PD law, robot-side wire format) live in include/wbc_extras.h."""
    main = open(os.path.join(ROOT, "include", "wbc.h")).read()
    extras = open(os.path.join(ROOT, "include", "wbc_extras.h")).read()
    for s in ("wbc_pd_step", "wbc_robot_state_decode", "wbc_robot_state_encode", "wbc_robot_states_unpack", "wbc_robot_controls_pack"):
        assert s not in main and s in extras, s
    for s in ("wbc_create", "wbc_step", "wbc_sync", "wbc_destroy", "wbc_stats_get", "wbc_stats_pack", "wbc_stats_reduce", "wbc_last_error"):
        assert s in main, s


def test_stats_reduce_folds_gathered_vectors_without_a_gpu():
    """wbc_stats_reduce: the host half of the one collective (a C caller brings its own ncclAllGather of wbc_stats_pack vectors):
    sums everywhere, the maximum on tau_abs_max; equal to the Python mirror's reduction (quadruped_drake_amd/stats.py)."""
    from quadruped_drake_amd import _lib, stats
    l = _lib.lib()
    rng = np.random.default_rng(3)
    world = 5
    g = rng.uniform(0.0, 100.0, (world, 22))
    out = _lib.WbcStats()
    assert l.wbc_stats_reduce(g.ctypes.data_as(_lib.c_double_p), world, C.byref(out)) == 0
    got = np.array([out.ticks, out.status_nonzero, out.iters_sum, out.tau_abs_sum, out.tau_abs_max, out.err_sum] + list(out.mask_count))
    want = g.sum(0); want[4] = g[:, 4].max()
    assert np.allclose(got, want, rtol=1e-15)
    per_rank = [stats.from_vector(g[r]) for r in range(world)]
    red = stats.reduce_vectors(g)
    assert red["tau_abs_max"] == g[:, 4].max() and abs(red["ticks"] - g[:, 0].sum()) < 1e-9 and red["mask_count"][15] == pytest.approx(g[:, 21].sum())
    assert per_rank[2]["err_sum"] == g[2, 5]
    assert stats.reduce_vectors_numpy(g) == red                 # the host-only fallback of stats.reduce_vectors: same fold, same bits
    assert l.wbc_stats_reduce(None, 1, C.byref(out)) < 0 and l.wbc_stats_reduce(g.ctypes.data_as(_lib.c_double_p), 0, C.byref(out)) < 0
    assert l.wbc_stats_pack(None, g.ctypes.data_as(_lib.c_double_p)) < 0


def test_params_default_matches_reference_literals():
    from quadruped_drake_amd import _lib
    l = _lib.lib()
    p = _lib.WbcParams()
    assert l.wbc_params_default(0, C.byref(p)) == 0
    # inverse_dynamics_controller.py:117-127, :19, :93
    assert (p.Kp_body_p, p.Kd_body_p, p.Kp_foot, p.Kd_foot, p.w_body, p.w_foot, p.mu, p.Kd_contact) == \
        (500.0, 50.0, 100.0, 20.0, 10.0, 1.0, 0.7, 100.0)
    assert l.wbc_params_default(1, C.byref(p)) == 0
    # mptc_controller.py:143-153
    assert (p.Kp_body_p, p.Kd_body_p, p.Kp_foot, p.Kd_foot, p.w_body, p.w_foot, p.mu) == (100.0, 10.0, 200.0, 20.0, 10.0, 1.0, 0.7)
    assert np.isinf(p.tau_max)
    assert l.wbc_params_default(2, C.byref(p)) == 0 and p.Kp_foot == 200.0     # PC uses the MPTC gains (pc_controller.py:69-81)
    assert l.wbc_params_default(3, C.byref(p)) == 0 and p.Kp_body_p == 500.0   # CLF derives from IDController
    assert l.wbc_params_default(7, C.byref(p)) < 0
    assert b"bad argument" in l.wbc_last_error()


def test_create_rejects_misuse_and_fails_loudly_without_gpu():
    import torch
    from quadruped_drake_amd import _lib, IDController
    l = _lib.lib()
    h = C.c_void_p()
    m = _lib.WbcModel()
    assert l.wbc_create(None, 0, None, 8, 0, 0, C.byref(h)) < 0
    from quadruped_drake_amd.controller import load_model
    m.flat[:] = load_model("mini_cheetah")["flat"]
    m.q_perm[:] = list(range(12)); m.act_perm[:] = list(range(12))
    assert l.wbc_create(C.byref(m), 5, None, 8, 0, 0, C.byref(h)) < 0        # bad kind
    assert l.wbc_create(C.byref(m), 0, None, 0, 0, 0, C.byref(h)) < 0        # bad batch
    hdr = open(os.path.join(ROOT, "include", "wbc.h")).read()
    max_ld = int(re.search(r"#define WBC_MAX_LD (\d+)", hdr).group(1))
    assert max_ld == 1 << 23
    assert l.wbc_create(C.byref(m), 0, None, max_ld + 1, 0, 0, C.byref(h)) < 0 and b"WBC_MAX_LD" in l.wbc_last_error()   # 32-bit row offsets in the kernels
    m.q_perm[3] = 4
    assert l.wbc_create(C.byref(m), 0, None, 8, 0, 0, C.byref(h)) < 0        # not a permutation
    assert b"permutation" in l.wbc_last_error()
    if not torch.cuda.is_available():
        with pytest.raises(_lib.WbcError):
            IDController(max_batch=4, host_ptrs=True)       # no CPU fallback: must raise


def test_pack_trunk_input_follows_planner_schema():
    """planners/simple.py:45-85 dict -> 54 rows + mask."""
    from quadruped_drake_amd import pack_trunk_input
    d = {"p_lf": [0.175, 0.11, 0], "p_rf": [0.175, -0.11, 0], "p_lh": [-0.2, 0.11, 0], "p_rh": [-0.2, -0.11, 0]}
    for f in ("lf", "rf", "lh", "rh"):
        d["pd_" + f] = np.zeros(3); d["pdd_" + f] = np.zeros(3)
    d.update(rpy_body=np.zeros(3), p_body=np.array([0, 0, 0.3]), rpyd_body=np.zeros(3), pd_body=np.zeros(3),
             rpydd_body=np.zeros(3), pdd_body=np.zeros(3), contact_states=[True, False, True, True],
             f_cj=np.zeros((3, 4)), u2_max=0.0)
    t, mask = pack_trunk_input(d)
    assert mask == 0b1101
    assert np.allclose(t[0:3], [0, 0, 0.3]) and np.allclose(t[18:21], [0.175, 0.11, 0]) and np.allclose(t[45:48], [-0.2, -0.11, 0])
    from quadruped_drake_amd import workloads
    assert np.allclose(workloads.standing_targets("mini_cheetah", 1)[:, 0], t)


def test_model_tables_are_committed_and_sane():
    from quadruped_drake_amd.controller import load_model
    for name, mass in (("mini_cheetah", 8.252), ("anymal_b", 30.4214)):
        t = load_model(name)
        assert len(t["flat"]) == 215 and abs(t["total_mass"] - mass) < 1e-3
        assert sorted(t["act_perm"]) == list(range(12))


def test_workload_generators_are_seeded_and_shaped():
    from quadruped_drake_amd import workloads
    a = workloads.make_batch(3, n=64); b = workloads.make_batch(3, n=64)
    assert all(np.array_equal(a[k], b[k]) for k in ("q", "v", "targets", "mask"))
    assert a["q"].shape == (19, 64) and a["v"].shape == (18, 64) and a["targets"].shape == (54, 64)
    assert set(np.unique(a["mask"])) <= {0b1001, 0b0110}
    assert np.allclose(np.linalg.norm(a["q"][:4], axis=0), 1.0)
    c = workloads.make_batch(5, n=32)
    assert c["mu"].min() >= 0.4 and c["mu"].max() <= 1.0 and c["mass_scale"].min() >= 0.8
    assert (workloads.make_batch(2, n=8)["mask"] == 0b1111).all()


def test_dpp_hazard_lint_flags_a_violation_and_passes_clean_code(tmp_path):
    """tools/dpp_lint.py (run by build() over the device assembly): a VALU write followed within two wait states by a DPP read
    of the same register is reported; the same code behind an s_nop, or with another register, is not."""
    import subprocess, sys
    tool = os.path.join(ROOT, "tools", "dpp_lint.py")
    dirty = tmp_path / "dirty.s"
    dirty.write_text("_Zk:\n\tv_add_f64 v[2:3], v[4:5], v[6:7]\n\tv_fmac_f64_dpp v[8:9], v[2:3], v[10:11] row_newbcast:3 row_mask:0xf bank_mask:0xf\n")
    clean = tmp_path / "clean.s"
    clean.write_text("_Zk:\n\tv_add_f64 v[2:3], v[4:5], v[6:7]\n\ts_nop 1\n\tv_fmac_f64_dpp v[8:9], v[2:3], v[10:11] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                     "\tv_add_f64 v[12:13], v[4:5], v[6:7]\n\tv_fmac_f64_dpp v[8:9], v[14:15], v[12:13] row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
    assert subprocess.run([sys.executable, tool, str(dirty)], capture_output=True).returncode == 1
    r = subprocess.run([sys.executable, tool, str(clean)], capture_output=True, text=True)
    assert r.returncode == 0 and "2 fused" in r.stdout
    # the same tool refuses a kernel that spills to scratch (code-object metadata)
    meta = "amdhsa.kernels:\n  - .agpr_count:     0\n    .name:           _Zk\n    .private_segment_fixed_size: %d\n    .vgpr_count: 12\n    .wavefront_size: 64\n"
    spill = tmp_path / "spill.s"
    spill.write_text(clean.read_text() + meta % 96)
    ok = tmp_path / "ok.s"
    ok.write_text(clean.read_text() + meta % 0)
    r = subprocess.run([sys.executable, tool, str(spill)], capture_output=True, text=True)
    assert r.returncode == 1 and "_Zk: 96 B/lane" in r.stdout
    assert subprocess.run([sys.executable, tool, str(ok)], capture_output=True).returncode == 0


def test_headers_compile_as_c99_and_struct_layouts_equal_the_ctypes_mirrors(tmp_path):
    """include/wbc.h and include/wbc_extras.h are C headers (`gcc -std=c99 -Wall -pedantic`), and every struct the ctypes
    mirror in quadruped_drake_amd/_lib.py re-declares by hand has the same size and the same offset for every field: a member
    added on one side only would otherwise be a silent mis-read across the boundary."""
    import subprocess
    from quadruped_drake_amd import _lib
    mirrors = {"wbc_model": _lib.WbcModel, "wbc_params": _lib.WbcParams, "wbc_stats": _lib.WbcStats,
               "wbc_trunk_state": _lib.WbcTrunkState, "wbc_robot_state": _lib.WbcRobotState}
    lines = []
    for cname, cls in mirrors.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf("%s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (cname, f, cname, f, cname, f))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "wbc.h"\n#include "wbc_extras.h"\n'
                   'int main(void) {\n  printf("NSTAT %d\\n", WBC_NSTAT);\n  ' + "\n  ".join(lines) + "\n  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)],
                   check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = {k: set() for k in mirrors}
    for ln in out:
        w = ln.split()
        if not w:
            continue
        if w[0] == "NSTAT":
            assert int(w[1]) == 22 == C.sizeof(_lib.WbcStats) // 8
        elif w[1] == "sizeof":
            assert int(w[2]) == C.sizeof(mirrors[w[0]]), ln
        else:
            fld = getattr(mirrors[w[0]], w[1])
            assert (int(w[2]), int(w[3])) == (fld.offset, fld.size), ln
            seen[w[0]].add(w[1])
    # ... and the other way round: every member the C struct declares is mirrored (member names parsed from the headers)
    hdr = open(os.path.join(ROOT, "include", "wbc.h")).read() + open(os.path.join(ROOT, "include", "wbc_extras.h")).read()
    for cname, cls in mirrors.items():
        body = re.search(r"typedef struct \{([^}]*)\}\s*%s;" % cname, hdr).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        members = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                members += [re.match(r"\**\s*([A-Za-z_0-9]+)", m.strip().split(" ")[-1]).group(1) for m in decl.split(",")]
        assert sorted(members) == sorted(f for f, _ in cls._fields_) == sorted(seen[cname]), (cname, members)


def test_window_generation_equals_sharding_the_whole_batch():
    """What every rank of a multi-GPU run generates (workloads.make_batch(..., window=shard_range(...))) is bit for bit the
    rank's shard of the whole batch (stats.shard_batch): BASELINE config 5, N = 32768 over 8 ranks; and a ragged split."""
    from quadruped_drake_amd import stats, workloads
    for cfg, n, world in ((5, 32768, 8), (3, 1003, 3)):
        whole = workloads.make_batch(cfg, n=n)
        covered = 0
        for r in range(world):
            lo, hi = stats.shard_range(n, r, world)
            win = workloads.make_batch(cfg, n=n, window=(lo, hi))
            ref = stats.shard_batch(whole, r, world)
            assert win["n"] == ref["n"] == hi - lo and win["n_total"] == n and win["window"] == (lo, hi)
            for k in ("q", "v", "targets", "mask", "mu", "mass_scale"):
                if ref[k] is None:
                    assert win[k] is None
                else:
                    assert win[k].dtype == ref[k].dtype and np.array_equal(win[k], ref[k]), (cfg, r, k)
            covered += hi - lo
        assert covered == n
