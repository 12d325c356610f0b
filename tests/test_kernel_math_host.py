"""The kernel's per-robot math (quadruped_drake_amd/csrc/wbc_tick.hpp) instantiated on the HOST
with double, against the oracle.  This pins the reduced 12-variable formulation to the literal
30+3nc-variable restatement without needing a GPU; the -m gpu tests then check the device build
of the very same header through the C ABI."""
import numpy as np
import pytest

import host_tick as ht
from oracle import oracle_py as orc
from quadruped_drake_amd import workloads

TOL = 1e-4   # north_star: torques within 1e-4 relative of the CPU reference


def rel_err(tau, tau_o):
    return np.abs(tau - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)


@pytest.mark.parametrize("cfg,kind", [(2, "id"), (3, "mptc"), (3, "id"), (2, "mptc"), (4, "mptc"), (5, "mptc"), (3, "pc"), (2, "pc"), (3, "clf"), (2, "clf"), (4, "clf")])
def test_host_kernel_math_matches_oracle(cfg, kind):
    b = workloads.make_batch(cfg, n=192)
    t = orc.load_model_json(b["model"])
    m = orc.model(b["model"]); p = orc.params(kind)
    tau_o, met_o, st_o = orc.step_batch(kind, m, p, b["q"], b["v"], b["targets"], b["mask"], b["mu"],
                                        b["mass_scale"], nthreads=4)
    tau, met, st, it = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"])
    assert (st == 0).all() and (st_o == 0).all()
    assert rel_err(tau, tau_o).max() < 1e-5 < TOL
    assert np.allclose(met, met_o, rtol=1e-7, atol=1e-8)


def test_all_contact_modes_and_params():
    b = workloads.make_batch(3, n=16)
    t = orc.load_model_json("mini_cheetah"); m = orc.model("mini_cheetah")
    for mask in range(16):
        mk = np.full(16, mask, np.uint8)
        for kind in ("id", "mptc", "pc", "clf"):
            p = orc.params(kind)
            tau_o, met_o, st_o = orc.step_batch(kind, m, p, b["q"], b["v"], b["targets"], mk)
            tau, met, st, it = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], mk)
            assert (st == 0).all() and (st_o == 0).all(), (mask, kind)
            assert rel_err(tau, tau_o).max() < 1e-5, (mask, kind, rel_err(tau, tau_o).max())
            assert np.allclose(met, met_o, rtol=1e-7, atol=1e-8), (mask, kind)


def test_torque_box_and_mu():
    b = workloads.make_batch(2, n=32)
    t = orc.load_model_json("mini_cheetah"); m = orc.model("mini_cheetah")
    p = orc.params("id"); p.tau_max = 12.0; p.mu = 0.45
    tau_o, _, st_o = orc.step_batch("id", m, p, b["q"], b["v"], b["targets"], b["mask"])
    pp = np.array([p.Kp_body_p, p.Kd_body_p, p.Kp_body_rpy, p.Kd_body_rpy, p.Kp_foot, p.Kd_foot, p.w_body,
                   p.w_foot, p.mu, p.Kd_contact, p.tau_max, p.tiebreak_eps2])
    tau, _, st, _ = ht.run("id", t["flat"], b["q"], b["v"], b["targets"], b["mask"], params12=pp)
    ok = (st == 0) & (st_o == 0)
    assert ok.sum() >= 24          # a 12 N.m box can be infeasible for violent states: both must agree
    assert np.array_equal(st == 0, st_o == 0)
    assert np.abs(tau[:, ok]).max() <= 12.0 + 1e-9
    assert rel_err(tau[:, ok], tau_o[:, ok]).max() < 1e-5


def test_permutations():
    b = workloads.make_batch(3, n=8)
    t = orc.load_model_json("mini_cheetah")
    rng = np.random.default_rng(0)
    qperm = rng.permutation(12); aperm = rng.permutation(12)
    q2 = b["q"].copy(); v2 = b["v"].copy()
    q2[7 + qperm] = b["q"][7:]            # canonical joint j lives in row 7 + q_perm[j]
    v2[6 + qperm] = b["v"][6:]
    tau, _, _, _ = ht.run("mptc", t["flat"], b["q"], b["v"], b["targets"], b["mask"])
    tau2, _, _, _ = ht.run("mptc", t["flat"], q2, v2, b["targets"], b["mask"], q_perm=qperm, act_perm=aperm)
    assert np.array_equal(tau2, tau[aperm])


@pytest.mark.parametrize("cfg,kind,n", [(2, "id", 24), (3, "mptc", 24), (4, "mptc", 8), (5, "mptc", 8), (3, "pc", 24), (3, "id", 12),
                                        (3, "clf", 24), (2, "clf", 16), (5, "clf", 8)])
def test_hex_kernel_math_emulated_on_host(cfg, kind, n):
    """wbc_hex.hpp (16 lanes = one DPP row per robot) with the row emulated by 16 lock-step fibres."""
    b = workloads.make_batch(cfg, n=n)
    t = orc.load_model_json(b["model"])
    m = orc.model(b["model"]); p = orc.params(kind)
    tau_o, met_o, st_o = orc.step_batch(kind, m, p, b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"])
    tau, met, st, it, vd = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"],
                                  hexv=True, want_vdot=True)
    tq, mq, sq, iq, vdq = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], b["mu"], b["mass_scale"],
                                 want_vdot=True)   # the scalar (one-lane-per-robot) instantiation of the same math
    assert (st == 0).all()
    assert rel_err(tau, tau_o).max() < 1e-5
    assert np.allclose(met, met_o, rtol=1e-6, atol=1e-7)
    assert np.allclose(vd, vdq, rtol=1e-5, atol=1e-6)          # generalized accelerations agree with the other mapping


def test_hex_all_contact_masks_on_host():
    b = workloads.make_batch(3, n=16)
    t = orc.load_model_json("mini_cheetah"); m = orc.model("mini_cheetah")
    mk = np.arange(16, dtype=np.uint8)
    for kind in ("id", "mptc", "pc", "clf"):
        tau_o, met_o, st_o = orc.step_batch(kind, m, orc.params(kind), b["q"], b["v"], b["targets"], mk)
        tau, met, st, it = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], mk, hexv=True)
        assert (st == 0).all()
        assert rel_err(tau, tau_o).max() < 1e-5
        assert np.allclose(met, met_o, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("cfg,kind,tmax,mu", [(2, "id", 12.0, 0.45), (3, "mptc", 10.0, 0.7), (3, "id", 8.0, 0.7), (3, "pc", 10.0, 0.7),
                                              (3, "clf", 12.0, 0.7), (4, "mptc", 25.0, 0.7), (3, "mptc", 2.0, 0.7)])
def test_hex_torque_box(cfg, kind, tmax, mu):
    """Optional torque box |tau_j| <= tau_max on the 16-lane mapping (second constraint slot per lane): same
    torques as the literal oracle QP with its 24 torque rows, same feasibility verdicts (a 2 N.m box is infeasible
    for some states), torques clamp at the bound, and the lane-per-robot mapping agrees."""
    b = workloads.make_batch(cfg, n=32)
    t = orc.load_model_json(b["model"]); m = orc.model(b["model"])
    p = orc.params(kind); p.tau_max = tmax; p.mu = mu
    tau_o, _, st_o = orc.step_batch(kind, m, p, b["q"], b["v"], b["targets"], b["mask"])
    pp = np.array([p.Kp_body_p, p.Kd_body_p, p.Kp_body_rpy, p.Kd_body_rpy, p.Kp_foot, p.Kd_foot, p.w_body,
                   p.w_foot, p.mu, p.Kd_contact, p.tau_max, p.tiebreak_eps2])
    tl, _, sl, _ = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], params12=pp)
    th, _, sh, _ = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], params12=pp, hexv=True)
    assert np.array_equal(sh == 0, st_o == 0) and np.array_equal(sh, sl)
    ok = sh == 0
    assert ok.sum() >= 24
    assert np.abs(th[:, ok]).max() <= tmax + 1e-9
    assert (np.abs(np.abs(th[:, ok]) - tmax) < 1e-6).sum() >= 20          # the box is active on this batch
    assert rel_err(th[:, ok], tau_o[:, ok]).max() < 1e-5
    assert rel_err(th[:, ok], tl[:, ok]).max() < 1e-5


def test_straight_knee():
    """The reduced (task-coordinate) formulation inverts every leg's 3x3 foot Jacobian (|det| ~ 0.04 |sin knee|).
    ID / CLF, swing leg: a straight knee is evaluated at |sin(knee)| = 1e-8 and agrees with the dense oracle -- which,
    like the reference's full QP, solves it -- to 1e-6 (profiles/r02/singular_envelope.md).  Stance legs and MPTC / PC
    (Lambda = (J M^-1 J')^-1 is singular there in the reference itself): status 2 with zero torques.  DESIGN.md section 3."""
    b = workloads.make_batch(3, n=8)
    t = orc.load_model_json("mini_cheetah")
    q = b["q"].copy()
    q[7 + 2, 0] = 0.0          # LF knee straight
    q[7 + 3 * 2 + 2, 1] = 0.0  # LH knee straight
    q[7 + 2, 2] = 1e-12        # LF knee nearly straight
    for kind in ("id", "clf"):
        tau, met, st, it = ht.run(kind, t["flat"], q, b["v"], b["targets"], b["mask"], hexv=True)
        tau_o, _, st_o = orc.step_batch(kind, orc.model("mini_cheetah"), orc.params(kind), q, b["v"], b["targets"], b["mask"])
        assert (st == 0).all() and (st_o == 0).all()
        assert rel_err(tau, tau_o).max() < 1e-5
    for kw in ({}, {"hexv": True}):
        tau, met, st, it = ht.run("mptc", t["flat"], q, b["v"], b["targets"], b["mask"], **kw)
        assert st.tolist()[:2] == [2, 2] and (st[3:] == 0).all()
        assert (tau[:, :2] == 0).all() and np.isfinite(tau[:, [0, 1, 3, 4, 5, 6, 7]]).all()


def test_nearly_straight_knee_is_reported_as_ill_conditioned():
    """MPTC / PC invert J M^-1 J' (mptc_controller.py:237-238): below |sin(knee)| = 1e-4 the tick is solved and written
    but reported as status 3 -- by the kernel math and by the oracle alike (include/wbc.h); the ID-type laws, which agree
    with the dense restatement down to 1e-8 rad, stay at status 0.  Just above the threshold the status is 0 and the
    torques agree (profiles/r03/singular_envelope.md)."""
    b = workloads.make_batch(3, n=8)
    t = orc.load_model_json("mini_cheetah")
    q = b["q"].copy()
    q[7 + 2, 0] = 5e-5          # LF knee below the threshold
    q[7 + 3 * 3 + 2, 1] = -3e-6   # RH knee, negative side
    q[7 + 2, 2] = 2e-4          # above it
    for kind in ("mptc", "pc"):
        tau_o, _, st_o = orc.step_batch(kind, orc.model("mini_cheetah"), orc.params(kind), q, b["v"], b["targets"], b["mask"])
        for kw in ({}, {"hexv": True}):
            tau, met, st, it = ht.run(kind, t["flat"], q, b["v"], b["targets"], b["mask"], **kw)
            assert st.tolist() == [3, 3, 0, 0, 0, 0, 0, 0] == st_o.tolist()
            assert np.isfinite(tau).all() and (np.abs(tau[:, :2]).max(0) > 0).all()   # written, not zeroed
        assert rel_err(tau[:, 2:], tau_o[:, 2:]).max() < 1e-5   # 16-lane form
    for kind in ("id", "clf"):
        tau, met, st, it = ht.run(kind, t["flat"], q, b["v"], b["targets"], b["mask"], hexv=True)
        tau_o, _, st_o = orc.step_batch(kind, orc.model("mini_cheetah"), orc.params(kind), q, b["v"], b["targets"], b["mask"])
        assert (st == 0).all() and (st_o == 0).all() and rel_err(tau, tau_o).max() < 1e-5


@pytest.mark.parametrize("kind", ["id", "mptc", "pc", "clf"])
def test_malformed_instances_are_reported_with_zero_outputs(kind):
    """include/wbc.h "Malformed instances" on the host instantiation of the kernel headers and on the oracle's mirror of the convention: every
    case of tests/poisons.py in one batch (half trot states, half saturated stands).  A malformed instance: status 2, zero torques and
    accelerations, finite metrics, on both sides; an odd-looking legal one (non-unit quaternion, contact-mask bits above 0xF): status 0 and the
    clean instance's outputs; nothing non-finite anywhere; every untouched instance keeps its bits."""
    from poisons import POISONS, copy_batch, expected_status
    names = list(POISONS)
    n = 2 * len(names) + 8
    bt, bs = workloads.make_batch(5, n=n), workloads.make_batch(2, n=n)
    base = copy_batch(bt, n)
    half = np.arange(n) >= n // 2
    for k in ("q", "v", "targets"):
        base[k][:, half] = bs[k][:, half]
    base["mask"][half] = 0xF
    bad = copy_batch(base, n)
    where = {}
    for j, name in enumerate(names):
        for i in (j, n // 2 + j):              # once on a trot state, once on a stand
            POISONS[name][0](bad, i)
            where[i] = name
    flat = orc.load_model_json("mini_cheetah")["flat"]
    m, p = orc.model("mini_cheetah"), orc.params(kind)
    run = lambda b: ht.run(kind, flat, b["q"], b["v"], b["targets"], b["mask"], mu=b["mu"], mass_scale=b["mass_scale"], hexv=True, want_vdot=True)
    tau0, met0, st0, it0, vd0 = run(base)
    tau, met, st, it, vd = run(bad)
    tau_o, met_o, st_o = orc.step_batch(kind, m, p, bad["q"], bad["v"], bad["targets"], bad["mask"], bad["mu"], bad["mass_scale"])
    assert np.isfinite(tau).all() and np.isfinite(met).all() and np.isfinite(vd).all() and np.isfinite(tau_o).all() and np.isfinite(met_o).all()
    clean = np.array([i not in where for i in range(n)])
    assert (st0 == 0).all()
    for a, b in ((tau, tau0), (met, met0), (vd, vd0)):
        assert np.array_equal(a[:, clean], b[:, clean])
    assert np.array_equal(st[clean], st0[clean]) and np.array_equal(it[clean], it0[clean])
    for i, name in where.items():
        want = expected_status(name, bad["mask"][i])
        assert st[i] == st_o[i], (name, i, st[i], st_o[i])
        if want is not None:
            assert st[i] == want, (name, i, st[i])
        if st[i] == 2:
            assert (tau[:, i] == 0).all() and (vd[:, i] == 0).all() and (tau_o[:, i] == 0).all() and (it[i] == 0 or want is None), name
            assert met[2, i] == 0 and met[3, i] == 0, name
        elif want == 0:
            # a legal input: the clean instance's outputs (the quaternion's scale cancels in 2 / |q|^2 up to rounding; mask bits above 0xF are not read)
            tol = 0.0 if name in ("mask_high_bits", "nan_foot_target", "inf_foot_rate_tgt") else 1e-6     # (a saturated stand amplifies the last-bit change of R by 1e4: measured 1e-8)
            assert np.abs(tau[:, i] - tau0[:, i]).max() <= tol * np.abs(tau0[:, i]).max(), name
            assert rel_err(tau[:, i:i + 1], tau_o[:, i:i + 1]).max() < 1e-5, name


def test_pc_enforces_passivity_where_mptc_does_not():
    """pc_controller.py: Vdot <= 0 is a hard row; MPTC only logs Vdot."""
    b = workloads.make_batch(3, n=128)
    t = orc.load_model_json("mini_cheetah")
    _, met_m, _, _ = ht.run("mptc", t["flat"], b["q"], b["v"], b["targets"], b["mask"])
    tau_p, met_p, st_p, _ = ht.run("pc", t["flat"], b["q"], b["v"], b["targets"], b["mask"])
    assert (met_m[3] > 1e-6).sum() >= 3            # the batch does contain Vdot > 0 cases
    assert (st_p == 0).all() and met_p[3].max() < 1e-9


def test_fast_and_generic_paths_round_identically():
    """A robot's result may not depend on its wave-mates: leaving the active set's compile-time-q fast path at any trip
    (what a wave-mate's drop causes) must give the same bits.  Host instantiation; the device A/B is tools/dump_tau.py
    on -DWBC_DEV_FORCE_BAIL=k builds (bit-identical)."""
    L = ht.lib()
    for cfg, kind in ((2, "id"), (3, "mptc"), (3, "id"), (3, "pc"), (2, "pc")):
        b = workloads.make_batch(cfg, n=32)
        t = orc.load_model_json(b["model"])
        L.host_gi_force_bail(-1)
        ref = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], hexv=True)
        try:
            for k in range(8):
                L.host_gi_force_bail(k)
                r = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], hexv=True)
                assert np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1]) and np.array_equal(r[3], ref[3]), (kind, k)
        finally:
            L.host_gi_force_bail(-1)


def test_evaluation_after_a_drop_and_graded_pivots_on_host(tmp_path):
    """Round 4's two repairs of the saturated-stand error (csrc/wbc_hex.hpp; profiles/r04/accuracy.md), each against a build without it:
    the evaluation of z from the rotated right-hand side for robots that dropped a row (-DWBC_NO_DROP_REFINE) and the graded pivot
    order of the QR factor (-DWBC_NATURAL_PIVOTS).  On 4-contact stands (768 of them, the seeds of the round's analysis) the default
    build is within 2e-7 of the oracle compiled in extended precision; a build without either repair is 10x further off on its worst
    robot.  On a trot batch the evaluation never runs (two feet down: one internal-force direction): same bits with and without it."""
    import ctypes as C
    import os
    import subprocess
    from oracle import oracle_ld as old
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dp = C.POINTER(C.c_double)

    def build(flags, name):
        return ht.build_variant(tmp_path, name, flags)

    def run(L, kind, b):
        n = b["q"].shape[1]
        t = orc.load_model_json(b["model"])
        q, v, tg = (np.ascontiguousarray(b[x]) for x in ("q", "v", "targets"))
        flat = np.ascontiguousarray(t["flat"], dtype=np.float64)
        tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
        rc = L.host_hex_batch({"id": 0, "mptc": 1}[kind], flat.ctypes.data_as(dp), None, None, None, n, n, q.ctypes.data_as(dp),
                              v.ctypes.data_as(dp), tg.ctypes.data_as(dp), b["mask"].ctypes.data_as(C.POINTER(C.c_ubyte)), None, None,
                              tau.ctypes.data_as(dp), met.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int)),
                              it.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0 and (st == 0).all()
        return tau

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:                          # the two compiles side by side
        fa = ex.submit(build, ["-DWBC_NO_DROP_REFINE", "-DWBC_NATURAL_PIVOTS"], "libhost_r3.so")
        fb = ex.submit(build, ["-DWBC_NO_DROP_REFINE"], "libhost_norefine.so")
        plain, norefine = fa.result(), fb.result()
    b = workloads.make_batch(2, n=768, seed=50002)
    tau_l, _, st_l = old.step_batch("id", old.model(b["model"]), old.params("id"), b["q"], b["v"], b["targets"], b["mask"])
    assert (st_l == 0).all()
    tau_l = tau_l.astype(np.float64)
    tau, _, st, _ = ht.run("id", orc.load_model_json(b["model"])["flat"], b["q"], b["v"], b["targets"], b["mask"], hexv=True)
    e_new, e_old = rel_err(tau, tau_l).max(), rel_err(run(plain, "id", b), tau_l).max()
    assert e_new < 2e-7 and e_old > 10.0 * e_new, (e_new, e_old)
    # trots: the evaluation does not run -- bit-identical with and without it
    bt = workloads.make_batch(3, n=128)
    tau_t, _, _, _ = ht.run("mptc", orc.load_model_json(bt["model"])["flat"], bt["q"], bt["v"], bt["targets"], bt["mask"], hexv=True)
    assert np.array_equal(tau_t, run(norefine, "mptc", bt))


def test_swing_row_compaction_changes_no_bit(tmp_path):
    """Task-space laws: a wavefront without a robot of three or four swing legs appends 24 rows (each robot's swing blocks
    moved up into the first two block slots, csrc/wbc_hex.hpp) instead of 30 with zero rows.  Blocks move by multiples of three rows, so every
    term keeps its accumulator and its place in the order of the sums: against a build without the compaction
    (-DWBC_NO_SWING_COMPACT) every contact mask gives the same bits."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    L = ht.build_variant(tmp_path, "libhost_tick_nocompact.so", ["-DWBC_NO_SWING_COMPACT"])
    dp = C.POINTER(C.c_double)
    n = 64
    b = workloads.make_batch(3, n=n)
    t = orc.load_model_json(b["model"])
    mk = (np.arange(n) % 16).astype(np.uint8)          # every contact mask, swing counts 0..4
    q, v, tg = (np.ascontiguousarray(b[x]) for x in ("q", "v", "targets"))
    flat = np.ascontiguousarray(t["flat"], dtype=np.float64)
    for kind, k in (("mptc", 1), ("pc", 2)):
        tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
        rc = L.host_hex_batch(k, flat.ctypes.data_as(dp), None, None, None, n, n, q.ctypes.data_as(dp), v.ctypes.data_as(dp),
                              tg.ctypes.data_as(dp), mk.ctypes.data_as(C.POINTER(C.c_ubyte)), None, None,
                              tau.ctypes.data_as(dp), met.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int)),
                              it.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0
        tau_c, met_c, st_c, it_c = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], mk, hexv=True)
        assert (st == 0).all() and (st_c == 0).all() and np.array_equal(it, it_c)
        assert np.array_equal(tau_c, tau) and np.array_equal(met_c, met)


@pytest.mark.parametrize("kind,tmax", [("pc", None), ("id", 8.0), ("pc", 8.0), ("mptc", 8.0)])
def test_evaluation_after_a_drop_with_inhomogeneous_rows_on_host(kind, tmax):
    """The evaluation after a drop (csrc/wbc_hex.hpp, end of hex_gi) has a branch for INHOMOGENEOUS active rows -- the PC law's dense
    row Vdot <= 0 and the torque box's rows put g = sum beta_a W_a into the used slots -- which the ID stands of the test above never
    enter.  4-contact stands (drop-heavy: 10 - 19 trips per robot) under the PC law and under a box of 8 N m that binds on most of
    them, against the oracle compiled in extended precision: within 2e-7 (measured 2e-8 .. 8e-8), every status equal.  Runs the
    host instantiation, whose asserts also check that lane (0, 3) -- which carries y = Q'b in its row slots -- contributes
    nothing to any 16-lane reduction."""
    from oracle import oracle_ld as old
    b = workloads.make_batch(2, n=256, seed=50002)
    pl = old.params(kind)
    po = orc.params(kind)
    names = ("Kp_body_p", "Kd_body_p", "Kp_body_rpy", "Kd_body_rpy", "Kp_foot", "Kd_foot", "w_body", "w_foot", "mu", "Kd_contact",
             "tau_max", "tiebreak_eps2")
    pp = np.array([getattr(po, k) for k in names])
    if tmax is not None:
        pl.tau_max = tmax; pp[10] = tmax
    tau_l, _, st_l = old.step_batch(kind, old.model(b["model"]), pl, b["q"], b["v"], b["targets"], b["mask"])
    tau, _, st, it = ht.run(kind, orc.load_model_json(b["model"])["flat"], b["q"], b["v"], b["targets"], b["mask"], params12=pp, hexv=True)
    assert np.array_equal(st, st_l) and (st == 0).all()
    assert it.mean() > 8                                                       # drop-heavy: the branch is exercised
    assert rel_err(tau, tau_l.astype(np.float64)).max() < 2e-7
    if tmax is not None:
        assert (np.abs(tau).max(0) > tmax * (1 - 1e-9)).sum() > 100 and np.abs(tau).max() <= tmax * (1 + 1e-9)


def test_apex_rule_and_deterministic_pick_on_host(tmp_path):
    """Round 5 (csrc/wbc_hex.hpp: hex_gi, hex_pack_pick; profiles/r05/apex_rule.md).  The four friction rows of a foot are linearly dependent
    (n_0 + n_1 = n_2 + n_3): with three active the fourth is identically zero, and with one x-side and one y-side row active the two others
    tie exactly.  Rounding used to decide both: noise-violated fourth rows were picked, found dependent and exchanged (a drop and an add), and
    the tie went either way.  Now a row whose three leg-mates are active is no candidate and the pick quantises its keys (lower index wins a
    tie).  Against a build without either (-DWBC_NO_APEX_RULE -DWBC_PICK_BITS=5): the same solutions, fewer trips on the saturated stands of
    BASELINE config 2, never more than a handful more on any robot; the trot batch keeps its solutions too."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dp = C.POINTER(C.c_double)
    old = ht.build_variant(tmp_path, "libhost_noapex.so", ["-DWBC_NO_APEX_RULE", "-DWBC_PICK_BITS=5"])

    def run_old(kind, b):
        n = b["q"].shape[1]
        t = orc.load_model_json(b["model"])
        q, v, tg = (np.ascontiguousarray(b[x]) for x in ("q", "v", "targets"))
        flat = np.ascontiguousarray(t["flat"], dtype=np.float64)
        tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
        rc = old.host_hex_batch({"id": 0, "mptc": 1}[kind], flat.ctypes.data_as(dp), None, None, None, n, n, q.ctypes.data_as(dp),
                                v.ctypes.data_as(dp), tg.ctypes.data_as(dp), b["mask"].ctypes.data_as(C.POINTER(C.c_ubyte)), None, None,
                                tau.ctypes.data_as(dp), met.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int)),
                                it.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0 and (st == 0).all()
        return tau, it

    b = workloads.make_batch(2, n=512)
    tau_new, _, st, it_new = ht.run("id", orc.load_model_json(b["model"])["flat"], b["q"], b["v"], b["targets"], b["mask"], hexv=True)
    tau_old, it_old = run_old("id", b)
    assert (st == 0).all()
    assert rel_err(tau_new, tau_old).max() < 1e-6                        # the same QP solution (two descriptions of an apex, at most)
    assert it_new.mean() < it_old.mean() - 0.2 and it_new.max() <= it_old.max()
    assert (it_new - it_old).max() <= 6                                  # no robot pays much for the other tie-break
    bt = workloads.make_batch(3, n=512)
    tau_new, _, st, it_new = ht.run("mptc", orc.load_model_json(bt["model"])["flat"], bt["q"], bt["v"], bt["targets"], bt["mask"], hexv=True)
    tau_old, it_old = run_old("mptc", bt)
    assert (st == 0).all() and rel_err(tau_new, tau_old).max() < 1e-6 and it_new.mean() <= it_old.mean() + 0.02


def test_pick_rule_by_contact_count_on_the_dense_row_laws(tmp_path):
    """Round 5 (csrc/wbc_hex.hpp: hex_gi, HYB; profiles/r05/hybrid_pick.md).  PC and CLF choose the row to add per robot: greatest dual gain with one
    or two feet down, the most violated row with three or four.  Against a build without it (-DWBC_HYBRID_PICK=0: most violated everywhere): the same
    solutions, fewer trips and fewer drops on the trot batch, bit-identical outputs on the 4-contact stands (every robot there is `deep`)."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dp = C.POINTER(C.c_double)
    old = ht.build_variant(tmp_path, "libhost_nohybrid.so", ["-DWBC_HYBRID_PICK=0"])

    def run_old(kind, b):
        n = b["q"].shape[1]
        t = orc.load_model_json(b["model"])
        q, v, tg = (np.ascontiguousarray(b[x]) for x in ("q", "v", "targets"))
        flat = np.ascontiguousarray(t["flat"], dtype=np.float64)
        tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
        rc = old.host_hex_batch({"pc": 2, "clf": 3}[kind], flat.ctypes.data_as(dp), None, None, None, n, n, q.ctypes.data_as(dp),
                                v.ctypes.data_as(dp), tg.ctypes.data_as(dp), b["mask"].ctypes.data_as(C.POINTER(C.c_ubyte)), None, None,
                                tau.ctypes.data_as(dp), met.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int)),
                                it.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0
        return tau, st, it

    for kind in ("pc", "clf"):
        bt = workloads.make_batch(3, n=256)
        tau_new, _, st, it_new = ht.run(kind, orc.load_model_json(bt["model"])["flat"], bt["q"], bt["v"], bt["targets"], bt["mask"], hexv=True)
        tau_old, st_old, it_old = run_old(kind, bt)
        assert (st == 0).all() and (st_old == 0).all()
        assert rel_err(tau_new, tau_old).max() < 1e-7                       # the same QP solution by another walk
        assert it_new.mean() < it_old.mean() - 0.03 and it_new.max() <= it_old.max()
        assert it_new.reshape(-1, 4).max(1).mean() < it_old.reshape(-1, 4).max(1).mean()      # lock step of four: what the device pays
        bs = workloads.make_batch(2, n=64)
        tau_new, _, st, it_new = ht.run(kind, orc.load_model_json(bs["model"])["flat"], bs["q"], bs["v"], bs["targets"], bs["mask"], hexv=True)
        tau_old, st_old, it_old = run_old(kind, bs)
        assert np.array_equal(tau_new, tau_old) and np.array_equal(it_new, it_old) and np.array_equal(st, st_old)


def test_clf_row_built_lazily(tmp_path):
    """Round 5 (csrc/wbc_hex.hpp: LAZY; profiles/r05/lazy_dense.md): CLF takes its dense row's VALUE fresh from z at every pick and builds the row's image
    only in a trip that adds it, instead of reflecting the image through every trip.  Against a build that reflects it (-DWBC_LAZY_DENSE=0): bit-identical
    where the row never binds (the BASELINE batches), the same solution to 1e-10 where it does (body targets pushed 20 x further out: the walk of a few
    robots adds the row), and both within 1e-7 of the oracle."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dp = C.POINTER(C.c_double)
    old = ht.build_variant(tmp_path, "libhost_nolazy.so", ["-DWBC_LAZY_DENSE=0"])

    def run_old(b, tg):
        n = b["q"].shape[1]
        t = orc.load_model_json(b["model"])
        q, v, tg = (np.ascontiguousarray(x) for x in (b["q"], b["v"], tg))
        flat = np.ascontiguousarray(t["flat"], dtype=np.float64)
        tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
        rc = old.host_hex_batch(3, flat.ctypes.data_as(dp), None, None, None, n, n, q.ctypes.data_as(dp), v.ctypes.data_as(dp), tg.ctypes.data_as(dp),
                                b["mask"].ctypes.data_as(C.POINTER(C.c_ubyte)), None, None, tau.ctypes.data_as(dp), met.ctypes.data_as(dp),
                                st.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0
        return tau, st, it

    b = workloads.make_batch(3, n=64)
    flat = orc.load_model_json(b["model"])["flat"]
    tau_new, _, st, it_new = ht.run("clf", flat, b["q"], b["v"], b["targets"], b["mask"], hexv=True)
    tau_old, st_old, it_old = run_old(b, b["targets"])
    assert np.array_equal(tau_new, tau_old) and np.array_equal(it_new, it_old) and np.array_equal(st, st_old)     # the row never binds here
    tg = b["targets"].copy()
    tg[0:6] = b["targets"][0:6] + (b["targets"][0:6] - b["targets"][0:6].mean(1, keepdims=True)) * 19.0
    tau_new, _, st, it_new = ht.run("clf", flat, b["q"], b["v"], tg, b["mask"], hexv=True)
    tau_old, st_old, it_old = run_old(b, tg)
    differ = (np.abs(tau_new - tau_old).max(0) > 0).sum()
    assert 1 <= differ <= 16 and rel_err(tau_new, tau_old).max() < 1e-10          # a few walks add the row: a freshly built image instead of a reflected one
    assert np.array_equal(it_new, it_old) and np.array_equal(st, st_old) and (st == 0).all()
    tau_o, _, st_o = orc.step_batch("clf", orc.model(b["model"]), orc.params("clf"), b["q"], b["v"], tg, b["mask"])
    assert (st_o == 0).all() and rel_err(tau_new, tau_o).max() < 1e-7 and rel_err(tau_old, tau_o).max() < 1e-7


def test_no_unwritten_storage_feeds_the_arithmetic(tmp_path):
    """Round 6: under LAZY the CLF law's dense-row image Dpc[] is written only where the row is being added or is active, and the
    evaluation after a drop multiplied it by a zero multiplier everywhere else -- 0 x stale storage, which is NaN whenever the stale bits are a NaN
    pattern (status 0, NaN torques on 178 of 256 stands when poisoned).  The host instantiation of the same header, compiled by clang once as it is
    and once with every automatic variable pre-filled with the 0xFF..FF pattern (a NaN for a double), must give the same bits on the drop-heavy
    stands and on the trots of every law, with and without the torque box: no value that was never written reaches an instruction that computes.
    (Both builds by the same compiler: gcc and clang round a handful of unannotated expressions differently, 1e-10 on the torques.)"""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        pytest.skip("no clang++ for -ftrivial-auto-var-init=pattern")
    builds = []
    for name, extra in (("plain", []), ("poison", ["-ftrivial-auto-var-init=pattern"])):      # the two compiles side by side
        so = str(tmp_path / ("libhost_tick_%s.so" % name))
        builds.append((so, subprocess.Popen([clang, "-O2", "-std=c++20", "-pthread", "-fPIC", "-shared", "-ffp-contract=off", "-DHOST_TICK_HEX_ONLY"] + extra +
                                            ["-o", so, os.path.join(root, "tools", "host_tick.cpp")])))
    libs = []
    for so, proc in builds:
        assert proc.wait() == 0
        libs.append(C.CDLL(so))
    dp = C.POINTER(C.c_double)
    names = ("Kp_body_p", "Kd_body_p", "Kp_body_rpy", "Kd_body_rpy", "Kp_foot", "Kd_foot", "w_body", "w_foot", "mu", "Kd_contact", "tau_max", "tiebreak_eps2")
    n = 128
    for cfg, kind, k, tmax in ((2, "clf", 3, None), (3, "clf", 3, None), (2, "pc", 2, None), (3, "pc", 2, None), (2, "id", 0, None), (2, "mptc", 1, None),
                               (3, "mptc", 1, None), (2, "clf", 3, 12.0), (2, "id", 0, 8.0)):
        b = workloads.make_batch(cfg, n=n)
        t = orc.load_model_json(b["model"])
        pp = None
        if tmax is not None:
            p = orc.params(kind); p.tau_max = tmax
            pp = np.array([getattr(p, f) for f in names], dtype=np.float64)
        q, v, tg = (np.ascontiguousarray(b[x]) for x in ("q", "v", "targets"))
        flat = np.ascontiguousarray(t["flat"], dtype=np.float64)
        out = []
        for L in libs:
            tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
            rc = L.host_hex_batch(k, flat.ctypes.data_as(dp), pp.ctypes.data_as(dp) if pp is not None else None, None, None, n, n, q.ctypes.data_as(dp),
                                  v.ctypes.data_as(dp), tg.ctypes.data_as(dp), b["mask"].ctypes.data_as(C.POINTER(C.c_ubyte)), None, None,
                                  tau.ctypes.data_as(dp), met.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)))
            assert rc == 0
            out.append((tau, met, st, it))
        (tau, met, st, it), (tau_p, met_p, st_p, it_p) = out
        assert np.isfinite(tau_p).all() and np.isfinite(met_p).all(), (cfg, kind, int(np.isnan(tau_p).any(0).sum()))
        assert np.array_equal(st, st_p) and np.array_equal(it, it_p), (cfg, kind, tmax)
        assert np.array_equal(tau, tau_p) and np.array_equal(met, met_p), (cfg, kind, tmax)
        if cfg == 2:
            assert it.mean() > 8          # drop-heavy: the evaluation after a drop runs


def test_closed_switches_patch_applies_and_is_neutral(tmp_path):
    """tools/lab/patches/closed_switches.patch puts the compile-time switches of rounds 1-5 (alternatives that were measured and closed; removed from
    the product source in round 6) back into a copy of the headers.  It must apply to the current tree, and without any -D flag the patched tree must
    still be the product's arithmetic: same bits as the ordinary host build on a stand and a trot of every law."""
    import ctypes as C
    L = ht.build_variant(tmp_path, "libhost_patched.so", [])
    dp = C.POINTER(C.c_double)
    n = 48
    for cfg in (2, 3):
        b = workloads.make_batch(cfg, n=n)
        t = orc.load_model_json(b["model"])
        q, v, tg = (np.ascontiguousarray(b[x]) for x in ("q", "v", "targets"))
        flat = np.ascontiguousarray(t["flat"], dtype=np.float64)
        for kind, k in (("id", 0), ("mptc", 1), ("pc", 2), ("clf", 3)):
            tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); it = np.zeros(n, np.int32)
            rc = L.host_hex_batch(k, flat.ctypes.data_as(dp), None, None, None, n, n, q.ctypes.data_as(dp), v.ctypes.data_as(dp), tg.ctypes.data_as(dp),
                                  b["mask"].ctypes.data_as(C.POINTER(C.c_ubyte)), None, None, tau.ctypes.data_as(dp), met.ctypes.data_as(dp),
                                  st.ctypes.data_as(C.POINTER(C.c_int)), it.ctypes.data_as(C.POINTER(C.c_int)))
            assert rc == 0
            tau_g, met_g, st_g, it_g = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], hexv=True)
            assert np.array_equal(tau, tau_g) and np.array_equal(met, met_g) and np.array_equal(st, st_g) and np.array_equal(it, it_g), (cfg, kind)
