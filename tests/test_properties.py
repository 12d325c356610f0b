"""Property tests (hypothesis): on states drawn from the synthetic distributions' envelope the
solution satisfies the reference QP's own rows -- dynamics residual, contact rows, friction pyramid -- and results do
not depend on the batch position.  CPU: the oracle's literal QP and the kernel math instantiated on the host;
GPU (-m gpu): the device outputs, with the contact forces RECOVERED from (tau, vd) through the oracle's dynamics
terms (M vd + Cv + tau_g = S'tau + sum J_c' f, basic_controller.py:106), so nothing of the kernel's own algebra is reused."""
import numpy as np
import pytest
from hypothesis import HealthCheck, assume, given, settings, strategies as st

import host_tick as ht
from oracle import oracle_py as orc
from quadruped_drake_amd import workloads

MASKS = [0b1111, 0b1001, 0b0110, 0b0111, 0b1011, 0b0011, 0b1000]


def draw_state(seed, mask, model="mini_cheetah", scale=1.0):
    """One state + targets in the envelope of workloads.make_batch (scale widens velocities and target offsets)."""
    rng = np.random.default_rng(seed)
    b = workloads.make_batch(3 if model == "mini_cheetah" else 4, n=1, seed=int(rng.integers(1 << 30)), model=model)
    b["v"] *= scale
    b["targets"][3:9] *= scale
    b["mask"][:] = mask
    # swing-foot height targets follow the mask (make_batch drew them for its own mask)
    for i in range(4):
        b["targets"][18 + 9 * i + 2] = 0.0 if (mask >> i) & 1 else 0.07
    return b


def recover_forces(m, q, v, tau_canon, vd, mask):
    """f_l = J_l^-T ((M vd + Cv + tau_g)_leg - tau_l) for contact legs; returns forces and the base-row residual."""
    M, Cv, tg = orc.calc_dynamics(m, q, v)
    lhs = M @ vd + Cv + tg
    f = np.zeros((4, 3))
    base = lhs[:6].copy()
    for l in range(4):
        _, J, _ = orc.foot_quantities(m, q, v, l)
        rows = slice(6 + 3 * l, 9 + 3 * l)
        r = lhs[rows] - tau_canon[3 * l:3 * l + 3]
        if (mask >> l) & 1:
            f[l] = np.linalg.solve(J[:, rows].T, r)
            base -= J[:, :6].T @ f[l]
        else:
            assert np.abs(r).max() < 1e-7 * (1 + np.abs(lhs).max())      # swing leg: joint rows close without a force
    return f, base


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), mask=st.sampled_from(MASKS), kind=st.sampled_from(["id", "mptc", "pc", "clf"]),
       mu=st.floats(0.3, 1.2), scale=st.floats(0.2, 2.0))
def test_oracle_solution_satisfies_the_reference_rows(seed, mask, kind, mu, scale):
    b = draw_state(seed, mask, scale=scale)
    m = orc.model("mini_cheetah"); p = orc.params(kind); p.mu = mu
    ct = [(mask >> k) & 1 for k in range(4)]
    tau, met, stt, qp = orc.control_law(kind, m, p, b["q"][:, 0], b["v"][:, 0], b["targets"][:, 0], ct, want_qp=True)
    if kind == "pc" and stt == 2:
        assume(False)      # the PC law's hard row Vdot <= 0 can be infeasible under the friction limits (the reference asserts there)
    assert stt == 0
    x = qp["x"]
    sc = 1.0 + np.abs(x).max()
    assert np.abs(qp["Aeq"] @ x - qp["beq"]).max() < 1e-9 * sc          # dynamics + contact rows
    assert (qp["Ain"] @ x - qp["bin"]).max() < 1e-9 * sc                 # friction pyramid (+ PC / CLF rows)
    nc = qp["nc"]
    f = x[30:30 + 3 * nc].reshape(nc, 3)
    assert (np.abs(f[:, 0]) <= mu * f[:, 2] + 1e-9 * sc).all() and (np.abs(f[:, 1]) <= mu * f[:, 2] + 1e-9 * sc).all()
    assert np.allclose(tau, x[18:30], atol=0)                            # identity act_perm: tau is the QP's tau block


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), mask=st.sampled_from(MASKS), kind=st.sampled_from(["id", "mptc", "pc", "clf"]),
       mu=st.floats(0.3, 1.2), pos=st.integers(0, 4))
def test_kernel_math_matches_oracle_and_is_batch_position_invariant(seed, mask, kind, mu, pos):
    """The 16-lane kernel math (host fibre emulation): same torques as the oracle; the same robot placed at another
    batch position, among different neighbours, gives bit-identical outputs."""
    b = draw_state(seed, mask)
    t = orc.load_model_json("mini_cheetah")
    m = orc.model("mini_cheetah"); p = orc.params(kind); p.mu = mu
    pp = np.array([p.Kp_body_p, p.Kd_body_p, p.Kp_body_rpy, p.Kd_body_rpy, p.Kp_foot, p.Kd_foot, p.w_body, p.w_foot,
                   p.mu, p.Kd_contact, p.tau_max, p.tiebreak_eps2])
    ct = [(mask >> k) & 1 for k in range(4)]
    tau_o, met_o, st_o = orc.control_law(kind, m, p, b["q"][:, 0], b["v"][:, 0], b["targets"][:, 0], ct)
    others = workloads.make_batch(3, n=5, seed=seed % 1000)
    q = others["q"].copy(); v = others["v"].copy(); tg = others["targets"].copy(); mk = others["mask"].copy()
    q[:, pos] = b["q"][:, 0]; v[:, pos] = b["v"][:, 0]; tg[:, pos] = b["targets"][:, 0]; mk[pos] = mask
    tau, met, stt, _ = ht.run(kind, t["flat"], q, v, tg, mk, params12=pp, hexv=True)
    tau1, met1, st1, _ = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], params12=pp, hexv=True)
    if kind == "pc" and st_o == 2:
        assert stt[pos] == 2           # infeasible passivity row: both report it
        return
    assert st_o == 0 and stt[pos] == 0
    assert np.array_equal(tau[:, pos], tau1[:, 0]) and np.array_equal(met[:, pos], met1[:, 0])
    assert np.abs(tau[:, pos] - tau_o).max() < 1e-4 * max(np.abs(tau_o).max(), 1e-3)


@pytest.mark.gpu
@settings(max_examples=12, deadline=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), kind=st.sampled_from(["id", "mptc", "pc", "clf"]), mu=st.floats(0.3, 1.2),
       scale=st.floats(0.2, 2.0))
def test_gpu_outputs_satisfy_dynamics_and_friction(seed, kind, mu, scale):
    """Device outputs (tau, vd) of a 28-robot batch covering every contact mask with >= 1 contact: the recovered
    contact forces close the base rows of the dynamics and lie inside the friction pyramid; swing legs need no force;
    permuting the batch permutes the outputs bit for bit."""
    import torch
    from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
    cls = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[kind]
    masks = [mk for mk in range(1, 16) if not (kind in ("mptc", "pc") and mk == 0)] * 2
    n = len(masks)
    parts = [draw_state(seed + 17 * i, mk, scale=scale) for i, mk in enumerate(masks)]
    q = np.concatenate([p_["q"] for p_ in parts], 1); v = np.concatenate([p_["v"] for p_ in parts], 1)
    tg = np.concatenate([p_["targets"] for p_ in parts], 1); mk = np.array(masks, dtype=np.uint8)
    ctrl = cls(max_batch=n, device=0, params={"mu": mu})
    up = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda:0")
    vd = torch.zeros((18, n), dtype=torch.float64, device="cuda:0")
    ctrl.set_vdot_output(vd)
    tau, met, stt = ctrl.step(up(q), up(v), up(tg), up(mk))
    ctrl.sync()
    tau = tau.cpu().numpy(); vdn = vd.cpu().numpy(); stn = stt.cpu().numpy()
    perm = np.random.default_rng(seed).permutation(n)
    tau2, _, _ = ctrl.step(up(q[:, perm]), up(v[:, perm]), up(tg[:, perm]), up(mk[perm]))
    ctrl.sync()
    assert np.array_equal(tau2.cpu().numpy(), tau[:, perm])
    ctrl.set_vdot_output(None)
    ctrl.close()
    m = orc.model("mini_cheetah")
    if kind != "pc":
        assert (stn == 0).all()
    for i in range(n):
        if stn[i] != 0:
            continue                   # PC: an infeasible passivity row is reported (zero torques), nothing to check
        f, base = recover_forces(m, q[:, i], v[:, i], tau[:, i], vdn[:, i], int(mk[i]))
        sc = 1.0 + np.abs(tau[:, i]).max() + np.abs(f).max()
        assert np.abs(base).max() < 1e-7 * sc, (i, masks[i], np.abs(base).max())          # base rows of the dynamics close
        for l in range(4):
            if (mk[i] >> l) & 1:
                assert abs(f[l, 0]) <= mu * f[l, 2] + 1e-7 * sc and abs(f[l, 1]) <= mu * f[l, 2] + 1e-7 * sc


# ---- wire formats and planner dictionaries (callers of the path): round-trip properties on arbitrary values
f32 = st.floats(width=32, allow_nan=False, allow_infinity=True)


@settings(max_examples=60, deadline=None)
@given(st.lists(f32, min_size=49, max_size=49))
def test_robot_state_codec_round_trips_bit_for_bit(vals):
    """robot_state_control_lcmt: C codec == numpy/struct oracle on arbitrary float32 payloads (infinities, zeros of both
    signs, denormals), and decode(encode(x)) == x."""
    import struct
    from oracle import traj_oracle as to
    from quadruped_drake_amd.lcm_io import decode_robot_state, encode_robot_state
    x = np.array(vals, dtype=np.float32)
    b = encode_robot_state(x[:19], x[19:37], x[37:])
    assert b == to.RS_FINGERPRINT + struct.pack(">49f", *x.tolist())
    d = decode_robot_state(b)
    assert np.concatenate([d["q"], d["v"], d["tau"]]).tobytes() == x.tobytes()
    q, v, tau = to.robot_state_decode(b)
    assert np.concatenate([q, v, tau]).astype(np.float32).tobytes() == x.tobytes()


@settings(max_examples=60, deadline=None)
@given(st.lists(st.floats(min_value=-1e6, max_value=1e6, allow_nan=False), min_size=54, max_size=54), st.integers(0, 15))
def test_trunk_dictionary_pack_unpack_are_inverse(vals, mask):
    from quadruped_drake_amd import pack_trunk_input, unpack_trunk_input
    t = np.array(vals)
    d = unpack_trunk_input(t, mask)
    assert set(d) == {"p_body", "pd_body", "pdd_body", "rpy_body", "rpyd_body", "rpydd_body", "contact_states", "f_cj", "u2_max"} | \
        {pre + f for f in ("lf", "rf", "lh", "rh") for pre in ("p_", "pd_", "pdd_")}        # the keys of planners/simple.py:45-85
    t2, m2 = pack_trunk_input(d)
    assert t2.tobytes() == t.tobytes() and m2 == mask
