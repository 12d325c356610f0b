"""Callers of the path: trunk_state_t wire decode and the nearest-timestamp
target lookup.  Byte/index work: BIT-EXACT.  The wire fixtures were produced with the reference's own
generated LCM encoder (tests/golden/make_trunk_state_golden.py), so this parity is pinned by the
reference itself."""
import os

import numpy as np
import pytest

from oracle import traj_oracle as to

HERE = os.path.dirname(os.path.abspath(__file__))


def load_msgs():
    raw = open(os.path.join(HERE, "golden", "trunk_state_msgs.bin"), "rb").read()
    assert len(raw) % 549 == 0
    exp = np.load(os.path.join(HERE, "golden", "trunk_state_expected.npz"))
    return [raw[i:i + 549] for i in range(0, len(raw), 549)], exp


def test_oracle_decode_matches_reference_encoder():
    msgs, exp = load_msgs()
    for k, b in enumerate(msgs):
        d = to.decode(b)
        assert d["timestamp"] == exp["timestamp"][k] and d["finished"] == bool(exp["finished"][k])
        body = np.stack([d[n] for n in ("base_p", "base_pd", "base_pdd", "base_rpy", "base_rpyd", "base_rpydd")])
        feet = np.stack([d[n] for n in ("lf_p", "rf_p", "lh_p", "rh_p", "lf_pd", "rf_pd", "lh_pd", "rh_pd",
                                        "lf_pdd", "rf_pdd", "lh_pdd", "rh_pdd")])
        assert np.array_equal(body, exp["body"][k]) and np.array_equal(feet, exp["feet"][k])
        assert d["contact"] == [bool(c) for c in exp["contact"][k]] and np.array_equal(d["foot_f"], exp["f"][k])


def test_c_decoder_is_bit_exact_and_rejects_bad_input():
    from quadruped_drake_amd.trajectory import decode_trunk_state
    msgs, exp = load_msgs()
    for k, b in enumerate(msgs):
        d = decode_trunk_state(b)
        o = to.decode(b)
        t_o, m_o = to.to_targets(o)
        assert d["timestamp"] == o["timestamp"] and d["finished"] == o["finished"]
        assert np.array_equal(d["targets"], t_o) and d["contact_mask"] == m_o        # bit-exact, incl. denormals / -0.0
        assert np.array_equal(d["foot_f"], o["foot_f"]) and d["contact"] == o["contact"]
        assert np.array_equal(d["foot_p"], np.stack([o[f + "_p"] for f in ("lf", "rf", "lh", "rh")]))
    assert np.signbit(decode_trunk_state(msgs[7])["base_p"][1])                    # -0.0 survives
    with pytest.raises(ValueError, match="Decode error"):                           # trunk_state_t.py:87-88
        decode_trunk_state(b"\x00" * 8 + msgs[0][8:])
    with pytest.raises(ValueError):
        decode_trunk_state(msgs[0][:548])


def test_lookup_oracle_semantics():
    """planners/towr.py:96-106: wait, nearest, first index on ties and duplicates, clamping."""
    ts = np.array([0.0, 0.25, 0.5, 0.5, 1.0])          # binary-exact values so that ties are real ties
    table = np.arange(5 * 54, dtype=float).reshape(5, 54); masks = np.array([1, 2, 3, 4, 5], np.uint8)
    st = -np.ones(54)
    out, mk = to.lookup([0.5, 1.0, 1.125, 1.375, 1.5, 1.75, 1.8125, 99.0], ts, table, masks, st, 15, 1.0)
    # 1.125: tie 0.0/0.25 -> first; 1.5: duplicate timestamps -> first; 1.75: tie 0.5/0.5/1.0 -> first
    assert list(mk) == [15, 1, 1, 2, 3, 3, 5, 5]
    assert np.array_equal(out[:, 0], st) and np.array_equal(out[:, 4], table[2])


@pytest.mark.gpu
def test_device_lookup_bit_exact_vs_oracle():
    import torch
    from quadruped_drake_amd.trajectory import TrunkTrajectory, decode_trunk_state
    msgs, _ = load_msgs()
    traj = TrunkTrajectory.from_messages(msgs[:60], wait_time=1.0, device=0)        # sorted part of the fixture
    dec = [to.decode(m) for m in msgs[:60]]
    ts = np.array([d["timestamp"] for d in dec]); tab = np.stack([to.to_targets(d)[0] for d in dec])
    mk = np.array([to.to_targets(d)[1] for d in dec], np.uint8)
    from quadruped_drake_amd import workloads
    st = workloads.standing_targets("mini_cheetah", 1)[:, 0]
    rng = np.random.default_rng(0)
    t = np.concatenate([rng.uniform(0, 1.1, 3000), 1.0 + ts, 1.0 + ts + 0.0005, [0.0, 0.999999, 1.0, 50.0]])
    tg, m = traj.lookup(torch.tensor(t, device="cuda:0"))
    torch.cuda.synchronize()
    out_o, mk_o = to.lookup(t, ts, tab, mk, st, 0b1111, 1.0)
    assert np.array_equal(tg.cpu().numpy(), out_o) and np.array_equal(m.cpu().numpy(), mk_o)
    # duplicates + non-uniform spacing
    ts2 = np.array([0.0, 0.25, 0.5, 0.5, 1.0]); tab2 = rng.normal(size=(5, 54)); mk2 = np.array([1, 2, 3, 4, 5], np.uint8)
    tr2 = TrunkTrajectory(ts2, tab2, mk2, wait_time=0.5, device=0)
    t2 = np.array([0.2, 0.5, 0.625, 0.875, 1.0, 1.25, 1.3125, 9.0, 0.55, 0.65, 0.8])
    tg2, m2 = tr2.lookup(torch.tensor(t2, device="cuda:0"))
    o2, k2 = to.lookup(t2, ts2, tab2, mk2, st, 0b1111, 0.5)
    assert np.array_equal(tg2.cpu().numpy(), o2) and np.array_equal(m2.cpu().numpy(), k2)
    with pytest.raises(Exception):
        TrunkTrajectory([0.2, 0.1], np.zeros((2, 54)), [1, 1], device=0)              # unsorted timestamps


@pytest.mark.gpu
def test_lookup_feeds_the_controller():
    """Closed loop of the two pieces: trajectory lookup -> wbc_step, all on the device."""
    import torch
    from quadruped_drake_amd import MPTCController, workloads
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    from oracle import oracle_py as orc
    msgs, _ = load_msgs()
    dec = [to.decode(m) for m in msgs[:60]]
    b = workloads.make_batch(3, n=128)
    # build a physically sensible table: the batch's own targets as 60 samples
    tab = b["targets"][:, :60].T.copy(); mk = b["mask"][:60].copy(); ts = np.array([d["timestamp"] for d in dec])
    traj = TrunkTrajectory(ts, tab, mk, wait_time=0.0, device=0)
    t = np.random.default_rng(1).uniform(0, 0.059, 128)
    tg, m = traj.lookup(torch.tensor(t, device="cuda:0"))
    ctrl = MPTCController(max_batch=128, device=0)
    tau, met, st = ctrl.step(torch.tensor(b["q"], device="cuda:0"), torch.tensor(b["v"], device="cuda:0"), tg, m)
    ctrl.sync()
    idx = np.array([int(np.abs(ts - x).argmin()) for x in t])
    tau_o, _, st_o = orc.step_batch("mptc", orc.model("mini_cheetah"), orc.params("mptc"), b["q"], b["v"], tab[idx].T.copy(), mk[idx])
    ok = (st.cpu().numpy() == 0) & (st_o == 0)
    assert ok.sum() > 100
    r = np.abs(tau.cpu().numpy() - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)
    assert r[ok].max() < 1e-4


def test_hinted_index_search_equals_numpy_argmin():
    """wbc_traj_dev.hpp::traj_index (galloping from a hint, then bisection) on the host: for ANY hint it returns
    what np.abs(ts - (t - wait)).argmin() returns -- duplicates, ties, both ends, the wait phase, empty table."""
    import ctypes as C
    import host_tick as ht
    L = ht.lib()
    L.host_traj_index.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double, C.c_int]
    rng = np.random.default_rng(5)
    for trial in range(40):
        K = int(rng.integers(1, 60))
        ts = np.sort(rng.choice(np.arange(0, 64) * 0.25, size=K, replace=True)).astype(np.float64)   # binary-exact, with duplicates
        wait = float(rng.choice([0.0, 0.5, 2.0]))
        p = ts.ctypes.data_as(C.POINTER(C.c_double))
        for t in np.concatenate([rng.uniform(-1, 20, 30), ts + wait, ts + wait + 0.125, ts + wait - 0.125]):
            want = -1 if t < wait else int(np.abs(ts - (t - wait)).argmin())
            for hint in (-5, 0, K // 2, K - 1, K + 7, int(rng.integers(0, K))):
                assert L.host_traj_index(p, K, wait, float(t), hint) == want, (K, wait, t, hint)
    empty = np.zeros(1)
    assert L.host_traj_index(empty.ctypes.data_as(C.POINTER(C.c_double)), 0, 0.0, 3.0, 0) == -1
