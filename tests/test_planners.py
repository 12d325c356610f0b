"""Callers of the path: the reference's trunk planners (planners/simple.py, planners/towr.py) against dictionaries the
reference's OWN planner code produced (tests/golden/make_planner_golden.py).  Data-format work: BIT-EXACT."""
import os

import numpy as np
import pytest

from oracle import traj_oracle as to

HERE = os.path.dirname(os.path.abspath(__file__))
VEC_KEYS = ["p_body", "pd_body", "pdd_body", "rpy_body", "rpyd_body", "rpydd_body"] + \
           [pre + f for f in ("lf", "rf", "lh", "rh") for pre in ("p_", "pd_", "pdd_")]


def gold():
    return np.load(os.path.join(HERE, "golden", "planner_golden.npz"))


def msgs():
    raw = open(os.path.join(HERE, "golden", "trunk_state_msgs.bin"), "rb").read()
    m = [raw[i:i + 549] for i in range(0, len(raw), 549)]
    del m[60]                                        # as make_planner_golden.py: the one out-of-order sample of the fixture
    return m


def same_dict(d, g, prefix, i=None):
    pick = (lambda a: a) if i is None else (lambda a: a[i])
    for k in VEC_KEYS:
        a, b = np.asarray(d[k], float), pick(g[prefix + k])
        assert a.shape == (3,) and a.tobytes() == b.tobytes(), (prefix, k, i, a, b)          # bit-exact, -0.0 included
    assert [bool(c) for c in d["contact_states"]] == [bool(c) for c in pick(g[prefix + "contact_states"])], (prefix, i)
    assert np.array_equal(np.asarray(d["f_cj"], float), pick(g[prefix + "f_cj"]))
    assert float(d["u2_max"]) == float(pick(g[prefix + "u2_max"]))


def test_basic_planner_scenarios_match_the_reference():
    from quadruped_drake_amd.planners import BasicTrunkPlanner
    g = gold()
    bp = BasicTrunkPlanner()
    same_dict(bp.output_dict, g, "basic_standing_")           # the constructor leaves the standing dictionary
    bp.EdgeTest(); same_dict(bp.output_dict, g, "basic_edge_")
    for i, t in enumerate(g["basic_orientation_times"]):
        bp.OrientationTest(float(t)); same_dict(bp.output_dict, g, "basic_orientation_", i)
    for i, t in enumerate(g["basic_raisefoot_times"]):
        bp.RaiseFoot(float(t)); same_dict(bp.output_dict, g, "basic_raisefoot_", i)
    same_dict(bp.SetTrunkOutputs(0.3), g, "basic_standing_")  # the port function as shipped: standing


def test_batched_scenarios_and_packing_slots():
    """scenario_targets is the batched form; pack / unpack are inverse and put every key in its include/wbc.h slot."""
    from quadruped_drake_amd import pack_trunk_input
    from quadruped_drake_amd.planners import scenario_targets, unpack_trunk_input
    g = gold()
    for name, pre in (("orientation", "basic_orientation_"), ("raise_foot", "basic_raisefoot_")):
        t = g[pre + "times"]
        tg, mk = scenario_targets(name, t)
        assert tg.shape == (54, t.size) and mk.dtype == np.uint8
        for i in range(t.size):
            d = {k: g[pre + k][i] for k in VEC_KEYS}
            d["contact_states"] = list(g[pre + "contact_states"][i])
            exp, em = pack_trunk_input(d)
            assert tg[:, i].tobytes() == exp.tobytes() and mk[i] == em
            same_dict(unpack_trunk_input(tg[:, i], mk[i]), g, pre, i)
    # slots: body p pd pdd rpy rpyd rpydd, then per foot [LF RF LH RH] p pd pdd
    d = {k: np.full(3, float(i)) for i, k in enumerate(VEC_KEYS)}
    d["contact_states"] = [True, False, False, True]
    t54, m = pack_trunk_input(d)
    assert np.array_equal(t54, np.repeat(np.arange(18.0), 3)) and m == 0b1001
    with pytest.raises(ValueError):
        scenario_targets("gallop", [0.0])


def test_lookup_oracle_is_pinned_by_the_reference_planner():
    """oracle/traj_oracle.py (decode + lookup + dict order) reproduces TowrTrunkPlanner.SetTrunkOutputs bit for bit."""
    from quadruped_drake_amd import workloads
    g = gold()
    dec = [to.decode(b) for b in msgs()]
    ts = np.array([d["timestamp"] for d in dec])
    tab = np.stack([to.to_targets(d)[0] for d in dec]); mk = np.array([to.to_targets(d)[1] for d in dec], np.uint8)
    st = workloads.standing_targets("mini_cheetah", 1)[:, 0]
    times = g["towr_times"]
    out, m = to.lookup(times, ts, tab, mk, st, 0b1111, 1.0)
    for i in range(times.size):
        exp = np.concatenate([g["towr_" + k][i] for k in VEC_KEYS])
        assert out[:, i].tobytes() == exp.tobytes(), (i, times[i])
        assert [bool((m[i] >> b) & 1) for b in range(4)] == list(g["towr_contact_states"][i])
    assert (times < 1.0).sum() == 3 and (g["towr_u2_max"][:3] == 0.0).all()      # the wait phase is the standing dictionary


def test_towr_planner_host_side_matches_the_reference():
    """lcm_handler (C decoder), ComputeMaxControlInputs, the host twin of the index search, f_cj."""
    from quadruped_drake_amd.planners import TowrTrunkPlanner
    g = gold()
    tp = TowrTrunkPlanner(msgs())
    assert tp.traj_finished and len(tp.towr_data) == 63
    assert tp.u2_max == float(g["towr_u2_max"][-1]) == float(gold()["towr_u2_max"].max())
    ts = np.array(tp.towr_timestamps)
    for i, t in enumerate(g["towr_times"]):
        if t < 1.0:
            continue
        k = tp.sample_index(float(t))
        assert k == int(np.abs(ts - (t - 1.0)).argmin()), (i, t)
        assert np.array_equal(tp.towr_data[k]["foot_f"].T, g["towr_f_cj"][i])
    with pytest.raises(ValueError, match="Decode error"):
        tp.lcm_handler("trunk_state", b"\x00" * 549)


@pytest.mark.gpu
def test_device_lookup_matches_the_reference_planner():
    import torch
    from quadruped_drake_amd.planners import TowrTrunkPlanner
    g = gold()
    tp = TowrTrunkPlanner(msgs())
    times = g["towr_times"]
    tg, mk = tp.trajectory(0).lookup(torch.tensor(times, device="cuda:0"))
    tg, mk = tg.cpu().numpy(), mk.cpu().numpy()
    for i in range(times.size):
        exp = np.concatenate([g["towr_" + k][i] for k in VEC_KEYS])
        assert tg[:, i].tobytes() == exp.tobytes(), (i, times[i])
        assert [bool((mk[i] >> b) & 1) for b in range(4)] == list(g["towr_contact_states"][i])
    for i in (0, 3, 4, 70, 130, 134, 198):                      # one robot's dictionary, f_cj and u2_max included
        same_dict(tp.SetTrunkOutputs(float(times[i])), g, "towr_", i)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["id", "mptc"])
def test_reference_scenarios_tick_like_the_oracle(kind):
    """The reference's manual scenarios at the simulate.py:171-176 initial state, one robot per scenario time."""
    import torch
    from oracle import oracle_py as orc
    from quadruped_drake_amd import IDController, MPTCController, workloads
    from quadruped_drake_amd.planners import scenario_targets
    cls = {"id": IDController, "mptc": MPTCController}[kind]
    for name, t in (("standing", [0.0]), ("orientation", np.linspace(0.0, 6.0, 13)), ("raise_foot", [0.5, 1.5, 2.0]), ("edge", [0.0])):
        tg, mk = scenario_targets(name, t)
        n = tg.shape[1]
        q, v = workloads.nominal_state("mini_cheetah", n)
        ctrl = cls(max_batch=n, device=0)
        up = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda:0")
        tau, met, st = ctrl.step(up(q), up(v), up(tg), up(mk)); ctrl.sync()
        tau, st = tau.cpu().numpy(), st.cpu().numpy()
        iters = ctrl.stats()["iters_sum"]
        ctrl.close()
        tau_o, met_o, st_o = orc.step_batch(kind, orc.model("mini_cheetah"), orc.params(kind), q, v, tg, mk)
        assert (st == 0).all() and (st_o == 0).all(), (name, st, st_o)
        err = np.abs(tau - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)
        assert err.max() < 1e-6, (name, err.max())
        assert np.allclose(met.cpu().numpy(), met_o, rtol=1e-5, atol=1e-6)
        if name == "edge":
            assert iters > 0, "EdgeTest is the scenario whose friction rows become active (planners/simple.py:110-115)"


def _rpy(q):
    w, x, y, z = q[0:4]
    return np.array([np.arctan2(2 * (y * z + w * x), 1 - 2 * (x * x + y * y)),
                     np.arctan2(-2 * (x * z - w * y), np.hypot(1 - 2 * (y * y + z * z), 2 * (x * y + w * z))),
                     np.arctan2(2 * (x * y + w * z), 1 - 2 * (y * y + z * z))])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,dt", [("id", 5e-3), ("mptc", 1e-3)])
def test_reference_scenarios_closed_loop(kind, dt):
    """The experiments the reference runs by editing SetTrunkOutputs (planners/simple.py:121-124), closed loop on the
    device from the simulate.py:171-176 initial state: lookup -> tick -> forward step in one persistent launch.
    OrientationTest: the body tracks the moving pitch / yaw target.  RaiseFoot: the body shifts over the support
    triangle, then the right-front foot leaves the ground (contact mask 0b1101) and reaches its target 0.1 m up."""
    import torch
    import energy_model as em
    from quadruped_drake_amd import IDController, MPTCController, workloads
    from quadruped_drake_amd.planners import scenario_trajectory
    cls = {"id": IDController, "mptc": MPTCController}[kind]
    model = em.load("mini_cheetah")
    n = 8
    rng = np.random.default_rng(11)
    for name, dur in (("orientation", 4.0), ("raise_foot", 2.0)):
        traj = scenario_trajectory(name, dur, dt)
        ctrl = cls(max_batch=n, device=0)
        q0, v0 = workloads.nominal_state("mini_cheetah", n)
        q0[7:, 1:] += rng.uniform(-0.02, 0.02, (12, n - 1))          # robot 0: exactly the reference's initial state
        feet0 = np.array([[f["p"] for f in em.bodies(model, q0[:, i])[1]] for i in range(n)])
        q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0")
        time = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        ctrl.stats(reset=True)
        steps = int(round(dur / dt))
        tau, met, st, tg, mk = ctrl.rollout(traj, steps, dt, q, v, time); ctrl.sync()
        s = ctrl.stats()
        assert s["ticks"] == steps * n and s["status_nonzero"] == 0, (name, s)
        qf, tgf, mkf = q.cpu().numpy(), tg.cpu().numpy(), mk.cpu().numpy()
        assert np.allclose(time.cpu().numpy(), dur, atol=1e-9)
        feet1 = np.array([[f["p"] for f in em.bodies(model, qf[:, i])[1]] for i in range(n)])
        if name == "orientation":
            for i in range(n):
                assert np.abs(_rpy(qf[:, i]) - tgf[9:12, i]).max() < 5e-3, (kind, i, _rpy(qf[:, i]), tgf[9:12, i])
            assert np.abs(qf[4:7] - np.array([0.0, 0.0, 0.3])[:, None]).max() < 5e-3
            assert np.abs(feet1 - feet0).max() < 5e-3 and (mkf == 0b1111).all()      # stance feet did not move
        else:
            assert (mkf == 0b1101).all()
            assert np.abs(feet1[:, 1] - np.array([0.175, -0.11, 0.1])).max() < 3e-3   # RF foot at its raised target
            assert np.abs(feet1[:, [0, 2, 3]] - feet0[:, [0, 2, 3]]).max() < 5e-3     # the three stance feet stayed
            tol = 2e-3 if kind == "id" else 0.06     # MPTC's low body gains (Kp 100, Kd 10) are still settling at t = 2 s
            assert np.abs(qf[4:7] - np.array([-0.1, 0.05, 0.3])[:, None]).max() < tol
        ctrl.close(); traj.close()


def test_simulate_harness_arguments():
    """The batch counterpart of the reference's entry script keeps its common parameters (simulate.py:9-25)."""
    from quadruped_drake_amd import simulate
    a = simulate.parse([])
    assert (a.control, a.planner, a.sim_time, a.dt) == ("ID", "basic", 6.0, 5e-3)
    assert simulate.parse(["--control", "MPTC"]).dt == 1e-3               # explicit-step stability, DESIGN.md section 9
    for bad in (["--planner", "towr"], ["--control", "B"], ["--n", "0"]):
        with pytest.raises(SystemExit):
            simulate.parse(bad)


@pytest.mark.gpu
def test_simulate_harness_runs_the_reference_experiments(tmp_path):
    from quadruped_drake_amd import simulate
    log = str(tmp_path / "log.npz")
    assert simulate.main(["--control", "ID", "--scenario", "raise_foot", "--n", "16", "--perturb", "0.02", "--sim-time", "2",
                          "--log", log]) == 0
    z = np.load(log)
    assert z["t"].shape == (40,) and z["err"].shape == (40, 16) and abs(z["t"][-1] - 2.0) < 1e-9
    assert z["err"][-1].max() < 1e-4 < z["err"][0].min()                   # the output error of simulate.py's third plot decays
    # a recorded trunk_state stream through the towr planner path (the golden wire messages; their targets are noise,
    # so only the plumbing is checked: waits 1 s standing, then serves samples)
    raw = open(os.path.join(HERE, "golden", "trunk_state_msgs.bin"), "rb").read()
    p = tmp_path / "msgs.bin"; p.write_bytes(raw[:60 * 549])
    r = simulate.run(simulate.parse(["--control", "ID", "--planner", "towr", "--messages", str(p), "--sim-time", "0.5"]))
    assert r["ticks"] == 100 and r["status_nonzero"] == 0
