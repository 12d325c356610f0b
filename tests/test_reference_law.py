"""The oracle, the host-instantiated kernel math and the HIP path against vectors produced by EXECUTING the reference's
own controller code (tests/golden/make_reference_law_golden.py: controllers/*.py imported from /root/reference over
stand-ins for the Drake plant and for MathematicalProgram / the solver).

What this pins: the arithmetic the reference's Python defines -- targets and gains, RPY handling, Lambda / Jbar / Q /
f_des, the Coriolis-matrix and Jdot definitions, QP assembly, logging -- for ID, MPTC, PC, CLF, every contact mask,
both robots, randomised mu / mass.  What it does not: Drake's rigid-body numbers (the stand-in plant takes them from
oracle/) and OSQP's pick among the optimal set (the stand-in solver applies this repository's tie-break).
Tolerances: solver-independent quantities (vd, metrics) 1e-7; torques 1e-5 relative (north_star: 1e-4) -- the
stand-in solves the Hessian form the reference assembles (J'J, G'WG), which carries less of the eps2 tie-break's
precision than the square-root form the oracle and the kernels factor; measured worst case 9e-7."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SETS = ["cfg2_id", "cfg3_id", "cfg3_mptc", "cfg4_anymal_mptc", "cfg5_rand_mptc", "cfg3_pc", "cfg2_pc", "cfg3_clf", "cfg2_clf",
        "masks16_id", "masks16_mptc", "masks16_pc", "masks16_clf",
        # simulate.py's initial state with the dictionaries of the reference's planner scenarios (planner + controller chained)
        "scen_id", "scen_mptc", "scen_pc", "scen_clf",
        # the same reference code over a plant backed by tests/energy_model.py instead of oracle/: nothing shared at all
        "indep_cfg2_id", "indep_cfg3_mptc", "indep_cfg4_anymal_mptc", "indep_cfg3_clf", "indep_cfg3_pc"]
PERM_SETS = ["perm_cfg2_id", "perm_cfg3_mptc", "perm_cfg4_anymal_mptc"]   # plant with its own joint / actuator numbering
TAU_TOL, IND_TOL = 1e-5, 1e-7


def load(name):
    z = np.load(os.path.join(HERE, "golden", "reference_law_golden.npz"))
    d = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + "_")}
    d["kind"], d["model"] = str(d["kind"]), str(d["model"])
    d["mu"] = d["mu"] if d["mu"].size else None
    d["mass_scale"] = d["mass_scale"] if d["mass_scale"].size else None
    return d


def rel(tau, ref):
    return np.abs(tau - ref).max(0) / np.maximum(np.abs(ref).max(0), 1e-3)


def check_metrics(met, ref, kind):
    """[V, err, res, Vdot]: res is each solver's own residual (not comparable); ID logs err only."""
    cols = [1] if kind == "id" else [0, 1, 3]
    for c in cols:
        assert np.allclose(met[c], ref[c], rtol=IND_TOL, atol=IND_TOL), (kind, c, np.abs(met[c] - ref[c]).max())


@pytest.mark.parametrize("name", SETS)
def test_oracle_matches_the_executed_reference_code(name):
    from oracle import oracle_py as orc
    g = load(name)
    n = g["q"].shape[1]
    p = orc.params(g["kind"])
    for i in range(n):
        m = orc.model(g["model"]) if g["mass_scale"] is None else orc.model_scaled(g["model"], float(g["mass_scale"][i]))
        if g["mu"] is not None:
            p.mu = float(g["mu"][i])
        ct = [(int(g["mask"][i]) >> k) & 1 for k in range(4)]
        tau, met, st, qp = orc.control_law(g["kind"], m, p, g["q"][:, i], g["v"][:, i], g["targets"][:, i], ct, want_qp=True)
        assert st == 0
        assert rel(tau[:, None], g["tau"][:, i:i + 1])[0] < TAU_TOL, (name, i)
        check_metrics(met[:, None], g["metrics"][:, i:i + 1], g["kind"])
        # tier (i), solver-independent: the QP's accelerations and contact forces' net effect
        assert np.abs(qp["x"][:18] - g["vd"][:, i]).max() < IND_TOL * (1.0 + np.abs(g["vd"][:, i]).max()), (name, i)
        nc = sum(ct)
        f = np.zeros(12)
        for j, k in enumerate([k for k in range(4) if ct[k]]):
            f[3 * k:3 * k + 3] = qp["x"][30 + 3 * j:33 + 3 * j]
        assert np.abs(f.reshape(4, 3).sum(0) - g["f"][:, i].reshape(4, 3).sum(0)).max() < 1e-6 * (1.0 + np.abs(f).max()), (name, i, nc)


@pytest.mark.parametrize("name", SETS)
def test_kernel_math_on_the_host_matches_the_executed_reference_code(name):
    import host_tick as ht
    from quadruped_drake_amd import load_model
    g = load(name)
    flat = np.array(load_model(g["model"])["flat"])
    tau, met, st, it, vd = ht.run(g["kind"], flat, g["q"], g["v"], g["targets"], g["mask"], g["mu"], g["mass_scale"],
                                  want_vdot=True, hexv=True)
    assert (st == 0).all()
    assert rel(tau, g["tau"]).max() < TAU_TOL, (name, rel(tau, g["tau"]).max())
    check_metrics(met, g["metrics"], g["kind"])
    assert np.abs(vd - g["vd"]).max() < IND_TOL * (1.0 + np.abs(g["vd"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", SETS)
def test_hip_path_matches_the_executed_reference_code(name):
    import torch
    from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
    g = load(name)
    cls = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[g["kind"]]
    n = g["q"].shape[1]
    ctrl = cls(model=g["model"], max_batch=n, device=0)
    up = lambda x: None if x is None else torch.tensor(np.ascontiguousarray(x), device="cuda:0")
    vd = torch.zeros((18, n), dtype=torch.float64, device="cuda:0")
    ctrl.set_vdot_output(vd)
    tau, met, st = ctrl.step(up(g["q"]), up(g["v"]), up(g["targets"]), up(g["mask"]), up(g["mu"]), up(g["mass_scale"]))
    ctrl.sync()
    tau, met, st, vd = tau.cpu().numpy(), met.cpu().numpy(), st.cpu().numpy(), vd.cpu().numpy()
    ctrl.close()
    assert (st == 0).all()
    assert rel(tau, g["tau"]).max() < TAU_TOL, (name, rel(tau, g["tau"]).max())
    check_metrics(met, g["metrics"], g["kind"])
    assert np.abs(vd - g["vd"]).max() < IND_TOL * (1.0 + np.abs(g["vd"]).max())


@pytest.mark.parametrize("name", PERM_SETS)
def test_permuted_plant_oracle_and_host_math(name):
    """basic_controller.py:310-313: the plant's joint and actuator numbering are its own.  The reference's code ran on a
    plant with breadth-first joints and random actuators; its torques are the canonical ones re-ordered by act_joint."""
    import host_tick as ht
    from oracle import oracle_py as orc
    from quadruped_drake_amd import load_model
    g = load(name)
    act = [int(x) for x in g["act_joint"]]
    table = dict(load_model(g["model"])); table["act_perm"] = act
    tau_o, met_o, st_o = orc.step_batch(g["kind"], orc.model(table), orc.params(g["kind"]), g["q"], g["v"], g["targets"], g["mask"])
    assert (st_o == 0).all() and rel(tau_o, g["tau"]).max() < TAU_TOL
    check_metrics(met_o, g["metrics"], g["kind"])
    order = np.array([int(x) for x in g["order"]])
    q2 = g["q"].copy(); v2 = g["v"].copy()
    q2[7 + order] = g["q"][7:]; v2[6 + order] = g["v"][6:]
    tau, met, st, it = ht.run(g["kind"], np.array(table["flat"]), q2, v2, g["targets"], g["mask"], q_perm=order, act_perm=act, hexv=True)
    assert (st == 0).all() and rel(tau, g["tau"]).max() < TAU_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", PERM_SETS)
def test_permuted_plant_hip_path(name):
    import torch
    from quadruped_drake_amd import IDController, MPTCController
    g = load(name)
    order = np.array([int(x) for x in g["order"]]); act = [int(x) for x in g["act_joint"]]
    q2 = g["q"].copy(); v2 = g["v"].copy()
    q2[7 + order] = g["q"][7:]; v2[6 + order] = g["v"][6:]
    n = q2.shape[1]
    ctrl = {"id": IDController, "mptc": MPTCController}[g["kind"]](model=g["model"], max_batch=n, device=0, q_perm=order, act_perm=act)
    up = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda:0")
    tau, met, st = ctrl.step(up(q2), up(v2), up(g["targets"]), up(g["mask"])); ctrl.sync()
    tau, st = tau.cpu().numpy(), st.cpu().numpy()
    ctrl.close()
    assert (st == 0).all() and rel(tau, g["tau"]).max() < TAU_TOL


def _closed(kind):
    z = np.load(os.path.join(HERE, "golden", "reference_law_golden.npz"))
    p = "closedloop_%s_" % kind
    return float(z[p + "dt"]), z[p + "times"], z[p + "q"], z[p + "v"], z[p + "tau"]


def test_closed_loop_pc_fixture_has_the_passivity_row_active():
    """The PC trajectory is not the MPTC one in disguise: the reference's logged Vdot (pc_controller.py:229-237, the row
    Vdot <= delta <= 0) sits AT zero on a run of ticks after the contact switch, and below zero before."""
    z = np.load(os.path.join(HERE, "golden", "reference_law_golden.npz"))
    vd, T = z["closedloop_pc_vdot_metric"], z["closedloop_pc_times"]
    at = np.abs(vd) < 1e-9
    assert at.sum() >= 10 and (T[at] > 1).all() and (vd[~at] < -1e-3).all() and (vd < 1e-9).all()
    assert not np.array_equal(z["closedloop_pc_q"], z["closedloop_mptc_q"])


@pytest.mark.parametrize("kind", ["id", "mptc", "pc", "clf"])
def test_closed_loop_oracle_follows_the_executed_reference(kind):
    """Reference planner scenario (RaiseFoot across its contact switch) -> reference controller code -> forward step,
    60 ticks (make_reference_law_golden.py): the oracle's tick + the numpy integrator retrace the same trajectory."""
    from oracle import oracle_py as orc
    from oracle import traj_oracle as to
    from quadruped_drake_amd.planners import scenario_targets
    dt, T, Q, V, TAU = _closed(kind)
    m, p = orc.model("mini_cheetah"), orc.params(kind)
    q, v = Q[:, 0].copy(), V[:, 0].copy()
    for k, t in enumerate(T):
        tg, mk = scenario_targets("raise_foot", [t])
        tau, met, st, qp = orc.control_law(kind, m, p, q, v, tg[:, 0], [(int(mk[0]) >> b) & 1 for b in range(4)], want_qp=True)
        assert st == 0 and np.abs(tau - TAU[:, k]).max() < 1e-5 * max(np.abs(TAU[:, k]).max(), 1e-3), (kind, k)
        qn, vn = to.integrate(q[:, None], v[:, None], qp["x"][:18][:, None], dt)
        q, v = qn[:, 0], vn[:, 0]
        assert np.abs(q - Q[:, k + 1]).max() < 1e-7 and np.abs(v - V[:, k + 1]).max() < 1e-6, (kind, k)
    assert (T <= 1).any() and (T > 1).any()          # the contact switch lies inside the window


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["id", "mptc", "pc", "clf"])
def test_device_rollout_follows_the_executed_reference(kind):
    """wbc_rollout (stored-trajectory lookup -> tick -> forward step, one persistent launch per chunk) against the same
    trajectory: the whole closed-loop chain of the device against the reference's executed planner + controller code."""
    import torch
    from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
    from quadruped_drake_amd.planners import scenario_targets
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    dt, T, Q, V, TAU = _closed(kind)
    tg, mk = scenario_targets("raise_foot", T)
    traj = TrunkTrajectory(T, np.ascontiguousarray(tg.T), mk, wait_time=0.0, device=0)
    ctrl = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[kind](max_batch=4, device=0)
    q = torch.tensor(np.tile(Q[:, :1], (1, 4)), device="cuda:0"); v = torch.tensor(np.tile(V[:, :1], (1, 4)), device="cuda:0")
    time = torch.full((4,), float(T[0]), dtype=torch.float64, device="cuda:0")
    for c in range(6):
        tau, met, st, _, mko = ctrl.rollout(traj, 10, dt, q, v, time); ctrl.sync()
        k = 10 * (c + 1)
        assert (st.cpu().numpy() == 0).all()
        assert np.abs(q.cpu().numpy()[:, 0] - Q[:, k]).max() < 1e-7 and np.abs(v.cpu().numpy()[:, 0] - V[:, k]).max() < 1e-6, (kind, k)
        assert np.abs(tau.cpu().numpy()[:, 0] - TAU[:, k - 1]).max() < 1e-5 * max(np.abs(TAU[:, k - 1]).max(), 1e-3)
        assert int(mko.cpu().numpy()[0]) == int(mk[k - 1])
    assert torch.equal(q[:, 0], q[:, 3])             # the four copies stay identical
    ctrl.close(); traj.close()
