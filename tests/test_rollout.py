"""generalized-acceleration output, forward step and the device-resident closed loop."""
import numpy as np
import pytest

import host_tick as ht
from oracle import oracle_py as orc
from oracle import traj_oracle as to
from quadruped_drake_amd import workloads


@pytest.mark.parametrize("cfg,kind", [(3, "mptc"), (2, "id"), (3, "pc"), (3, "clf")])
def test_vdot_of_kernel_math_equals_oracle_qp_vd(cfg, kind):
    """vd is solver-independent (unique): the kernels' reduced solution must reproduce x[:18] of the literal QP."""
    b = workloads.make_batch(cfg, n=24)
    t = orc.load_model_json(b["model"]); m = orc.model(b["model"]); p = orc.params(kind)
    for hexv in (False, True):
        _, _, st, _, vd = ht.run(kind, t["flat"], b["q"], b["v"], b["targets"], b["mask"], hexv=hexv, want_vdot=True)
        for i in range(24):
            ct = [(int(b["mask"][i]) >> k) & 1 for k in range(4)]
            _, _, _, qp = orc.control_law(kind, m, p, b["q"][:, i], b["v"][:, i], b["targets"][:, i], ct, want_qp=True)
            assert np.abs(vd[:, i] - qp["x"][:18]).max() < 1e-9 * (1 + np.abs(qp["x"][:18]).max())


def test_integrator_oracle_properties():
    rng = np.random.default_rng(0)
    b = workloads.make_batch(3, n=16)
    vd = rng.normal(0, 5, (18, 16))
    q1, v1 = to.integrate(b["q"], b["v"], vd, 5e-3)
    assert np.allclose(np.linalg.norm(q1[:4], axis=0), 1.0, atol=1e-15)
    assert np.allclose(v1, b["v"] + 5e-3 * vd)
    assert np.allclose(q1[4:7], b["q"][4:7] + 5e-3 * v1[3:6]) and np.allclose(q1[7:], b["q"][7:] + 5e-3 * v1[6:])
    # pure rotation about world z by w dt
    q0 = np.zeros((19, 1)); q0[0] = 1; v0 = np.zeros((18, 1)); v0[2] = 2.0
    q2, _ = to.integrate(q0, v0, np.zeros((18, 1)), 0.1)
    assert np.allclose(q2[:4, 0], [np.cos(0.1), 0, 0, np.sin(0.1)])


@pytest.mark.gpu
def test_gpu_vdot_and_integrate_match_oracles():
    import torch
    from quadruped_drake_amd import MPTCController
    b = workloads.make_batch(3, n=512)
    ctrl = MPTCController(max_batch=512, device=0)
    up = lambda x: torch.tensor(x, device="cuda:0")
    q, v, tg, mk = up(b["q"]), up(b["v"]), up(b["targets"]), up(b["mask"])
    vd = torch.zeros((18, 512), dtype=torch.float64, device="cuda:0")
    ctrl.set_vdot_output(vd)
    tau, met, st = ctrl.step(q, v, tg, mk)
    ctrl.sync()
    t = orc.load_model_json("mini_cheetah")
    _, _, st_h, _, vd_h = ht.run("mptc", t["flat"], b["q"], b["v"], b["targets"], b["mask"], want_vdot=True)
    assert np.abs(vd.cpu().numpy() - vd_h).max() < 1e-8 * (1 + np.abs(vd_h).max())
    ctrl.integrate(q, v, vd, 5e-3)
    ctrl.sync()
    q_o, v_o = to.integrate(b["q"], b["v"], vd.cpu().numpy(), 5e-3)
    assert np.allclose(q.cpu().numpy(), q_o, rtol=0, atol=1e-14) and np.allclose(v.cpu().numpy(), v_o, rtol=0, atol=1e-14)
    ctrl.set_vdot_output(None)
    ctrl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,kind", [(3, "mptc"), (2, "id"), (3, "pc"), (3, "clf"), (5, "mptc")])
def test_gpu_vdot_equals_oracle_qp_vd(cfg, kind):
    """The device's generalized accelerations against x[:18] of the ORACLE's literal 30+3nc-variable QP (the
    solver-independent, unique part of the solution), for every law."""
    import torch
    from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
    n = 96
    b = workloads.make_batch(cfg, n=n)
    cls = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[kind]
    ctrl = cls(model=b["model"], max_batch=n, device=0)
    up = lambda x: None if x is None else torch.tensor(x, device="cuda:0")
    vd = torch.full((18, n), float("nan"), dtype=torch.float64, device="cuda:0")
    ctrl.set_vdot_output(vd)
    tau, met, st = ctrl.step(up(b["q"]), up(b["v"]), up(b["targets"]), up(b["mask"]), up(b["mu"]), up(b["mass_scale"]))
    ctrl.sync()
    ctrl.set_vdot_output(None)
    ctrl.close()
    vd = vd.cpu().numpy()
    assert (st.cpu().numpy() == 0).all() and np.isfinite(vd).all()
    m = orc.model(b["model"])
    for i in range(n):
        p = orc.params(kind)
        if b["mu"] is not None:
            p.mu = float(b["mu"][i])
        mi = m if b["mass_scale"] is None else orc.model_scaled(b["model"], float(b["mass_scale"][i]))
        ct = [(int(b["mask"][i]) >> k) & 1 for k in range(4)]
        _, _, st_o, qp = orc.control_law(kind, mi, p, b["q"][:, i], b["v"][:, i], b["targets"][:, i], ct, want_qp=True)
        assert st_o == 0
        assert np.abs(vd[:, i] - qp["x"][:18]).max() < 1e-8 * (1 + np.abs(qp["x"][:18]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["id", "mptc"])
def test_rollout_with_a_straight_knee_stays_finite(kind):
    """A straight knee on a STANCE leg is reported (status 2) with zero torques AND zero accelerations, so the closed loop
    integrates defined values -- q, v stay finite and frozen, the caller's (uninitialised) vdot buffer is never read."""
    import torch
    from quadruped_drake_amd import IDController, MPTCController
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    n = 8
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    q0[7 + 2, 0] = 0.0                               # robot 0: LF knee straight -> singular leg Jacobian
    st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
    traj = TrunkTrajectory(np.zeros(0), np.zeros((0, 54)), np.zeros(0, np.uint8), wait_time=1e9, device=0,
                           standing_targets=st_t, standing_mask=0b1111)
    cls = IDController if kind == "id" else MPTCController
    ctrl = cls(max_batch=n, device=0)
    q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0")
    time = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    tau, met, st, tg, mk = ctrl.rollout(traj, 5, 1e-3, q, v, time)
    ctrl.sync()
    st = st.cpu().numpy(); qf = q.cpu().numpy(); vf = v.cpu().numpy(); tau = tau.cpu().numpy()
    assert np.isfinite(qf).all() and np.isfinite(vf).all() and np.isfinite(tau).all()
    assert st[0] == 2 and (st[1:] == 0).all()
    assert np.array_equal(qf[:, 0], q0[:, 0]) and np.array_equal(vf[:, 0], v0[:, 0])    # zero accelerations from rest: frozen
    assert (tau[:, 0] == 0).all()
    ctrl.close()


@pytest.mark.gpu
def test_closed_loop_standing_rollout_id():
    """simulate.py:171-179 initial state, planners/simple.py standing targets, 200 ticks of dt = 5e-3
    (simulate.py:21) with the ID law: the robot keeps standing, feet stay put, every tick is solved."""
    import torch
    from quadruped_drake_amd import IDController
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    import energy_model as em
    n = 64
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(3)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n))           # slightly perturbed joints
    q0[6] += rng.uniform(-0.01, 0.01, n)                   # and body height
    v0[0:6] = rng.normal(0, 0.1, (6, n))
    st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
    model = em.load("mini_cheetah")
    feet0 = np.array([[f["p"] for f in em.bodies(model, q0[:, i])[1]] for i in range(n)])
    traj = TrunkTrajectory(np.zeros(0), np.zeros((0, 54)), np.zeros(0, np.uint8), wait_time=1e9, device=0,
                           standing_targets=st_t, standing_mask=0b1111)
    ctrl = IDController(max_batch=n, device=0)
    q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0")
    time = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    ctrl.stats(reset=True)
    tau, met, st, tg, mk = ctrl.rollout(traj, 200, 5e-3, q, v, time)
    ctrl.sync()
    s = ctrl.stats()
    assert s["ticks"] == 200 * n and s["status_nonzero"] == 0
    qf = q.cpu().numpy(); vf = v.cpu().numpy()
    assert np.allclose(time.cpu().numpy(), 1.0, atol=1e-9)
    assert np.isfinite(qf).all() and np.allclose(np.linalg.norm(qf[:4], axis=0), 1.0, atol=1e-12)
    assert np.abs(qf[6] - 0.3).max() < 2e-3 and np.abs(qf[4:6]).max() < 2e-3        # body converged to the target
    assert np.abs(vf).max() < 1e-2                                                  # and came to rest
    feet1 = np.array([[f["p"] for f in em.bodies(model, qf[:, i])[1]] for i in range(n)])
    assert np.abs(feet1 - feet0).max() < 5e-3                                       # stance feet did not slide
    assert (mk.cpu().numpy() == 0b1111).all() and np.array_equal(tg.cpu().numpy()[:, 0], st_t)
    ctrl.close()


@pytest.mark.gpu
def test_closed_loop_mptc_is_passive():
    """MPTC's storage function V decreases along the closed loop and Vdot <= 0 (RSS'20 p077 property) --
    with dt = 1e-3: the law's roll damping rate Kd/Lambda_roll is ~885 1/s, so the explicit forward
    step is only stable for dt < 2/885; at the reference's 5e-3 it is not."""
    import torch
    from quadruped_drake_amd import MPTCController
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    n = 32
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(4)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); q0[6] += 0.01; v0[0] = 0.3
    st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
    traj = TrunkTrajectory(np.zeros(0), np.zeros((0, 54)), np.zeros(0, np.uint8), wait_time=1e9, device=0,
                           standing_targets=st_t, standing_mask=0b1111)
    ctrl = MPTCController(max_batch=n, device=0)
    q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0")
    time = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    Vs = []
    for chunk in range(6):
        tau, met, st, tg, mk = ctrl.rollout(traj, 100, 1e-3, q, v, time)
        ctrl.sync()
        m = met.cpu().numpy()
        assert (st.cpu().numpy() == 0).all() and (m[3] <= 1e-9).all()            # Vdot <= 0 at every sampled tick
        Vs.append(m[0].copy())
    Vs = np.array(Vs)
    assert (np.diff(Vs, axis=0) <= 1e-12).all() and (Vs[-1] < 0.6 * Vs[0]).all()    # V decreases monotonically
    assert np.abs(v.cpu().numpy()).max() < 1.0
    ctrl.close()


def _trot_trajectory(K=300, dt=1e-3, seed=7):
    """A synthetic stored trunk trajectory: alternating diagonal-pair contacts every 60 samples, targets = the
    standing targets with a slow body sway and lifted swing feet (what a TOWR trot would stream)."""
    rng = np.random.default_rng(seed)
    st_t = workloads.standing_targets("mini_cheetah", 1)[:, 0]
    ts = np.arange(K) * dt
    ts[40] = ts[39]                                     # a duplicated timestamp (first index must win)
    tg = np.tile(st_t, (K, 1))
    tg[:, 0] += 0.01 * np.sin(2 * np.pi * ts / 0.3)     # body x sway
    tg[:, 3] = 0.01 * 2 * np.pi / 0.3 * np.cos(2 * np.pi * ts / 0.3)
    masks = np.where((np.arange(K) // 60) % 2 == 0, 0b1001, 0b0110).astype(np.uint8)
    for f in range(4):
        sw = ((masks >> f) & 1) == 0
        tg[sw, 18 + 9 * f + 2] += 0.03                  # swing foot z target
    return ts, tg, masks, st_t


@pytest.mark.gpu
@pytest.mark.parametrize("box", [False, True])
@pytest.mark.parametrize("kind", ["mptc", "id", "pc", "clf"])
def test_persistent_rollout_equals_launch_per_stage(kind, box):
    """wbc_rollout on the 16-lane mapping is ONE persistent launch (state in LDS between ticks, lookup with an
    index hint, in-kernel forward step); it must reproduce the launch-per-stage loop (wbc_traj_lookup, wbc_step,
    wbc_integrate, time += dt) bit for bit -- contact switches, per-robot time offsets and the wait phase included.
    All EIGHT instantiations wbc_rollout dispatches (four laws x torque box off / on: each has its own register allocation,
    462 - 488 registers) against the tick kernels of the same law and option.  The box (6 N m under the task-space laws, 12 N m under
    the ID-type ones: what binds on about half of this rollout's ticks and stays feasible -- host emulation) must really bind."""
    import torch
    from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    n, steps, dt = 203, 120, 1e-3
    ts, tg, masks, st_t = _trot_trajectory()
    traj = TrunkTrajectory(ts, tg, masks, wait_time=0.03, device=0, standing_targets=st_t, standing_mask=0b1111)
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(11)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
    t0 = rng.uniform(0.0, 0.25, n)                       # some robots start inside the wait phase, some past the table's end
    cls = {"mptc": MPTCController, "id": IDController, "pc": PCController, "clf": CLFController}[kind]
    tau_max = {"mptc": 6.0, "pc": 6.0, "id": 12.0, "clf": 12.0}[kind] if box else None
    prm = None if tau_max is None else {"tau_max": tau_max}
    dev = "cuda:0"
    # (a) persistent
    ca = cls(max_batch=n, device=0, params=prm); ca.set_variant("hex")
    qa = torch.tensor(q0, device=dev); va = torch.tensor(v0, device=dev); ta = torch.tensor(t0, device=dev)
    tau_a, met_a, st_a, tg_a, mk_a = ca.rollout(traj, steps, dt, qa, va, ta)
    ca.sync()
    sa = ca.stats()
    # (b) one launch per stage, same kernels' arithmetic
    cb = cls(max_batch=n, device=0, params=prm); cb.set_variant("hex")
    qb = torch.tensor(q0, device=dev); vb = torch.tensor(v0, device=dev); tb = torch.tensor(t0, device=dev)
    vd = torch.zeros((18, n), dtype=torch.float64, device=dev)
    cb.set_vdot_output(vd)
    tbox_active = 0
    for _ in range(steps):
        tg_b, mk_b = traj.lookup(tb)
        torch.cuda.synchronize()
        tau_b, met_b, st_b = cb.step(qb, vb, tg_b, mk_b)
        cb.integrate(qb, vb, vd, dt)
        cb.sync()
        tb += dt
        if tau_max is not None:
            tbox_active += int((tau_b.abs().max(0).values > tau_max * (1 - 1e-6)).sum())
    sb = cb.stats()
    if tau_max is not None:
        # the box binds along the rollout, and holds to the solver's accuracy on these drop-heavy ticks (measured: 6.00000005 under PC, i.e.
        # 8e-9 relative -- the level at which such ticks agree with the extended-precision oracle, tests/test_kernel_math_host.py)
        assert float(tau_b[:, st_b == 0].abs().max()) <= tau_max * (1 + 1e-6) and tbox_active > steps * n // 10
    for a, b in ((qa, qb), (va, vb), (ta, tb), (tau_a, tau_b), (met_a, met_b), (tg_a, tg_b)):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert np.array_equal(st_a.cpu().numpy(), st_b.cpu().numpy()) and np.array_equal(mk_a.cpu().numpy(), mk_b.cpu().numpy())
    assert sa["ticks"] == sb["ticks"] == steps * n and sa["status_nonzero"] == sb["status_nonzero"]
    assert sa["iters_sum"] == sb["iters_sum"] and sa["mask_count"] == sb["mask_count"]
    assert len(set(mk_a.cpu().numpy().tolist())) >= 2     # the batch really was in different contact phases
    ca.close(); cb.close()


@pytest.mark.gpu
def test_persistent_rollout_long_and_ragged():
    """More ticks than one bounded launch carries (1024) and a batch that does not fill its last wavefront:
    still identical to the launch-per-stage loop, and the table's last sample is held after its end."""
    import torch
    from quadruped_drake_amd import IDController
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    n, steps, dt = 9, 1100, 1e-3
    ts, tg, masks, st_t = _trot_trajectory(K=200)
    masks[:] = 0b1111                                   # keep the robots standing; the targets still vary with time
    traj = TrunkTrajectory(ts, tg, masks, wait_time=0.01, device=0, standing_targets=st_t, standing_mask=0b1111)
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    t0 = np.linspace(0.0, 0.05, n)
    dev = "cuda:0"
    ca = IDController(max_batch=16, device=0); ca.set_variant("hex")
    qa = torch.tensor(q0, device=dev); va = torch.tensor(v0, device=dev); ta = torch.tensor(t0, device=dev)
    tau_a, met_a, st_a, tg_a, mk_a = ca.rollout(traj, steps, dt, qa, va, ta)
    ca.sync()
    cb = IDController(max_batch=16, device=0); cb.set_variant("hex")
    qb = torch.tensor(q0, device=dev); vb = torch.tensor(v0, device=dev); tb = torch.tensor(t0, device=dev)
    vd = torch.zeros((18, n), dtype=torch.float64, device=dev)
    cb.set_vdot_output(vd)
    for _ in range(steps):
        tg_b, mk_b = traj.lookup(tb)
        torch.cuda.synchronize()
        tau_b, _, _ = cb.step(qb, vb, tg_b, mk_b)
        cb.integrate(qb, vb, vd, dt)
        cb.sync()
        tb += dt
    for a, b in ((qa, qb), (va, vb), (ta, tb), (tau_a, tau_b), (tg_a, tg_b)):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert np.array_equal(tg_a.cpu().numpy()[:, 0], tg[-1])          # past the end: the last sample
    assert ca.stats()["ticks"] == steps * n
    ca.close(); cb.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tau_max", [None, 12.0])
def test_no_product_kernel_spills(tau_max):
    """The runtime's view of what build() checks in the code object: the tick kernel AND the persistent rollout kernel of every
    law run without a private segment (the rollout kernels carried 100 - 300 B/lane of scratch until the resource report
    looked at them: wbc_rollout_kernel_info)."""
    from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
    for cls in (IDController, MPTCController, PCController, CLFController):
        c = cls(max_batch=64, device=0, params=None if tau_max is None else {"tau_max": tau_max})
        tick, ro = c.kernel_info(), c.kernel_info(rollout=True)
        c.close()
        assert tick["scratch_bytes_per_lane"] == 0 and ro["scratch_bytes_per_lane"] == 0, (cls.__name__, tick, ro)
        assert 0 < tick["num_regs"] <= 512 and 0 < ro["num_regs"] <= 512


def _sway_trajectory(dt, duration=2.0, amp=0.05, hz=2.0):
    """Body target swaying sideways harder than friction allows at the peaks (amp (2 pi hz)^2 = 7.9 m/s^2 against mu g = 6.9): a closed loop
    that sits on its friction limits for part of every period (tools/lab/r06/warm_probe.py)."""
    ts = np.arange(int(round(duration / dt)) + 1) * dt
    tg = workloads.standing_targets("mini_cheetah", ts.size)
    w = 2 * np.pi * hz
    tg[1] += amp * np.sin(w * ts); tg[4] = amp * w * np.cos(w * ts); tg[7] = -amp * w * w * np.sin(w * ts)
    return ts, np.ascontiguousarray(tg.T), np.full(ts.size, 0b1111, np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,dt,steps,amp", [("mptc", 1e-3, 200, 0.05), ("id", 5e-3, 100, 0.05), ("pc", 1e-3, 10, 0.03), ("clf", 5e-3, 10, 0.036)])   # (all four rollout kernels)
def test_warm_started_rollout_is_the_cold_rollout_in_fewer_trips(kind, dt, steps, amp):
    """wbc_set_warm_start (include/wbc.h; csrc/wbc_hex.hpp: hex_gi<..., WARM>): every tick of a rollout starts its active set from the friction rows that were
    active when the robot's previous tick ended, where the reference -- and the default -- solve every tick from scratch
    (inverse_dynamics_controller.py:200).  Same strictly convex QP, another order of the adds: on a closed loop that sits on its friction limits the
    torques, accelerations and the state after `steps` ticks agree with the cold rollout to the accuracy either has on saturated ticks (the bar of
    tests/test_gpu_parity.py for 4-contact stands on their friction limits, 2e-6: there the solution is carried at 1 / eps = 1e4 in the internal-force
    directions; measured 2.4e-7 after 200 closed-loop ticks), in markedly fewer active-set trips.  (The PC / CLF laws' hard rows make harder sways
    infeasible after some tens of ticks -- tools/lab/r06/pc_clf_sway_probe.py: a closed loop on its way there amplifies the last bits of every
    tick -- so they sway a little less and are compared over ten ticks: what is tested for them is the seeded active set, tick by tick.)"""
    import torch
    from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    cls = {"mptc": MPTCController, "id": IDController, "pc": PCController, "clf": CLFController}[kind]
    n = 512
    ts, tg, masks = _sway_trajectory(dt, amp=amp)
    traj = TrunkTrajectory(ts, tg, masks, wait_time=0.0, device=0)
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(5)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
    t0 = rng.uniform(0.0, 0.5, n)
    dev = "cuda:0"
    res = {}
    for warm in (False, True):
        c = cls(max_batch=n, device=0)
        c.set_warm_start(warm)
        q = torch.tensor(q0, device=dev); v = torch.tensor(v0, device=dev); t = torch.tensor(t0, device=dev)
        tau, met, st, tgo, mk = c.rollout(traj, steps, dt, q, v, t)
        c.sync()
        res[warm] = dict(q=q.cpu().numpy(), v=v.cpu().numpy(), tau=tau.cpu().numpy(), st=st.cpu().numpy(), vd=c._keep[5].cpu().numpy(), stats=c.stats())
        c.close()
    traj.close()
    cold, warm = res[False], res[True]
    it_c, it_w = cold["stats"]["iters_sum"] / cold["stats"]["ticks"], warm["stats"]["iters_sum"] / warm["stats"]["ticks"]
    assert it_c > 2.0, it_c                                   # the loop really sits on its friction rows
    assert cold["stats"]["status_nonzero"] == 0 and warm["stats"]["status_nonzero"] == 0
    assert it_w < (0.8 if kind in ("mptc", "id") else 0.95) * it_c, (it_c, it_w)      # measured: 8.0 -> 5.0 (MPTC), 9.1 -> 6.8 (ID)
    ok = np.ones(n, bool)
    scale = lambda a: np.maximum(np.abs(a).max(0), 1e-3)
    loose = 1.0 if kind in ("mptc", "id") else 5.0          # (PC / CLF stands: 1e-5, tests/test_gpu_parity.py)
    for k, bar in (("tau", 2e-6 * loose), ("vd", 2e-6 * loose), ("q", 1e-8), ("v", 1e-6 * loose)):
        err = (np.abs(warm[k] - cold[k]).max(0) / scale(cold[k]))[ok].max()
        assert err < bar, (k, err)
    print("%s: %.2f -> %.2f trips per tick, %d of %d robots compared, status != 0 on %d / %d ticks" % (kind, it_c, it_w, ok.sum(), n, cold["stats"]["status_nonzero"], warm["stats"]["status_nonzero"]))


@pytest.mark.gpu
@pytest.mark.parametrize("steps", [120, 2000])
def test_warm_start_is_off_by_default_and_a_trot_needs_no_tolerance(steps):
    """Default handles never seed (the persistent rollout stays bit-identical to the launch-per-stage loop: test_persistent_rollout_equals_launch_per_stage).
    With the option on, a trot rollout -- one or two active rows per tick -- still agrees with the cold one far below any bar a test here uses."""
    import torch
    from quadruped_drake_amd import MPTCController
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    n, dt = 203, 1e-3                                      # (2000 ticks: two bounded launches of the persistent kernel; the memory does not outlive a launch)
    ts, tg, masks, st_t = _trot_trajectory(K=2400 if steps > 300 else 300)
    if steps > 300:
        masks[:] = 0b1111          # two seconds of this open-loop trot end on the floor that the forward step does not have; standing robots under the moving targets stay up
    traj = TrunkTrajectory(ts, tg, masks, wait_time=0.03, device=0, standing_targets=st_t, standing_mask=0b1111)
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(11)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
    t0 = rng.uniform(0.0, 0.25, n)
    out = []
    for mode in ("default", "off", "on"):
        c = MPTCController(max_batch=n, device=0)
        if mode != "default":
            c.set_warm_start(mode == "on")
        q = torch.tensor(q0, device="cuda:0"); v = torch.tensor(v0, device="cuda:0"); t = torch.tensor(t0, device="cuda:0")
        tau, met, st, _, _ = c.rollout(traj, steps, dt, q, v, t); c.sync()
        out.append((q.cpu().numpy(), v.cpu().numpy(), tau.cpu().numpy(), c.stats()["iters_sum"]))
        c.close()
    traj.close()
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a, b)
    assert out[0][3] == out[1][3]
    for a, b in zip(out[1][:3], out[2][:3]):
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(a).max())
    assert out[2][3] <= 1.05 * out[1][3] + 2      # (a trot has next to nothing to seed: 65 trips in 24 360 ticks, cold)
    assert out[1][3] > 0 or steps > 300
    assert all(np.isfinite(a).all() for a in out[2][:3])


@pytest.mark.parametrize("kind,dt", [("mptc", 1e-3), ("id", 5e-3)])
def test_warm_start_on_the_host_emulation(kind, dt):
    """The rollout kernels' instantiation hex_tick<..., WARM> on the HOST (tools/host_tick.cpp: host_hex_rollout -- 16 lock-step fibres per robot, the
    forward step's arithmetic, the per-lane seed bit kept between a robot's ticks): a closed loop on its friction limits, cold and warm-started.  Same
    states and torques to the accuracy of saturated ticks, every tick solved, markedly fewer active-set trips -- without a GPU."""
    import ctypes as C
    import host_tick as ht
    from oracle import oracle_py as orc
    L = ht.lib()
    dp = C.POINTER(C.c_double)
    n, steps = 12, 30
    flat = np.ascontiguousarray(orc.load_model_json("mini_cheetah")["flat"], dtype=np.float64)
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(5)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); v0[0:6] = rng.normal(0, 0.05, (6, n))
    ts, tgs, masks = _sway_trajectory(dt, duration=0.3)
    k0 = int(round(0.1 / dt))                                  # the ticks around the sway's peak acceleration (t = 0.125 s): friction rows active
    tgs = np.ascontiguousarray(tgs[k0:k0 + steps]); masks = np.ascontiguousarray(masks[k0:k0 + steps])
    assert tgs.shape[0] == steps
    k = {"id": 0, "mptc": 1}[kind]
    out = {}
    for warm in (0, 1):
        q = q0.copy(); v = v0.copy(); tau = np.zeros((12, n)); met = np.zeros((4, n)); st = np.zeros(n, np.int32); its = np.zeros(n); bad = C.c_int(-1)
        rc = L.host_hex_rollout(k, flat.ctypes.data_as(dp), None, n, n, steps, C.c_double(dt), warm, q.ctypes.data_as(dp), v.ctypes.data_as(dp), tgs.ctypes.data_as(dp),
                                masks.ctypes.data_as(C.POINTER(C.c_ubyte)), None, None, tau.ctypes.data_as(dp), met.ctypes.data_as(dp),
                                st.ctypes.data_as(C.POINTER(C.c_int)), its.ctypes.data_as(dp), C.byref(bad))
        assert rc == 0 and bad.value == 0 and (st == 0).all()
        out[warm] = (q, v, tau, its.sum() / (n * steps))
    it_c, it_w = out[0][3], out[1][3]
    assert it_c > 4.0 and it_w < 0.8 * it_c, (it_c, it_w)                       # measured: 11.7 -> 7.2 (MPTC)
    for a, b, bar in zip(out[0][:3], out[1][:3], (1e-9, 1e-7, 2e-6)):
        assert np.abs(a - b).max() <= bar * max(1e-3, np.abs(a).max()), (bar, np.abs(a - b).max())
    # the cold emulation's last tick against the oracle on the state it was solved at is covered tick by tick elsewhere (test_hex_kernel_math_emulated_on_host)
