#!/usr/bin/env python3
"""Golden vectors produced by EXECUTING the reference's own controller code
(/root/reference/controllers/{inverse_dynamics,mptc,pc,clf}_controller.py + basic_controller.py + helpers.py:
imported from where they lie, never copied) on this repository's seeded inputs.

What runs and what does not.  The controllers are Drake LeafSystems; their ControlLaw is numpy arithmetic around three
services of pydrake, none of which is in this image: a MultibodyPlant (rigid-body terms), MathematicalProgram (a
container for numeric costs / constraints) and OsqpSolver.  tests/fake_pydrake supplies stand-ins:
  * refplant.RefPlant      rigid-body terms from oracle/ in Drake's documented conventions; the reference's two autodiff
                           recipes (Coriolis matrix = 1/2 d(Cv)/dv, Jdot = dJ/dq N(q) v) are served as derivatives of
                           the oracle's Cv and J, not with the oracle's own C / Jdot;
  * mathprog               records the QP exactly as the reference's Add* builders state it, then solves it exactly
                           (extended-precision KKT active set, independent of oracle/ and of the kernels) under this
                           repository's tie-break (+ 1/2 eps2 |[tau; f (; delta)]|^2) -- NOT OSQP.
So these fixtures pin, against the reference's executed code: target / gain arithmetic, RPY handling, the task-space
terms (Lambda, Jbar, Q, f_des), the Coriolis-matrix and Jdot definitions, QP assembly (cost forms, signs, rows,
variable order) and the logged metrics.  They do NOT pin Drake's rigid-body numbers (supplied by oracle/; checked
separately against tests/energy_model.py) or OSQP's selection among the optimal set: parity at the Drake / OSQP
boundary stays unpinned.

Output: reference_law_golden.npz -- per case set `<set>_*`: inputs (q, v, targets, mask[, mu, mass_scale]) and what the
reference's code returned: tau[12], metrics [V, err, res, Vdot] (res = the stand-in solver's equality residual), the
QP solution's vd[18] and contact forces f[12] (LF RF LH RH, zero for swing feet), active-set size.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(HERE, "..", "fake_pydrake"))
sys.path.insert(0, os.path.join(HERE, ".."))      # tests/energy_model.py (the "energy" plant backend)
sys.path.insert(0, ROOT)
sys.modules["lcm"] = types.ModuleType("lcm")      # import-only stub: use_lcm=False everywhere below
np.object = object                                # helpers.py:19 uses the alias numpy removed in 1.24
sys.path.insert(0, "/root/reference")
import pydrake.all as fake                         # noqa: E402
from pydrake.mathprog import OsqpSolver            # noqa: E402
from controllers import BasicController, IDController, MPTCController, PCController, CLFController   # noqa: E402  (reference code)
from oracle import oracle_py as orc                # noqa: E402  (rigid-body terms of the stand-in plant)

LAWS = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}
FEET = ("lf", "rf", "lh", "rh")
BODY = ("p_body", "pd_body", "pdd_body", "rpy_body", "rpyd_body", "rpydd_body")


def trunk_dict(t54, mask):
    """include/wbc.h target rows -> the planner dictionary of planners/simple.py:45-85"""
    d = {k: t54[3 * i:3 * i + 3].copy() for i, k in enumerate(BODY)}
    for i, f in enumerate(FEET):
        for j, pre in enumerate(("p_", "pd_", "pdd_")):
            d[pre + f] = t54[18 + 9 * i + 3 * j:21 + 9 * i + 3 * j].copy()
    d["contact_states"] = [bool((int(mask) >> i) & 1) for i in range(4)]
    d["f_cj"] = np.zeros((3, 4)); d["u2_max"] = 0.0
    return d


def run_set(kind, model, q, v, tg, mask, mu=None, mass_scale=None, order=None, act_joint=None, backend="oracle"):
    """q, v in CANONICAL joint order; `order` / `act_joint`: the plant's own numbering (refplant.RefPlant)."""
    plant = fake.RefPlant(model, body_frame="body", order=order, act_joint=act_joint, backend=backend)   # "body": basic_controller.py:65
    if order is not None:
        qd, vd_ = q.copy(), v.copy()
        for j in range(12):
            qd[7 + order[j]] = q[7 + j]; vd_[6 + order[j]] = v[6 + j]
        q, v = qd, vd_
    ctrl = LAWS[kind](plant, 5e-3)
    n = q.shape[1]
    out = dict(tau=np.zeros((12, n)), metrics=np.zeros((4, n)), vd=np.zeros((18, n)), f=np.zeros((12, n)),
               nactive=np.zeros(n, np.int32))
    for i in range(n):
        if mu is not None:
            ctrl.mu = float(mu[i])                        # the reference's friction coefficient is this attribute
        if mass_scale is not None:
            plant.m = orc.model_scaled(model, float(mass_scale[i])); ctrl.plant_autodiff.m = plant.m
        ctx = ctrl.CreateDefaultContext()
        ctrl.get_input_port(0).FixValue(ctx, np.concatenate([q[:, i], v[:, i]]))
        ctrl.get_input_port(1).FixValue(ctx, trunk_dict(tg[:, i], mask[i]))
        ctrl.V = ctrl.err = ctrl.res = ctrl.Vdot = 0
        out["tau"][:, i] = ctrl.get_output_port(0).Eval(ctx)            # DoSetControlTorques -> ControlLaw
        out["metrics"][:, i] = ctrl.get_output_port(1).Eval(ctx)        # SetLoggingOutputs
        last = OsqpSolver.last
        x = last["x"]
        out["vd"][:, i] = x[:18][plant.pv]            # back to canonical order
        ct = [k for k in range(4) if (int(mask[i]) >> k) & 1]
        for j, k in enumerate(ct):
            out["f"][3 * k:3 * k + 3, i] = x[30 + 3 * j:33 + 3 * j]
        out["nactive"][i] = len(last["active"])
    return out


gold = {}
SETS = [("cfg2_id", "id", 16), ("cfg3_id", "id", 16), ("cfg3_mptc", "mptc", 16), ("cfg4_anymal_mptc", "mptc", 8),
        ("cfg5_rand_mptc", "mptc", 8), ("cfg3_pc", "pc", 16), ("cfg2_pc", "pc", 8), ("cfg3_clf", "clf", 16),
        ("cfg2_clf", "clf", 8), ("masks16_id", "id", 16), ("masks16_mptc", "mptc", 16), ("masks16_pc", "pc", 16),
        ("masks16_clf", "clf", 16)]
for name, kind, n in SETS:
    z = np.load(os.path.join(HERE, name + ".npz"))
    sel = np.arange(n)
    if name.startswith("masks16") and kind in ("mptc", "pc"):
        sel = sel[z["mask"][:n] != 0]      # the reference's MPTC / PC cannot run in flight (np.vstack of no rows, mptc_controller.py:303)
    q, v, tg, mk = z["q"][:, sel], z["v"][:, sel], z["targets"][:, sel], z["mask"][sel]
    mu = z["mu"][sel] if z["mu"].size else None
    ms = z["mass_scale"][sel] if z["mass_scale"].size else None
    r = run_set(kind, str(z["model"]), q, v, tg, mk, mu, ms)
    gold[name + "_kind"] = kind; gold[name + "_model"] = str(z["model"])
    gold[name + "_q"], gold[name + "_v"], gold[name + "_targets"], gold[name + "_mask"] = q, v, tg, mk
    gold[name + "_mu"] = np.zeros(0) if mu is None else mu
    gold[name + "_mass_scale"] = np.zeros(0) if ms is None else ms
    for k, a in r.items():
        gold[name + "_" + k] = a
    # how the oracle's own outputs for the same inputs (already committed in <name>.npz) compare
    e = np.abs(r["tau"] - z["tau"][:, sel]).max(0) / np.maximum(np.abs(z["tau"][:, sel]).max(0), 1e-3)
    print("%-18s %-4s n=%2d  tau vs oracle: max rel %.2e  median %.2e   active rows %s" %
          (name, kind, len(sel), e.max(), np.median(e), np.bincount(r["nactive"]).tolist()))
# the plant numbers its joints breadth-first (all abduction joints first) and its actuators at random: what
# basic_controller.py:310-313 warns about.  tau comes back in ACTUATOR order; inputs are stored in canonical order.
ORDER = [4 * (j % 3) + j // 3 for j in range(12)]
ACT = [int(x) for x in np.random.default_rng(2).permutation(12)]
for name, kind, n in (("cfg2_id", "id", 8), ("cfg3_mptc", "mptc", 8), ("cfg4_anymal_mptc", "mptc", 4)):
    z = np.load(os.path.join(HERE, name + ".npz"))
    q, v, tg, mk = z["q"][:, :n], z["v"][:, :n], z["targets"][:, :n], z["mask"][:n]
    r = run_set(kind, str(z["model"]), q, v, tg, mk, order=ORDER, act_joint=ACT)
    pn = "perm_" + name
    gold[pn + "_kind"] = kind; gold[pn + "_model"] = str(z["model"])
    gold[pn + "_q"], gold[pn + "_v"], gold[pn + "_targets"], gold[pn + "_mask"] = q, v, tg, mk
    gold[pn + "_mu"] = np.zeros(0); gold[pn + "_mass_scale"] = np.zeros(0)
    gold[pn + "_order"], gold[pn + "_act_joint"] = np.array(ORDER), np.array(ACT)
    for k, a in r.items():
        gold[pn + "_" + k] = a
    e = np.abs(r["tau"] - z["tau"][:, :n][ACT]).max(0) / np.maximum(np.abs(z["tau"][:, :n]).max(0), 1e-3)
    print("%-18s %-4s n=%2d  permuted plant, tau vs oracle[act]: max rel %.2e" % (pn, kind, n, e.max()))
# the reference's own experiments: simulate.py:171-179 initial state with the dictionaries its BasicTrunkPlanner produces
# for its scenarios (planner_golden.npz, made by make_planner_golden.py from planners/simple.py) -- planner code and
# controller code of the reference chained, as simulate.py wires them
pg = np.load(os.path.join(HERE, "planner_golden.npz"))
PKEYS = list(BODY) + [pre + f for f in FEET for pre in ("p_", "pd_", "pdd_")]


def planner_case(prefix, i=None):
    pick = (lambda a: a) if i is None else (lambda a: a[i])
    t = np.concatenate([pick(pg[prefix + k]) for k in PKEYS])          # include/wbc.h row order == PKEYS order
    return t, sum(1 << b for b, c in enumerate(pick(pg[prefix + "contact_states"])) if c)


cases = [planner_case("basic_standing_"), planner_case("basic_edge_"), planner_case("basic_raisefoot_", 1),
         planner_case("basic_raisefoot_", 4), planner_case("basic_orientation_", 3), planner_case("basic_orientation_", 6)]
q0 = np.array([1.0, 0, 0, 0, 0, 0, 0.3] + [0.0, -0.8, 1.6] * 4); v0 = np.zeros(18)     # simulate.py:171-179
for kind in ("id", "mptc", "pc", "clf"):
    n = len(cases)
    q = np.tile(q0[:, None], (1, n)); v = np.tile(v0[:, None], (1, n))
    tg = np.stack([c[0] for c in cases], axis=1); mk = np.array([c[1] for c in cases], np.uint8)
    r = run_set(kind, "mini_cheetah", q, v, tg, mk)
    pn = "scen_" + kind
    gold[pn + "_kind"] = kind; gold[pn + "_model"] = "mini_cheetah"
    gold[pn + "_q"], gold[pn + "_v"], gold[pn + "_targets"], gold[pn + "_mask"] = q, v, tg, mk
    gold[pn + "_mu"] = np.zeros(0); gold[pn + "_mass_scale"] = np.zeros(0)
    for k, a in r.items():
        gold[pn + "_" + k] = a
    tau_o, _, st_o = orc.step_batch(kind, orc.model("mini_cheetah"), orc.params(kind), q, v, tg, mk)
    e = np.abs(r["tau"] - tau_o).max(0) / np.maximum(np.abs(tau_o).max(0), 1e-3)
    print("%-22s %-4s n=%2d  reference scenarios at q0, tau vs oracle: max rel %.2e   active rows %s" %
          (pn, kind, n, e.max(), r["nactive"].tolist()))

# closed loop: the reference's planner scenario (RaiseFoot across its contact switch at t > 1) -> the reference's
# controller code -> the forward step of include/wbc.h (oracle/traj_oracle.integrate, semi-implicit Euler on the QP's own
# accelerations), 60 ticks.  Pins the chain lookup -> tick -> integrate of wbc_rollout against the executed reference.
from oracle import traj_oracle as to_      # noqa: E402
sys.path.insert(0, ROOT)
from planners.simple import BasicTrunkPlanner   # noqa: E402  (reference code)
bp = BasicTrunkPlanner({"trunk": 0, "lf": 1, "rf": 2, "lh": 3, "rh": 4})
for kind, dt, t_start in (("id", 5e-3, 0.85), ("mptc", 1e-3, 0.97), ("pc", 1e-3, 0.97), ("clf", 5e-3, 0.85)):
    plant = fake.RefPlant("mini_cheetah", body_frame="body")
    ctrl = LAWS[kind](plant, dt)
    steps = 60
    # start from a state already shifted over the support triangle (what the scenario has reached by then)
    q = q0.copy(); q[4:7] = [-0.1, 0.05, 0.3]; v = v0.copy()
    if kind == "pc":
        # a start velocity under which the passivity row Vdot <= 0 (pc_controller.py:229-237) becomes ACTIVE in closed loop -- ticks
        # 47 .. 59 of the window, after the contact switch (with v = 0 it never is and the trajectory equals MPTC's bit for bit)
        rng_pc = np.random.default_rng(329)
        v[:3] = rng_pc.normal(0, 1.0, 3); v[3:6] = rng_pc.normal(0, 0.6, 3); v[6:] = rng_pc.normal(0, 2.0, 12)
    Q, V, T, TAU, VD = [q.copy()], [v.copy()], [], [], []
    for k in range(steps):
        t = t_start + k * dt
        bp.RaiseFoot(t)
        d = dict(bp.output_dict)
        ctx = ctrl.CreateDefaultContext()
        ctrl.get_input_port(0).FixValue(ctx, np.concatenate([q, v]))
        ctrl.get_input_port(1).FixValue(ctx, d)
        tau = ctrl.get_output_port(0).Eval(ctx)
        vd = OsqpSolver.last["x"][:18]
        qn, vn = to_.integrate(q[:, None], v[:, None], vd[:, None], dt)
        q, v = qn[:, 0], vn[:, 0]
        Q.append(q.copy()); V.append(v.copy()); T.append(t); TAU.append(np.array(tau)); VD.append(float(ctrl.Vdot))
    pn = "closedloop_" + kind
    gold[pn + "_dt"] = dt; gold[pn + "_times"] = np.array(T)
    gold[pn + "_q"] = np.array(Q).T; gold[pn + "_v"] = np.array(V).T; gold[pn + "_tau"] = np.array(TAU).T
    gold[pn + "_vdot_metric"] = np.array(VD)
    print("%-22s %d ticks from t = %.3f, dt = %g: RF foot contact %s -> %s, |v|max %.3f" %
          (pn, steps, t_start, dt, True, bool(d["contact_states"][1]), np.abs(np.array(V)).max()))

# the joint-space PD law (control method "B", BasicController.ControlLaw): random states, some far enough from the nominal
# pose for the +-150 clip, on the identity plant and on the permuted one
z = np.load(os.path.join(HERE, "cfg2_id.npz"))
rng_pd = np.random.default_rng(31)
qpd, vpd = z["q"][:, :24].copy(), z["v"][:, :24].copy()
qpd[7:, 12:] += rng_pd.uniform(-8.0, 8.0, (12, 12)); vpd[6:, 18:] *= 40.0
for pn, order, act in (("pd_identity", None, None), ("pd_perm", ORDER, ACT)):
    plant = fake.RefPlant("mini_cheetah", body_frame="body", order=order, act_joint=act)
    ctrl = BasicController(plant, 5e-3)
    U = np.zeros((12, qpd.shape[1]))
    for i in range(qpd.shape[1]):
        qd, vd_ = qpd[:, i].copy(), vpd[:, i].copy()
        if order is not None:
            for j in range(12):
                qd[7 + order[j]] = qpd[7 + j, i]; vd_[6 + order[j]] = vpd[6 + j, i]
        ctx = ctrl.CreateDefaultContext()
        ctrl.get_input_port(0).FixValue(ctx, np.concatenate([qd, vd_]))
        U[:, i] = ctrl.get_output_port(0).Eval(ctx)
    gold[pn + "_q"], gold[pn + "_v"], gold[pn + "_u"] = qpd, vpd, U
    gold[pn + "_order"] = np.arange(12) if order is None else np.array(order)
    gold[pn + "_act_joint"] = np.arange(12) if act is None else np.array(act)
    print("%-22s PD law, %d ticks, clipped entries: %d" % (pn, U.shape[1], int((np.abs(U) == 150.0).sum())))

# NOTHING SHARED: the same reference code over a plant whose rigid-body terms come from tests/energy_model.py (plain FK +
# Kane projection, numerically differentiated twists) instead of oracle/ -- reference law code + independent dynamics +
# independent solver.  Looser by construction (finite differences inside the dynamics): tests allow 1e-5.
for name, kind, n in (("cfg2_id", "id", 6), ("cfg3_mptc", "mptc", 6), ("cfg4_anymal_mptc", "mptc", 3), ("cfg3_clf", "clf", 3),
                      ("cfg3_pc", "pc", 3)):
    z = np.load(os.path.join(HERE, name + ".npz"))
    q, v, tg, mk = z["q"][:, :n], z["v"][:, :n], z["targets"][:, :n], z["mask"][:n]
    r = run_set(kind, str(z["model"]), q, v, tg, mk, backend="energy")
    pn = "indep_" + name
    gold[pn + "_kind"] = kind; gold[pn + "_model"] = str(z["model"])
    gold[pn + "_q"], gold[pn + "_v"], gold[pn + "_targets"], gold[pn + "_mask"] = q, v, tg, mk
    gold[pn + "_mu"] = np.zeros(0); gold[pn + "_mass_scale"] = np.zeros(0)
    for k, a in r.items():
        gold[pn + "_" + k] = a
    e = np.abs(r["tau"] - z["tau"][:, :n]).max(0) / np.maximum(np.abs(z["tau"][:, :n]).max(0), 1e-3)
    print("%-22s %-4s n=%2d  independent dynamics, tau vs oracle: max rel %.2e  median %.2e" % (pn, kind, n, e.max(), np.median(e)))
np.savez_compressed(os.path.join(os.environ.get("GOLDEN_OUT", HERE), "reference_law_golden.npz"), **gold)   # GOLDEN_OUT: tests/test_fixture_freshness.py
print("wrote reference_law_golden.npz")
