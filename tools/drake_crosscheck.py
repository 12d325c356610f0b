#!/usr/bin/env python3
"""The one script that closes the parity gap at the Drake / OSQP boundary -- on a machine that HAS pydrake.

    python3 tools/drake_crosscheck.py --reference /path/to/quadruped_drake            # real Drake + OSQP
    python3 tools/drake_crosscheck.py --reference /root/reference --fake             # plumbing test (this image: no pydrake)

The reference's arithmetic for the hot path lives in Drake's MultibodyPlant and in OSQP;
neither is in this repository's build image, so every parity statement made here is against oracle/ -- a restatement.  This
script removes that qualifier wherever pydrake is installed.  It is NOT shipped to the GPU box with reference code; it only
imports the reference from the path it is given.

Stage A -- rigid-body numbers.  The two URDFs are loaded the way simulate.py:31-64 loads
  them (MultibodyPlant(time_step=dt), Parser.AddModelFromFile, ground half-space, Finalize) and, on the inputs of the
  committed fixtures tests/golden/cfg*.npz, the Drake calls the controllers make are evaluated:
    basic_controller.py:110-113   CalcMassMatrixViaInverseDynamics, CalcBiasTerm, CalcGravityGeneralizedForces, MakeActuationMatrix
    basic_controller.py:180-195   CalcPointsPositions, CalcJacobianTranslationalVelocity, CalcBiasTranslationalAcceleration (4 feet)
    basic_controller.py:253-267   CalcRelativeTransform, CalcJacobianSpatialVelocity, CalcBiasSpatialAcceleration (body)
  and compared with oracle/ (M, Cv, tau_g, S, foot p / J / Jdv, body R / p / J / Jdv), joint order mapped through the plant's
  own numbering (velocity_start()).
Stage B -- the controllers with the real solver.  The reference's IDController / MPTCController (/ PC / CLF) are run through
  their LeafSystem ports on the same inputs (OsqpSolver as constructed at inverse_dynamics_controller.py:23), and the
  solver-independent quantities -- tier (i) of DESIGN.md section 2: the accelerations v-dot = x[:18] and the QP's cost --
  are compared with oracle/'s literal QP solved under this repository's tie-break; the torques (tier ii: OSQP picks its own
  element of the optimal set) are printed for information.  With --gpu the HIP path is compared as well.

Exit code 0 when every Stage-A quantity agrees within --tol-plant (1e-9 relative) and every Stage-B acceleration within
--tol-vd (1e-3 relative: OSQP's own stock tolerances).  With --fake the plant stand-in answers from oracle/ itself, so the
numbers agree trivially: that run only proves the plumbing (joint renumbering, port protocol, solution capture).
"""
import argparse
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FEET = ("LF_FOOT", "RF_FOOT", "LH_FOOT", "RH_FOOT")
FEET_L = ("lf", "rf", "lh", "rh")
BODY_KEYS = ("p_body", "pd_body", "pdd_body", "rpy_body", "rpyd_body", "rpydd_body")
URDF = {"mini_cheetah": "models/mini_cheetah/mini_cheetah_mesh.urdf",                      # simulate.py:31
        "anymal_b": "models/anymal_b_simple_description/urdf/anymal_drake.urdf"}
BODY_FRAME = {"mini_cheetah": "body", "anymal_b": "base"}                                  # basic_controller.py:65 comment


def parse():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default="/root/reference", help="checkout of vincekurtz/quadruped_drake")
    ap.add_argument("--fake", action="store_true", help="use tests/fake_pydrake instead of pydrake (plumbing test)")
    ap.add_argument("--cases", nargs="*", default=["cfg2_id", "cfg3_mptc", "cfg3_id", "cfg4_anymal_mptc", "cfg5_rand_mptc", "cfg3_pc", "cfg3_clf"])
    ap.add_argument("--n", type=int, default=8, help="instances per fixture")
    ap.add_argument("--tol-plant", type=float, default=1e-9)
    ap.add_argument("--tol-vd", type=float, default=1e-3)
    ap.add_argument("--gpu", action="store_true", help="also run the HIP path (needs a GPU and the built library)")
    ap.add_argument("--dt", type=float, default=5e-3)
    return ap.parse_args()


def import_pydrake(fake):
    sys.path.insert(0, ROOT)
    if fake:
        sys.path.insert(0, os.path.join(ROOT, "tests", "fake_pydrake"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import pydrake.all as pd
    except ImportError:
        sys.exit("drake_crosscheck: pydrake is not importable here.  Run this on a machine with Drake installed, or pass --fake "
                 "for the plumbing test against tests/fake_pydrake.")
    return pd


def build_plant(pd, ref_root, model, dt, fake):
    """simulate.py:31-64: MultibodyPlant(time_step=dt) + URDF + ground half-space (mu = 1) + Finalize."""
    if fake:   # breadth-first joints, shuffled actuators: the renumbering hazard of basic_controller.py:310-313 is part of the plumbing
        order = [4 * (j % 3) + j // 3 for j in range(12)]
        act = [int(x) for x in np.random.default_rng(2).permutation(12)]
        return pd.RefPlant(model, body_frame="body", order=order, act_joint=act)     # "body": the name the reference hard-codes
    builder = pd.DiagramBuilder()
    scene_graph = builder.AddSystem(pd.SceneGraph())
    plant = builder.AddSystem(pd.MultibodyPlant(time_step=dt))
    plant.RegisterAsSourceForSceneGraph(scene_graph)
    urdf = os.path.join(ref_root, URDF[model])
    parser = pd.Parser(plant=plant)
    if hasattr(parser, "AddModelFromFile"):
        parser.AddModelFromFile(urdf, "quad")          # the call simulate.py:40 makes (Drake of 2021-22)
    else:
        parser.AddModels(urdf)                         # its successor in current Drake
    try:       # the ground of simulate.py:43-52; it takes no part in the quantities compared here, and its signature moved between releases
        plant.RegisterCollisionGeometry(plant.world_body(), pd.RigidTransform(), pd.HalfSpace(), "ground_collision",
                                        pd.CoulombFriction(static_friction=1.0, dynamic_friction=1.0))
    except Exception as e:   # noqa: BLE001
        print("drake_crosscheck: ground half-space not registered (%s): irrelevant for stage A / B" % type(e).__name__, file=sys.stderr)
    plant.Finalize()
    return plant


def plant_order(plant, table):
    """canonical (leg-major) joint j sits at velocity 6 + order[j] of the plant; actuator k drives canonical joint act[k]."""
    names = [l["joint"] for leg in table["legs"] for l in leg["links"]]
    order = [plant.GetJointByName(nm).velocity_start() - 6 for nm in names]
    B = np.asarray(plant.MakeActuationMatrix())
    act = [order.index(int(np.argmax(np.abs(B[6:, k])))) for k in range(12)]
    return order, act


def to_plant(q, v, order):
    qp, vp = q.copy(), v.copy()
    for j in range(12):
        qp[7 + order[j]] = q[7 + j]; vp[6 + order[j]] = v[6 + j]
    return qp, vp


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def stage_a(pd, plant, model, order, q, v, orc):
    """One state: every Drake call of the path against oracle/.  Returns {quantity: relative difference}."""
    pv = np.array(list(range(6)) + [6 + o for o in order])          # canonical i <-> plant pv[i]
    ctx = plant.CreateDefaultContext()
    qp, vp = to_plant(q, v, order)
    plant.SetPositions(ctx, qp); plant.SetVelocities(ctx, vp)
    W = plant.world_frame()
    m = orc.model(model)
    out = {}
    M = np.asarray(plant.CalcMassMatrixViaInverseDynamics(ctx))[np.ix_(pv, pv)]     # basic_controller.py:110
    Cv = np.asarray(plant.CalcBiasTerm(ctx))[pv]                                       # :111
    tg = -np.asarray(plant.CalcGravityGeneralizedForces(ctx))[pv]                      # :112 (note the sign)
    Mo, Cvo, tgo = orc.calc_dynamics(m, q, v)[:3]
    out["M"], out["Cv"], out["tau_g"] = rel(M, Mo), rel(Cv, Cvo), rel(tg, tgo)
    kV = pd.JacobianWrtVariable.kV
    for f, name in enumerate(FEET):
        fr = plant.GetFrameByName(name)
        p = np.asarray(plant.CalcPointsPositions(ctx, fr, np.zeros(3), W)).reshape(3)                       # :180-183
        J = np.asarray(plant.CalcJacobianTranslationalVelocity(ctx, kV, fr, np.zeros(3), W, W))[:, pv]      # :184-189
        Jdv = np.asarray(plant.CalcBiasTranslationalAcceleration(ctx, kV, fr, np.zeros(3), W, W)).reshape(3)  # :190-195
        po, Jo, Jdvo = orc.foot_quantities(m, q, v, f)[:3]
        out["p_" + FEET_L[f]], out["J_" + FEET_L[f]], out["Jdv_" + FEET_L[f]] = rel(p, po), rel(J, Jo), rel(Jdv, Jdvo)
    fb = plant.GetFrameByName("body" if hasattr(plant, "backend") else BODY_FRAME[model])
    X = plant.CalcRelativeTransform(ctx, W, fb)                                                              # :253-255
    Jb = np.asarray(plant.CalcJacobianSpatialVelocity(ctx, kV, fb, np.zeros(3), W, W))[:, pv]                # :256-261
    Jdvb = np.asarray(plant.CalcBiasSpatialAcceleration(ctx, kV, fb, np.zeros(3), W, W).get_coeffs())        # :262-267
    bq = orc.body_quantities(m, q, v)
    Ro, po, Jbo, Jdvbo = bq[0], bq[1], bq[2], bq[3]
    out["R_body"] = rel(np.asarray(X.rotation().matrix()), np.asarray(Ro).reshape(3, 3))
    out["p_body"] = rel(np.asarray(X.translation()), po)
    out["J_body"] = rel(Jb, Jbo)
    out["Jdv_body"] = float(np.abs(np.asarray(Jdvb) - np.asarray(Jdvbo)).max())          # both are ~0: absolute
    return out


def trunk_dict(t54, mask):
    """include/wbc.h target rows -> the planner dictionary of planners/simple.py:45-85"""
    d = {k: t54[3 * i:3 * i + 3].copy() for i, k in enumerate(BODY_KEYS)}
    for i, f in enumerate(FEET_L):
        for j, pre in enumerate(("p_", "pd_", "pdd_")):
            d[pre + f] = t54[18 + 9 * i + 3 * j:21 + 9 * i + 3 * j].copy()
    d["contact_states"] = [bool((int(mask) >> i) & 1) for i in range(4)]
    d["f_cj"] = np.zeros((3, 4)); d["u2_max"] = 0.0
    return d


class SolveSpy:
    """Wraps the controller's solver (inverse_dynamics_controller.py:23,223): keeps the last result's full solution."""

    def __init__(self, solver):
        self.solver, self.x, self.cost = solver, None, None

    def Solve(self, prog, *a, **k):
        r = self.solver.Solve(prog, *a, **k)
        try:
            self.x = np.asarray(r.GetSolution(prog.decision_variables()), float)
            self.cost = float(r.get_optimal_cost())
        except AttributeError:                       # tests/fake_pydrake: the stand-in keeps its last solve
            last = type(self.solver).last
            self.x, self.cost = np.asarray(last["x"], float), None
        return r


def stage_b(pd, ref, plant, kind, model, order, act, q, v, tg, mask, mu, orc):
    """One tick of the reference's controller (ports and all) with the real solver; tier-(i) comparison with oracle/."""
    cls = {"id": ref.IDController, "mptc": ref.MPTCController, "pc": ref.PCController, "clf": ref.CLFController}[kind]
    ctrl = cls(plant, 5e-3)
    if mu is not None:
        ctrl.mu = float(mu)
    spy = SolveSpy(ctrl.solver)
    ctrl.solver = spy
    ctx = ctrl.CreateDefaultContext()
    qp, vp = to_plant(q, v, order)
    ctrl.get_input_port(0).FixValue(ctx, np.concatenate([qp, vp]))
    ctrl.get_input_port(1).FixValue(ctx, pd.AbstractValue.Make(trunk_dict(tg, mask)) if not hasattr(plant, "backend") else trunk_dict(tg, mask))
    tau = np.asarray(ctrl.get_output_port(0).Eval(ctx), float).reshape(-1)
    pv = np.array(list(range(6)) + [6 + o for o in order])
    vd = spy.x[:18][pv]
    p = orc.params(kind)
    if mu is not None:
        p.mu = float(mu)
    ct = [(int(mask) >> j) & 1 for j in range(4)]
    tau_o, met_o, st_o, qp_o = orc.control_law(kind, orc.model(model), p, q, v, tg, ct, want_qp=True)
    tau_can = np.zeros(12); tau_can[np.array(act)] = tau               # actuator order -> canonical joints
    return {"vd": rel(vd, qp_o["x"][:18]), "tau_info": rel(tau_can, tau_o), "oracle_status": int(st_o)}


def main():
    a = parse()
    pd = import_pydrake(a.fake)
    sys.path.insert(0, ROOT)
    from oracle import oracle_py as orc
    from quadruped_drake_amd import load_model
    have_ref = os.path.isdir(os.path.join(a.reference, "controllers"))
    ref = None
    if have_ref:
        if a.fake:
            sys.modules.setdefault("lcm", types.ModuleType("lcm"))   # import-only stub: use_lcm=False everywhere
            np.object = object                                       # helpers.py:19 uses the alias numpy removed in 1.24
        sys.path.insert(0, a.reference)
        import controllers as ref                                     # the reference's own code, imported where it lies
    worstA, worstB = {}, {}
    plants = {}
    print("| fixture | law | model | stage A worst (quantity) | stage B v-dot | torque (info) |")
    print("|---|---|---|---|---|---|")
    for name in a.cases:
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        kind, model = str(z["kind"]), str(z["model"])
        if model not in plants:
            plant = build_plant(pd, a.reference, model, a.dt, a.fake)
            plants[model] = (plant,) + plant_order(plant, load_model(model))
        plant, order, act = plants[model]
        n = min(a.n, z["q"].shape[1])
        wa, wb, wt = ("", 0.0), 0.0, 0.0
        for i in range(n):
            if z["mass_scale"].size:                # a scaled plant would have to be re-parsed from a scaled URDF: plant numbers
                continue                            # are checked on the unscaled fixtures
            r = stage_a(pd, plant, model, order, z["q"][:, i], z["v"][:, i], orc)
            k = max(r, key=r.get)
            if r[k] > wa[1]:
                wa = (k, r[k])
            for kk, vv in r.items():
                worstA[kk] = max(worstA.get(kk, 0.0), vv)
            runnable = ref is not None and (a.fake or model == "mini_cheetah")   # the reference hard-codes the frame name "body"
            if runnable and not (kind in ("mptc", "pc") and int(z["mask"][i]) == 0):
                mu = float(z["mu"][i]) if z["mu"].size else None
                rb = stage_b(pd, ref, plant, kind, model, order, act, z["q"][:, i], z["v"][:, i], z["targets"][:, i], z["mask"][i], mu, orc)
                wb, wt = max(wb, rb["vd"]), max(wt, rb["tau_info"])
        worstB[name] = wb
        print("| %s | %s | %s | %.1e (%s) | %.1e | %.1e |" % (name, kind, model, wa[1], wa[0], wb, wt))
    print("\nstage A, worst relative difference per quantity over all fixtures:")
    for k in sorted(worstA):
        print("  %-10s %.2e" % (k, worstA[k]))
    if a.gpu:
        import torch
        from quadruped_drake_amd import IDController, MPTCController, PCController, CLFController
        for name in a.cases:
            z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
            cls = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}[str(z["kind"])]
            n = z["q"].shape[1]
            c = cls(model=str(z["model"]), max_batch=n, device=0)
            up = lambda x: None if x is None or x.size == 0 else torch.tensor(x, device="cuda:0")
            tau, met, st = c.step(up(z["q"]), up(z["v"]), up(z["targets"]), up(z["mask"]), up(z["mu"]), up(z["mass_scale"])); c.sync()
            print("HIP path vs fixture %s: torques %.1e" % (name, rel(tau.cpu().numpy(), z["tau"])))
            c.close()
    okA = all(v <= a.tol_plant for k, v in worstA.items() if k != "Jdv_body") and worstA.get("Jdv_body", 0.0) <= 1e-9
    okB = all(v <= a.tol_vd for v in worstB.values())
    print("\nstage A (Drake's rigid-body numbers vs oracle/): %s   stage B (controllers + real solver, v-dot): %s%s" % (
        "AGREE" if okA else "DIFFER", "AGREE" if okB else "DIFFER",
        "   [--fake: plumbing only, the stand-in plant answers from oracle/]" if a.fake else ""))
    if ref is None:
        print("stage B skipped: no reference checkout at %s" % a.reference)
    sys.exit(0 if (okA and okB) else 1)


if __name__ == "__main__":
    main()
