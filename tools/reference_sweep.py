#!/usr/bin/env python3
"""THIS CONTAINER ONLY (needs /root/reference): many fresh random ticks through the reference's own controller code
(executed over the stand-ins of tests/fake_pydrake, exactly as tests/golden/make_reference_law_golden.py does) against
the oracle.  Prints one line per law / config: ticks, worst and median relative torque deviation, worst deviation of the
QP's accelerations and of the logged metrics, solver failures.   python tools/reference_sweep.py [ticks per case]"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "fake_pydrake")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
sys.modules["lcm"] = types.ModuleType("lcm")
np.object = object
sys.path.insert(0, "/root/reference")
import pydrake.all as fake                                   # noqa: E402
from pydrake.mathprog import OsqpSolver                      # noqa: E402
from controllers import IDController, MPTCController, PCController, CLFController   # noqa: E402  (reference code)
from oracle import oracle_py as orc                          # noqa: E402
from quadruped_drake_amd import workloads                    # noqa: E402
from quadruped_drake_amd.planners import unpack_trunk_input  # noqa: E402

LAWS = {"id": IDController, "mptc": MPTCController, "pc": PCController, "clf": CLFController}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dump = sys.argv[3] if len(sys.argv) > 3 else None            # .npz: inputs + what the reference's code returned (for tools/reference_sweep_gpu.py)
dumped = {}
backend = sys.argv[2] if len(sys.argv) > 2 else "oracle"     # "energy": the plant of tests/energy_model.py (nothing shared with oracle/)
for kind, cfg in (("id", 2), ("id", 3), ("mptc", 3), ("mptc", 4), ("mptc", 5), ("pc", 3), ("pc", 2), ("clf", 3), ("clf", 2)):
    if backend == "energy" and cfg == 5:
        continue              # the energy model has no mass-scale hook: config 5 runs on the oracle backend only
    b = workloads.make_batch(cfg, n=n, seed=90000 + 13 * cfg + len(kind))
    plant = fake.RefPlant(b["model"], body_frame="body", backend=backend)
    ctrl = LAWS[kind](plant, 5e-3)
    p = orc.params(kind)
    rel, dvd, dmet, fails, infeasible = [], [], [], 0, 0
    keep, taus, vds, mets = [], [], [], []
    for i in range(n):
        if b["mu"] is not None:
            ctrl.mu = float(b["mu"][i]); p.mu = float(b["mu"][i])
        m = orc.model(b["model"])
        if b["mass_scale"] is not None:
            m = orc.model_scaled(b["model"], float(b["mass_scale"][i])); plant.m = m; ctrl.plant_autodiff.m = m

        ctx = ctrl.CreateDefaultContext()
        ctrl.get_input_port(0).FixValue(ctx, np.concatenate([b["q"][:, i], b["v"][:, i]]))
        ctrl.get_input_port(1).FixValue(ctx, unpack_trunk_input(b["targets"][:, i], int(b["mask"][i])))
        ctrl.V = ctrl.err = ctrl.res = ctrl.Vdot = 0
        ct = [(int(b["mask"][i]) >> k) & 1 for k in range(4)]
        tau_o, met_o, st_o, qp = orc.control_law(kind, m, p, b["q"][:, i], b["v"][:, i], b["targets"][:, i], ct, want_qp=True)
        try:
            tau = ctrl.get_output_port(0).Eval(ctx)
        except AssertionError:                       # the reference's `assert result.is_success()`
            fails += 1
            infeasible += int(st_o != 0)
            continue
        met = ctrl.get_output_port(1).Eval(ctx)
        if st_o != 0:
            fails += 1
            continue
        keep.append(i); taus.append(np.array(tau)); vds.append(OsqpSolver.last["x"][:18].copy()); mets.append(np.array(met))
        rel.append(np.abs(tau - tau_o).max() / max(np.abs(tau_o).max(), 1e-3))
        dvd.append(np.abs(OsqpSolver.last["x"][:18] - qp["x"][:18]).max() / (1.0 + np.abs(qp["x"][:18]).max()))
        cols = [1] if kind == "id" else [0, 1, 3]
        dmet.append(max(abs(met[c] - met_o[c]) / (1.0 + abs(met_o[c])) for c in cols))
    rel, dvd, dmet = np.array(rel), np.array(dvd), np.array(dmet)
    if dump:
        k = "%s_cfg%d_" % (kind, cfg)
        keep = np.array(keep)
        dumped.update({k + "q": b["q"][:, keep], k + "v": b["v"][:, keep], k + "targets": b["targets"][:, keep], k + "mask": b["mask"][keep],
                       k + "mu": np.zeros(0) if b["mu"] is None else b["mu"][keep],
                       k + "mass_scale": np.zeros(0) if b["mass_scale"] is None else b["mass_scale"][keep],
                       k + "model": b["model"], k + "tau": np.array(taus).T, k + "vd": np.array(vds).T, k + "metrics": np.array(mets).T})
    print("%-4s cfg %d  %4d ticks: torque rel dev worst %.2e median %.2e | accelerations worst %.2e | metrics worst %.2e | "
          "not compared %d (of which infeasible for both: %d)" % (kind, cfg, rel.size, rel.max(), np.median(rel), dvd.max(), dmet.max(), fails, infeasible), flush=True)
if dump:
    np.savez_compressed(dump, **dumped)
    print("dumped", dump)
