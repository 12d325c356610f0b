"""The LeafSystem adapter (quadruped_drake_amd.controller.make_leaf_system) executed end to end on the GPU against
a TEST-ONLY fake `pydrake.all` (tests/fake_pydrake): the four ports of controllers/basic_controller.py:33-50 /
inverse_dynamics_controller.py:14-16, the q_perm / act_perm derivation from the plant (basic_controller.py:310-313),
`quad_torques` and `output_metrics` against the oracle."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle_py as orc
from quadruped_drake_amd import workloads
from quadruped_drake_amd.controller import load_model

FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_pydrake")


@pytest.fixture
def fake_pydrake():
    assert "pydrake" not in sys.modules or getattr(sys.modules["pydrake"], "__file__", "").startswith(FAKE)
    sys.path.insert(0, FAKE)
    try:
        yield
    finally:
        sys.path.remove(FAKE)
        for k in [k for k in sys.modules if k == "pydrake" or k.startswith("pydrake.")]:
            del sys.modules[k]


def _trunk_dict(tg, mask):
    d = {}
    for i, k in enumerate(("p_body", "pd_body", "pdd_body", "rpy_body", "rpyd_body", "rpydd_body")):
        d[k] = tg[3 * i:3 * i + 3].copy()
    for i, f in enumerate(("lf", "rf", "lh", "rh")):
        d["p_" + f] = tg[18 + 9 * i:21 + 9 * i].copy()
        d["pd_" + f] = tg[21 + 9 * i:24 + 9 * i].copy()
        d["pdd_" + f] = tg[24 + 9 * i:27 + 9 * i].copy()
    d["contact_states"] = [bool((mask >> i) & 1) for i in range(4)]
    d["f_cj"] = np.zeros((3, 4)); d["u2_max"] = 0.0          # present in the planner's dict, never read by these laws
    return d


def test_fake_plant_orders_are_what_the_adapter_expects(fake_pydrake):
    from pydrake.all import FakePlant
    t = load_model("mini_cheetah")
    names = [l["joint"] for leg in t["legs"] for l in leg["links"]]
    order = [4 * (j % 3) + j // 3 for j in range(12)]
    plant = FakePlant(names, order, list(range(12)))
    assert plant.GetJointByName(names[1]).velocity_start() == 6 + 4       # LF hip comes after the four abduction joints
    B = plant.MakeActuationMatrix()
    assert B.shape == (18, 12) and (B.sum(0) == 1).all() and (B[:6] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("method,cfg,model", [("ID", 2, "mini_cheetah"), ("MPTC", 3, "mini_cheetah"), ("MPTC", 4, "anymal_b")])
def test_leaf_system_ports_against_the_oracle(fake_pydrake, method, cfg, model):
    from pydrake.all import FakePlant
    from quadruped_drake_amd.controller import make_leaf_system
    t = load_model(model)
    names = [l["joint"] for leg in t["legs"] for l in leg["links"]]
    order = [4 * (j % 3) + j // 3 for j in range(12)]            # breadth-first v numbering
    act_joint = list(np.random.default_rng(2).permutation(12))    # actuator k drives canonical joint act_joint[k]
    plant = FakePlant(names, order, act_joint)
    sys_ = make_leaf_system(plant, 5e-3, control_method=method, model=model)
    assert [p.name for p in sys_._in] == ["quad_state", "trunk_input"]
    assert [p.name for p in sys_._out] == ["quad_torques", "output_metrics"]
    b = workloads.make_batch(cfg, n=6, model=model)
    kind = method.lower()
    table = dict(t); table["act_perm"] = [int(x) for x in act_joint]
    m = orc.model(table); p = orc.params(kind)
    for i in range(6):
        q = b["q"][:, i]; v = b["v"][:, i]
        qd = q.copy(); vd = v.copy()
        for j in range(12):                                        # canonical joint j lives at Drake index order[j]
            qd[7 + order[j]] = q[7 + j]; vd[6 + order[j]] = v[6 + j]
        ctx = sys_.CreateDefaultContext()
        sys_.get_input_port(0).FixValue(ctx, np.concatenate([qd, vd]))
        sys_.get_input_port(1).FixValue(ctx, _trunk_dict(b["targets"][:, i], int(b["mask"][i])))
        u = sys_.get_output_port(0).Eval(ctx)
        met = sys_.get_output_port(1).Eval(ctx)
        ct = [(int(b["mask"][i]) >> k) & 1 for k in range(4)]
        u_o, met_o, st_o = orc.control_law(kind, m, p, q, v, b["targets"][:, i], ct)
        assert st_o == 0
        assert np.abs(u - u_o).max() < 1e-4 * max(np.abs(u_o).max(), 1e-3)
        assert np.allclose(met, met_o, rtol=1e-5, atol=1e-6)
    sys_.ctrl.close()


@pytest.mark.gpu
def test_leaf_system_raises_like_the_reference_assert(fake_pydrake):
    """inverse_dynamics_controller.py:224 `assert result.is_success()`: a non-zero status raises."""
    from pydrake.all import FakePlant
    from quadruped_drake_amd.controller import make_leaf_system, SolverError
    t = load_model("mini_cheetah")
    names = [l["joint"] for leg in t["legs"] for l in leg["links"]]
    plant = FakePlant(names, list(range(12)), list(range(12)))
    sys_ = make_leaf_system(plant, 5e-3, control_method="MPTC")
    q, v = workloads.nominal_state("mini_cheetah", 1)
    q[7 + 2, 0] = 0.0                                           # straight LF knee: Lambda = (J M^-1 J')^-1 is singular -> status 2
    ctx = sys_.CreateDefaultContext()
    sys_.get_input_port(0).FixValue(ctx, np.concatenate([q[:, 0], v[:, 0]]))
    sys_.get_input_port(1).FixValue(ctx, _trunk_dict(workloads.standing_targets("mini_cheetah", 1)[:, 0], 0b1111))
    with pytest.raises(SolverError):
        sys_.get_output_port(0).Eval(ctx)
    sys_.ctrl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["perm_cfg2_id", "perm_cfg3_mptc", "perm_cfg4_anymal_mptc"])
def test_leaf_system_reproduces_the_executed_reference_on_a_permuted_plant(fake_pydrake, name):
    """The same wiring the reference has -- Controller(plant, dt), quad_state / trunk_input in, quad_torques out -- on a
    plant that numbers its joints breadth-first and its actuators at random, against what the reference's own
    controller code returned on such a plant (tests/golden/make_reference_law_golden.py, `perm_*` sets)."""
    from pydrake.all import FakePlant
    from quadruped_drake_amd.controller import make_leaf_system
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_law_golden.npz"))
    g = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + "_")}
    model, kind = str(g["model"]), str(g["kind"])
    order, act = [int(x) for x in g["order"]], [int(x) for x in g["act_joint"]]
    t = load_model(model)
    names = [l["joint"] for leg in t["legs"] for l in leg["links"]]
    sys_ = make_leaf_system(FakePlant(names, order, act), 5e-3, control_method=kind.upper(), model=model)
    for i in range(g["q"].shape[1]):
        qd = g["q"][:, i].copy(); vd = g["v"][:, i].copy()
        for j in range(12):
            qd[7 + order[j]] = g["q"][7 + j, i]; vd[6 + order[j]] = g["v"][6 + j, i]
        ctx = sys_.CreateDefaultContext()
        sys_.get_input_port(0).FixValue(ctx, np.concatenate([qd, vd]))
        sys_.get_input_port(1).FixValue(ctx, _trunk_dict(g["targets"][:, i], int(g["mask"][i])))
        u = sys_.get_output_port(0).Eval(ctx)
        met = sys_.get_output_port(1).Eval(ctx)
        ref = g["tau"][:, i]
        assert np.abs(u - ref).max() < 1e-5 * max(np.abs(ref).max(), 1e-3), (name, i)
        cols = [1] if kind == "id" else [0, 1, 3]
        assert np.allclose(met[cols], g["metrics"][cols, i], rtol=1e-7, atol=1e-7)
    sys_.ctrl.close()
