#!/usr/bin/env python3
"""Golden target dictionaries, produced HERE by running the reference's OWN trunk planners
(/root/reference/planners/simple.py, planners/towr.py: imported, not copied).

The planners are Drake LeafSystems and TowrTrunkPlanner talks LCM; neither library is in this image.  Their
arithmetic, however, is plain numpy, so the generator imports them over two stand-ins that supply NO arithmetic:
  * tests/fake_pydrake  -- LeafSystem / AbstractValue / FramePoseVector / RigidTransform containers,
  * a stub `lcm` module  -- only so that `import lcm` succeeds; TowrTrunkPlanner.__init__ (which launches the TOWR
    binary and listens on LCM) is NOT run: the object is created bare and its own `lcm_handler` is fed the golden
    wire messages of trunk_state_msgs.bin (made by the reference's encoder, make_trunk_state_golden.py), exactly what
    `self.lc.handle()` would deliver.
Everything stored is data the reference's code computed:
  planner_golden.npz
    basic_<scenario>_*        output_dict of BasicTrunkPlanner after SimpleStanding / OrientationTest(t) / RaiseFoot(t) /
                              EdgeTest  (planners/simple.py:39-115)
    towr_times, towr_*        output_dict of TowrTrunkPlanner.SetTrunkOutputs at each query time (planners/towr.py:92-148)
    towr_u2_max               ComputeMaxControlInputs (planners/towr.py:71-90)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "fake_pydrake"))
sys.path.insert(0, os.path.join(HERE, "..", ".."))      # the fake's plant module imports oracle/ (unused by the planners)
sys.modules["lcm"] = types.ModuleType("lcm")          # import-only stub (see the docstring)
sys.path.insert(0, "/root/reference")
import pydrake.all as fake                              # noqa: E402
from planners.simple import BasicTrunkPlanner           # noqa: E402  (reference code)
from planners.towr import TowrTrunkPlanner              # noqa: E402  (reference code)

VEC_KEYS = ["p_body", "pd_body", "pdd_body", "rpy_body", "rpyd_body", "rpydd_body"] + \
           [pre + f for f in ("lf", "rf", "lh", "rh") for pre in ("p_", "pd_", "pdd_")]


def snapshot(d):
    out = {k: np.array(d[k], dtype=float) for k in VEC_KEYS}
    out["contact_states"] = np.array([bool(c) for c in d["contact_states"]])
    out["f_cj"] = np.array(d["f_cj"], dtype=float)
    out["u2_max"] = np.float64(d["u2_max"])
    return out


def stack(snaps):
    return {k: np.stack([s[k] for s in snaps]) for k in snaps[0]}


gold = {}

# ---- BasicTrunkPlanner scenarios (real __init__ over the container fakes)
frame_ids = {"trunk": 0, "lf": 1, "rf": 2, "lh": 3, "rh": 4}
bp = BasicTrunkPlanner(frame_ids)
bp.SimpleStanding()
for k, v in snapshot(bp.output_dict).items():
    gold["basic_standing_" + k] = v
bp.EdgeTest()
for k, v in snapshot(bp.output_dict).items():
    gold["basic_edge_" + k] = v
t_or = np.array([0.0, 0.1, 0.5, 1.0, np.pi / 2, 2.0, 3.7, 10.0])
snaps = []
for t in t_or:
    bp.OrientationTest(float(t)); snaps.append(snapshot(bp.output_dict))
gold["basic_orientation_times"] = t_or
for k, v in stack(snaps).items():
    gold["basic_orientation_" + k] = v
t_rf = np.array([0.0, 0.5, 1.0, 1.0 + 2.0 ** -40, 1.5, 4.0])      # the switch is `t > 1` (strict)
snaps = []
for t in t_rf:
    bp.RaiseFoot(float(t)); snaps.append(snapshot(bp.output_dict))
gold["basic_raisefoot_times"] = t_rf
for k, v in stack(snaps).items():
    gold["basic_raisefoot_" + k] = v
# the port function itself: SetTrunkOutputs through the declared abstract output port
ctx = fake.Context(); ctx.time = 0.3
d = bp.get_output_port(0).Eval(ctx)
assert all(np.array_equal(np.asarray(d[k], float), gold["basic_standing_" + k]) for k in VEC_KEYS)

# ---- TowrTrunkPlanner: stored trajectory from the golden wire messages, through the planner's own lcm_handler
raw = open(os.path.join(HERE, "trunk_state_msgs.bin"), "rb").read()
msgs = [raw[i:i + 549] for i in range(0, len(raw), 549)]
del msgs[60]     # the fixture's one out-of-order sample (0.059 after 59 * 0.001): TOWR publishes in time order, and the
                 # device table requires it (wbc_traj_create rejects decreasing timestamps); duplicates (0.0625 twice) stay
tp = TowrTrunkPlanner.__new__(TowrTrunkPlanner)      # bare: no TOWR subprocess, no LCM socket
fake.LeafSystem.__init__(tp)
tp.output_dict = {}
tp.traj_finished = False
tp.towr_timestamps, tp.towr_data = [], []
for b in msgs:
    tp.lcm_handler("trunk_state", b)
assert tp.traj_finished and len(tp.towr_data) == len(msgs)
tp.u2_max = tp.ComputeMaxControlInputs()
tp.wait_time = 1.0
ts = np.array(tp.towr_timestamps)
mid = 0.5 * (ts[:-1] + ts[1:])                        # near-ties between neighbours (rounding decides, as in numpy)
times = np.concatenate([[0.0, 0.5, 1.0 - 2.0 ** -50, 1.0], 1.0 + ts[:62], 1.0 + mid[:61],
                        1.0 + np.array([0.0005, 0.0615, 0.0625, 0.07, 1.0, 5e8, 1e9, 2e9]),
                        1.0 + np.random.default_rng(7).uniform(0.0, 0.07, 64)])


class _Out:
    def __init__(self):
        self.d = {}

    def get_mutable_value(self):
        return self.d


snaps = []
for t in times:
    ctx = fake.Context(); ctx.time = float(t)
    o = _Out()
    tp.SetTrunkOutputs(ctx, o)
    snaps.append(snapshot(o.d))
gold["towr_times"] = times
gold["towr_u2_max"] = np.float64(tp.u2_max)
for k, v in stack(snaps).items():
    gold["towr_" + k] = v

np.savez_compressed(os.path.join(os.environ.get("GOLDEN_OUT", HERE), "planner_golden.npz"), **gold)   # GOLDEN_OUT: tests/test_fixture_freshness.py
print("wrote planner_golden.npz:", len(gold), "arrays,", len(times), "TOWR query times, u2_max =", tp.u2_max)
