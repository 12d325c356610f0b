"""Host-side mirrors of the reference's trunk planners -- the callers that feed `trunk_input` to the hot path.

  BasicTrunkPlanner   planners/simple.py:4-139   standing targets and the three manual scenarios the reference toggles
                                                 by editing SetTrunkOutputs (OrientationTest, RaiseFoot, EdgeTest)
  TowrTrunkPlanner    planners/towr.py:10-148    stored TOWR trajectory: lcm_handler -> samples, ComputeMaxControlInputs,
                                                 per-tick nearest-timestamp lookup

Same method names and the same `output_dict` schema (planners/simple.py:45-85), no Drake / LCM: the scenario generators
are batched (`scenario_targets(name, t[N]) -> targets[54, N], mask[N]`, what a batch of N robots consumes); the wire
messages are parsed by the C decoder of libwbc_hip.so; the per-tick lookup is the device kernel behind
`TrunkTrajectory.lookup`.  Results are pinned bit-for-bit by dictionaries the reference's own planner code produced
(tests/golden/make_planner_golden.py -> planner_golden.npz).
"""
import numpy as np

from . import workloads
from .controller import FEET, TRUNK_KEYS_BODY, pack_trunk_input

SCENARIOS = ("standing", "orientation", "raise_foot", "edge")
_P, _RPY, _RPYD, _RPYDD = 0, 9, 12, 15            # rows of the 54-vector (include/wbc.h): body p, ..., rpy, rpyd, rpydd
_FOOT = lambda i: 18 + 9 * i                      # foot i: p, pd, pdd


def scenario_targets(name, t, model="mini_cheetah"):
    """Targets of one of the reference's scenarios at times t[N] -> (targets [54, N], contact_mask [N] uint8).

    standing     planners/simple.py:39-85    four feet down, body at 0.3 m
    orientation  planners/simple.py:87-95    pitch / yaw targets 0.4 sin t, 0.4 cos t with their rates and accelerations
    raise_foot   planners/simple.py:97-108   body shifted by (-0.1, 0.05, 0); after t > 1 the RF foot swings, 0.1 m up
    edge         planners/simple.py:110-115  body shifted by (-0.1, 0.63, 0): friction rows become active
    """
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    tg = workloads.standing_targets(model, t.size)
    mask = np.full(t.size, 0b1111, dtype=np.uint8)
    if name == "standing":
        pass
    elif name == "orientation":
        s, c = 0.4 * np.sin(t), 0.4 * np.cos(t)
        tg[_RPY + 1], tg[_RPY + 2] = s, c
        tg[_RPYD + 1], tg[_RPYD + 2] = c, -s
        tg[_RPYDD + 1], tg[_RPYDD + 2] = -s, -c
    elif name == "raise_foot":
        tg[_P:_P + 3] += np.array([-0.1, 0.05, 0.0])[:, None]
        up = t > 1
        tg[_FOOT(1) + 2] += np.where(up, 0.1, 0.0)
        mask[up] = 0b1101
    elif name == "edge":
        tg[_P:_P + 3] += np.array([-0.1, 0.63, 0.0])[:, None]
    else:
        raise ValueError("unknown scenario %r (one of %s)" % (name, ", ".join(SCENARIOS)))
    return tg, mask


def scenario_trajectory(name, duration, dt, model="mini_cheetah", device=0):
    """A scenario tabulated at the tick times k*dt, k = 0 .. duration/dt, as a device-resident stored trajectory
    (wait_time 0): what `wbc_rollout` consumes, so that the reference's manual experiments run closed loop on the GPU."""
    from .trajectory import TrunkTrajectory
    ts = np.arange(int(round(duration / dt)) + 1) * dt
    tg, mk = scenario_targets(name, ts, model)
    return TrunkTrajectory(ts, np.ascontiguousarray(tg.T), mk, model=model, wait_time=0.0, device=device)


def unpack_trunk_input(targets, contact_mask, f_cj=None, u2_max=0.0):
    """(targets[54], mask) -> the planner's dictionary (inverse of controller.pack_trunk_input)."""
    t = np.asarray(targets, dtype=np.float64).reshape(54)
    d = {k: t[3 * i:3 * i + 3].copy() for i, k in enumerate(TRUNK_KEYS_BODY)}
    for i, f in enumerate(FEET):
        for j, pre in enumerate(("p_", "pd_", "pdd_")):
            d[pre + f] = t[_FOOT(i) + 3 * j:_FOOT(i) + 3 * j + 3].copy()
    d["contact_states"] = [bool((int(contact_mask) >> i) & 1) for i in range(4)]
    d["f_cj"] = np.zeros((3, 4)) if f_cj is None else np.asarray(f_cj, dtype=np.float64)
    d["u2_max"] = u2_max
    return d


class BasicTrunkPlanner:
    """planners/simple.py:4-139 without the LeafSystem plumbing: `output_dict` after each scenario call is the
    dictionary the reference's port would carry; `targets()` is its packed form for the C ABI."""

    def __init__(self, model="mini_cheetah"):
        self.model = model
        self.output_dict = {}
        self.SimpleStanding()

    def _set(self, name, t=0.0):
        tg, mk = scenario_targets(name, [t], self.model)
        self.output_dict.update(unpack_trunk_input(tg[:, 0], mk[0]))

    def SimpleStanding(self):
        self._set("standing")

    def OrientationTest(self, t):
        self._set("orientation", t)

    def RaiseFoot(self, t):
        self._set("raise_foot", t)

    def EdgeTest(self):
        self._set("edge")

    def SetTrunkOutputs(self, t=0.0):
        """The reference's port function (planners/simple.py:117-124) as shipped: standing, whatever the time."""
        self.SimpleStanding()
        return self.output_dict

    def targets(self):
        return pack_trunk_input(self.output_dict)


class TowrTrunkPlanner(BasicTrunkPlanner):
    """planners/towr.py:10-148.  The TOWR optimisation itself (trunk_mpc) is out of scope: the
    planner is fed the `trunk_state` messages a recorded run published -- `lcm_handler(channel, data)` is the
    reference's subscriber callback -- and serves the same per-tick lookup from a device-resident table."""

    def __init__(self, messages=(), model="mini_cheetah", wait_time=1.0):
        BasicTrunkPlanner.__init__(self, model)
        self.traj_finished = False
        self.towr_timestamps = []
        self.towr_data = []
        self.wait_time = wait_time
        self.u2_max = 0.0
        self._traj = None
        for m in messages:
            self.lcm_handler("trunk_state", m)
        if self.towr_data:
            self.u2_max = self.ComputeMaxControlInputs()

    def lcm_handler(self, channel, data):
        from .trajectory import decode_trunk_state
        msg = decode_trunk_state(data)                # raises ValueError("Decode error") like trunk_state_t.decode
        self.towr_timestamps.append(msg["timestamp"])
        self.towr_data.append(msg)
        self.traj_finished = msg["finished"]
        self._traj = None
        if self.traj_finished:          # the reference computes it once the stream has finished (planners/towr.py:31-32, 67-68)
            self.u2_max = self.ComputeMaxControlInputs()

    def ComputeMaxControlInputs(self):
        """max over the samples of the 2-norm of [foot accelerations (LF RF LH RH); rpydd; pdd]  (planners/towr.py:71-90)."""
        best = 0
        for d in self.towr_data:
            u2 = np.linalg.norm(np.concatenate([d["foot_pdd"].reshape(-1), d["base_rpydd"], d["base_pdd"]]))
            best = max(best, u2)
        return best

    def trajectory(self, device=0):
        """The stored samples as a device-resident table (TrunkTrajectory), built on first use."""
        from .trajectory import TrunkTrajectory
        if self._traj is None or self._traj.device != device:
            self._traj = TrunkTrajectory(self.towr_timestamps, np.stack([d["targets"] for d in self.towr_data]),
                                         [d["contact_mask"] for d in self.towr_data], model=self.model,
                                         wait_time=self.wait_time, device=device)
        return self._traj

    def SetTrunkOutputs(self, t, device=0):
        """One robot's dictionary at time t through the device lookup (planners/towr.py:92-148)."""
        import torch
        tg, mk = self.trajectory(device).lookup(torch.tensor([float(t)], dtype=torch.float64, device="cuda:%d" % device))
        tg, mk = tg.cpu().numpy()[:, 0], int(mk.cpu().numpy()[0])
        if t < self.wait_time:
            self.output_dict.update(unpack_trunk_input(tg, mk))
        else:
            k = self.sample_index(t)
            self.output_dict.update(unpack_trunk_input(tg, mk, f_cj=self.towr_data[k]["foot_f"].T, u2_max=self.u2_max))
        return self.output_dict

    def sample_index(self, t):
        """Index of the stored sample the lookup serves at time t >= wait_time: bisection + nearer neighbour, first index
        on ties and duplicates (the host twin of the device search; `f_cj` is not part of the 54 targets)."""
        ts = self.towr_timestamps
        x = t - self.wait_time
        lo, hi = 0, len(ts)
        while lo < hi:                                   # first index with ts >= x
            mid = (lo + hi) // 2
            if ts[mid] < x:
                lo = mid + 1
            else:
                hi = mid
        if lo == len(ts):
            lo -= 1
        if lo > 0 and abs(ts[lo - 1] - x) <= abs(ts[lo] - x):
            lo -= 1
        while lo > 0 and abs(ts[lo - 1] - x) == abs(ts[lo] - x):
            lo -= 1
        return lo
