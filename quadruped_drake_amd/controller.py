"""Host-side mirror of the reference's controller interface for the whole-body-QP hot path.

`IDController` / `MPTCController` keep the reference's names (controllers/inverse_dynamics_controller.py:3,
controllers/mptc_controller.py:3) and the meaning of its ports (controllers/basic_controller.py:33-50):

    quad_state   [q(19); v(18)]          -> q[19, N], v[18, N]
    trunk_input  dict of task targets    -> targets[54, N], contact_mask[N]   (planners/simple.py:45-85)
    quad_torques tau(12), actuator order -> tau[12, N]
    output_metrics [V, err, res, Vdot]   -> metrics[4, N]

but step a whole batch of N independent robots in one launch of the HIP kernel behind the C ABI
(include/wbc.h).  PyTorch is used only to own device memory and the stream.  There is no CPU path.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))

TRUNK_KEYS_BODY = ["p_body", "pd_body", "pdd_body", "rpy_body", "rpyd_body", "rpydd_body"]
FEET = ["lf", "rf", "lh", "rh"]


def load_model(name_or_path):
    path = name_or_path if os.path.exists(name_or_path) else os.path.join(_HERE, "models", name_or_path + ".json")
    with open(path) as f:
        return json.load(f)


def pack_trunk_input(trunk_data):
    """planners/simple.py:45-85 dict -> (targets[54], contact_mask).  `f_cj`/`u2_max` are ignored,
    exactly as ID and MPTC ignore them."""
    t = np.zeros(54)
    for i, k in enumerate(TRUNK_KEYS_BODY):
        t[3 * i:3 * i + 3] = np.asarray(trunk_data[k], dtype=float).reshape(3)
    for i, f in enumerate(FEET):
        for j, pre in enumerate(("p_", "pd_", "pdd_")):
            t[18 + 9 * i + 3 * j:21 + 9 * i + 3 * j] = np.asarray(trunk_data[pre + f], dtype=float).reshape(3)
    mask = sum((1 << i) for i, c in enumerate(trunk_data["contact_states"]) if c)
    return t, mask


class SolverError(AssertionError):
    """Mirrors `assert result.is_success()` (inverse_dynamics_controller.py:224).  args = (text, status, tau).  Raised for
    status 1 (iteration cap) and 2 (singular / infeasible).  Status 3 (MPTC / PC with a nearly straight knee: the law's own
    inv(J M^-1 J') is ill-conditioned) is a solved tick -- the reference's assert passes there and it applies the torques -- so
    the mirror returns them with an IllConditionedWarning and `last_status = 3`; `strict=True` turns that into this error too."""


class IllConditionedWarning(RuntimeWarning):
    """A tick of a task-space law with |sin(knee)| < 1e-4 on a leg (status 3): torques computed, not vouched for."""


STATUS_TEXT = {1: "iteration cap", 2: "not answerable -- singular, infeasible, or a malformed instance (a non-finite input, a bad mu / mass scale, an overflow): torques, accelerations and metrics are zero",
               3: "ill-conditioned: |sin(knee)| < 1e-4 on a leg under a task-space law; torques written but not trustworthy"}


class BatchedController:
    kind = None

    def __init__(self, model="mini_cheetah", max_batch=4096, device=0, params=None, host_ptrs=False,
                 q_perm=None, act_perm=None, use_torch_stream=True, strict=False):
        self.strict = bool(strict)      # ControlLaw: raise on status 3 as well (default: warn and return the torques)
        self.last_status = 0
        self.n_illcond = 0              # ControlLaw ticks flagged with status 3 so far (every one of them also warns)
        self.table = load_model(model) if isinstance(model, str) else model
        self.max_batch = int(max_batch)
        self.device = int(device)
        self.host_ptrs = bool(host_ptrs)
        L = _lib.lib()
        m = _lib.WbcModel()
        flat = np.asarray(self.table["flat"], dtype=np.float64)
        assert flat.size == 215
        m.flat[:] = flat.tolist()
        m.q_perm[:] = list(range(12)) if q_perm is None else [int(x) for x in q_perm]
        m.act_perm[:] = [int(x) for x in (self.table.get("act_perm", range(12)) if act_perm is None else act_perm)]
        p = _lib.WbcParams()
        _lib.check(L.wbc_params_default(self.kind, C.byref(p)))
        for k, val in (params or {}).items():
            if not hasattr(p, k):
                raise KeyError("unknown parameter %r" % k)
            setattr(p, k, float(val))
        self.params = p
        h = C.c_void_p()
        _lib.check(L.wbc_create(C.byref(m), self.kind, C.byref(p), self.max_batch, self.device,
                                _lib.HOST_PTRS if host_ptrs else _lib.DEVICE_PTRS, C.byref(h)))
        self._h = h
        self._L = L
        self._torch_stream = (not host_ptrs) and use_torch_stream
        self._bound_stream = None
        self._bind_stream()

    def _bind_stream(self):
        """Launch on torch's CURRENT stream of this device (re-read on every call: a controller used under
        `with torch.cuda.stream(s)` must not race with the stream it was created on)."""
        if not self._torch_stream:
            return None
        import torch
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._bound_stream:
            _lib.check(self._L.wbc_set_stream(self._h, C.c_void_p(s)))
            self._bound_stream = s
        return s

    def close(self):
        if getattr(self, "_h", None):
            self._L.wbc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- pointers ---------------------------------------------------------------------------
    def _ptr(self, a, rows, n, dtype, name, optional=False):
        if a is None:
            if optional:
                return None, None
            raise ValueError(name + " is required")
        if self.host_ptrs:
            arr = np.ascontiguousarray(a, dtype=dtype)
            if arr.shape != ((rows, n) if rows else (n,)):
                raise ValueError("%s: expected shape %s, got %s" % (name, (rows, n) if rows else (n,), arr.shape))
            return arr, C.c_void_p(arr.ctypes.data)
        import torch
        tdt = {np.float64: torch.float64, np.uint8: torch.uint8, np.int32: torch.int32}[dtype]
        if not (isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == tdt and a.is_contiguous()):
            raise ValueError("%s: expected a contiguous CUDA tensor of dtype %s" % (name, tdt))
        if a.device.index != self.device:
            raise ValueError("%s: tensor lives on cuda:%s, this controller's handle on cuda:%d" % (name, a.device.index, self.device))
        if tuple(a.shape) != ((rows, n) if rows else (n,)):
            raise ValueError("%s: expected shape %s, got %s" % (name, (rows, n) if rows else (n,), tuple(a.shape)))
        return a, C.c_void_p(a.data_ptr())

    def _out(self, rows, n, dtype):
        if self.host_ptrs:
            arr = np.zeros((rows, n) if rows else (n,), dtype=dtype)
            return arr, C.c_void_p(arr.ctypes.data)
        import torch
        tdt = {np.float64: torch.float64, np.int32: torch.int32}[dtype]
        t = torch.empty((rows, n) if rows else (n,), dtype=tdt, device="cuda:%d" % self.device)
        return t, C.c_void_p(t.data_ptr())

    def _args(self, q, v, targets, contact_mask, mu, mass_scale, out):
        n = int(q.shape[1])
        keep = []
        ptrs = []
        for a, rows, dt, name, opt in ((q, 19, np.float64, "q", False), (v, 18, np.float64, "v", False),
                                       (targets, 54, np.float64, "targets", False),
                                       (contact_mask, 0, np.uint8, "contact_mask", False),
                                       (mu, 0, np.float64, "mu", True), (mass_scale, 0, np.float64, "mass_scale", True)):
            k, p = self._ptr(a, rows, n, dt, name, opt)
            keep.append(k); ptrs.append(p)
        if out is None:
            tau, pt = self._out(12, n, np.float64)
            met, pm = self._out(4, n, np.float64)
            st, ps = self._out(0, n, np.int32)
        else:
            tau, met, st = out
            tau, pt = self._ptr(tau, 12, n, np.float64, "tau")
            met, pm = self._ptr(met, 4, n, np.float64, "metrics")
            st, ps = self._ptr(st, 0, n, np.int32, "status")
        return n, keep, ptrs + [pt, pm, ps], (tau, met, st)

    # -- the hot path -----------------------------------------------------------------------
    def step(self, q, v, targets, contact_mask, mu=None, mass_scale=None, out=None):
        """One control tick for the batch.  Asynchronous in device mode (call sync() or use torch)."""
        n, keep, ptrs, outs = self._args(q, v, targets, contact_mask, mu, mass_scale, out)
        self._bind_stream()
        _lib.check(self._L.wbc_step(self._h, n, n, *ptrs))
        if self.host_ptrs:
            self.sync()
        self._keep = keep
        return outs

    def time_steps(self, steps, q, v, targets, contact_mask, mu=None, mass_scale=None, out=None):
        """`steps` back-to-back launches timed with HIP events on the launch stream -> ms per launch."""
        n, keep, ptrs, outs = self._args(q, v, targets, contact_mask, mu, mass_scale, out)
        ms = C.c_float(0)
        self._bind_stream()
        _lib.check(self._L.wbc_time_steps(self._h, int(steps), n, n, *ptrs, C.byref(ms)))
        return ms.value, outs

    def time_steps_each(self, steps, q, v, targets, contact_mask, mu=None, mass_scale=None, out=None):
        """The same launches with one HIP event between every two: per-launch device milliseconds (numpy array)."""
        n, keep, ptrs, outs = self._args(q, v, targets, contact_mask, mu, mass_scale, out)
        ms = (C.c_float * int(steps))()
        self._bind_stream()
        _lib.check(self._L.wbc_time_steps_each(self._h, int(steps), n, n, *ptrs, ms))
        return np.array(ms[:], dtype=np.float64), outs

    def bind(self, q, v, targets, contact_mask, mu=None, mass_scale=None, out=None):
        """Validate the tensors ONCE and return a callable bundle for repeated ticks on the same buffers (a control loop
        steps the same device arrays every tick; the per-call checks of step() cost ~10 us of Python each)."""
        if self.host_ptrs:
            raise ValueError("bind(): device-pointer handles only (host-pointer steps stage and synchronise every call)")
        n, keep, ptrs, outs = self._args(q, v, targets, contact_mask, mu, mass_scale, out)
        ctrl = self

        class Bound:
            outputs = outs
            _keep = keep

            def step(self):
                ctrl._bind_stream()
                _lib.check(ctrl._L.wbc_step(ctrl._h, n, n, *ptrs))
                return outs

            def time_steps(self, steps, wait=True):
                """wait=False only queues the launches and their two events; time_steps_result() reads them later."""
                ms = C.c_float(0)
                ctrl._bind_stream()
                _lib.check(ctrl._L.wbc_time_steps(ctrl._h, int(steps), n, n, *ptrs, C.byref(ms) if wait else None))
                return ms.value if wait else None

            def time_steps_result(self):
                ms = C.c_float(0)
                _lib.check(ctrl._L.wbc_time_steps_result(ctrl._h, C.byref(ms)))
                return ms.value

        return Bound()

    def sync(self):
        _lib.check(self._L.wbc_sync(self._h))

    def stats(self, reset=False):
        s = _lib.WbcStats()
        _lib.check(self._L.wbc_stats_get(self._h, C.byref(s)))
        d = dict(ticks=s.ticks, status_nonzero=s.status_nonzero, iters_sum=s.iters_sum, tau_abs_sum=s.tau_abs_sum,
                 tau_abs_max=s.tau_abs_max, err_sum=s.err_sum, mask_count=list(s.mask_count))
        if reset:
            _lib.check(self._L.wbc_stats_reset(self._h))
        return d

    def stats_reset(self):
        """Zero the device-side statistics, asynchronously on the handle's stream (wbc_stats_reset): no wait, unlike stats(reset=True)."""
        self._bind_stream()
        _lib.check(self._L.wbc_stats_reset(self._h))

    # -- closed-loop rollouts ---------------------------------------------------
    def set_vdot_output(self, vdot, n=None):
        """CUDA float64 [18, N] tensor that every step() fills with the QP's generalized accelerations (None: off)."""
        if vdot is None:
            self._vdot = None
            _lib.check(self._L.wbc_set_vdot_output(self._h, None))
            return
        n = int(vdot.shape[1]) if n is None else int(n)
        k, p = self._ptr(vdot, 18, n, np.float64, "vdot")
        self._vdot = k
        _lib.check(self._L.wbc_set_vdot_output(self._h, p))

    def integrate(self, q, v, vdot, dt):
        """Semi-implicit Euler step, in place on q [19, N] and v [18, N]."""
        n = int(q.shape[1])
        _, pq = self._ptr(q, 19, n, np.float64, "q")
        _, pv = self._ptr(v, 18, n, np.float64, "v")
        _, pd = self._ptr(vdot, 18, n, np.float64, "vdot")
        self._bind_stream()
        _lib.check(self._L.wbc_integrate(self._h, n, n, float(dt), pq, pv, pd))

    def rollout(self, traj, steps, dt, q, v, time, mu=None, mass_scale=None):
        """`steps` closed-loop ticks on the device: lookup(traj, time) -> step -> integrate.  Updates q, v, time in
        place; returns the last (tau, metrics, status, targets, mask)."""
        import torch
        n = int(q.shape[1])
        _, pq = self._ptr(q, 19, n, np.float64, "q")
        _, pv = self._ptr(v, 18, n, np.float64, "v")
        _, pt = self._ptr(time, 0, n, np.float64, "time")
        _, pmu = self._ptr(mu, 0, n, np.float64, "mu", optional=True)
        _, pms = self._ptr(mass_scale, 0, n, np.float64, "mass_scale", optional=True)
        dev = q.device
        tg = torch.empty((54, n), dtype=torch.float64, device=dev); mk = torch.empty((n,), dtype=torch.uint8, device=dev)
        tau = torch.empty((12, n), dtype=torch.float64, device=dev); met = torch.empty((4, n), dtype=torch.float64, device=dev)
        st = torch.empty((n,), dtype=torch.int32, device=dev); vd = torch.empty((18, n), dtype=torch.float64, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        self._bind_stream()
        _lib.check(self._L.wbc_rollout(self._h, traj._h, int(steps), float(dt), n, n, pq, pv, pt, p(tg), p(mk),
                                       pmu, pms, p(tau), p(met), p(st), p(vd)))
        self._keep = (tg, mk, tau, met, st, vd, mu, mass_scale)
        return tau, met, st, tg, mk

    def set_warm_start(self, on=True):
        """rollout(): start every tick's active set from the rows that were active when the robot's previous tick ended (wbc_set_warm_start).  The
        reference solves every tick from scratch (inverse_dynamics_controller.py:200); the result is the same QP solution to rounding, reached in fewer
        active-set trips on closed loops that sit on their friction limits.  Off by default: a cold rollout equals the launch-per-stage loop bit for bit."""
        _lib.check(self._L.wbc_set_warm_start(self._h, 1 if on else 0))

    def set_variant(self, variant):
        """"auto" (0) or "hex" (3): the 16-lanes-per-robot kernel is the one product kernel family."""
        v = {"auto": 0, "hex": 3}.get(variant, variant)
        _lib.check(self._L.wbc_set_variant(self._h, int(v)))

    def variant_for(self, n):
        """Name of the kernel variant a step of n instances runs (always "hex")."""
        v = self._L.wbc_variant_for(self._h, int(n))
        if v < 0:
            _lib.check(v)
        return {3: "hex"}[v]

    def kernel_info(self, rollout=False):
        """Registers, scratch bytes per lane and LDS bytes of the tick kernel (rollout=True: of the persistent closed-loop kernel)."""
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        fn = self._L.wbc_rollout_kernel_info if rollout else self._L.wbc_kernel_info
        _lib.check(fn(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(num_regs=a.value, scratch_bytes_per_lane=b.value, lds_bytes=c.value, block_threads=d.value)

    # -- single-robot convenience with the reference's signature --------------------------
    def ControlLaw(self, q, v, trunk_data):
        """controllers/*_controller.py ControlLaw(context, q, v) for ONE robot given the planner dict.
        Raises SolverError on status 1 / 2, like the reference's assert; status 3 (ill-conditioned, torques written) warns
        and returns the torques, as the reference -- whose solver reports success there -- would apply them."""
        t, mask = pack_trunk_input(trunk_data)
        qq = np.asarray(q, float).reshape(19, 1); vv = np.asarray(v, float).reshape(18, 1)
        if self.host_ptrs:
            tau, met, st = self.step(qq, vv, t.reshape(54, 1), np.array([mask], np.uint8))
            self.sync()                       # the ABI hands the outputs over at wbc_sync
            tau, met, st = tau[:, 0], met[:, 0], int(st[0])
        else:
            import torch
            dev = "cuda:%d" % self.device
            outs = self.step(torch.tensor(qq, device=dev), torch.tensor(vv, device=dev),
                             torch.tensor(t.reshape(54, 1), device=dev),
                             torch.tensor([mask], dtype=torch.uint8, device=dev))
            self.sync()
            tau, met, st = outs[0][:, 0].cpu().numpy(), outs[1][:, 0].cpu().numpy(), int(outs[2][0])
        self.last_status = st
        if st == 3 and not self.strict:
            import warnings
            # n_illcond counts EVERY flagged tick; the warning is throttled to the 1st, 2nd, 4th, 8th ... of them (a 1 kHz loop in an
            # ill-conditioned pose would otherwise print a thousand lines a second).  The text carries the count, so Python's default
            # "once per location and text" filter does not swallow the later ones; the location is the caller's (stacklevel 2).
            self.n_illcond += 1
            if self.n_illcond & (self.n_illcond - 1) == 0:
                warnings.warn("whole-body QP tick flagged with status 3 (%s) [flagged tick #%d of this controller -- every flagged tick is "
                              "counted in n_illcond and visible in last_status; this message appears at #1, #2, #4, #8, ...]"
                              % (STATUS_TEXT[3], self.n_illcond), IllConditionedWarning, stacklevel=2)
        elif st != 0:
            raise SolverError("whole-body QP failed with status %d (%s)" % (st, STATUS_TEXT.get(st, "unknown")), st, tau)
        self.V, self.err, self.res, self.Vdot = (float(x) for x in met)
        return tau


class IDController(BatchedController):
    """controllers/inverse_dynamics_controller.py:3-234, batched."""
    kind = _lib.KIND_ID


class MPTCController(BatchedController):
    """controllers/mptc_controller.py:3-310, batched."""
    kind = _lib.KIND_MPTC


class PCController(BatchedController):
    """controllers/pc_controller.py:3-255 (MPTC + passivity constraint Vdot <= 0), batched."""
    kind = _lib.KIND_PC


class CLFController(BatchedController):
    """controllers/clf_controller.py:3-234 (CLF-QP inverse dynamics), batched; 13 reduced variables [z; delta]
    (the slack lives on a spare sub-lane of the 16-lane kernel)."""
    kind = _lib.KIND_CLF


def make_leaf_system(plant, dt, control_method="ID", model="mini_cheetah", use_lcm=False, **kw):
    """Drop-in for `IDController(plant, dt, use_lcm)` / `MPTCController(...)` in simulate.py:106-118:
    a pydrake LeafSystem with the reference's four ports (basic_controller.py:33-50,
    inverse_dynamics_controller.py:14-16) that evaluates the HIP path with N = 1.

    Import-guarded: pydrake is not part of this image.  The joint order of q/v and the actuator
    order are read from the plant (they depend on the Drake release: basic_controller.py:310-313)."""
    from pydrake.all import AbstractValue, BasicVector, LeafSystem  # noqa: guarded import

    if use_lcm:
        raise NotImplementedError("the LCM bridge of basic_controller.py:55-61,307-314 is out of scope")
    table = load_model(model)
    joint_names = [l["joint"] for leg in table["legs"] for l in leg["links"]]
    q_perm = [plant.GetJointByName(nm).velocity_start() - 6 for nm in joint_names]
    B = plant.MakeActuationMatrix()  # nv x nu
    act_perm = [q_perm.index(int(np.argmax(B[6:, k]))) for k in range(12)]
    cls = {"ID": IDController, "MPTC": MPTCController, "PC": PCController, "CLF": CLFController}[control_method]
    ctrl = cls(model=table, max_batch=1, host_ptrs=True, q_perm=q_perm, act_perm=act_perm, **kw)

    class _Leaf(LeafSystem):
        def __init__(self):
            LeafSystem.__init__(self)
            self.dt = dt
            self.ctrl = ctrl
            self._metrics = np.zeros(4)
            self.DeclareVectorInputPort("quad_state", BasicVector(plant.num_positions() + plant.num_velocities()))
            self.DeclareAbstractInputPort("trunk_input", AbstractValue.Make({}))
            self.DeclareVectorOutputPort("quad_torques", BasicVector(plant.num_actuators()), self.DoSetControlTorques)
            self.DeclareVectorOutputPort("output_metrics", BasicVector(4), self.SetLoggingOutputs)

        def DoSetControlTorques(self, context, output):
            state = self.EvalVectorInput(context, 0).get_value()
            q = state[:plant.num_positions()]
            v = state[-plant.num_velocities():]
            trunk_data = self.EvalAbstractInput(context, 1).get_value()
            u = ctrl.ControlLaw(q, v, trunk_data)
            self._metrics = np.array([ctrl.V, ctrl.err, ctrl.res, ctrl.Vdot])
            output.SetFromVector(u)

        def SetLoggingOutputs(self, context, output):
            output.SetFromVector(self._metrics)

    return _Leaf()
