"""include/wbc.h threading contract (SURVEY 8b "one host thread per GPU"): a handle is not thread-safe, DISTINCT handles are
independent, wbc_last_error() is thread-local.  Two handles are stepped concurrently from two host threads (ctypes releases the
GIL around every call into the library) -- one on a caller-owned stream (wbc_set_stream: torch's), one on its own -- and
must produce, bit for bit, what each produces when run alone."""
import ctypes as C
import threading

import numpy as np
import pytest

from quadruped_drake_amd import workloads


def _start(n, seed):
    q0, v0 = workloads.nominal_state("mini_cheetah", n)
    rng = np.random.default_rng(seed)
    q0[7:] += rng.uniform(-0.05, 0.05, (12, n)); q0[6] += rng.uniform(-0.01, 0.01, n)
    v0[0:6] = rng.normal(0, 0.1, (6, n))
    tg = workloads.standing_targets("mini_cheetah", n)
    tg[0:2] += rng.normal(0, 0.01, (2, n))
    return q0, v0, tg, np.full(n, 0b1111, np.uint8)


def _closed_loop(ctrl, q0, v0, tg, mk, steps, dt, log=None):
    """steps x (wbc_step -> wbc_integrate -> wbc_sync) on device tensors; returns the final state and the last torques."""
    import torch
    dev = "cuda:0"
    q = torch.tensor(q0, device=dev); v = torch.tensor(v0, device=dev); t = torch.tensor(tg, device=dev); m = torch.tensor(mk, device=dev)
    n = q0.shape[1]
    vd = torch.zeros((18, n), dtype=torch.float64, device=dev)
    out = (torch.zeros((12, n), dtype=torch.float64, device=dev), torch.zeros((4, n), dtype=torch.float64, device=dev),
           torch.zeros((n,), dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    ctrl.set_vdot_output(vd)
    for s in range(steps):
        ctrl.step(q, v, t, m, out=out)
        ctrl.integrate(q, v, vd, dt)
        ctrl.sync()
        if log is not None:
            log.append(s)
    ctrl.set_vdot_output(None)
    return q.cpu().numpy(), v.cpu().numpy(), out[0].cpu().numpy(), out[2].cpu().numpy()


@pytest.mark.gpu
def test_two_handles_stepped_from_two_threads_equal_their_serial_runs():
    import torch
    from quadruped_drake_amd import MPTCController
    steps, dt = 200, 1e-3
    A = _start(256, 21); B = _start(192, 22)
    mk = lambda own: MPTCController(max_batch=256, device=0, use_torch_stream=not own)
    # serial: each handle alone
    ca, cb = mk(False), mk(True)
    ref_a = _closed_loop(ca, *A, steps, dt)
    ref_b = _closed_loop(cb, *B, steps, dt)
    sa, sb = ca.stats(), cb.stats()
    ca.close(); cb.close()
    assert (ref_a[3] == 0).all() and (ref_b[3] == 0).all()
    assert not np.array_equal(ref_a[0], A[0])                   # the loop really moved the state
    # concurrent: two fresh handles, two threads, started together
    ca, cb = mk(False), mk(True)
    res, logs, errs = {}, {"a": [], "b": []}, []
    gate = threading.Barrier(2)

    def run(name, ctrl, data):
        try:
            torch.cuda.set_device(0)
            gate.wait()
            res[name] = _closed_loop(ctrl, *data, steps, dt, log=logs[name])
        except Exception as e:      # noqa: BLE001 -- reported by the assert below
            errs.append((name, repr(e)))

    ta = threading.Thread(target=run, args=("a", ca, A)); tb = threading.Thread(target=run, args=("b", cb, B))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    for got, ref in ((res["a"], ref_a), (res["b"], ref_b)):
        for x, y in zip(got, ref):
            assert np.array_equal(x, y)
    ta_, tb_ = ca.stats(), cb.stats()
    assert ta_ == sa and tb_ == sb                               # per-handle statistics: nothing of the other handle's ticks
    assert sa["ticks"] == steps * 256 and sb["ticks"] == steps * 192
    assert len(logs["a"]) == steps and len(logs["b"]) == steps
    ca.close(); cb.close()


@pytest.mark.gpu
def test_last_error_is_thread_local():
    """Each thread provokes its own API misuse at the same moment; each reads back ITS message, the main thread's stays."""
    from quadruped_drake_amd import _lib, MPTCController
    L = _lib.lib()
    ctrl = MPTCController(max_batch=8, device=0)
    assert L.wbc_params_default(99, None) < 0
    main_msg = L.wbc_last_error()
    assert b"wbc_params_default" in main_msg
    gate = threading.Barrier(2)
    seen = {}

    def misuse_step():
        gate.wait()
        rc = L.wbc_step(ctrl._h, 9999, 9999, None, None, None, None, None, None, None, None, None)   # n > max_batch
        gate.wait()
        seen["step"] = (rc, L.wbc_last_error())

    def misuse_integrate():
        gate.wait()
        rc = L.wbc_integrate(None, 1, 1, 1e-3, None, None, None)
        gate.wait()
        seen["integrate"] = (rc, L.wbc_last_error())

    ts = [threading.Thread(target=misuse_step), threading.Thread(target=misuse_integrate)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert seen["step"][0] < 0 and b"wbc_step" in seen["step"][1] and b"out of range" in seen["step"][1]
    assert seen["integrate"][0] < 0 and b"wbc_integrate" in seen["integrate"][1]
    assert L.wbc_last_error() == main_msg
    ctrl.close()


@pytest.mark.gpu
def test_calls_leave_the_callers_current_device_alone():
    """include/wbc.h "Current device": every wbc_* call runs on its handle's device and restores the calling thread's current HIP device (round 5's
    calls left it on the handle's: a second handle on another GPU, or torch on the same thread, was silently moved).  Two handles are stepped on ONE
    thread and hipGetDevice is read around every call.  With two or more GPUs visible the handles sit on different devices and the thread's current
    device is a third choice; on a one-GPU box the assertion is the same (device 0 before and after) and the calls are checked to have gone through
    hipGetDevice / hipSetDevice at all by the library exporting nothing that bypasses the guard (every GPU entry point is exercised here)."""
    import torch
    from quadruped_drake_amd import _lib, MPTCController, IDController
    from quadruped_drake_amd.trajectory import TrunkTrajectory
    hip = C.CDLL("libamdhip64.so")
    ndev = torch.cuda.device_count()
    dev_a, dev_b = 0, (1 if ndev >= 2 else 0)
    home = (ndev - 1) if ndev >= 2 else 0                       # the device the thread calls "current"

    def current():
        d = C.c_int(-1)
        assert hip.hipGetDevice(C.byref(d)) == 0
        return d.value

    assert hip.hipSetDevice(home) == 0
    n = 32
    q0, v0, tg, mk = _start(n, 3)
    ctrls = [(MPTCController(max_batch=n, device=dev_a), dev_a), (IDController(max_batch=n, device=dev_b), dev_b)]
    assert current() == home
    for ctrl, d in ctrls:
        with torch.cuda.device(d):
            dev = "cuda:%d" % d
            q = torch.tensor(q0, device=dev); v = torch.tensor(v0, device=dev); t = torch.tensor(tg, device=dev); m = torch.tensor(mk, device=dev)
            tm = torch.zeros(n, dtype=torch.float64, device=dev)
            vd = torch.zeros((18, n), dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
        assert hip.hipSetDevice(home) == 0
        calls = [lambda: ctrl.step(q, v, t, m), lambda: ctrl.sync(), lambda: ctrl.set_vdot_output(vd), lambda: ctrl.step(q, v, t, m),
                 lambda: ctrl.integrate(q, v, vd, 1e-3), lambda: ctrl.stats(), lambda: ctrl.stats_reset(), lambda: ctrl.kernel_info(),
                 lambda: ctrl.kernel_info(rollout=True), lambda: ctrl.time_steps(3, q, v, t, m), lambda: ctrl.sync()]
        for k, call in enumerate(calls):
            call()
            assert current() == home, (d, k)
        ts, tgt, masks = np.arange(8) * 1e-3, np.tile(tg[:, 0], (8, 1)), np.full(8, 0b1111, np.uint8)
        traj = TrunkTrajectory(ts, tgt, masks, wait_time=0.0, device=d, standing_targets=tg[:, 0], standing_mask=0b1111)
        assert current() == home
        ctrl.rollout(traj, 5, 1e-3, q, v, tm); ctrl.sync()
        assert current() == home
        traj.close()
        assert current() == home
    for ctrl, d in ctrls:
        ctrl.close()
        assert current() == home
    assert hip.hipSetDevice(0) == 0
