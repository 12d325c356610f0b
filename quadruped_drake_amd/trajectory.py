"""Callers of the hot path: the stored trunk trajectory of planners/towr.py.

`decode_trunk_state` parses one 549-byte `trunk_state_t` LCM message (lcm_types/trunk_state_t.lcm,
lcm_types/trunklcm/trunk_state_t.py:83-121) with the C decoder of libwbc_hip.so -- no `lcm` needed.
`TrunkTrajectory` keeps the decoded samples on the GPU and reproduces, per robot instance and in one
kernel, the per-tick lookup of `TowrTrunkPlanner.SetTrunkOutputs` (planners/towr.py:92-148):
standing targets for t < wait_time, else the sample nearest to t - wait_time.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import workloads


def decode_trunk_state(buf):
    """bytes -> dict with the fields of trunk_state_t (+ 'targets' [54] and 'contact_mask')."""
    L = _lib.lib()
    s = _lib.WbcTrunkState()
    rc = L.wbc_trunk_state_decode(bytes(buf), len(buf), C.byref(s))
    if rc != 0:
        raise ValueError("Decode error" if rc == -3 else "short trunk_state_t buffer")   # trunk_state_t.py:87-88
    t = np.zeros(54); m = C.c_uint8(0)
    _lib.check(L.wbc_trunk_state_to_targets(C.byref(s), t.ctypes.data_as(_lib.c_double_p), C.byref(m)))
    a = lambda x: np.array(x[:])
    a2 = lambda x: np.array([list(r) for r in x])
    return dict(timestamp=s.timestamp, finished=bool(s.finished), base_p=a(s.base_p), base_pd=a(s.base_pd),
                base_pdd=a(s.base_pdd), base_rpy=a(s.base_rpy), base_rpyd=a(s.base_rpyd), base_rpydd=a(s.base_rpydd),
                foot_p=a2(s.foot_p), foot_pd=a2(s.foot_pd), foot_pdd=a2(s.foot_pdd),
                contact=[bool(c) for c in s.contact], foot_f=a2(s.foot_f), targets=t, contact_mask=int(m.value))


class TrunkTrajectory:
    """Device-resident trajectory table + nearest-timestamp lookup (planners/towr.py:92-106)."""

    def __init__(self, timestamps, targets, masks, model="mini_cheetah", wait_time=1.0, device=0,
                 standing_targets=None, standing_mask=0b1111):
        ts = np.ascontiguousarray(timestamps, dtype=np.float64)
        tg = np.ascontiguousarray(targets, dtype=np.float64).reshape(ts.size, 54)
        mk = np.ascontiguousarray(masks, dtype=np.uint8)
        st = workloads.standing_targets(model, 1)[:, 0] if standing_targets is None else np.asarray(standing_targets, float)
        st = np.ascontiguousarray(st, dtype=np.float64)
        self.device = device
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.wbc_traj_create(device, ts.size, ts.ctypes.data_as(_lib.c_double_p), tg.ctypes.data_as(_lib.c_double_p),
                                     mk.ctypes.data_as(_lib.c_u8_p), st.ctypes.data_as(_lib.c_double_p), standing_mask,
                                     float(wait_time), C.byref(h))
        _lib.check(rc)   # message from wbc_last_error(): bad argument / non-decreasing timestamps / the HIP failure
        self._h = h

    @classmethod
    def from_messages(cls, messages, **kw):
        """messages: iterable of 549-byte trunk_state_t buffers (what planners/towr.py:37-48 collects)."""
        d = [decode_trunk_state(m) for m in messages]
        return cls([x["timestamp"] for x in d], np.stack([x["targets"] for x in d]), [x["contact_mask"] for x in d], **kw)

    def lookup(self, time, out=None):
        """time: CUDA float64 tensor [n] -> (targets [54, n], contact_mask [n]) on the current stream."""
        import torch
        if not (isinstance(time, torch.Tensor) and time.is_cuda and time.dtype == torch.float64 and time.is_contiguous()
                and time.dim() == 1 and time.device.index == self.device):
            raise ValueError("time: expected a contiguous float64 CUDA tensor [n] on device %d" % self.device)
        n = int(time.shape[0])
        if out is not None:
            tg, mk = out
            ok = (tg.is_cuda and tg.dtype == torch.float64 and tg.is_contiguous() and tuple(tg.shape) == (54, n) and
                  mk.is_cuda and mk.dtype == torch.uint8 and mk.is_contiguous() and tuple(mk.shape) == (n,))
            if not ok:
                raise ValueError("out: expected (float64 [54, n], uint8 [n]) contiguous CUDA tensors")
        if out is None:
            dev = time.device
            out = (torch.empty((54, n), dtype=torch.float64, device=dev), torch.empty((n,), dtype=torch.uint8, device=dev))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.wbc_traj_lookup(self._h, C.c_void_p(stream), n, n, C.c_void_p(time.data_ptr()),
                                           C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr())))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._L.wbc_traj_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
