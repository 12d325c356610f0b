"""The robot-side wire format of the reference's `use_lcm` path (controllers/basic_controller.py:52-61,79-87,289-314):
`robot_state_control_lcmt` messages (lcm_types/robot_state_control_lcmt.lcm: float q[19], v[18], tau[12]; 204 bytes).

  decode_robot_state / encode_robot_state     one message on the host (the C codec of libwbc_hip.so, no `lcm` needed)
  unpack_states(messages)                     N "robot_current_state" messages -> q[19, N], v[18, N] on the device
  pack_controls(tau, q_perm, act_perm)        tau[12, N] (actuator order, as wbc_step writes it) -> N
                                              "robot_control_input" messages: tau = (S'u)[-12:] in the plant's joint
                                              order, q and v zero -- what basic_controller.py:308-314 publishes

Transport (LCM sockets, channels) is out of scope; these are the bytes either side of the tick, batched.
Bit-exact against the reference's own generated codec (tests/golden/make_robot_state_golden.py)."""
import ctypes as C

import numpy as np

from . import _lib

MESSAGE_BYTES = 204


def decode_robot_state(buf):
    """bytes -> dict(q [19], v [18], tau [12]) of float32 values; ValueError("Decode error") on a foreign fingerprint."""
    s = _lib.WbcRobotState()
    rc = _lib.lib().wbc_robot_state_decode(bytes(buf), len(buf), C.byref(s))
    if rc != 0:
        raise ValueError("Decode error" if rc == -3 else "short robot_state_control_lcmt buffer")
    return dict(q=np.array(s.q[:], dtype=np.float32), v=np.array(s.v[:], dtype=np.float32), tau=np.array(s.tau[:], dtype=np.float32))


def encode_robot_state(q=None, v=None, tau=None):
    """-> 204 bytes; values are rounded to float32 (round to nearest even, like the reference's struct.pack('>f'))."""
    s = _lib.WbcRobotState()
    for name, n, val in (("q", 19, q), ("v", 18, v), ("tau", 12, tau)):
        a = np.zeros(n, dtype=np.float32) if val is None else np.asarray(val, dtype=np.float64).reshape(n).astype(np.float32)
        getattr(s, name)[:] = a.tolist()
    out = C.create_string_buffer(MESSAGE_BYTES)
    if _lib.lib().wbc_robot_state_encode(C.byref(s), out, MESSAGE_BYTES) != MESSAGE_BYTES:
        raise ValueError("encode failed")
    return out.raw


def _msgs_tensor(messages, device):
    import torch
    if isinstance(messages, torch.Tensor):
        t = messages
    else:
        raw = messages if isinstance(messages, (bytes, bytearray)) else b"".join(messages)
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.dim() == 1 and t.numel() % MESSAGE_BYTES == 0):
        raise ValueError("messages: a contiguous uint8 CUDA tensor (or bytes) holding whole 204-byte messages")
    return t


def unpack_states(messages, device=0, out=None):
    """messages: bytes / list of bytes / uint8 CUDA tensor [N * 204] -> (q [19, N], v [18, N], ok [N] uint8) on the device,
    asynchronously on torch's current stream.  ok[i] = 0: foreign fingerprint, that robot's columns are left as they were."""
    import torch
    dev = "cuda:%d" % device
    t = _msgs_tensor(messages, dev)
    n = t.numel() // MESSAGE_BYTES
    if out is None:
        out = (torch.zeros((19, n), dtype=torch.float64, device=dev), torch.zeros((18, n), dtype=torch.float64, device=dev))
    q, v = out
    if not all(x.is_cuda and x.dtype == torch.float64 and x.is_contiguous() for x in (q, v)) or \
            tuple(q.shape) != (19, n) or tuple(v.shape) != (18, n):
        raise ValueError("out: (float64 [19, N], float64 [18, N]) contiguous CUDA tensors")
    ok = torch.empty((n,), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(device).cuda_stream
    _lib.check(_lib.lib().wbc_robot_states_unpack(device, C.c_void_p(stream), n, n, C.c_void_p(t.data_ptr()),
                                                  C.c_void_p(q.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(ok.data_ptr())))
    return q, v, ok


def pack_controls(tau, q_perm=None, act_perm=None):
    """tau: float64 CUDA tensor [12, N] in actuator order -> uint8 CUDA tensor [N * 204] of "robot_control_input" messages."""
    import torch
    if not (isinstance(tau, torch.Tensor) and tau.is_cuda and tau.dtype == torch.float64 and tau.is_contiguous() and
            tau.dim() == 2 and tau.shape[0] == 12):
        raise ValueError("tau: a contiguous float64 CUDA tensor [12, N]")
    n = int(tau.shape[1])
    device = tau.device.index
    msgs = torch.empty((n * MESSAGE_BYTES,), dtype=torch.uint8, device=tau.device)
    arr = lambda p: None if p is None else (C.c_int * 12)(*[int(x) for x in p])
    stream = torch.cuda.current_stream(device).cuda_stream
    _lib.check(_lib.lib().wbc_robot_controls_pack(device, C.c_void_p(stream), n, n, C.c_void_p(tau.data_ptr()), arr(q_perm),
                                                  arr(act_perm), C.c_void_p(msgs.data_ptr())))
    return msgs
