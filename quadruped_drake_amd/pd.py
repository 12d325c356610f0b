"""The reference's joint-space PD law -- control method "B", `BasicController.ControlLaw`
(controllers/basic_controller.py:322-352) -- batched on the device: u = clip(S (-Kp N+(q)(q - q_nom) - Kd v), +-150).
S selects the joint rows, where N+ is the identity, so the law is twelve multiply-subtracts per robot; it is here for the
completeness of the controller family (simulate.py:13-17), not for its cost.  Bit-exact against the reference's executed
code (tests/test_pd.py)."""
import ctypes as C

import numpy as np

from . import _lib

Q_NOM = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.3] + [0.0, -0.8, 1.6] * 4)      # basic_controller.py:333-340


class BasicController:
    """Same name and gains as the reference's class (Kp 30, Kd 1.5, clip 150); `q_nom` is in the plant's own joint order, as
    the reference's literal is.  q_perm / act_perm as for the QP controllers (plant numbering of joints / actuators)."""

    def __init__(self, device=0, kp=30.0, kd=1.5, u_max=150.0, q_nom=None, q_perm=None, act_perm=None):
        self.device, self.kp, self.kd, self.u_max = int(device), float(kp), float(kd), float(u_max)
        self.q_nom = np.ascontiguousarray(Q_NOM if q_nom is None else q_nom, dtype=np.float64).reshape(19)
        arr = lambda p: None if p is None else (C.c_int * 12)(*[int(x) for x in p])
        self._qp, self._ap = arr(q_perm), arr(act_perm)

    def step(self, q, v, out=None):
        """q [19, N], v [18, N] float64 CUDA tensors -> tau [12, N] in actuator order, asynchronously on torch's current stream."""
        import torch
        n = int(q.shape[1])
        for t, rows, name in ((q, 19, "q"), (v, 18, "v")):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
                    and tuple(t.shape) == (rows, n) and t.device.index == self.device):
                raise ValueError("%s: expected a contiguous float64 CUDA tensor [%d, N] on device %d" % (name, rows, self.device))
        if out is None:
            out = torch.empty((12, n), dtype=torch.float64, device=q.device)
        elif not (out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and tuple(out.shape) == (12, n)):
            raise ValueError("out: expected a contiguous float64 CUDA tensor [12, N]")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().wbc_pd_step(self.device, C.c_void_p(stream), n, n, C.c_void_p(q.data_ptr()), C.c_void_p(v.data_ptr()),
                                          self.q_nom.ctypes.data_as(_lib.c_double_p), self.kp, self.kd, self.u_max, self._qp, self._ap,
                                          C.c_void_p(out.data_ptr())))
        return out

    def ControlLaw(self, q, v):
        """One robot, the reference's signature (context dropped): numpy in, numpy u[12] out."""
        import torch
        dev = "cuda:%d" % self.device
        u = self.step(torch.tensor(np.asarray(q, float).reshape(19, 1), device=dev), torch.tensor(np.asarray(v, float).reshape(18, 1), device=dev))
        return u.cpu().numpy()[:, 0]
