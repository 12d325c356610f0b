"""TEST-ONLY stand-in for pydrake.autodiffutils: a value + derivative-vector CONTAINER (no arithmetic).  The
reference's helpers.jacobian2 (helpers.py:5-33) seeds such objects, hands them to the plant and reads
`.derivatives()` back; the fake plant (refplant.py) is what fills the derivatives in."""
import numpy as np


class AutoDiffXd:
    def __init__(self, value, derivatives=()):
        self._v = float(value)
        self._d = np.asarray(derivatives, dtype=float).reshape(-1)

    def value(self):
        return self._v

    def derivatives(self):
        return self._d

    def __float__(self):
        return self._v


def split(x):
    """object array of AutoDiffXd (or plain numbers) -> (values, seed matrix [x.size, nd] or None)."""
    flat = np.asarray(x, dtype=object).reshape(-1)
    vals = np.array([e.value() if isinstance(e, AutoDiffXd) else float(e) for e in flat])
    nd = max([e.derivatives().size for e in flat if isinstance(e, AutoDiffXd)] + [0])
    if nd == 0:
        return vals, None
    D = np.zeros((flat.size, nd))
    for i, e in enumerate(flat):
        if isinstance(e, AutoDiffXd) and e.derivatives().size:
            D[i] = e.derivatives()
    return vals, D


def join(values, dvalues):
    """values [shape], dvalues [shape + (nd,)] -> object array of AutoDiffXd."""
    values = np.asarray(values, dtype=float)
    out = np.empty(values.shape, dtype=object)
    for idx in np.ndindex(values.shape):
        out[idx] = AutoDiffXd(values[idx], dvalues[idx])
    return out
