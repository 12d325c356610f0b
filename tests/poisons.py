"""Malformed instances (include/wbc.h "Malformed instances"): the cases the host test (tests/test_kernel_math_host.py) and the device tests
(tests/test_robustness_gpu.py) put into chosen slots of a batch.  Each entry: name -> (function that damages column i of the batch arrays in place,
expected status of that instance: 2 = reported with zero outputs, 0 = a legal input that only looks odd, None = 0 or 2 (finite inputs that may
overflow on the way: what matters is that nothing non-finite comes out and nobody else is touched), "rf_swing" = 2 when the RF foot (bit 1 of the
contact mask) swings and 0 -- the value is not read, as in the reference -- when it stands)."""
import numpy as np


def _set(arr, row, val):
    def f(b, i):
        b[arr][row, i] = val
    return f


def _zero_quat(b, i):
    b["q"][0:4, i] = 0.0


def _scale_quat(b, i):
    b["q"][0:4, i] *= 3.7


def _tiny_quat(b, i):
    b["q"][0:4, i] *= 1e-170          # |q|^2 underflows


def _mask_high_bits(b, i):
    b["mask"][i] |= 0xF0


def _mu(val):
    def f(b, i):
        b["mu"][i] = val
    return f


def _ms(val):
    def f(b, i):
        b["mass_scale"][i] = val
    return f


POISONS = {
    "nan_quat":          (_set("q", 0, np.nan), 2),
    "nan_position":      (_set("q", 5, np.nan), 2),
    "nan_joint":         (_set("q", 7 + 4, np.nan), 2),
    "nan_base_rate":     (_set("v", 1, np.nan), 2),
    "nan_joint_rate":    (_set("v", 6 + 7, np.nan), 2),
    "nan_body_target":   (_set("targets", 2, np.nan), 2),
    "nan_body_rate_tgt": (_set("targets", 10, np.nan), 2),
    "nan_foot_target":   (_set("targets", 18 + 9 * 1 + 2, np.nan), "rf_swing"),     # RF foot: swinging in half of the trot batch, in contact in the other half and on every stand
    "inf_foot_rate_tgt": (_set("targets", 18 + 9 * 1 + 4, np.inf), "rf_swing"),
    "nan_last_row":      (lambda b, i: (b["targets"].__setitem__((53, i), np.nan), b["mask"].__setitem__(i, b["mask"][i] & 0x7)), 2),   # (RH foot made to swing: row 53 is read)
    "inf_position":      (_set("q", 4, np.inf), 2),
    "neg_inf_rate":      (_set("v", 4, -np.inf), 2),
    "inf_target":        (_set("targets", 13, np.inf), 2),
    "zero_quat":         (_zero_quat, 2),
    "tiny_quat":         (_tiny_quat, 2),
    "nonunit_quat":      (_scale_quat, 0),
    "mask_high_bits":    (_mask_high_bits, 0),
    "nan_mu":            (_mu(np.nan), 2),
    "inf_mu":            (_mu(np.inf), 2),
    "negative_mu":       (_mu(-0.7), 2),
    "zero_mass_scale":   (_ms(0.0), 2),
    "inf_mass_scale":    (_ms(np.inf), 2),
    "huge_target":       (_set("targets", 0, 1e200), None),
    "huge_rate":         (_set("v", 3, 1e160), None),
}


def copy_batch(b, n):
    """Arrays of a workloads.make_batch dictionary, copied; mu / mass_scale always present (the handle's mu, 1.0)."""
    out = {k: np.array(b[k], copy=True) for k in ("q", "v", "targets", "mask")}
    out["mu"] = np.array(b["mu"], copy=True) if b.get("mu") is not None else np.full(n, 0.7)
    out["mass_scale"] = np.array(b["mass_scale"], copy=True) if b.get("mass_scale") is not None else np.ones(n)
    return out


def expected_status(name, mask_i):
    want = POISONS[name][1]
    if want == "rf_swing":
        return 0 if (int(mask_i) >> 1) & 1 else 2
    return want
