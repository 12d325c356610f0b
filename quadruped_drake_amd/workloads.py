"""Synthetic randomised-state batches for BASELINE.json's configs.

All arrays are SoA with the batch index fastest: q[19, N], v[18, N], targets[54, N],
mask[N] (bit i = foot i of [LF, RF, LH, RH] in contact).

State conventions are the reference's (simulate.py:171-176, planners/simple.py:45-85):
  q = [qw qx qy qz | x y z | 12 joints leg-major], v = [w_WB | v_WBo | 12 joint rates],
  targets = body p, pd, pdd, rpy, rpyd, rpydd (18) then per foot p, pd, pdd (9 each).
"""
import numpy as np

# planners/simple.py:45-52 (mini cheetah / ANYmal standing foot set), [LF RF LH RH]
STAND_FEET = {
    "mini_cheetah": np.array([[0.175, 0.11, 0.0], [0.175, -0.11, 0.0], [-0.2, 0.11, 0.0], [-0.2, -0.11, 0.0]]),
    "anymal_b": np.array([[0.34, 0.19, 0.0], [0.34, -0.19, 0.0], [-0.34, 0.19, 0.0], [-0.34, -0.19, 0.0]]),
}
# simulate.py:171-176 nominal joints; ANYmal: X-configuration that realises TOWR's nominal
# stance (+-0.34, +-0.19, -0.42) (towr/include/towr/models/examples/anymal_model.h:46-53)
NOMINAL_JOINTS = {
    "mini_cheetah": np.tile([0.0, -0.8, 1.6], 4),
    "anymal_b": np.array([0.0, 0.4, -0.8, 0.0, 0.4, -0.8, 0.0, -0.4, 0.8, 0.0, -0.4, 0.8]),
}
NOMINAL_HEIGHT = {"mini_cheetah": 0.30, "anymal_b": 0.50}

TROT_MASKS = (0b1001, 0b0110)  # {LF,RH} and {RF,LH}: towr/src/quadruped_gait_generator.cc:58-59


def rpy_to_quat(rpy):
    r, p, y = rpy[0] * 0.5, rpy[1] * 0.5, rpy[2] * 0.5
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.stack([cr * cp * cy + sr * sp * sy,
                     sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy])


def standing_targets(model="mini_cheetah", n=1):
    """planners/simple.py:39-85 SimpleStanding, broadcast to a batch."""
    t = np.zeros((54, n))
    t[2] = NOMINAL_HEIGHT[model] if model != "mini_cheetah" else 0.3
    for i in range(4):
        t[18 + 9 * i:21 + 9 * i] = STAND_FEET[model][i][:, None]
    return t


def nominal_state(model="mini_cheetah", n=1):
    """simulate.py:171-179 initial state, broadcast."""
    q = np.zeros((19, n)); v = np.zeros((18, n))
    q[0] = 1.0
    q[6] = NOMINAL_HEIGHT[model]
    q[7:] = NOMINAL_JOINTS[model][:, None]
    return q, v


def make_batch(config, n=None, seed=None, model=None, window=None):
    """config in {2, 3, 4, 5} (BASELINE.json configs[1..4]).  Returns a dict.
    window = (lo, hi): only instances lo..hi-1 of the n-instance batch are KEPT (identical values: every array is drawn
    from the one seeded stream exactly as for the whole batch and cut right after its draw, so a rank of a sharded run never
    holds more than one full-width array at a time; out["n"] = hi - lo, out["n_total"] = n)."""
    defaults = {2: (1024, 1001, "mini_cheetah", "id"), 3: (4096, 1002, "mini_cheetah", "mptc"),
                4: (4096, 1003, "anymal_b", "mptc"), 5: (32768, 1004, "mini_cheetah", "mptc")}
    dn, dseed, dmodel, kind = defaults[config]
    n = dn if n is None else n
    seed = dseed if seed is None else seed
    model = dmodel if model is None else model
    rng = np.random.default_rng(seed)
    trot = config in (3, 4, 5)
    vsig = 0.5 if trot else 0.3
    lo, hi = (0, n) if window is None else (int(window[0]), int(window[1]))
    assert 0 <= lo <= hi <= n
    w = hi - lo
    cut = lambda a: a[..., lo:hi]          # every draw has the full width (same stream), only the window is kept

    rpy = cut(rng.uniform(-0.2, 0.2, (3, n)))
    q = np.zeros((19, w))
    q[0:4] = rpy_to_quat(rpy)
    q[4:6] = cut(rng.uniform(-1.0, 1.0, (2, n)))
    h0 = NOMINAL_HEIGHT[model]
    q[6] = cut(rng.uniform(h0 - 0.03, h0 + 0.03, n))
    q[7:] = NOMINAL_JOINTS[model][:, None] + cut(rng.uniform(-0.25, 0.25, (12, n)))
    v = np.ascontiguousarray(cut(rng.normal(0.0, vsig, (18, n))))

    t = np.zeros((54, w))
    t[0:3] = q[4:7] + cut(rng.normal(0, 0.02, (3, n)))         # p_body
    t[3:9] = cut(rng.normal(0, 0.1, (6, n)))                    # pd, pdd
    t[9:12] = cut(rng.normal(0, 0.05, (3, n)))                  # rpy target
    t[12:18] = cut(rng.normal(0, 0.1, (6, n)))                  # rpyd, rpydd
    if trot:
        mask = np.where(cut(rng.random(n)) < 0.5, TROT_MASKS[0], TROT_MASKS[1]).astype(np.uint8)
    else:
        mask = np.full(w, 0b1111, dtype=np.uint8)
    for i in range(4):
        p = np.tile(STAND_FEET[model][i][:, None], (1, w))
        p[0:2] += q[4:6]
        swing = ((mask >> i) & 1) == 0
        p[2] = np.where(swing, cut(rng.uniform(0.05, 0.10, n)), 0.0)
        t[18 + 9 * i:21 + 9 * i] = p
        sig = np.where(swing, 0.3, 0.1)
        t[21 + 9 * i:24 + 9 * i] = cut(rng.normal(0, 1.0, (3, n))) * sig
        t[24 + 9 * i:27 + 9 * i] = cut(rng.normal(0, 0.1, (3, n)))
    out = dict(config=config, n=w, n_total=n, window=(lo, hi), seed=seed, model=model, kind=kind, q=q, v=v, targets=t, mask=mask,
               mu=None, mass_scale=None)
    if config == 5:
        out["mu"] = np.ascontiguousarray(cut(rng.uniform(0.4, 1.0, n)))
        out["mass_scale"] = np.ascontiguousarray(cut(rng.uniform(0.8, 1.2, n)))
    return out


def dump_batch(path, batch, model_table=None):
    """The batch as ONE little-endian binary file for a host without Python (examples/wbc_host.cpp reads it):
    "WBCBATCH" | int32 version = 1, kind, n, has_mu | wbc_model (flat[215] doubles, q_perm[12], act_perm[12] int32) |
    q[19][n] v[18][n] targets[54][n] doubles, batch index fastest | mask[n] bytes, zero-padded to a multiple of 8 |
    [mu[n] mass_scale[n] doubles]."""
    import json
    import os
    if model_table is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", batch["model"] + ".json")) as f:
            model_table = json.load(f)
    n = int(batch["n"])
    kind = {"id": 0, "mptc": 1, "pc": 2, "clf": 3}[batch["kind"]]
    has_mu = batch["mu"] is not None
    with open(path, "wb") as f:
        f.write(b"WBCBATCH")
        f.write(np.array([1, kind, n, int(has_mu)], dtype="<i4").tobytes())
        f.write(np.asarray(model_table["flat"], dtype="<f8").tobytes())
        f.write(np.arange(12, dtype="<i4").tobytes())
        f.write(np.asarray(model_table.get("act_perm", range(12)), dtype="<i4").tobytes())
        for k, rows in (("q", 19), ("v", 18), ("targets", 54)):
            a = np.ascontiguousarray(batch[k], dtype="<f8")
            assert a.shape == (rows, n)
            f.write(a.tobytes())
        mk = np.zeros((n + 7) // 8 * 8, dtype=np.uint8)
        mk[:n] = batch["mask"]
        f.write(mk.tobytes())
        if has_mu:
            f.write(np.ascontiguousarray(batch["mu"], dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(batch["mass_scale"], dtype="<f8").tobytes())
    return path
