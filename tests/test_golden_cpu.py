"""Oracle and host-instantiated kernel math against the committed golden vectors (CPU)."""
import glob
import os

import numpy as np
import pytest

import host_tick as ht
from oracle import oracle_py as orc

GOLD = sorted(f for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
              if os.path.basename(f).startswith(("cfg", "masks16")))   # the tick fixtures of make_golden.py


def load(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["model"] = str(d["model"]); d["kind"] = str(d["kind"])
    d["mu"] = d["mu"] if d["mu"].size else None
    d["mass_scale"] = d["mass_scale"] if d["mass_scale"].size else None
    return d


def test_golden_exist():
    assert len(GOLD) >= 7


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_oracle_reproduces_golden(path):
    g = load(path)
    m = orc.model(g["model"]); p = orc.params(g["kind"])
    tau, met, st = orc.step_batch(g["kind"], m, p, g["q"], g["v"], g["targets"], g["mask"], g["mu"], g["mass_scale"])
    assert np.array_equal(st, g["status"])
    assert np.allclose(tau, g["tau"], rtol=1e-9, atol=1e-9)
    assert np.allclose(met, g["metrics"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_kernel_math_on_host_matches_golden(path):
    g = load(path)
    t = orc.load_model_json(g["model"])
    tau, met, st, it = ht.run(g["kind"], t["flat"], g["q"], g["v"], g["targets"], g["mask"], g["mu"], g["mass_scale"])
    assert np.array_equal(st, g["status"])
    rel = np.abs(tau - g["tau"]).max(0) / np.maximum(np.abs(g["tau"]).max(0), 1e-3)
    assert rel.max() < 1e-5
