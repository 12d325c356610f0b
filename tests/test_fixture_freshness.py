"""The committed fixtures that were produced by executing reference code are regenerated and compared, whenever the
reference is at hand (this container; it does not exist on the GPU box, where these tests skip): a fixture can then
not silently drift from its committed generator, from the stand-ins or from the reference."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/controllers"), reason="reference sources not present")


@pytest.mark.parametrize("script,fixture", [("make_reference_law_golden.py", "reference_law_golden.npz"),
                                            ("make_planner_golden.py", "planner_golden.npz"),
                                            ("make_robot_state_golden.py", "robot_state_expected.npz")])
def test_generator_reproduces_the_committed_fixture(tmp_path, script, fixture):
    env = dict(os.environ, GOLDEN_OUT=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(HERE, "golden", script)], check=True, env=env, capture_output=True, timeout=600)
    new = np.load(os.path.join(str(tmp_path), fixture)); old = np.load(os.path.join(HERE, "golden", fixture))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        a, b = new[k], old[k]
        if a.dtype.kind in "fc":
            assert a.shape == b.shape and np.allclose(a, b, rtol=1e-9, atol=1e-10), k     # same machine: in practice bit-equal
        else:
            assert np.array_equal(a, b), k
