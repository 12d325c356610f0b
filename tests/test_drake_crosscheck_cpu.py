"""tools/drake_crosscheck.py is the script that turns "parity unpinned at the Drake / OSQP boundary" into a measurement on
any machine with pydrake.  pydrake is not in this image, so only its PLUMBING is tested here, against tests/fake_pydrake
(a plant that numbers its joints breadth-first and its actuators at random): joint renumbering, the Drake call sequence of
basic_controller.py:110-113,180-195,253-267, the controllers' port protocol and the capture of the solver's solution."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_crosscheck_plumbing_against_the_stand_in_plant():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "drake_crosscheck.py"), "--fake", "--n", "2",
                        "--cases", "cfg2_id", "cfg3_mptc", "cfg4_anymal_mptc", "cfg3_pc"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "stage A (Drake's rigid-body numbers vs oracle/): AGREE" in r.stdout
    if os.path.isdir("/root/reference/controllers"):      # stage B executes the reference's controllers where they lie
        assert "stage B (controllers + real solver, v-dot): AGREE" in r.stdout and "stage B skipped" not in r.stdout


def test_crosscheck_without_pydrake_says_so():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "drake_crosscheck.py"), "--n", "1"], capture_output=True, text=True, timeout=120)
    try:
        import pydrake  # noqa: F401
    except ImportError:
        assert r.returncode != 0 and "pydrake is not importable" in (r.stderr + r.stdout)
