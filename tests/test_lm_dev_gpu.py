"""Device primitives of the per-frame LM kernels (vdo_slam_amd/csrc/lm_dev.hpp), checked by the stand-alone program
tools/lm_dev_check (built by __graft_entry__.build()): the register butterfly reduction against exact integer sums, and the
lane-parallel pivoted 6x6 LDLT against the one-lane routine - bit for bit over 20 000 systems (SPD, zero, negative definite,
badly scaled, tied diagonal entries, a zero row / column, indefinite, tiny pivots)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lm_device_primitives():
    exe = os.path.join(ROOT, "tools", "lm_dev_check")
    if not os.path.exists(exe):
        pytest.fail("tools/lm_dev_check is missing: run __graft_entry__.build()")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MISMATCH" not in r.stdout
    assert r.stdout.count("ok") >= 5, r.stdout
