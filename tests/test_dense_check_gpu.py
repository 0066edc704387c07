"""The dense reduced-camera solver (vdo_slam_amd/csrc/ba_dense.hip: blocked Cholesky on the fp64 MFMA units + substitutions) on its own,
checked by the stand-alone program tools/dense_check (built by __graft_entry__.build()): random SPD systems of 1 .. 34 blocks of 64 (incl.
padded ones) through ALL launch sequences (VDO_BA_DENSE = 1 .. 5) against a long-double host Cholesky (solution <= 1e-10 relative, residual <= 1e-12 componentwise),
and an indefinite matrix, for which the failure flag must rise (g2o's "Cholesky failure", g2o/solvers/linear_solver_dense.h:65-113); the one-workgroup
solver of small systems (k_dense_small) against the same host Cholesky."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dense_solver_against_host_cholesky():
    exe = os.path.join(ROOT, "tools", "dense_check")
    if not os.path.exists(exe):
        pytest.fail("tools/dense_check is missing: run __graft_entry__.build()")
    r = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "BAD" not in r.stdout and "dense_check: 0 bad" in r.stdout, r.stdout
    assert r.stdout.count(" ok") >= 46, r.stdout
    # (round 6) k_dense_small - <= 128 unknowns in one workgroup: 18 .. 126 unknowns and an indefinite matrix, S zeroed behind the read
    small = [l for l in r.stdout.splitlines() if l.startswith("k_dense_small")]
    assert len(small) >= 6 and all(l.rstrip().endswith("ok") and "S zeroed 1" in l for l in small), small
