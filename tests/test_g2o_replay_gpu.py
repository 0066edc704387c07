"""`.g2o` replay on the GPU: a graph written in the reference's dialect, read back and optimised by the HIP
solver gives the oracle's LM (iterations, trials, chi2, estimates) - through the Python reader and through the stand-alone
tool (tools/ba_replay.py, GPU mode), which writes `<name>_after_opt.g2o` like the reference (src/Optimizer.cc:1934-1936)."""
import ctypes as C
import subprocess
import sys

import numpy as np
import pytest

from vdo_slam_amd import _capi as K
from vdo_slam_amd import g2o_io, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("digits", [17, 6])
def test_g2o_write_read_hip_lm_matches_oracle_lm(oracle, tmp_path, digits):
    """digits = 6 is what the reference's own writer emits (default ostream precision): the replayed problem is then the ROUNDED
    graph - both solvers are given that same graph."""
    from vdo_slam_amd.ba import BatchBA, Context
    g = synth.make_ba_graph(10, 400, 2, 40, seed=13)
    p = tmp_path / "g.g2o"
    g2o_io.write_g2o(p, g, digits=digits)
    h = g2o_io.read_g2o(p, huber_eb=g.huber_eb, huber_et=g.huber_et, huber_ep=g.huber_ep)
    assert (h.n_pose, h.n_point, h.n_eb, h.n_et) == (g.n_pose, g.n_point, g.n_eb, g.n_et)
    gc, keep = K.graph_to_c(h)
    opt = K.LMOptionsC(40, 1e-4, 0, 0, 0.0, 0)
    st_o = K.LMStatsC()
    pose_o = np.zeros_like(h.pose); point_o = np.zeros_like(h.point)
    assert oracle.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), K._dp(pose_o), K._dp(point_o), C.byref(st_o)) == 0
    ctx = Context(0)
    ba = BatchBA(ctx, h)
    st = ba.optimize(max_iterations=40, gain_threshold=1e-4)
    pose, point = ba.estimates()
    ba.close()
    assert (st.iterations, st.total_trials) == (st_o.iterations, st_o.total_trials) and st.iterations >= 2
    assert abs(st.final_chi2 - st_o.final_chi2) <= 1e-6 * st_o.final_chi2 and st.final_chi2 < st.initial_chi2
    assert np.abs(pose[:, :9] - pose_o[:, :9]).max() <= 1e-4
    assert np.abs(pose[:, 9:] - pose_o[:, 9:]).max() <= 1e-4 * np.abs(pose_o[:, 9:]).max()
    assert np.abs(point - point_o).max() <= 1e-4 * np.abs(point_o).max()
    if digits == 17:
        # the stand-alone tool in GPU mode on the same file
        r = subprocess.run([sys.executable, "tools/ba_replay.py", str(p), "--iterations", "40", "--huber", repr(float(g.huber_eb))], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        after = g2o_io.read_g2o(tmp_path / "g_after_opt.g2o")
        assert np.abs(after.pose[:, 9:] - pose_o[:, 9:]).max() <= 1e-4 * np.abs(pose_o[:, 9:]).max()
        assert np.abs(after.point - point_o).max() <= 1e-4 * np.abs(point_o).max()
