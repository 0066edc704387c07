"""`.g2o` reader/writer in the reference's dialect."""
import ctypes as C
import subprocess
import sys

import numpy as np

from vdo_slam_amd import _capi as K
from vdo_slam_amd import g2o_io, synth


def _same_graph(a, b, tol):
    for f in ("eb_pose", "eb_point", "et_p1", "et_p2", "et_pose", "ep_i", "ep_j", "pr_pose"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for f in ("pose", "point", "eb_z", "eb_w", "et_z", "et_w", "ep_z", "ep_info", "pr_z", "pr_info"):
        x, y = np.asarray(getattr(a, f), float), np.asarray(getattr(b, f), float)
        assert x.shape == y.shape, f
        if x.size:
            assert np.abs(x - y).max() <= tol * max(1.0, np.abs(x).max()), f


def test_round_trip_is_lossless_at_17_digits(tmp_path):
    g = synth.make_ba_graph(8, 120, 2, 12, seed=6)
    p = tmp_path / "g.g2o"
    g2o_io.write_g2o(p, g)
    h = g2o_io.read_g2o(p)
    _same_graph(g, h, 1e-14)          # rotations pass through a unit quaternion: a few ulp
    txt = open(p).read().split("\n")
    assert txt[0].startswith("PARAMS_SE3OFFSET 0") and sum(l.startswith("EDGE_SE3_MOTION") for l in txt) == g.n_et


def test_reader_accepts_the_reference_precision_and_foreign_ids(tmp_path):
    """The reference streams 6 significant digits and numbers vertices in visiting order (cameras, points and
    motions interleaved): ids are remapped in ascending order per vertex type."""
    g = synth.make_ba_graph(6, 60, 1, 8, seed=7)
    p = tmp_path / "ref_style.g2o"
    g2o_io.write_g2o(p, g, digits=6)
    lines = open(p).read().strip().split("\n")
    remap = lambda i: 3 * int(i) + 11                     # monotone, so per-type order is preserved
    out = []
    for l in lines:
        t = l.split()
        if t[0].startswith("VERTEX"):
            t[1] = str(remap(t[1]))
        elif t[0] == "EDGE_SE3_PRIOR":
            t[1] = str(remap(t[1]))
        elif t[0] in ("EDGE_SE3:QUAT", "EDGE_SE3_TRACKXYZ"):
            t[1], t[2] = str(remap(t[1])), str(remap(t[2]))
        elif t[0] == "EDGE_SE3_MOTION":
            t[1], t[2], t[3] = str(remap(t[1])), str(remap(t[2])), str(remap(t[3]))
        out.append(" ".join(t))
    out.insert(3, "FIX 11")
    open(p, "w").write("\n".join(out) + "\n")
    h = g2o_io.read_g2o(p)
    _same_graph(g, h, 2e-5)


def test_replayed_graph_optimises_like_the_original(tmp_path, oracle):
    g = synth.make_ba_graph(8, 150, 1, 15, seed=8)
    p = tmp_path / "g.g2o"
    g2o_io.write_g2o(p, g)
    h = g2o_io.read_g2o(p)
    res = []
    for graph in (g, h):
        gc, keep = K.graph_to_c(graph)
        opt = K.LMOptionsC(30, 1e-4, 0, 0, 0.0, 0)
        st = K.LMStatsC()
        pose = np.zeros_like(graph.pose); point = np.zeros_like(graph.point)
        assert oracle.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), K._dp(pose), K._dp(point), C.byref(st)) == 0
        res.append((st.iterations, st.final_chi2, pose))
    assert res[0][0] == res[1][0] and abs(res[0][1] - res[1][1]) <= 1e-9 * res[0][1]
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-9
    # the stand-alone tool (oracle mode: no GPU here)
    r = subprocess.run([sys.executable, "tools/ba_replay.py", str(p), "--iterations", "30", "--huber", repr(float(g.huber_eb)), "--oracle"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    after = g2o_io.read_g2o(tmp_path / "g_after_opt.g2o")
    assert np.abs(after.pose - res[0][2]).max() < 1e-9
