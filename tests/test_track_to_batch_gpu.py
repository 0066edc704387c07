"""Track() -> Map -> FullBatchOptimization end to end on the GPU: FramePipeline fills a VDO_SLAM::Map frame by frame
("Save Graph Structure", src/Tracking.cc:1046-1110), the C++ graph builder + GPU solver optimise it, and the same Map goes
through the Python restatement of the reference's builder + the oracle's LM."""
import ctypes as C

import numpy as np
import pytest

from vdo_slam_amd import _capi as K
from tests import map_builder_ref as SM
from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ
from vdo_slam_amd.ba import Context
from vdo_slam_amd.pipeline import FramePipeline, kitti_params

pytestmark = pytest.mark.gpu
W, H = synth.KITTI_W, synth.KITTI_H


@pytest.mark.parametrize("defer,worker", [(0, False), (1, True)])
def test_track_then_full_batch_matches_the_oracle(oracle, defer, worker):
    import torch
    n_frames = 10
    Ts = SQ.camera_poses(n_frames)
    objs = SQ.default_objects()
    ctx, ctx_lm, ctx_obj = Context(0), Context(0), Context(0)
    ctx_w = Context(0) if worker else None
    pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1, defer_objects=defer), ctx_obj, ctx_w)
    pipe.attach_map()
    gt = []
    for k in range(n_frames):
        fr = SQ.render_frame(k, Ts, objs, flow_sigma=0.1)
        d = {q: torch.from_numpy(np.ascontiguousarray(fr[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")}
        torch.cuda.synchronize()
        pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
        gt.append(np.linalg.inv(fr["Tcw"]))
    pipe.flush()
    pipe.finalize_map()
    m = pipe.export_map(synth.KITTI_K)
    # the Map is what Track() saw: one entry per frame, tracklets long enough to become landmarks, object motions with labels
    assert m["n_frames"] == n_frames and len(m["rigid_motion"]) == n_frames - 1
    assert all(len(f["sta_uv"]) >= 1000 for f in m["feats"]) and all(len(f["dyn_uv"]) > 500 for f in m["feats"])
    assert sum(len(t) >= 3 for t in m["tr_sta"]) > 1000 and sum(len(t) >= 3 for t in m["tr_dyn"]) > 300
    assert all(l[0] == 0 for l in m["rm_label"]) and max(len(l) for l in m["rm_label"]) >= 3
    np.testing.assert_allclose(m["cam_pose"][-1], np.linalg.inv(pipe.pose().astype(np.float64)), atol=1e-5)
    # GPU: C++ builder + solver
    st = pipe.full_batch(synth.KITTI_K)
    m_rf = pipe.export_map(synth.KITTI_K, refined=True)
    # the same optimisation built straight from the pipeline's flat GraphStore (no Map, no per-point cv::Mat): the same graph
    # (two solves of one graph agree to the rounding of the pose-pose atomics, not bit for bit)
    st2 = pipe.full_batch_store()
    assert (st2.iterations, st2.total_trials) == (st.iterations, st.total_trials) and abs(st2.final_chi2 - st.final_chi2) <= 1e-9 * st.final_chi2
    np.testing.assert_allclose(pipe.store_poses(refined=True), m_rf["cam_pose"], rtol=0, atol=1e-6)
    assert np.array_equal(pipe.store_poses(), m["cam_pose"])
    # oracle: Python restatement of the builder + the oracle's LM on the same Map
    g, info = SM.map_to_graph(m)
    gc, keep = K.graph_to_c(g)
    opt = K.LMOptionsC(300, 1e-4, 0, 0, 0.0, 0)
    st_o = K.LMStatsC()
    pose_o = np.zeros_like(g.pose); point_o = np.zeros_like(g.point)
    assert oracle.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), K._dp(pose_o), K._dp(point_o), C.byref(st_o)) == 0
    assert g.n_eb > 10000 and g.n_et > 1000
    assert st.iterations == st_o.iterations and st.total_trials == st_o.total_trials and st.iterations >= 1
    assert abs(st.final_chi2 - st_o.final_chi2) <= 1e-6 * st_o.final_chi2 and st.final_chi2 < st.initial_chi2
    for i in range(1, n_frames):
        ref = pose_o[info["cam_idx"][i]]
        np.testing.assert_allclose(m_rf["cam_pose"][i][:3, :3].ravel(), ref[:9], atol=2e-6)
        np.testing.assert_allclose(m_rf["cam_pose"][i][:3, 3], ref[9:], rtol=1e-4, atol=1e-5)
    # the batch optimisation keeps the trajectory at the centimetre level of the tracker
    err_before = max(np.abs(m["cam_pose"][i][:3, 3] - gt[i][:3, 3]).max() for i in range(n_frames))
    err_after = max(np.abs(m_rf["cam_pose"][i][:3, 3] - gt[i][:3, 3]).max() for i in range(n_frames))
    assert err_before < 0.05 and err_after < 0.05
    pipe.close()


def test_partial_batch_optimization_runs_on_the_reference_schedule():
    """WINDOW_SIZE 8 / OVERLAP_SIZE 2: the windowed optimisation fires at f_id 7 and 13 (src/Tracking.cc:1169), refines the
    Map's window in place and leaves the tracker's own state alone."""
    import torch
    n_frames = 14
    Ts = SQ.camera_poses(n_frames)
    objs = SQ.default_objects()

    def run(window):
        ctx, ctx_lm, ctx_obj = Context(0), Context(0), Context(0)
        pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1,
                                                       window_size=window, overlap_size=2 if window else 0), ctx_obj)
        pipe.attach_map()
        fired, poses = [], []
        for k in range(n_frames):
            fr = SQ.render_frame(k, Ts, objs, flow_sigma=0.1)
            d = {q: torch.from_numpy(np.ascontiguousarray(fr[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")}
            torch.cuda.synchronize()
            pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
            fired.append(pipe.partial_batches()); poses.append(pipe.pose().copy())
        pipe.finalize_map()
        m = pipe.export_map(synth.KITTI_K)
        pipe.close()
        return fired, poses, m

    fired, poses, m = run(8)
    fired0, poses0, m0 = run(0)
    assert [b - a for a, b in zip([0] + fired[:-1], fired)] == [1 if k in (7, 13) else 0 for k in range(n_frames)]
    assert fired0[-1] == 0
    for a, b in zip(poses, poses0):
        assert np.array_equal(a, b)                       # Track() itself does not read the refined Map
    moved = max(np.abs(m["cam_pose"][i] - m0["cam_pose"][i]).max() for i in range(n_frames))
    assert 0 < moved < 0.05                              # the window's poses were refined (a little)
    err = max(np.abs(m["cam_pose"][i][:3, 3] - Ts[i][:3, 3]).max() for i in range(n_frames))
    assert err < 0.06
    assert len(m["tr_sta"]) == len(m0["tr_sta"]) and len(m["tr_dyn"]) == len(m0["tr_dyn"])
