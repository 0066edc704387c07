"""Empty / degenerate inputs through the C-ABI on the GPU: every entry point must return cleanly
(the reference's behaviour on such inputs is "nothing happens" or identity / 0 inliers)."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

from tests import tracking_ref as T
from vdo_slam_amd import _capi as K
from vdo_slam_amd import pose_only as PO
from vdo_slam_amd import synth, synth_frames as SF
from vdo_slam_amd import tracking as TR
from vdo_slam_amd.ba import BatchBA, Context
from vdo_slam_amd.flow2 import Flow2Batch
from vdo_slam_amd.frontend import FrameImages, ORBextractor

pytestmark = pytest.mark.gpu
W, H = 1242, 375
E = np.zeros(0, np.float32)
EI = np.zeros(0, np.int32)


@pytest.fixture(scope="module")
def ctx():
    return Context(0)


def test_featureless_image_and_empty_filters(ctx, oracle):
    orb = ORBextractor(ctx, W, H)
    kp = orb(np.full((H, W), 90, np.uint8))
    assert kp["x"].size == 0
    im = FrameImages(ctx, W, H)
    fr = SF.make_frame(seed=3)
    depth = np.zeros((H, W), np.float32)                      # no valid depth anywhere
    im.upload(depth, fr["flow"], np.zeros((H, W), np.int32))
    assert im.static_filter(E, E, SF.TH_DEPTH_BG)["keep_idx"].size == 0
    rng = np.random.default_rng(0)
    kx = rng.uniform(0, W, 500).astype(np.float32); ky = rng.uniform(0, H, 500).astype(np.float32)
    assert im.static_filter(kx, ky, SF.TH_DEPTH_BG)["keep_idx"].size == 0     # depth <= 0 everywhere: all dropped
    assert im.object_sample(SF.TH_DEPTH_OBJ)["label"].size == 0               # mask all background


def test_tracking_entry_points_with_no_points(ctx, oracle):
    fr = SF.make_frame(seed=4)
    depth = (SF.BF / np.maximum(fr["depth_raw"] / SF.DEPTH_MAP_FACTOR, 1e-9)).astype(np.float32)
    im = FrameImages(ctx, W, H); im.upload(depth, fr["flow"], fr["mask"])
    last = FrameImages(ctx, W, H); last.upload(depth, fr["flow"], fr["mask"])
    assert TR.propagate_object(im, E, E, 25.0)[0].size == 0
    assert TR.mask_at(im, E, E).size == 0
    K4 = np.array(synth.KITTI_K, np.float32); I4 = np.eye(4, dtype=np.float32)
    assert TR.get3d_world(ctx, E, E, E, K4, I4).shape == (0, 3)
    fl, ol = TR.scene_flow(ctx, (E, E, E, EI), I4, (E, E, E, EI), I4, K4, EI)
    assert fl.shape == (0, 3) and ol.size == 0
    out = TR.renew_static(im, EI, E, E, E, E, 1200)
    assert out["key_x"].size == 0
    # no inliers but ORB keypoints to top up from: same as the oracle
    rng = np.random.default_rng(1)
    ox = rng.uniform(0, W, 800).astype(np.float32); oy = rng.uniform(0, H, 800).astype(np.float32)
    got = TR.renew_static(im, np.full(10, -1, np.int32), E, E, ox, oy, 300)
    exp = T.renew_static(oracle, np.full(10, -1, np.int32), E, E, ox, oy, fr["mask"], depth, fr["flow"], 300)
    for k in exp:
        assert np.array_equal(got[k], exp[k])
    tmp = dict(x=E, y=E, depth=E, label=EI, flow_x=E, flow_y=E, corr_x=E, corr_y=E)
    assert TR.renew_object(im, [], np.zeros(0, np.uint8), EI, EI, E, E, EI, tmp, 800)["key_x"].size == 0
    before = TR.download_mask(im)
    assert TR.update_mask(im, last, EI, E, E) == 0
    assert np.array_equal(TR.download_mask(im), before)
    prm = TR.DynObjParamsC(W, H, 25, 50, 0.12, 0.3, 25.0, 3)
    r = TR.dyn_obj_tracking(prm, EI, EI, E, E, E, np.zeros((0, 3), np.float32), EI, EI, EI, np.zeros(0, np.uint8), 5)
    assert len(r["objects"]) == 0 and r["max_id"] == 5


def test_lm_problems_below_three_correspondences(ctx):
    for n in (0, 1, 2):
        p = synth.make_flow2_problem(max(n, 1), seed=1)
        if n == 0:
            p = dataclasses.replace(p, obs=p.obs[:0], flow=p.flow[:0], depth=p.depth[:0])
        b = Flow2Batch(ctx, [p]); b.run(); r = b.fetch()[0]
        assert r["n_inliers"] == 0 and np.array_equal(r["T"], np.eye(4)) and not r["inliers"].any()
        assert np.array_equal(r["flow"], np.asarray(p.flow, np.float64).reshape(-1, 2))      # flows untouched
        q = PO.make_pose_problem(max(n, 1), seed=1)
        if n == 0:
            q = dataclasses.replace(q, obs=q.obs[:0], Xw=q.Xw[:0])
        pb = PO.PoseBatch(ctx, [q]); pb.run(); r = pb.fetch()[0]
        assert r["n_inliers"] == 0 and np.array_equal(r["T"], np.eye(4))


def test_pure_pose_graph_and_single_track(ctx, oracle):
    """Batch BA with no landmarks at all (only EdgeSE3 + prior), and with exactly one static point."""
    g = synth.make_ba_graph(10, 60, 1, 6, seed=4)
    z = np.zeros(0, np.int32)
    bare = dataclasses.replace(g, point=np.zeros((0, 3)), eb_pose=z, eb_point=z, eb_z=np.zeros((3, 0)), eb_w=np.zeros(0),
                               et_p1=z, et_p2=z, et_pose=z, et_z=np.zeros((3, 0)), et_w=np.zeros(0), point_gt=None)
    keep = g.eb_point == 0
    one = dataclasses.replace(bare, point=g.point[:1].copy(), eb_pose=g.eb_pose[keep], eb_point=g.eb_point[keep],
                              eb_z=np.ascontiguousarray(g.eb_z[:, keep]), eb_w=g.eb_w[keep])
    for graph in (bare, one):
        gc, ka = K.graph_to_c(graph)
        opt = K.LMOptionsC(10, -1.0, 0, 0, 0.0, 0)
        st_o = K.LMStatsC()
        pose_o = np.zeros_like(graph.pose); point_o = np.zeros((max(graph.n_point, 1), 3))
        assert oracle.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), K._dp(pose_o), K._dp(point_o), C.byref(st_o)) == 0
        ba = BatchBA(ctx, graph)
        st = ba.optimize(max_iterations=10, gain_threshold=-1.0)
        pose, point = ba.estimates()
        assert st.iterations == st_o.iterations
        assert abs(st.final_chi2 - st_o.final_chi2) <= 1e-6 * max(st_o.final_chi2, 1e-12)
        assert np.abs(pose - pose_o).max() <= 1e-4 * max(1.0, np.abs(pose_o).max())
        ba.close()
