"""UseSampleFeature = 1 (example/omd.yaml): Frame::SampleKeyPoints instead of ORB, the sampled branch of the static filter,
top-up from the filtered samples - product vs oracle."""
import ctypes as C

import numpy as np
import pytest

from tests import frontend_ref as R
from vdo_slam_amd import _capi as K
from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ

OMD_W, OMD_H = 640, 480
OMD_K = (618.3587, 618.5786, 328.9866, 237.7507)          # example/omd.yaml


def _sample_product(rows, cols, seed):
    L = K.lib()
    L.vdo_sample_keypoints.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, K.c_float_p, K.c_float_p, C.POINTER(C.c_int)]
    x = np.zeros(3008, np.float32); y = np.zeros(3008, np.float32); n = C.c_int()
    K.check(L.vdo_sample_keypoints(rows, cols, seed, 3008, R._fp(x), R._fp(y), C.byref(n)))
    return x[:n.value], y[:n.value]


@pytest.mark.parametrize("rows,cols,seed", [(OMD_H, OMD_W, 1), (375, 1242, 1600000000), (480, 640, 0)])
def test_sample_keypoints_matches_the_oracle(oracle, rows, cols, seed):
    """Host-only entry point (no GPU needed): same cv::RNG stream, same grid order."""
    x, y = _sample_product(rows, cols, seed)
    ox = np.zeros(3000, np.float32); oy = np.zeros(3000, np.float32)
    oracle.vdo_oracle_sample_keypoints.argtypes = [C.c_int, C.c_int, C.c_ulonglong, K.c_float_p, K.c_float_p]
    n = oracle.vdo_oracle_sample_keypoints(rows, cols, seed, R._fp(ox), R._fp(oy))
    assert n == x.size == 3000
    assert np.array_equal(x, ox) and np.array_equal(y, oy)
    assert x.min() > 0 and y.min() > 0 and x.max() < cols and y.max() < rows
    cell = (x // (cols // 20)).astype(int) * 20 + (y // (rows // 20)).astype(int)
    assert np.all(np.diff(cell) >= 0) and np.bincount(cell, minlength=400).min() >= 4        # grid-cell order, every cell populated (7-8 per cell; a 0 coordinate is rejected)


@pytest.mark.gpu
def test_static_filter_sampled_matches_the_oracle(oracle):
    from vdo_slam_amd.ba import Context
    from vdo_slam_amd.frontend import FrameImages
    ctx = Context(0)
    Ts = SQ.camera_poses(2)
    fr = SQ.render_frame(0, Ts, SQ.default_objects(), w=OMD_W, h=OMD_H, K4=OMD_K, flow_sigma=0.3, invalid_depth=0.02, zero_flow=0.01)
    depth = fr["depth_raw"].copy()
    oracle.vdo_oracle_depth_preprocess(R._fp(depth), depth.size, SF.BF, SF.DEPTH_MAP_FACTOR)
    x, y = _sample_product(OMD_H, OMD_W, 7)
    im = FrameImages(ctx, OMD_W, OMD_H)
    im.upload(depth, fr["flow"], fr["mask"])
    L = K.lib()
    L.vdo_frame_static_filter_sampled.argtypes = [C.c_void_p, C.c_int, K.c_float_p, K.c_float_p, C.c_float, K.c_int32_p] + [K.c_float_p] * 5 + [C.POINTER(C.c_int)]
    n = x.size
    idx = np.zeros(n, np.int32); f = [np.zeros(n, np.float32) for _ in range(5)]; m = C.c_int()
    K.check(L.vdo_frame_static_filter_sampled(im._h, n, R._fp(x), R._fp(y), SF.TH_DEPTH_BG, R._ip(idx), *[R._fp(a) for a in f], C.byref(m)))
    oidx = np.zeros(n, np.int32); of = [np.zeros(n, np.float32) for _ in range(5)]
    oracle.vdo_oracle_frame_static_filter_sampled.argtypes = [C.c_int, K.c_float_p, K.c_float_p, K.c_int32_p, K.c_float_p, K.c_float_p, C.c_int, C.c_int, C.c_float, K.c_int32_p] + [K.c_float_p] * 5
    om = oracle.vdo_oracle_frame_static_filter_sampled(n, R._fp(x), R._fp(y), R._ip(fr["mask"]), R._fp(depth), R._fp(fr["flow"]), OMD_W, OMD_H, SF.TH_DEPTH_BG, R._ip(oidx), *[R._fp(a) for a in of])
    assert m.value == om and 1000 < om < n
    assert np.array_equal(idx[:om], oidx[:om])
    for a, b in zip(f, of):
        assert np.array_equal(a[:om], b[:om])


@pytest.mark.gpu
def test_omd_shaped_track_with_sampled_features_matches_the_oracle(oracle):
    """640 x 480, OMD intrinsics, UseSampleFeature 1, SFMgThres 0.02: the full Track() of FramePipeline == oracle-composed Track()."""
    import torch
    from tests.pipeline_ref import OraclePipeline
    from vdo_slam_amd.ba import Context
    from vdo_slam_amd.pipeline import FramePipeline, kitti_params
    n_frames = 6
    Ts = SQ.camera_poses(n_frames, step=0.25)
    objs = [dict(c=np.array([-1.2, 0.6, 6.0]), hw=0.7, hh=0.5, v=np.array([0.0, 0.0, 0.33])),
            dict(c=np.array([1.5, 0.6, 8.0]), hw=0.8, hh=0.55, v=np.array([0.01, 0.0, 0.2]))]
    ctx, ctx_lm, ctx_obj, ctx_w = Context(0), Context(0), Context(0), Context(0)
    prm = kitti_params(OMD_W, OMD_H, OMD_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1, use_sample_feature=1, sample_seed=11, sf_mg_thres=0.02, n_features=3000)
    pipe = FramePipeline(ctx, ctx_lm, prm, ctx_obj, ctx_w)
    ref = OraclePipeline(oracle, build_lm=True, K4=OMD_K, use_sample=True, sample_seed=11, sf_mg=0.02)
    keys = ("n_orb", "n_static_new", "n_object_samples", "n_static_tracked", "n_object_tracked", "n_objects", "n_static_tracks", "n_dynamic_tracks",
            "n_ransac_cam", "n_motion_model_cam", "n_ransac_obj", "n_cam_inliers", "cam_lm_iterations", "n_mm_inliers_obj", "n_motion_model_obj")
    for k in range(n_frames):
        fr = SQ.render_frame(k, Ts, objs, w=OMD_W, h=OMD_H, K4=OMD_K, flow_sigma=0.05)
        d = {q: torch.from_numpy(np.ascontiguousarray(fr[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")}
        torch.cuda.synchronize()
        got = pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
        exp = ref.step(fr)
        assert {q: got[q] for q in keys} == {q: exp[q] for q in keys}, (k, got, exp)
        np.testing.assert_allclose(pipe.pose(), ref.Tl, rtol=0, atol=5e-6)
    assert got["n_orb"] == 3000 and got["n_static_tracked"] >= 1000 and got["n_objects"] >= 1
    gt = fr["Tcw"]
    assert np.abs(pipe.pose()[:3, 3] - gt[:3, 3]).max() < 0.03
    pipe.close()
