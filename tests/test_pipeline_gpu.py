"""The C++ FramePipeline (per-frame sequence of GrabImageRGBD + Track over the C-ABI, vdo_slam_amd/host/)
against the same sequence composed from the oracle's functions (tests/pipeline_ref.py): chained over several
frames, every count of every frame must agree — any single differing keypoint, label, depth or selection
anywhere upstream changes them."""
import numpy as np
import pytest

from tests.pipeline_ref import OraclePipeline
from vdo_slam_amd import synth, synth_frames as SF
from vdo_slam_amd.ba import Context
from vdo_slam_amd.flow2 import Flow2Batch
from vdo_slam_amd.pipeline import FramePipeline, kitti_params

pytestmark = pytest.mark.gpu
W, H = synth.KITTI_W, synth.KITTI_H
KEYS = ("n_orb", "n_static_new", "n_object_samples", "n_static_tracked", "n_object_tracked", "n_objects", "n_recovered_masks",
        "n_static_tracks", "n_dynamic_tracks", "n_ransac_cam", "n_motion_model_cam", "n_ransac_obj", "n_mm_inliers_obj", "n_motion_model_obj")


def _dev(fr):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(fr[k])).cuda() for k in ("gray", "depth_raw", "flow", "mask")}


def test_pipeline_counts_match_the_oracle_sequence(oracle):
    import torch
    ctx, ctx_lm = Context(0), Context(0)
    pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ))
    ref = OraclePipeline(oracle)
    frames = [SF.make_frame(seed=40 + k) for k in range(3)]
    seq = [0, 1, 2, 1, 0, 2]
    for i, k in enumerate(seq):
        d = _dev(frames[k])
        torch.cuda.synchronize()
        got = pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())   # no LM: identity poses, all inliers
        exp = ref.step(frames[k])
        assert {q: got[q] for q in KEYS} == {q: exp[q] for q in KEYS}, (i, got, exp)
    assert got["n_static_tracked"] > 500 and got["n_object_tracked"] > 1000 and got["n_static_tracks"] > 1000
    ms = pipe.section_ms()
    assert all(v >= 0 for v in ms.values()) and ms["orb"] > 0
    pipe.close()


def test_pipeline_with_pose_problems_on_the_second_stream():
    """With the frame's LM problems attached (camera on its own stream while ORB runs, objects while RenewFrameInfo
    runs) the sequence completes and the LM results are those of the stand-alone batches."""
    import torch
    ctx, ctx_lm = Context(0), Context(0)
    pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ))
    cam = Flow2Batch(ctx_lm, [synth.make_flow2_problem(1200, seed=4)])
    objs = [synth.make_flow2_problem(n, seed=30 + j, is_object=True) for j, n in enumerate((600, 300))]
    ob = Flow2Batch(ctx_lm, objs)
    frames = [SF.make_frame(seed=50 + k) for k in range(2)]
    for k in (0, 1, 0, 1):
        d = _dev(frames[k])
        torch.cuda.synchronize()
        c = pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr(), cam, ob, 1200, 2)
    assert c["n_orb"] > 2000 and c["n_static_tracked"] > 0
    r = cam.fetch()[0]
    alone = Flow2Batch(ctx, [synth.make_flow2_problem(1200, seed=4)]); alone.run(); ra = alone.fetch()[0]
    assert r["iterations"] == ra["iterations"] and np.array_equal(r["T"], ra["T"]) and np.array_equal(r["inliers"], ra["inliers"])
    pipe.close()
