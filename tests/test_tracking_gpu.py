"""GPU parity of the Tracking-side gathers (K11-K15, vdo_slam_amd/csrc/tracking.hip) through the
C-ABI against oracle/tracking_oracle.cpp: bit-exact, including the fp32 back-projections
(same cv::gemm double-accumulate rounding on both sides)."""
import numpy as np
import pytest

from tests import tracking_ref as T
from tests.test_oracle_tracking import _frame, _points, _rand_pose, _renew_inputs
from vdo_slam_amd import synth_frames as SF
from vdo_slam_amd import tracking as TR
from vdo_slam_amd.ba import Context
from vdo_slam_amd.frontend import FrameImages
from vdo_slam_amd.synth import KITTI_K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return Context(0)


def _images(ctx, depth, flow, mask):
    h, w = mask.shape
    im = FrameImages(ctx, w, h)
    im.upload(depth, flow, mask)
    return im


def test_propagate_and_mask_at(ctx, oracle):
    fr, depth = _frame(3)
    h, w = depth.shape
    im = _images(ctx, depth, fr["flow"], fr["mask"])
    rng = np.random.default_rng(0)
    for n in (1, 777, 20000):
        kx, ky = _points(rng, n, w, h)
        assert np.array_equal(TR.propagate_static(im, kx, ky), T.propagate_static(oracle, kx, ky, depth))
        d, lab = TR.propagate_object(im, kx, ky, SF.TH_DEPTH_OBJ)
        d_o, lab_o = T.propagate_object(oracle, kx, ky, depth, fr["mask"], SF.TH_DEPTH_OBJ)
        assert np.array_equal(d, d_o) and np.array_equal(lab, lab_o)
        assert np.array_equal(TR.mask_at(im, kx, ky), T.mask_at(oracle, kx, ky, fr["mask"]))
    assert TR.propagate_static(im, np.zeros(0, np.float32), np.zeros(0, np.float32)).size == 0


def test_backprojection_and_scene_flow_bit_exact(ctx, oracle):
    rng = np.random.default_rng(2)
    K4 = np.array(KITTI_K, np.float32)
    n = 12000
    u0, v0 = rng.uniform(0, 1242, n).astype(np.float32), rng.uniform(0, 375, n).astype(np.float32)
    u1, v1 = (u0 + rng.normal(0, 8, n)).astype(np.float32), (v0 + rng.normal(0, 3, n)).astype(np.float32)
    z0, z1 = rng.uniform(3, 60, n).astype(np.float32), rng.uniform(3, 60, n).astype(np.float32)
    Tcw0, Tcw1 = _rand_pose(rng, 3.0).astype(np.float32), _rand_pose(rng, 3.0).astype(np.float32)
    X = TR.get3d_world(ctx, u0, v0, z0, K4, Tcw0)
    assert np.array_equal(X, T.get3d_world(oracle, u0, v0, z0, K4, Tcw0))
    lab = rng.integers(-1, 4, n).astype(np.int32); lab_l = rng.integers(0, 4, n).astype(np.int32)
    ol0 = rng.integers(0, 6, n).astype(np.int32)
    fl, ol = TR.scene_flow(ctx, (u1, v1, z1, lab), Tcw1, (u0, v0, z0, lab_l), Tcw0, K4, ol0)
    fl_o, ol_o = T.scene_flow(oracle, (u1, v1, z1, lab), Tcw1, (u0, v0, z0, lab_l), Tcw0, K4, ol0)
    assert np.array_equal(ol, ol_o)
    assert np.array_equal(fl, fl_o)


@pytest.mark.parametrize("case", [(5, 300, 500, 0.7, 400), (6, 700, 300, 0.9, 200), (7, 50, 3000, 0.5, 600), (8, 0, 100, 0.5, 50),
                                  (11, 1500, 2500, 0.8, 1600), (12, 1500, 2500, 0.2, 1600)])
def test_renew_static(ctx, oracle, case):
    seed, n_stat, n_orb, frac, max_num = case
    fr, depth, tm, sx, sy, ox, oy = _renew_inputs(seed, n_stat, n_orb, frac)
    im = _images(ctx, depth, fr["flow"], fr["mask"])
    got = TR.renew_static(im, tm, sx, sy, ox, oy, max_num)
    exp = T.renew_static(oracle, tm, sx, sy, ox, oy, fr["mask"], depth, fr["flow"], max_num)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k


def test_mask_warp(ctx, oracle):
    fr, depth = _frame(9)
    cur = SF.make_frame(seed=10)
    last_im = _images(ctx, depth, fr["flow"], fr["mask"])
    cur_im = _images(ctx, depth, cur["flow"], cur["mask"])
    exp = cur["mask"]
    for lab in (1, 3, 4):
        TR.mask_warp(cur_im, last_im, lab)
        exp = T.mask_warp(oracle, fr["mask"], fr["flow"], lab, exp)
    assert np.array_equal(TR.download_mask(cur_im), exp)
