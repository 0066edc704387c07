"""GPU parity of the Tracking-side gathers (K11-K15, vdo_slam_amd/csrc/tracking.hip) through the
C-ABI against oracle/tracking_oracle.cpp: bit-exact, including the fp32 back-projections
(same cv::gemm rounding rules on both sides: float fast path for untransposed 3-wide products, double accumulation for transposed ones)."""
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
    # the one-pass form with the 3-D points: same set, xyz = K12 of its keys (bit for bit)
    K4 = np.array(KITTI_K, np.float32)
    Twc = np.eye(4, dtype=np.float32); Twc[:3, 3] = (0.3, -0.1, 2.0); Twc[0, 1], Twc[1, 0] = -0.02, 0.02
    got3 = TR.renew_static(im, tm, sx, sy, ox, oy, max_num, world=(K4, Twc))
    for k in exp:
        assert np.array_equal(got3[k], exp[k]), k
    assert np.array_equal(got3["xyz"], TR.get3d_world(ctx, got3["key_x"], got3["key_y"], got3["depth"], K4, Twc))


def _object_case(oracle, seed):
    from tests import frontend_ref as R
    fr, depth = _frame(seed)
    rng = np.random.default_rng(seed)
    ob = R.object_sample(oracle, fr["mask"], depth, fr["flow"], SF.TH_DEPTH_OBJ)          # K10 sampling of the new image
    tmp = dict(x=ob["key_x"], y=ob["key_y"], depth=ob["depth"], label=ob["label"], flow_x=ob["flow_x"], flow_y=ob["flow_y"],
               corr_x=ob["corr_x"], corr_y=ob["corr_y"])
    # current object points: pixels on the masks (plus strays), sub-pixel positions as they come out of the flow propagation
    ys, xs = np.nonzero(fr["mask"])
    pick = rng.permutation(ys.size)[:3000]
    cx = (xs[pick] + rng.uniform(-0.4, 1.4, pick.size)).astype(np.float32)
    cy = (ys[pick] + rng.uniform(-0.4, 1.4, pick.size)).astype(np.float32)
    cx[:40] = rng.uniform(-5, 1250, 40).astype(np.float32); cy[:40] = rng.uniform(-5, 380, 40).astype(np.float32)
    lab_at = fr["mask"][np.clip(cy.astype(int), 0, 374), np.clip(cx.astype(int), 0, 1241)]
    col = rng.integers(1, 9, cx.size).astype(np.int32)
    labels = [l for l in np.unique(ob["label"]) if l > 0]
    tracked = labels[:3] if len(labels) >= 3 else labels
    inl = [np.nonzero((lab_at == l) & (rng.random(cx.size) < 0.8))[0].astype(np.int32) for l in tracked]
    stat = np.ones(len(tracked), np.uint8)
    if len(tracked) > 1:
        stat[1] = 0
    return fr, depth, tmp, cx, cy, col, inl, stat, np.array(tracked, np.int32), np.arange(11, 11 + len(tracked)).astype(np.int32)


@pytest.mark.parametrize("seed,max_num", [(21, 800), (22, 60), (23, 100000)])
def test_renew_object(ctx, oracle, seed, max_num):
    fr, depth, tmp, cx, cy, col, inl, stat, sem_pos, mod = _object_case(oracle, seed)
    im = _images(ctx, depth, fr["flow"], fr["mask"])
    got = TR.renew_object(im, inl, stat, sem_pos, mod, cx, cy, col, tmp, max_num)
    exp = T.renew_object(oracle, inl, stat, sem_pos, mod, cx, cy, col, tmp, fr["mask"], depth, fr["flow"], max_num)
    assert exp["key_x"].size > 200
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    assert (exp["obj_label"] == -2).sum() > 0 and (exp["inlier_id"] >= 0).sum() > 0
    if max_num >= 800:                       # with the small cap the carried points already fill the objects: no top-up
        assert ((exp["inlier_id"] == -1) & (exp["obj_label"] > 0)).sum() > 0
    K4 = np.array(KITTI_K, np.float32)
    Twc = np.eye(4, dtype=np.float32); Twc[:3, 3] = (0.3, -0.1, 2.0); Twc[0, 1], Twc[1, 0] = -0.02, 0.02
    got3 = TR.renew_object(im, inl, stat, sem_pos, mod, cx, cy, col, tmp, max_num, world=(K4, Twc))
    for k in exp:
        assert np.array_equal(got3[k], exp[k]), k
    assert np.array_equal(got3["xyz"], TR.get3d_world(ctx, got3["key_x"], got3["key_y"], got3["depth"], K4, Twc))


def test_update_mask_recovers_a_dropped_mask(ctx, oracle):
    """The current mask misses one object (Mask-RCNN dropout): its last-frame points land on background,
    the vote says 0, the previous mask is warped in; labels that are still there are left alone."""
    from tests import frontend_ref as R
    fr, depth = _frame(31)
    rng = np.random.default_rng(31)
    ob = R.object_sample(oracle, fr["mask"], depth, fr["flow"], SF.TH_DEPTH_OBJ)
    labels = np.unique(ob["label"])
    assert labels.size >= 3
    cur_mask = fr["mask"].copy()
    # "current" segmentation: every object shifted by its mean flow, one of them missing, one too small to vote
    cur_mask[:] = 0
    for l in labels:
        ys, xs = np.nonzero(fr["mask"] == l)
        if l == labels[0]:
            continue                                                        # dropped mask
        fx = int(np.median(fr["flow"][ys, xs, 0])); fy = int(np.median(fr["flow"][ys, xs, 1]))
        ok = (xs + fx > 0) & (xs + fx < 1242) & (ys + fy > 0) & (ys + fy < 375)
        cur_mask[(ys + fy)[ok], (xs + fx)[ok]] = l
    keep = np.ones(ob["label"].size, bool)
    small = np.nonzero(ob["label"] == labels[1])[0]
    keep[small[60:]] = False                                                # fewer than 100 votes: skipped whatever they say
    sl, cx, cy = ob["label"][keep], ob["corr_x"][keep], ob["corr_y"][keep]
    last_im = _images(ctx, depth, fr["flow"], fr["mask"])
    cur_im = _images(ctx, depth, fr["flow"], cur_mask)
    rec = TR.update_mask(cur_im, last_im, sl, cx, cy)
    exp, rec_o = T.update_mask(oracle, sl, cx, cy, fr["mask"], fr["flow"], cur_mask)
    assert rec == rec_o and rec >= 1
    got = TR.download_mask(cur_im)
    assert np.array_equal(got, exp)
    assert (got == labels[0]).sum() > 500 and (cur_mask == labels[0]).sum() == 0


@pytest.mark.parametrize("n_labels", [4, 70])
def test_update_mask_order_dependence_matches_the_sequential_reference(ctx, oracle, n_labels):
    """The warp of an earlier (smaller) label is visible to the votes of the later ones.  Constructed so that it matters both
    ways: label 2's samples land on background that label 1's warp covers (2 would be recovered, is not), and label 3's samples
    land on label 9 pixels that label 1's warp partly overwrites so that background becomes the most frequent label (3 would
    not be recovered, is).  Result == the oracle's label-after-label passes; 70 labels take the label-after-label launches
    (more than the 64 one-pass slots), 4 the three-launch path."""
    h, w = 120, 400
    rng = np.random.default_rng(3)
    last = np.zeros((h, w), np.int32); cur = np.zeros((h, w), np.int32)
    flow = np.zeros((h, w, 2), np.float32)
    depth = np.full((h, w), 10.0, np.float32)
    # label 1: a block whose warp (flow +40 px in x) lands on columns 100..139; dropped in the current mask
    last[20:80, 60:100] = 1; flow[20:80, 60:100, 0] = 40.0
    # label 2: samples on columns ~105..134 (inside 1's warp), currently background there
    last[20:80, 200:230] = 2; flow[20:80, 200:230, 0] = -95.3
    # label 3: samples on columns ~130..159: 130..147 carry label 9 in the current mask (18 of 30 columns: 9 wins), 148..159 are
    # background; label 1's warp overwrites 130..139, leaving 1: 10, 9: 8, 0: 12 columns - background wins afterwards
    last[20:80, 300:330] = 3; flow[20:80, 300:330, 0] = -170.2
    cur[20:80, 130:148] = 9
    cur[20:80, 163:190] = 9                      # (label 9 also lives elsewhere)
    sl, cx, cy = [], [], []
    for lab in (1, 2, 3):
        ys, xs = np.nonzero(last == lab)
        pick = rng.choice(ys.size, 400, replace=False)
        sl += [lab] * 400; cx += list(xs[pick] + flow[ys[pick], xs[pick], 0]); cy += list(ys[pick].astype(np.float32))
    for k in range(n_labels - 3):                # extra labels with too few samples to vote (and, at 70, the fallback path)
        lab = 20 + k
        last[100 + (k % 10), 5 + 5 * (k // 10):8 + 5 * (k // 10)] = lab
        sl += [lab] * 3; cx += [10.0, 11.0, 12.0]; cy += [100.0, 100.0, 100.0]
    sl = np.array(sl, np.int32); cx = np.array(cx, np.float32); cy = np.array(cy, np.float32)
    order = rng.permutation(sl.size)             # samples arrive unsorted
    sl, cx, cy = sl[order], cx[order], cy[order]
    last_im = _images(ctx, depth, flow, last)
    cur_im = _images(ctx, depth, flow, cur)
    rec = TR.update_mask(cur_im, last_im, sl, cx, cy)
    exp, rec_o = T.update_mask(oracle, sl, cx, cy, last, flow, cur)
    got = TR.download_mask(cur_im)
    assert rec == rec_o and np.array_equal(got, exp)
    # the construction did what it says: 1 recovered, 2 not (although its samples see background before 1's warp), 3 recovered
    alone2, r2 = T.update_mask(oracle, sl[sl == 2], cx[sl == 2], cy[sl == 2], last, flow, cur)
    alone3, r3 = T.update_mask(oracle, sl[sl == 3], cx[sl == 3], cy[sl == 3], last, flow, cur)
    assert r2 == 1 and r3 == 0 and rec == 2 and (got == 1).sum() > 1000 and (got == 3).sum() > 500 and (got == 2).sum() == 0
    # a second call on the same images works on a clean candidate image
    cur_im2 = _images(ctx, depth, flow, cur)
    assert TR.update_mask(cur_im2, last_im, sl, cx, cy) == rec and np.array_equal(TR.download_mask(cur_im2), exp)
    assert TR.update_mask(cur_im2, last_im, sl, cx, cy) >= 0                 # (idempotence is not claimed; it must simply run)


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


def test_object_chain_equals_the_three_separate_calls(ctx, oracle):
    """vdo_object_chain = vdo_update_mask + vdo_propagate_object + vdo_scene_flow (one upload / download / sync): same mask in
    HBM, same depths, labels, scene flow and object labels, bit for bit - on the dropped-mask scenario."""
    from tests import frontend_ref as R
    fr, depth = _frame(31)
    ob = R.object_sample(oracle, fr["mask"], depth, fr["flow"], SF.TH_DEPTH_OBJ)
    labels = np.unique(ob["label"])
    cur_mask = fr["mask"].copy()
    cur_mask[:] = 0
    for l in labels[1:]:                                                    # labels[0]: dropped mask
        ys, xs = np.nonzero(fr["mask"] == l)
        fx = int(np.median(fr["flow"][ys, xs, 0])); fy = int(np.median(fr["flow"][ys, xs, 1]))
        ok = (xs + fx > 0) & (xs + fx < 1242) & (ys + fy > 0) & (ys + fy < 375)
        cur_mask[(ys + fy)[ok], (xs + fx)[ok]] = l
    sl, cx, cy = ob["label"], ob["corr_x"], ob["corr_y"]
    rng = np.random.default_rng(5)
    Tl = np.eye(4, dtype=np.float32)
    Tc = np.eye(4, dtype=np.float32); Tc[:3, 3] = [0.02, -0.01, -0.8]
    K4 = np.array(KITTI_K, np.float32)
    # separate calls
    last_im = _images(ctx, depth, fr["flow"], fr["mask"])
    cur_a = _images(ctx, depth, fr["flow"], cur_mask)
    rec_a = TR.update_mask(cur_a, last_im, sl, cx, cy)
    d_a, sem_a = TR.propagate_object(cur_a, cx, cy, SF.TH_DEPTH_OBJ)
    fl_a, ol_a = TR.scene_flow(ctx, (cx, cy, d_a, sem_a), Tc, (ob["key_x"], ob["key_y"], ob["depth"], sl), Tl, K4, np.full(sl.size, -2, np.int32))
    # one call
    cur_b = _images(ctx, depth, fr["flow"], cur_mask)
    rec_b, d_b, sem_b, fl_b, ol_b = TR.object_chain(cur_b, last_im, sl, cx, cy, SF.TH_DEPTH_OBJ, Tc, ob["key_x"], ob["key_y"], ob["depth"], Tl, K4)
    assert rec_a == rec_b >= 1
    assert np.array_equal(TR.download_mask(cur_a), TR.download_mask(cur_b))
    assert np.array_equal(d_a, d_b) and np.array_equal(sem_a, sem_b) and np.array_equal(fl_a, fl_b) and np.array_equal(ol_a, ol_b)
    assert (ol_b == -1).sum() > 0 and (sem_b == labels[0]).sum() > 100
    # round 5: the last frame's inputs staged a frame ahead (vdo_object_chain_prestage) - same results; and a stage that is NOT what the chain is then called
    # with (other correspondences) is ignored, not used
    for stale in (False, True):
        cur_c = _images(ctx, depth, fr["flow"], cur_mask)
        TR.object_chain_prestage(ctx, sl, (cx + 3.0) if stale else cx, cy, ob["key_x"], ob["key_y"], ob["depth"])
        rec_c, d_c, sem_c, fl_c, ol_c = TR.object_chain(cur_c, last_im, sl, cx, cy, SF.TH_DEPTH_OBJ, Tc, ob["key_x"], ob["key_y"], ob["depth"], Tl, K4)
        assert rec_c == rec_b and np.array_equal(TR.download_mask(cur_c), TR.download_mask(cur_b)), stale
        assert np.array_equal(d_c, d_b) and np.array_equal(sem_c, sem_b) and np.array_equal(fl_c, fl_b) and np.array_equal(ol_c, ol_b), stale
