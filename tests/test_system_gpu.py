"""System::TrackRGBD (vdo_slam_amd/host/System.{h,cc}: the reference's entry API as a shell around FramePipeline) on host
images: settings file, colour conversion, in-place depth conversion, ground-truth gate of the object tracker, Map, the final
FullBatchOptimization."""
import ctypes as C
import os

import numpy as np
import pytest

from vdo_slam_amd import _capi as K
from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ
from vdo_slam_amd.ba import Context
from vdo_slam_amd.pipeline import FramePipeline, kitti_params

pytestmark = pytest.mark.gpu
W, H = synth.KITTI_W, synth.KITTI_H

YAML = """%YAML:1.0
Camera.fx: {fx}
Camera.fy: {fy}
Camera.cx: {cx}
Camera.cy: {cy}
Camera.k1: 0.0
Camera.width: {w}
Camera.height: {h}
Camera.fps: 10.0
Camera.bf: {bf}
Camera.RGB: 1
ChooseData: 2
DepthMapFactor: {dmf}
ThDepthBG: {thbg}
ThDepthOBJ: {thobj}
MaxTrackPointBG: 1200 # 1200 1500 2000
MaxTrackPointOBJ: 800
SFMgThres: 0.12 # 0.05
SFDsThres: 0.3
WINDOW_SIZE: 20
OVERLAP_SIZE: 4
UseSampleFeature: 0
ORBextractor.nFeatures: 2500
ORBextractor.scaleFactor: 1.2
ORBextractor.nLevels: 8
ORBextractor.iniThFAST: 20
ORBextractor.minThFAST: 7
"""


@pytest.fixture(scope="module")
def host():
    L = K.load_host_lib()
    L.host_system_create.restype = C.c_void_p
    L.host_system_create.argtypes = [C.c_char_p]
    L.host_system_destroy.argtypes = [C.c_void_p]
    L.host_system_track.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.host_system_motions.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.host_system_refined_poses.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.host_system_save.argtypes = [C.c_void_p, C.c_char_p]
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_system_trackrgbd_on_host_images(host, oracle, tmp_path):
    import torch
    from tests import frontend_ref as R
    n_frames = 7
    fx, fy, cx, cy = synth.KITTI_K
    cfg = tmp_path / "kitti.yaml"
    cfg.write_text(YAML.format(fx=fx, fy=fy, cx=cx, cy=cy, w=W, h=H, bf=SF.BF, dmf=SF.DEPTH_MAP_FACTOR, thbg=SF.TH_DEPTH_BG, thobj=SF.TH_DEPTH_OBJ))
    Ts = SQ.camera_poses(n_frames)
    objs = SQ.default_objects()
    frames = [SQ.render_frame(k, Ts, objs, flow_sigma=0.1) for k in range(n_frames)]

    def run_system(gt_labels):
        sys_ = host.host_system_create(str(cfg).encode())
        assert sys_
        poses, motions = [], []
        for k, fr in enumerate(frames):
            rgb = np.ascontiguousarray(np.repeat(fr["gray"][:, :, None], 3, axis=2))          # 3 equal channels: cvtColor returns the channel
            depth = fr["depth_raw"].copy(); mask = fr["mask"].copy()
            rows = np.array([[k, lab, 0, 0, 0, 0, 0, 0, 0, 0] for lab in gt_labels], np.float32).reshape(-1, 10)
            T = np.zeros(16, np.float32)
            assert host.host_system_track(sys_, _ptr(rgb), 3, _ptr(depth), _ptr(fr["flow"]), _ptr(mask), W, H, _ptr(rows) if len(rows) else None, len(rows), 10, n_frames, _ptr(T)) == 0
            if k == 0:                                                                          # the caller's depth map is converted in place (K1)
                ref = fr["depth_raw"].copy()
                oracle.vdo_oracle_depth_preprocess(R._fp(ref), ref.size, SF.BF, SF.DEPTH_MAP_FACTOR)
                assert np.array_equal(depth, ref)
            sl = np.zeros(16, np.int32); Hm = np.zeros((16, 16), np.float32)
            nm = host.host_system_motions(sys_, 16, _ptr(sl), _ptr(Hm))
            poses.append(T.reshape(4, 4).copy()); motions.append(sorted(int(x) for x in sl[:nm]))
        rf = np.zeros((n_frames, 16), np.float32)
        assert host.host_system_refined_poses(sys_, n_frames, _ptr(rf)) == n_frames             # FullBatchOptimization ran at the last frame
        prefix = str(tmp_path / "res_")                                                          # SaveResults takes a path PREFIX, as in the reference (src/System.cc:75-78)
        host.host_system_save(sys_, prefix.encode())
        ini = np.loadtxt(prefix + "initial_stereo_new.txt"); ref_ = np.loadtxt(prefix + "refined_stereo_new.txt"); gt_ = np.loadtxt(prefix + "cam_pose_gt_stereo.txt")
        assert ini.shape == ref_.shape == gt_.shape == (n_frames, 17) and np.array_equal(ini[:, 0], np.arange(n_frames))
        np.testing.assert_allclose(ini[-1, 1:13].reshape(3, 4), np.linalg.inv(T.reshape(4, 4).astype(np.float64))[:3], atol=2e-5)      # T_wc of the last frame, 9 digits
        np.testing.assert_allclose(ref_[:, 1:], rf.reshape(n_frames, 16), atol=1e-6)
        np.testing.assert_allclose(gt_[:, 1:], np.tile(np.eye(4).ravel(), (n_frames, 1)), atol=0)    # (this test hands identity ground-truth poses in)
        mot = np.loadtxt(prefix + "obj_mot_world_new.txt").reshape(-1, 18)
        assert mot.shape[0] >= 1 and set(mot[:, 1].astype(int)) <= {1, 2, 3, 4, 5, 6, 7, 8}
        assert os.path.exists(prefix + "obj_mot_world_rf_new.txt")
        host.host_system_destroy(sys_)
        return poses, motions, rf.reshape(-1, 4, 4)

    poses, motions, rf = run_system([1, 2, 3])
    # the same sequence through FramePipeline with device inputs
    ctx, ctx_lm, ctx_obj = Context(0), Context(0), Context(0)
    pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1), ctx_obj)
    for k, fr in enumerate(frames):
        d = {q: torch.from_numpy(np.ascontiguousarray(fr[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")}
        torch.cuda.synchronize()
        pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
        assert np.array_equal(pipe.pose(), poses[k]), k
        assert sorted(m["sem_label"] for m in pipe.motions()) == motions[k] or k == 0
    pipe.close()
    assert motions[-1] and set(motions[-1]) <= {1, 2, 3}
    gt_last = np.linalg.inv(frames[-1]["Tcw"])
    assert np.abs(rf[-1][:3, 3] - gt_last[:3, 3]).max() < 0.05
    # no ground-truth row for object 2: it is not tracked (src/Tracking.cc:791-841), the others are
    poses2, motions2, _ = run_system([1, 3])
    assert all(2 not in m for m in motions2) and any(m for m in motions2)
    assert all(np.array_equal(a, b) for a, b in zip(poses, poses2))        # the camera pose does not depend on the object gate


def test_system_rejects_frames_that_do_not_match_the_settings(tmp_path):
    """Camera.width/height size the device images; a frame of another size, element type or with padded rows is refused
    (empty pose, message) instead of being read out of bounds; OMD settings (ChooseData 1) run no global batch at the end."""
    from vdo_slam_amd.system import System, write_settings
    cfg = write_settings(tmp_path / "k.yaml", W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, choose_data=1)
    n_frames = 4
    Ts = SQ.camera_poses(n_frames)
    frames = [SQ.render_frame(k, Ts, SQ.default_objects(), flow_sigma=0.1) for k in range(n_frames)]
    s = System(cfg)
    fr = frames[0]
    small = {q: np.ascontiguousarray(fr[q][:300, :1000]) for q in ("gray", "depth_raw", "flow", "mask")}
    assert s.track_rgbd(small["gray"], small["depth_raw"].copy(), small["flow"], small["mask"].copy()) is None
    for k, f in enumerate(frames):
        d = f["depth_raw"].copy()
        T = s.track_rgbd(f["gray"], d, f["flow"], f["mask"].copy(), n_images=n_frames)
        assert T is not None and np.isfinite(T).all()
        assert np.isfinite(d).mean() > 0.9 and d[np.isfinite(d)].max() < 1e4 and not np.array_equal(d, f["depth_raw"])         # converted in place (metres; bf/0 = inf where the disparity is 0)
    assert np.abs(T[:3, 3] - frames[-1]["Tcw"][:3, 3]).max() < 0.05
    rf = s.refined_poses(n_frames)
    # OMD: FullBatchOptimization does not run (src/Tracking.cc:1198) - the refined poses are the unrefined ones
    assert rf.shape[0] == n_frames
    s.close()
    with pytest.raises(K.VdoError):
        System(tmp_path / "missing.yaml")


def test_tracking_frame_state_matches_the_oracle(host, oracle, tmp_path):
    """Tracking's public per-frame containers (reference include/Tracking.h:116-198, include/Frame.h:126-180), materialised on request by
    Tracking::SyncFrameState(): after every TrackRGBD call mCurrentFrame holds what the reference's Track() leaves there - the renewed
    static and object sets of RenewFrameInfo, their 3-D points, the per-object vectors, max_id - and they are the oracle pipeline's."""
    from tests.pipeline_ref import OraclePipeline
    host.host_system_frame_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    n_frames = 4
    fx, fy, cx, cy = synth.KITTI_K
    cfg = tmp_path / "kitti.yaml"
    cfg.write_text(YAML.format(fx=fx, fy=fy, cx=cx, cy=cy, w=W, h=H, bf=SF.BF, dmf=SF.DEPTH_MAP_FACTOR, thbg=SF.TH_DEPTH_BG, thobj=SF.TH_DEPTH_OBJ).replace("WINDOW_SIZE: 20", "WINDOW_SIZE: 0"))
    Ts = SQ.camera_poses(n_frames)
    objs = SQ.default_objects()
    sys_ = host.host_system_create(str(cfg).encode())
    assert sys_
    ref = OraclePipeline(oracle, build_lm=True)

    def state(what, rows):
        n = host.host_system_frame_state(sys_, what, None, 0)
        assert n >= 0
        buf = np.zeros(max(rows * n, 1), np.float32)
        assert host.host_system_frame_state(sys_, what, _ptr(buf), buf.size) == n
        return n, buf[:rows * n]

    for k in range(n_frames):
        fr = SQ.render_frame(k, Ts, objs)
        depth = fr["depth_raw"].copy(); mask = fr["mask"].copy()
        rows = np.array([[k, lab, 0, 0, 0, 0, 0, 0, 0, 0] for lab in (1, 2, 3)], np.float32)
        T = np.zeros(16, np.float32)
        assert host.host_system_track(sys_, _ptr(fr["gray"]), 1, _ptr(depth), _ptr(fr["flow"]), _ptr(mask), W, H, _ptr(rows), 3, 10, 1 << 30, _ptr(T)) == 0
        exp = ref.step(fr)
        L = ref.last
        n, s = state(0, 10)
        st = s.reshape(10, n)
        assert n == L["st"]["key_x"].size == exp["n_static_tracked"]
        for row, q in enumerate(("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y", "depth")):
            assert np.array_equal(st[row], L["st"][q]), (k, q)
        assert np.array_equal(st[7:10].T, np.asarray(L["st"]["xyz"], np.float32).reshape(-1, 3)), k
        n, s = state(1, 12)
        ob = s.reshape(12, n)
        assert n == L["ob"]["key_x"].size == exp["n_object_tracked"]
        for row, q in enumerate(("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y", "depth")):
            assert np.array_equal(ob[row], L["ob"][q]), (k, q)
        assert np.array_equal(ob[7:10].T, np.asarray(L["ob"]["xyz"], np.float32).reshape(-1, 3)), k
        assert np.array_equal(ob[10].astype(np.int32), L["ob"]["label"]), k                      # vSemObjLabel
        if k > 0:
            assert np.array_equal(ob[11].astype(np.int32), ref.result["objects"]["obj_label"]), k    # vObjLabel
        n, s = state(2, 19)
        po = s.reshape(n, 19)
        assert n == len(L["sem_pos"])
        assert np.array_equal(po[:, 0].astype(np.int32), L["sem_pos"]) and np.array_equal(po[:, 1].astype(np.int32), L["mod"]) and np.array_equal(po[:, 2].astype(np.uint8), L["stat"])
        for a in range(n):
            Hm = po[a, 3:].reshape(4, 4)
            if L["stat"][a]:
                np.testing.assert_allclose(Hm, L["H"][a], rtol=0, atol=5e-6)
            else:
                assert np.array_equal(Hm, np.eye(4, dtype=np.float32))
        n, s = state(3, 8)
        assert n == exp["n_object_samples"]
        _, sc = state(4, 17)
        assert int(sc[0]) == ref.max_id
        np.testing.assert_allclose(sc[1:].reshape(4, 4), ref.Tl, rtol=0, atol=2e-6)
        assert np.array_equal(sc[1:], T)
    host.host_system_destroy(sys_)


@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("noise", [0.0, 0.1])
def test_system_trackrgbd_equals_the_reference_source(host, noise, full, tmp_path):
    """full = True (round 5): against oracle/_ref/libref_full.so - the WHOLE reference, its real Optimizer.cc and all of g2o included, compiled verbatim
    against a mini-Eigen (tests/test_ref_g2o.py, tests/test_ref_full.py): no oracle anywhere between the product and the reference's own source.
    full = False: The PRODUCT's System::TrackRGBD (HIP kernels under the reference's class signatures) against oracle/_ref/libref_track.so = the reference's OWN
    System.cc / Tracking.cc / Frame.cc / Map.cc / ORBextractor.cc compiled verbatim (oracle/ref/, tests/test_ref_track.py): the pose TrackRGBD returns,
    the renewed static and object sets with their 3-D points and labels, the per-object vectors, the recovered mask and the converted depth map, frame
    by frame, bit for bit (object motions to 5e-6: the kernel's LM against the oracle's behind the reference's Optimizer statics).  The reference side
    runs in a child process (tests/ref_track.py): its code reads members it never initialises, which does not mix with the HIP runtime's threads."""
    from tests import oracle_lib
    from tests.ref_track import run_sequence_in_subprocess
    if (oracle_lib.load_ref_full() if full else oracle_lib.load_ref_track()) is None:
        pytest.skip("parity unpinned: oracle/_ref/libref_%s.so absent" % ("full" if full else "track"))
    host.host_system_frame_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    n_frames = 6
    fx, fy, cx, cy = synth.KITTI_K
    cfg = tmp_path / "kitti.yaml"
    cfg.write_text(YAML.format(fx=fx, fy=fy, cx=cx, cy=cy, w=W, h=H, bf=SF.BF, dmf=SF.DEPTH_MAP_FACTOR, thbg=SF.TH_DEPTH_BG, thobj=SF.TH_DEPTH_OBJ))
    Ts = SQ.camera_poses(n_frames)
    objs = SQ.default_objects()
    drop = {3: {1}, 4: {1}} if noise else {}
    labels = (1, 2, 3, 4, 5, 6, 7, 8)
    frames = [SQ.render_frame(k, Ts, objs, flow_sigma=noise, drop_masks=drop) for k in range(n_frames)]
    ref = run_sequence_in_subprocess(cfg, frames, tmp_path, labels=labels, full=full)
    sys_ = host.host_system_create(str(cfg).encode())
    assert sys_

    def state(what, rows):
        n = host.host_system_frame_state(sys_, what, None, 0)
        assert n >= 0
        buf = np.zeros(max(rows * n, 1), np.float32)
        assert host.host_system_frame_state(sys_, what, _ptr(buf), buf.size) == n
        return n, buf[:rows * n]

    for k, fr in enumerate(frames):
        depth = fr["depth_raw"].copy(); mask = fr["mask"].copy()
        rows = np.array([[k, lab, 0, 0, 0, 0, 0, 0, 0, 0] for lab in labels], np.float32)
        T = np.zeros(16, np.float32)
        assert host.host_system_track(sys_, _ptr(fr["gray"]), 1, _ptr(depth), _ptr(fr["flow"]), _ptr(mask), W, H, _ptr(rows), len(labels), 10, 1 << 30, _ptr(T)) == 0
        assert np.array_equal(T.reshape(4, 4), ref[f"T_{k}"]), (k, np.abs(T.reshape(4, 4) - ref[f"T_{k}"]).max())
        assert np.array_equal(depth, ref[f"depth_{k}"]), (k, "depth converted in place")
        assert np.array_equal(mask, ref[f"mask_{k}"]), (k, "mask after UpdateMask")
        for what, rows_ in ((0, 10), (1, 12), (3, 8)):
            if k == 0 and what == 3:
                continue                                   # (the reference fills mvTmpObj* from the first tracked frame on, src/Tracking.cc:870-872)
            n, a = state(what, rows_)
            assert n == int(ref[f"n{what}_{k}"]) and np.array_equal(a, ref[f"s{what}_{k}"]), (k, what, n)
        n, a = state(2, 19)
        assert n == int(ref[f"n2_{k}"])
        a, b = a.reshape(n, 19), ref[f"s2_{k}"].reshape(n, 19)
        assert np.array_equal(a[:, :3], b[:, :3]), (k, "nSemPosition / nModLabel / bObjStat")
        np.testing.assert_allclose(a[:, 3:], b[:, 3:], rtol=0, atol=5e-6)
        if k > 0:
            _, sa = state(4, 17)
            assert sa[0] == ref[f"s4_{k}"][0], (k, "max_id")
    host.host_system_destroy(sys_)
