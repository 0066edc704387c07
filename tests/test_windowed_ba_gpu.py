"""End to end with the batch optimisers INSIDE Track(), against the whole reference (oracle/_ref/libref_full.so, CPU child process): System::TrackRGBD over 24 frames of the bench
sequence with the reference's settings WINDOW_SIZE 20 / OVERLAP_SIZE 4 - Optimizer::PartialBatchOptimization fires at f_id 19 (src/Tracking.cc:1168-1181) - and nImage = 24, so that
Optimizer::FullBatchOptimization runs at StopFrame (:1189-1207).  Compared: everything of tests/bench_parity.py frame by frame (Track() does not read what the optimisers refine:
the windowed run must leave the per-frame results bit for bit where they were) AND the Map the two optimisers leave behind - vmCameraPose (the window refined in place),
vmCameraPose_RF, vmRigidMotion / vmRigidMotion_RF with their labels, vp3DPointSta / vp3DPointDyn - on the same data, built by each side's own Track().  The reference's g2o runs
BlockSolverX + LinearSolverCSparse on the whole system; the product builds the graph from its GraphStore and solves by Schur complement + chain-preconditioned PCG on the GPU."""
import os

import numpy as np
import pytest

from tests import bench_parity as BP
from tests import oracle_lib
from vdo_slam_amd import synth_seq as SQ

pytestmark = pytest.mark.gpu


# 24 frames: the first window (f_id 19: the only one with a gauge prior, two Levenberg iterations) + the final batch.  40 frames (round 6): also the second window (f_id 35) -
# no gauge prior (src/Optimizer.cc:227-236), ~27 iterations, the path on which the product's PCG gives way to its dense MFMA solver (ba_lm.hip kDenseTinyUnknowns).
@pytest.mark.parametrize("N", [24, 40])
def test_trackrgbd_with_windowed_and_final_batch_equals_the_whole_reference(tmp_path, N):
    from tests.ref_track import MAP_PARTS, _digest, finish_sequence, start_sequence_from_dir
    from vdo_slam_amd.system import System
    L = oracle_lib.load_ref_full()
    if L is None or not hasattr(L, "vdo_ref_system_map_export"):
        pytest.skip("parity unpinned: oracle/_ref/libref_full.so absent (or built before round 6)")
    spec = SQ.bench_spec(5, N - 5)
    labels = BP.labels_of(spec)
    fdir = str(tmp_path / "frames")
    frames = SQ.render_bench_sequence(spec, fdir)
    cfg = BP.write_bench_settings(str(tmp_path / "kitti.yaml"), window=20, overlap=4)
    proc = start_sequence_from_dir(cfg, fdir, N, str(tmp_path / "ref.npz"), n_images=N, labels=labels, full=True, export_map=True)
    # ---- the product: the same calls, the same nImage
    sysm = System(cfg)
    got = {"n": N}
    try:
        for k, fr in enumerate(frames):
            depth = fr["depth_raw"].copy(); mask = fr["mask"].copy()
            rows = np.array([[k, lab, 0, 0, 0, 0, 0, 0, 0, 0] for lab in labels], np.float32)
            T = sysm.track_rgbd(fr["gray"], depth, fr["flow"], mask, rows, n_images=N)
            assert T is not None, k
            got[f"T_{k}"] = T; got[f"depth_sha_{k}"] = _digest(depth); got[f"mask_sha_{k}"] = _digest(mask)
            for what, rows_ in BP.STATE:
                cnt, a = sysm.frame_state(what, rows_)
                got[f"s{what}_{k}"] = a.copy(); got[f"n{what}_{k}"] = cnt
        for which, name in ((0, "sta"), (1, "dyn")):
            off, fr_, ft_, ob_ = sysm.tracks(bool(which))
            got[f"tr_{name}_off"] = off; got[f"tr_{name}_frame"] = fr_; got[f"tr_{name}_feat"] = ft_
            if ob_ is not None:
                got[f"tr_{name}_obj"] = ob_
        for what, name in enumerate(MAP_PARTS):
            got["map_" + name] = sysm.map_export(what)
    finally:
        sysm.close()
    ref = finish_sequence(proc, str(tmp_path / "ref.npz"), timeout_s=1200)
    ref = {q: ref[q] for q in ref.files}
    # ---- per frame: nothing of Track() may move because the optimisers ran
    par = BP.compare(ref, got)
    assert BP.assert_parity(par) == [], par
    # ---- the Map
    assert np.array_equal(ref["map_rm_count"], got["map_rm_count"]) and np.array_equal(ref["map_rm_label"], got["map_rm_label"])
    cp_r, cp_g = ref["map_cam_pose"].reshape(N, 4, 4), got["map_cam_pose"].reshape(N, 4, 4)
    rf_r, rf_g = ref["map_cam_pose_rf"].reshape(N, 4, 4), got["map_cam_pose_rf"].reshape(N, 4, 4)
    moved_window = float(np.abs(cp_r - np.stack([np.linalg.inv(ref[f"T_{k}"].astype(np.float64)) for k in range(N)])).max())
    assert moved_window > 1e-6, "the windowed optimisation refined vmCameraPose in place"
    assert float(np.abs(rf_r - cp_r).max()) > 1e-7, "the final batch optimisation wrote vmCameraPose_RF"

    def close(a, b, what):
        a = a.astype(np.float64); b = b.astype(np.float64)
        rot = np.abs(a[:, :3, :3] - b[:, :3, :3]).max(); tr = np.abs(a[:, :3, 3] - b[:, :3, 3]).max() / max(1.0, np.abs(b[:, :3, 3]).max())
        print(f"{what}: rotation {rot:.2e}, translation (relative) {tr:.2e}")
        assert rot <= 1e-5 and tr <= 1e-5, (what, rot, tr)               # north star: 1e-4 relative; CV_32F storage 6e-8, the two solvers' stop rules 1e-6
    close(cp_g, cp_r, "vmCameraPose (window refined in place)")
    close(rf_g, rf_r, "vmCameraPose_RF")
    close(got["map_rigid_motion"].reshape(-1, 4, 4), ref["map_rigid_motion"].reshape(-1, 4, 4), "vmRigidMotion")
    close(got["map_rigid_motion_rf"].reshape(-1, 4, 4), ref["map_rigid_motion_rf"].reshape(-1, 4, 4), "vmRigidMotion_RF")
    for q in ("map_points_sta", "map_points_dyn"):
        a, b = got[q].astype(np.float64), ref[q].astype(np.float64)
        assert a.shape == b.shape and b.size > 10000
        d = np.abs(a - b).max() / max(1.0, np.abs(b).max())
        print(f"{q}: {b.size // 3} points, max relative difference {d:.2e}")
        assert d <= 1e-5, (q, d)
