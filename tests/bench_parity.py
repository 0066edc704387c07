"""Parity of the sequence bench.py times against THE REFERENCE ITSELF (oracle/_ref/libref_full.so = the reference's System / Tracking / Frame / Map /
ORBextractor / Optimizer / Converter sources and its vendored g2o, compiled where they lie; oracle/ref/Makefile).  TEST INFRASTRUCTURE: the checker of
bench.py's `parity` key and of tests/test_bench_sequence_gpu.py, never the thing measured.

Reference side: a child process (CPU) runs System::TrackRGBD over the frames of a directory (tests/ref_track.py).  Product side: the product's
System::TrackRGBD (vdo_slam_amd/host/System.cc over the C-ABI, HIP kernels) on the same host buffers, synchronous - every frame complete when the call
returns.  No borrowed seeds: each side runs its own RANSAC + EPnP + LM.  Compared per frame: the pose TrackRGBD returns (src/Tracking.cc:646), the
converted depth map and the mask after UpdateMask (digests), the renewed static set and object set with their 3-D points and labels, the object samples,
nSemPosition / nModLabel / bObjStat, max_id and every object motion vObjMod; at the end every (frame, feature) pair of every tracklet."""
import os

import numpy as np

STATE = ((0, 10), (1, 12), (2, 19), (3, 8), (4, 17))
MOTION_REL_TOL = 1e-4          # north_star: "within 1e-4 relative on final SE(3) camera poses and object motions"


def labels_of(spec):
    from vdo_slam_amd import synth_seq as SQ
    return tuple(range(1, len(SQ.bench_objects(spec)) + 1))


def write_bench_settings(path, window=1 << 20, overlap=4):
    """KITTI-0000's settings (example/kitti-0000-0013.yaml); window: WINDOW_SIZE (the default never fires PartialBatchOptimization - Track() does not read
    what it refines, src/Tracking.cc:1168-1181 - so the per-frame parity leaves it out on both sides; the windowed leg has its own test)"""
    from vdo_slam_amd import synth, synth_frames as SF
    from vdo_slam_amd.system import write_settings
    return write_settings(path, synth.KITTI_W, synth.KITTI_H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, window=window, overlap=overlap)


def start_reference(settings, frames_dir, n, out_npz, labels, n_images=1 << 30):
    from tests import oracle_lib
    from tests.ref_track import start_sequence_from_dir
    if oracle_lib.load_ref_full() is None:
        return None
    return start_sequence_from_dir(settings, frames_dir, n, out_npz, n_images=n_images, labels=labels, full=True)


def product_sequence(settings, frames, labels, n_images=1 << 30):
    """The product's System::TrackRGBD over the frames (host buffers, synchronous); returns the same per-frame record as tests/ref_track.dir_worker_main"""
    from tests.ref_track import _digest
    from vdo_slam_amd.system import System
    sysm = System(settings)
    out = {"n": len(frames)}
    try:
        for k, fr in enumerate(frames):
            depth = np.ascontiguousarray(fr["depth_raw"], np.float32).copy(); mask = np.ascontiguousarray(fr["mask"], np.int32).copy()
            rows = np.array([[k, lab, 0, 0, 0, 0, 0, 0, 0, 0] for lab in labels], np.float32)
            T = sysm.track_rgbd(np.ascontiguousarray(fr["gray"]), depth, np.ascontiguousarray(fr["flow"], np.float32), mask, rows, n_images=n_images)
            if T is None:
                raise RuntimeError(f"product TrackRGBD returned an empty pose at frame {k}")
            out[f"T_{k}"] = T; out[f"depth_sha_{k}"] = _digest(depth); out[f"mask_sha_{k}"] = _digest(mask)
            for what, rows_ in STATE:
                cnt, a = sysm.frame_state(what, rows_)
                out[f"s{what}_{k}"] = a.copy(); out[f"n{what}_{k}"] = cnt
        for which, name in ((0, "sta"), (1, "dyn")):
            off, fr_, ft_, ob_ = sysm.tracks(bool(which))
            out[f"tr_{name}_off"] = off; out[f"tr_{name}_frame"] = fr_; out[f"tr_{name}_feat"] = ft_
            if ob_ is not None:
                out[f"tr_{name}_obj"] = ob_
    finally:
        sysm.close()
    return out


def compare(ref, got, n=None):
    """-> the `parity` record.  Bit-exact parts: pose, depth / mask digests, static set, object set (points, labels), samples, per-object labels / flags, max_id,
    tracklets.  Floating-point part: the object motions (relative to the largest entry of the reference's matrix)."""
    n = int(ref["n"]) if n is None else n
    first_div = None
    pose_max_abs = 0.0; pose_max_rel = 0.0; pose_equal = 0
    sets_equal_frames = 0
    n_mot = 0; n_mot_ok = 0; mot_max_rel = 0.0; mot_max_abs = 0.0; mot_bit_equal = 0
    worst = []
    why_first = None
    parts = ("pose", "depth", "mask", "static set", "object set", "object samples", "max_id", "object count", "nSemPosition / nModLabel / bObjStat")
    part_diff = {q: 0 for q in parts}
    first_bad_motion = None
    for k in range(n):
        bad = []
        Tr, Tg = np.asarray(ref[f"T_{k}"], np.float64), np.asarray(got[f"T_{k}"], np.float64)
        d = float(np.abs(Tr - Tg).max())
        pose_max_abs = max(pose_max_abs, d); pose_max_rel = max(pose_max_rel, d / max(1.0, float(np.abs(Tr).max())))
        if np.array_equal(ref[f"T_{k}"], got[f"T_{k}"]):
            pose_equal += 1
        else:
            bad.append("pose")
        if not np.array_equal(ref[f"depth_sha_{k}"], got[f"depth_sha_{k}"]): bad.append("depth")
        if not np.array_equal(ref[f"mask_sha_{k}"], got[f"mask_sha_{k}"]): bad.append("mask")
        for what in (0, 1, 3):
            if k == 0 and what == 3:
                continue                                   # (the reference fills mvTmpObj* from the first tracked frame on, src/Tracking.cc:870-872)
            if int(ref[f"n{what}_{k}"]) != int(got[f"n{what}_{k}"]) or not np.array_equal(ref[f"s{what}_{k}"], got[f"s{what}_{k}"]):
                bad.append(("static set", "object set", "", "object samples")[what])
        if k > 0 and ref[f"s4_{k}"][0] != got[f"s4_{k}"][0]:
            bad.append("max_id")
        no = int(ref[f"n2_{k}"])
        if no != int(got[f"n2_{k}"]):
            bad.append("object count")
        else:
            a, b = np.asarray(got[f"s2_{k}"]).reshape(no, 19), np.asarray(ref[f"s2_{k}"]).reshape(no, 19)
            if not np.array_equal(a[:, :3], b[:, :3]):
                bad.append("nSemPosition / nModLabel / bObjStat")
            for j in range(no):
                if not b[j, 2]:
                    continue                               # (bObjStat false: vObjMod is the identity on both sides, covered by the flag comparison)
                n_mot += 1
                da = float(np.abs(a[j, 3:].astype(np.float64) - b[j, 3:].astype(np.float64)).max())
                rel = da / max(1e-30, float(np.abs(b[j, 3:]).max()))
                mot_max_abs = max(mot_max_abs, da); mot_max_rel = max(mot_max_rel, rel)
                mot_bit_equal += int(np.array_equal(a[j, 3:], b[j, 3:]))
                if rel <= MOTION_REL_TOL:
                    n_mot_ok += 1
                else:
                    worst.append({"frame": k, "label": int(b[j, 0]), "rel": rel})
                    if first_bad_motion is None:
                        first_bad_motion = {"frame": k, "label": int(b[j, 0]), "rel": rel}
        for q in bad:
            part_diff[q] += 1
        if not bad:
            sets_equal_frames += 1
        elif first_div is None:
            first_div = k; why_first = bad
    tr_equal = all(np.array_equal(ref[q], got[q]) for q in ("tr_sta_off", "tr_sta_frame", "tr_sta_feat", "tr_dyn_off", "tr_dyn_frame", "tr_dyn_feat", "tr_dyn_obj"))
    worst.sort(key=lambda w: -w["rel"])
    return {"against": "oracle/_ref/libref_full.so: the reference's own System / Tracking / Frame / Map / ORBextractor / Optimizer / Converter sources + vendored g2o compiled verbatim (CPU, child process); "
                       "product = System::TrackRGBD on the same host buffers, synchronous; each side runs its own RANSAC + EPnP + LM (no borrowed seeds)",
            "frames": n, "pose_bit_equal_frames": pose_equal, "pose_max_abs": pose_max_abs, "pose_max_rel": pose_max_rel,
            "index_sets_equal": bool(sets_equal_frames == n), "frames_all_bit_exact_parts_equal": sets_equal_frames, "first_divergence_frame": first_div, "first_divergence_what": why_first,
            "frames_equal_by_part": {q: n - part_diff[q] for q in parts}, "first_object_motion_outside": first_bad_motion,
            "labels_outside": sorted({w["label"] for w in worst}),
            "tracklets_equal": bool(tr_equal), "static_tracklets": int(len(ref["tr_sta_off"]) - 1), "dynamic_tracklets": int(len(ref["tr_dyn_off"]) - 1),
            "object_motions": n_mot, "object_motions_within_1e-4": n_mot_ok, "object_motions_bit_equal": mot_bit_equal, "object_motion_max_rel": mot_max_rel, "object_motion_max_abs": mot_max_abs,
            "object_motions_outside": worst[:8],
            "reference_seconds": float(ref["seconds"]) if "seconds" in ref else None}


def assert_parity(par):
    """What must hold before any timing is reported: the bit-exact parts over the whole sequence and every object motion within the north star's bar."""
    bad = []
    if not par["index_sets_equal"]:
        bad.append(f"bit-exact parts differ from frame {par['first_divergence_frame']} on ({par['first_divergence_what']})")
    if par["pose_bit_equal_frames"] != par["frames"]:
        bad.append(f"pose differs on {par['frames'] - par['pose_bit_equal_frames']} frames (max rel {par['pose_max_rel']:.2e})")
    if not par["tracklets_equal"]:
        bad.append("tracklets differ")
    if par["object_motions_within_1e-4"] != par["object_motions"]:
        bad.append(f"{par['object_motions'] - par['object_motions_within_1e-4']} of {par['object_motions']} object motions outside 1e-4 (max rel {par['object_motion_max_rel']:.2e}: {par['object_motions_outside'][:3]})")
    return bad


def check_long_sequence(par):
    """What holds on the KITTI-0000-length sequence whatever the last bits of the EPnP seeds are (tests/test_bench_sequence_gpu.py, tests/test_bench_sequence_ref.py): the camera
    trajectory, the depth maps and the static sets bit for bit over all frames; objects equal far beyond the driver's window; at least 90 % of the object motions within 1e-4."""
    n = par["frames"]; eq = par["frames_equal_by_part"]
    bad = []
    if par["pose_bit_equal_frames"] != n: bad.append(f"camera pose differs on {n - par['pose_bit_equal_frames']} frames (max rel {par['pose_max_rel']:.2e})")
    for q in ("depth", "static set", "max_id", "object count"):
        if eq[q] != n: bad.append(f"{q} differs on {n - eq[q]} frames")
    if par["first_divergence_frame"] is not None and par["first_divergence_frame"] < min(60, n): bad.append(f"first divergence at frame {par['first_divergence_frame']} ({par['first_divergence_what']})")
    if par["object_motions_within_1e-4"] < 0.9 * par["object_motions"]: bad.append(f"only {par['object_motions_within_1e-4']} of {par['object_motions']} object motions within 1e-4")
    return bad
