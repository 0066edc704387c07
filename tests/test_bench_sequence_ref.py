"""CPU half of the bench-sequence parity (round 6): over the sequence bench.py times - 5 boxes, 0.3 px flow noise, invalid pixels, a dropped mask, an object leaving and one
entering (vdo_slam_amd/synth_seq.bench_spec) -
  (1) the oracle-composed Track() (tests/pipeline_ref.py: what every GPU sequence test compares the product with) equals THE WHOLE REFERENCE (oracle/_ref/libref_full.so:
      src/*.cc + vendored g2o compiled verbatim; cv::solvePnPRansac = the oracle's restatement, OpenCV being absent) entry for entry, every object motion bit for bit;
  (2) the reference ITSELF is not stable to one ulp of its input on this sequence: raising the optical flow inside the instance mask of object 1 in frame 105 by one float
      ulp sends that object's motion estimates 1e-2 away in the next frame and of order one a few frames later (it sits at the edge of ThDepthObj with few inliers: the F3
      joint LM, 2-DoF flow vertices aliased onto 3x3 blocks, is chaotic there - tests/test_oracle_flow2.py::test_f3_lm_is_chaotic_in_the_seed shows the mechanism on one
      problem).  That is the size of the difference tests/test_bench_sequence_gpu.py finds between the product and the reference from frame 106 on, where the product's EPnP
      (a separately written restatement, equal to the oracle's to 1e-12 - tests/test_epnp_independent.py) rounds one seed to the neighbouring float: the north star's 1e-4 on
      object motions is not attainable there by ANY implementation whose last bits differ from OpenCV's - the camera poses, the static sets and every other object are not affected."""
import os

import numpy as np
import pytest

from tests import bench_parity as BP
from tests import oracle_lib
from vdo_slam_amd import synth_seq as SQ

N_FRAMES = 118          # the first 118 of the 153 frames: up to where object 1 has left the reference's trajectory for good
PERTURB_FRAME, PERTURB_LABEL = 105, 1


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    if oracle_lib.load_ref_full() is None:
        pytest.skip("parity unpinned: oracle/_ref/libref_full.so absent and no reference checkout to build it from")
    from tests.ref_track import finish_sequence
    td = tmp_path_factory.mktemp("bench_seq")
    spec = SQ.bench_spec(5, SQ.KITTI0000_FRAMES - 5)
    assert spec["n_seq"] == 153 and spec["leave_at"] == 60 and spec["enter_at"] == 80 and sorted(spec["drop_masks"]) == [8, 9]
    fdir = str(td / "frames")
    frames = SQ.render_bench_sequence(spec, fdir, n=N_FRAMES)
    # the perturbed copy: every frame linked, frame 105 rewritten with its flow inside the mask of object 1 one float ulp up
    pdir = str(td / "frames_ulp"); os.makedirs(pdir)
    for k in range(N_FRAMES):
        if k != PERTURB_FRAME:
            os.symlink(os.path.join(fdir, f"f{k}.npz"), os.path.join(pdir, f"f{k}.npz"))
    fr = dict(frames[PERTURB_FRAME]); fl = fr["flow"].copy()
    sel = np.repeat((fr["mask"] == PERTURB_LABEL)[..., None], 2, -1)
    assert sel.sum() > 1000
    fl[sel] = np.nextafter(fl[sel], np.float32(np.inf)); fr["flow"] = fl
    np.savez(os.path.join(pdir, f"f{PERTURB_FRAME}.npz"), **{q: fr[q] for q in SQ.FRAME_KEYS})
    cfg = BP.write_bench_settings(str(td / "kitti.yaml"))
    labels = BP.labels_of(spec)
    p0 = BP.start_reference(cfg, fdir, N_FRAMES, str(td / "ref.npz"), labels)
    p1 = BP.start_reference(cfg, pdir, N_FRAMES, str(td / "ref_ulp.npz"), labels)
    from tests.oracle_record import record_frame
    from tests.pipeline_ref import OraclePipeline
    ora = OraclePipeline(oracle_lib.load(), build_lm=True)
    got = {"n": N_FRAMES}
    for k, f in enumerate(frames):
        record_frame(got, k, ora, ora.step(f))
    ref = finish_sequence(p0, str(td / "ref.npz"), timeout_s=900); ref = {q: ref[q] for q in ref.files}
    ulp = finish_sequence(p1, str(td / "ref_ulp.npz"), timeout_s=900); ulp = {q: ulp[q] for q in ulp.files}
    return ref, ulp, got, ora


def test_oracle_track_equals_the_whole_reference_on_the_bench_sequence(runs):
    from tests import tracking_ref as TR
    ref, _, got, ora = runs
    got = dict(got)
    for k in range(N_FRAMES):                        # (not in the oracle's record: the converted depth map - tests/test_ref_track.py compares it - and the samples' contents)
        got[f"depth_sha_{k}"] = ref[f"depth_sha_{k}"]; got[f"s3_{k}"] = ref[f"s3_{k}"]
    got["s1_0"] = ref["s1_0"]                        # (vObjLabel of the first frame: nothing is tracked yet, tests/test_ref_track.py::compare_frame)
    o = oracle_lib.load()
    ts = TR.build_tracks(o, ora.assos_s); td_ = TR.build_tracks(o, ora.assos_d, ora.labs_d)
    got.update(tr_sta_off=ts[0], tr_sta_frame=ts[1], tr_sta_feat=ts[2], tr_dyn_off=td_[0], tr_dyn_frame=td_[1], tr_dyn_feat=td_[2], tr_dyn_obj=td_[3])
    par = BP.compare(ref, got)
    assert BP.assert_parity(par) == [], par
    assert par["object_motions"] > 300 and par["object_motions_bit_equal"] == par["object_motions"]
    assert par["static_tracklets"] > 30000 and par["dynamic_tracklets"] > 50000


def test_the_reference_itself_is_not_stable_to_one_ulp_of_its_input(runs):
    ref, ulp, _, _ = runs
    par = BP.compare(ref, ulp)
    print({q: par[q] for q in ("first_divergence_frame", "first_divergence_what", "object_motions", "object_motions_within_1e-4", "object_motion_max_rel", "labels_outside",
                               "first_object_motion_outside", "frames_equal_by_part")})
    # nothing differs before the perturbed frame is consumed; the camera pose, the static set never differ
    assert par["first_divergence_frame"] is not None and par["first_divergence_frame"] >= PERTURB_FRAME
    assert par["frames_equal_by_part"]["pose"] == N_FRAMES and par["frames_equal_by_part"]["static set"] == N_FRAMES
    # object 1 - and only it - leaves the unperturbed run's trajectory: 1e-2 relative in the first frame after the perturbation, order one later
    assert par["labels_outside"] == [PERTURB_LABEL]
    fo = par["first_object_motion_outside"]
    assert fo["frame"] == PERTURB_FRAME + 1 and fo["rel"] > 1e-3
    assert par["object_motion_max_rel"] > 0.5
    assert par["object_motions"] - par["object_motions_within_1e-4"] >= 5
