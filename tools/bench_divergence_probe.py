"""GPU probe: where and why the product leaves the oracle-composed Track() (== the whole reference, tests/test_ref_full.py) on the bench sequence.
usage: python tools/bench_divergence_probe.py [steps] [warmup] [own|product]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ
from vdo_slam_amd.ba import Context
from vdo_slam_amd.pipeline import FramePipeline, kitti_params
from tests import oracle_lib
from tests.pipeline_ref import OraclePipeline
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 5
seed_mode = sys.argv[3] if len(sys.argv) > 3 else "own"
W, H = synth.KITTI_W, synth.KITTI_H
spec = SQ.bench_spec(warmup, steps)
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    frames = SQ.render_bench_sequence(spec, os.path.join(td, "f"))
ctx, ctx_lm, ctx_obj = Context(0), Context(0), Context(0)
pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1), ctx_obj)
ref = OraclePipeline(oracle_lib.load(), build_lm=True, seed_refit=None if seed_mode == "own" else "product")
keys = ("n_orb", "n_static_new", "n_object_samples", "n_static_tracked", "n_object_tracked", "n_objects", "n_recovered_masks", "n_static_tracks",
        "n_dynamic_tracks", "n_ransac_cam", "n_motion_model_cam", "n_ransac_obj", "n_cam_inliers", "cam_lm_iterations", "n_mm_inliers_obj", "n_motion_model_obj")
ndiff = 0
for k, fr in enumerate(frames):
    d = {q: torch.from_numpy(np.ascontiguousarray(fr[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")}
    torch.cuda.synchronize()
    got = pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
    exp = ref.step(fr)
    cd = {q: (got[q], exp[q]) for q in keys if got[q] != exp[q]}
    dp = float(np.abs(pipe.pose() - ref.Tl).max())
    ms, mo = pipe.motions(), ref.motions
    md = []
    for a, b in zip(ms, mo):
        md.append((a["sem_label"], b["sem_label"], a["n_inliers"], b["n_inliers"], float(np.abs(a["H"] - b["H"]).max())))
    bad = bool(cd) or dp > 0 or len(ms) != len(mo) or any(m[4] > 0 or m[2] != m[3] for m in md)
    print(f"frame {k}: pose diff {dp:.2e} counts {cd} motions(sem got/exp, inl got/exp, dH) {[(m[0], m[1], m[2], m[3], f'{m[4]:.1e}') for m in md]} {'<-- DIFF' if bad else ''}", flush=True)
    if bad:
        ndiff += 1
        if ndiff >= 4:
            break
if ref.epnp_log:
    log = ref.epnp_log
    print("epnp log:", len(log), "refits; same inliers", sum(c["same_inliers"] for c in log), "same float seed", sum(c["same_float_seed"] for c in log), "max dT", max(c["dT"] for c in log))
    for i, c in enumerate(log):
        if not c["same_float_seed"] or not c["same_inliers"]:
            print("  refit", i, c)
