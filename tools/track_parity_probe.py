"""Per-frame GPU FramePipeline (build_lm) vs oracle-composed Track() over a long sequence: prints the first differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ
from vdo_slam_amd.ba import Context
from vdo_slam_amd.pipeline import FramePipeline, kitti_params
from tests import oracle_lib
from tests.pipeline_ref import OraclePipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65
W, H = synth.KITTI_W, synth.KITTI_H
Ts = SQ.camera_poses(n); objs = SQ.default_objects()
ctx, ctx_lm = Context(0), Context(0)
pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1))
ref = OraclePipeline(oracle_lib.load(), build_lm=True)
nd = 0
for k in range(n):
    fr = SQ.render_frame(k, Ts, objs)
    d = {q: torch.from_numpy(np.ascontiguousarray(fr[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")}
    torch.cuda.synchronize()
    got = pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
    exp = ref.step(fr)
    diff = {q: (got[q], exp[q]) for q in exp if got[q] != exp[q]}
    dp = float(np.abs(pipe.pose() - ref.Tl).max())
    gt = fr["Tcw"]
    eg = float(np.abs(pipe.pose()[:3, 3] - gt[:3, 3]).max()); eo = float(np.abs(ref.Tl[:3, 3] - gt[:3, 3]).max())
    if k % 4 == 0 or k > 44:
        print(f"frame {k}: pose diff {dp:.2e}  gt err gpu {eg:.4f} oracle {eo:.4f}  count diffs {diff}")
        nd += bool(diff) or dp > 2e-6
    if nd > 1200:
        break
