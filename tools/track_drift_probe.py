"""Bench-like configurations of the full Track() sequence: per-frame ground-truth error (debugging aid).
env: TS=1 ctx on a torch side stream; RES=1 all frames resident; SYNC=1 torch.cuda.synchronize() before every step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ
from vdo_slam_amd.ba import Context
from vdo_slam_amd.pipeline import FramePipeline, kitti_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65
TS, RES, SYNC = (int(os.environ.get(q, "0")) for q in ("TS", "RES", "SYNC"))
W, H = synth.KITTI_W, synth.KITTI_H
Ts = SQ.camera_poses(n); objs = SQ.default_objects()
if TS:
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    ctx = Context(0, stream.cuda_stream)
else:
    ctx = Context(0)
ctx_lm = Context(0)
mk = lambda: FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1))
LATE = int(os.environ.get("LATE", "0"))
if not LATE:
    pipe = mk()
frames = [SQ.render_frame(k, Ts, objs) for k in range(n)]
up = lambda fr: {q: torch.from_numpy(np.ascontiguousarray(fr[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")}
dev = [up(f) for f in frames] if RES else None
if LATE:
    pipe = mk()
torch.cuda.synchronize()
errs = []
for k in range(n):
    d = dev[k] if RES else up(frames[k])
    if SYNC or not RES:
        torch.cuda.synchronize()
    c = pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
    if os.environ.get("NOPOSE") and k < n - 1:
        continue
    e = float(np.abs(pipe.pose()[:3, 3] - frames[k]["Tcw"][:3, 3]).max())
    errs.append(e)
    if len(errs) > 1 and e > 2.0 * errs[-2] + 1e-3:
        print("jump at frame", k, e, errs[-2], c)
print(f"TS={TS} RES={RES} SYNC={SYNC}: final err {errs[-1]:.4f}", [round(x, 4) for x in errs[::8]])
