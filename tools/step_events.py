"""Where inside a Step each milestone of a frame is reached (VDO_PIPE_EVENTS): the bench sequence through FramePipeline, synchronous
(the reference's TrackRGBD semantics) and deferred.  Usage (GPU box): python tools/step_events.py [steps]"""
import os, sys, time
os.environ["VDO_PIPE_EVENTS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ
from vdo_slam_amd.ba import Context
from vdo_slam_amd.pipeline import FramePipeline, kitti_params

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
warm = 5
W, H = synth.KITTI_W, synth.KITTI_H
import tempfile
spec = SQ.bench_spec(warm, steps)
n_seq = spec["n_seq"]
frames = SQ.render_bench_sequence(spec, tempfile.mkdtemp(prefix="vdo_step_events_", dir="/tmp"))
dev = [{q: torch.from_numpy(np.ascontiguousarray(f[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")} for f in frames]
torch.cuda.synchronize()
os.environ.setdefault("VDO_ORB_THREADS", "5")
import gc
gc.collect(); gc.freeze(); gc.disable()
for defer in (1, 0, 1, 0):
    ctxs = [Context(0) for _ in range(4)]
    # VDO_ORB_ON_LM_STREAM=1: the ORB context on the camera LM's STREAM (two contexts, one stream): the two alternate in time anyway, and four streams get a hardware queue each
    ctxs.append(Context(0, stream=ctxs[1].stream_ptr) if os.environ.get("VDO_ORB_ON_LM_STREAM") else Context(0))
    pipe = FramePipeline(ctxs[0], ctxs[1], kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1, defer_objects=defer), ctxs[2], ctxs[3], ctxs[4])
    pipe.keep_graph()
    for i in range(warm):
        d = dev[i]; pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
    pipe.events_ms()
    t0 = time.perf_counter()
    for i in range(warm, n_seq):
        d = dev[i]; pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
    pipe.flush()
    dt = (time.perf_counter() - t0) / steps * 1e3
    ev = pipe.events_ms()
    print(f"defer_objects={defer}: {dt:.3f} ms/step ({1e3 / dt:.0f} frames/s)")
    for k, v in sorted(ev.items(), key=lambda kv: kv[1]):
        if v >= 0: print(f"   {v:7.3f} ms  {k}")
    print("   sections:", {k: round(v / n_seq, 4) for k, v in pipe.section_ms().items()})
    pipe.close()
