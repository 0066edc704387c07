"""Debug probe: device stage vs host quadtree time of vdo_orb_extract on frames of the bench sequence (python tools/orb_probe.py [n])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth, synth_seq as SQ
spec = SQ.bench_spec(2, 6, seed=0)
frames = SQ.render_bench_sequence(spec, "/tmp/orb_probe_seq", workers=4)
import torch
from vdo_slam_amd.ba import Context
from vdo_slam_amd.frontend import ORBextractor
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = Context(0, stream.cuda_stream)
W, H = synth.KITTI_W, synth.KITTI_H
orb = ORBextractor(ctx, W, H)
dev = [torch.from_numpy(f["gray"]).cuda() for f in frames]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
a = b = 0.0; tt = 0.0
for i in range(n + 5):
    d = dev[i % len(dev)]
    t0 = time.perf_counter(); kp = orb.extract_device(d.data_ptr(), W); t1 = time.perf_counter()
    if i >= 5:
        x, y = orb.last_timing(); a += x; b += y; tt += t1 - t0
print("threads", os.environ.get("VDO_ORB_THREADS", "default"), "keypoints", len(kp["x"]), "cand", [orb.level_info(l)[3] for l in range(8)] if hasattr(orb, "level_info") else "")
print(f"  orb: device stage + D2H {a / n:.3f} ms, host quadtree {b / n:.3f} ms, call {tt / n * 1e3:.3f} ms")
