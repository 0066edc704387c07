"""Developer probe: per-frame LM kernel latency (camera problem, object batch) on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vdo_slam_amd import synth
from vdo_slam_amd.ba import Context
from vdo_slam_amd.flow2 import Flow2Batch
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = Context(0, stream.cuda_stream)
cam = Flow2Batch(ctx, [synth.make_flow2_problem(1200, seed=4)])
obj = Flow2Batch(ctx, [synth.make_flow2_problem(n, seed=30 + k, is_object=True) for k, n in enumerate([800, 600, 400, 300, 200])])
for name, b in (("camera 1200", cam), ("5 objects", obj)):
    for _ in range(3): b.run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.run()
    e1.record(); torch.cuda.synchronize()
    r = b.fetch()
    print(name, "ms/launch %.3f" % (e0.elapsed_time(e1) / 20), "its", [x["iterations"] for x in r], "trials", [x["trials"] for x in r])
