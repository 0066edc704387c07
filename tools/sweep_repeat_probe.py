"""Debug probe: the K18 sweep and the whole linearisation timed with growing / shrinking repeat counts in one process (clock ramp, warm-up effects).
usage: python tools/sweep_repeat_probe.py [static landmarks] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import synth
from vdo_slam_amd.ba import BatchBA, Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2200000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = synth.make_ba_graph(frames, n, 10, 1500, seed=7)
ctx = Context(0)
ba = BatchBA(ctx, g)
print("edges", g.n_eb, g.n_et, "points", g.n_point, ba.dims())
for rep in (10, 30, 100, 10, 30, 3):
    a, b, _ = ba.profile_linearize(rep)
    print(rep, "sweep %.4f lin %.4f" % (a, b))
