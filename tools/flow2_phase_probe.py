"""Debug probe: per-phase shader-clock cycles of k_flow2_lm (needs flow2.hip built with -DF2_PROFILE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth
from vdo_slam_amd.ba import Context
from vdo_slam_amd.flow2 import Flow2Batch
ctx = Context(0)
names = ["schur sums", "serial ldlt+exp", "sweep (solve+err+build)", "accept/ctl", "init", "-", "-", "-"]
for label, probs in (("camera 1200", [synth.make_flow2_problem(1200, seed=4)]), ("object 1000", [synth.make_flow2_problem(1000, seed=33, is_object=True)])):
    b = Flow2Batch(ctx, probs)
    b.run(); b.run()
    r = b.fetch()[0]
    cyc = np.array(r["T"]).ravel()[:8]
    print(label, "its", r["iterations"], "trials", r["trials"], "total cycles %.0f" % cyc.sum())
    for n, c in zip(names, cyc):
        print("   %-14s %9.0f cycles  %5.1f %%" % (n, c, 100 * c / cyc.sum()))
