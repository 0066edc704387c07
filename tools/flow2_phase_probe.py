"""Debug probe: per-phase shader-clock cycles of k_flow2_lm (needs flow2.hip built with -DF2_PROFILE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth
from vdo_slam_amd.ba import Context
from vdo_slam_amd.flow2 import Flow2Batch
ctx = Context(0)
names = ["schur sums", "serial tail", "sweep (solve+err+build)", "accept/ctl", "init", "serial: Hs", "serial: ldlt", "serial: exp+scale"]
sizes = [int(a) for a in sys.argv[1:]] or [1200, 1000]
for label, probs in [("n=%d" % n, [synth.make_flow2_problem(n, seed=4)]) for n in sizes]:
    b = Flow2Batch(ctx, probs)
    b.run(); b.run()
    r = b.fetch()[0]
    cyc = np.array(r["T"]).ravel()[:8]
    print(label, "its", r["iterations"], "trials", r["trials"], "total cycles %.0f" % cyc.sum())
    print("   per trial: " + ", ".join("%s %.0f" % (n, c / max(1, r["trials"])) for n, c in zip(names, cyc) if n != "init"))
