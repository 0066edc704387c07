"""Debug probe: per-phase shader-clock cycles of k_flow2_lm (needs the -DF2_PROFILE build: tools/build_profiled_flow2.sh, then
VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_prof.so python tools/flow2_phase_probe.py 1200 o400 ...; "oN" = an object problem)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth
from vdo_slam_amd.ba import Context
from vdo_slam_amd.flow2 import Flow2Batch
ctx = Context(0)
names = ["schur: exchange", "serial tail", "sweep: exchange", "accept/ctl", "init", "serial: Hs", "serial: ldlt", "serial: exp+scale",
         "sweep: loads+backsubst", "sweep: project+err", "sweep: huber+J+sums", "sweep: reduce", "schur: sums", "schur: reduce", "-", "-"]
args = sys.argv[1:] or ["1200", "o400"]
for a in args:
    is_obj = a.startswith("o")
    n = int(a.lstrip("o"))
    p = synth.make_flow2_problem(n, seed=4, is_object=is_obj)
    p.ref_quirks = 1
    b = Flow2Batch(ctx, [p])
    b.run(); b.run()
    r = b.fetch()[0]
    cyc = np.array(r["T"]).ravel()[:16]
    print(a, "its", r["iterations"], "trials", r["trials"], "cycles per trial %.0f" % ((cyc.sum() - cyc[4]) / max(1, r["trials"])))
    print("   per trial: " + ", ".join("%s %.0f" % (nm, c / max(1, r["trials"])) for nm, c in zip(names, cyc) if nm not in ("init", "-")))
