"""Debug probe: which XCD the workgroups of an LM cluster run on (needs the -DF2_PROFILE build: tools/build_profiled_flow2.sh, then
VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_prof.so python tools/xcc_probe.py).  Prints, per problem, (XCC_ID + 100 * block id) of
workgroups 0 and 1: with the id-mod-8 placement of k_flow2_lm both sit on the same XCD (measured: block p and block p + 8 -> XCC p)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from vdo_slam_amd import synth
from vdo_slam_amd.ba import Context
from vdo_slam_amd.flow2 import Flow2Batch
ctx = Context(0)
for nprob in (1, 5):
    probs = []
    for k in range(nprob):
        p = synth.make_flow2_problem(400, seed=4 + k, is_object=True); p.ref_quirks = 1; probs.append(p)
    b = Flow2Batch(ctx, probs)
    for rep in range(3):
        b.run(); r = b.fetch()
        print(nprob, [tuple(np.array(x["T"]).ravel()[14:16]) for x in r])
