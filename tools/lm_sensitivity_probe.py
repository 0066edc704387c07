"""Every LM problem the oracle-composed Track() builds on the noisy 5-object sequence, solved by the oracle and by the GPU
kernel: iterations / trials / pose difference per problem (debug aid: where do the two LM trajectories part?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth, synth_seq as SQ
from vdo_slam_amd.ba import Context
from vdo_slam_amd.flow2 import Flow2Batch
from tests import oracle_lib
import tests.pipeline_ref as PR
import tests.test_oracle_flow2 as TF
o = oracle_lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
Ts = SQ.camera_poses(n); objs = SQ.default_objects(5)
ref = PR.OraclePipeline(o, build_lm=True)
probs = []
orig = TF.run_oracle
def rec(oracle, prob):
    r = orig(oracle, prob); probs.append((prob, r)); return r
TF.run_oracle = rec
for k in range(n):
    ref.step(SQ.render_frame(k, Ts, objs, flow_sigma=0.3, invalid_depth=0.02, zero_flow=0.01, drop_masks={5: {2}, 6: {2}}))
ctx = Context(0)
for j, (p, (T, flow, inl, ninl, st)) in enumerate(probs):
    b = Flow2Batch(ctx, [p]); b.run(); (r,) = b.fetch(); b.close()
    dT = np.abs(r["T"] - T).max()
    flag = "" if (r["iterations"] == st.iterations and r["trials"] == st.total_trials and dT < 1e-6) else "   <-- differs"
    print(f"problem {j:3d} n={p.n:5d} oracle its/trials {st.iterations:3d}/{st.total_trials:3d} gpu {r['iterations']:3d}/{r['trials']:3d}  dT {dT:.2e}  inliers {ninl}/{r['n_inliers']} chi2 {st.final_chi2:.6g}/{r['final_chi2']:.6g}{flag}")
