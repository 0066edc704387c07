"""Developer probe: LM verbose trace + timing on a synthetic batch graph (GPU)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth
from vdo_slam_amd.ba import BatchBA, Context
frames, static, objs, dyn = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (60, 30000, 5, 800))]
its = int(sys.argv[5]) if len(sys.argv) > 5 else 3
verbose = int(sys.argv[6]) if len(sys.argv) > 6 else 2
g = synth.make_ba_graph(frames, static, objs, dyn, seed=1)
print("P", g.n_pose, "L", g.n_point, "Eb", g.n_eb, "Et", g.n_et, "Ep", g.n_ep, "sweep MB", g.sweep_bytes() / 1e6)
ctx = Context(0)
ba = BatchBA(ctx, g)
ba.linearize()
for rep in (20, 50):
    ms = ba.linearize(repeat=rep, timed=True)
    print("sweep_eb ms", ms, "GB/s", 208 * g.n_eb / ms / 1e6)
t = time.perf_counter()
st = ba.optimize(max_iterations=its, gain_threshold=-1.0, verbose=verbose)
dt = time.perf_counter() - t
print("LM its", st.iterations, "trials", st.total_trials, "ms/iter", dt * 1e3 / st.iterations, "lin ms", st.ms_linearize, "solve ms", st.ms_solve)
