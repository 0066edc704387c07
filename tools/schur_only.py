"""Runs one Levenberg iteration (linearisation, factorisations, PCG: k_schur_tile<0> once per CG iteration) and then 20 Schur mat-vec launches alone on the roofline graph
(for rocprofv3 --kernel-trace / --pmc passes) and prints the graph, the tile layout and the byte model of the mat-vec (vdo_slam_amd/ba.py schur_byte_model)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import synth
from vdo_slam_amd.ba import BatchBA, Context, schur_byte_model
n_static = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
g = synth.make_ba_graph(200, n_static, 10, 1500, seed=7)
ctx = Context(0)
ba = BatchBA(ctx, g)
ms_sweep, ms_lin, dims = ba.profile_linearize(3)
st = ba.optimize(max_iterations=1, gain_threshold=-1.0)
ms = ba.profile_schur(20)
print("n_eb", g.n_eb, "n_et", g.n_et, "n_point", g.n_point, "tiles", dims["tiles"], "eb_entries", dims["eb_entries"], "ms", ms)
print("n_pose", g.n_pose, "dims", dims, "ms_schur_matvec", ms, "lm", st.iterations, st.total_trials)
print("model_bytes", schur_byte_model(g, dims, n_static))
