"""Runs the K18 sweep kernel and the whole linearisation a few times on the roofline graph (for rocprofv3 --kernel-trace / --pmc
passes) and prints the byte model of DESIGN.md 4.1 for this graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import synth
from vdo_slam_amd.ba import BatchBA, Context, linearize_byte_model
g = synth.make_ba_graph(200, int(sys.argv[1]) if len(sys.argv) > 1 else 600000, 10, 1500, seed=7)
ctx = Context(0)
ba = BatchBA(ctx, g)
ms_sweep, ms_lin, dims = ba.profile_linearize(10)
m = linearize_byte_model(g, dims)
print("n_eb", g.n_eb, "n_et", g.n_et, "n_point", g.n_point, "tiles", dims["tiles"], "eb_entries", dims["eb_entries"], "alg_bytes", 208 * g.n_eb + 452 * g.n_et + 96 * g.n_point, "ms", ms_sweep)
print("n_pose", g.n_pose, "dims", dims, "ms_sweep", ms_sweep, "ms_linearize", ms_lin)
print("model_bytes", m)
