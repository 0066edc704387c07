"""Runs only the K18 sweep kernel a few times on the roofline graph (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import synth
from vdo_slam_amd.ba import BatchBA, Context
g = synth.make_ba_graph(200, int(sys.argv[1]) if len(sys.argv) > 1 else 600000, 10, 1500, seed=7)
ctx = Context(0)
ba = BatchBA(ctx, g)
ms = ba.linearize(repeat=10, timed=True)
print("n_eb", g.n_eb, "n_et", g.n_et, "n_point", g.n_point, "alg_bytes", 208 * g.n_eb + 452 * g.n_et + 96 * g.n_point, "ms", ms)
