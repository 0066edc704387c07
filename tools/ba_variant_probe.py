"""Debug probe (A/B of builds or of VDO_BA_TILE_EPT): sweep, linearisation and ms per LM iteration on the bench's three graph shapes.
usage: [VDO_HIP_LIB=...] [VDO_BA_TILE_EPT=n] python tools/ba_variant_probe.py [bench|config3|large|roof ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import synth
from vdo_slam_amd.ba import BatchBA, Context
ctx = Context(0)
shapes = {"bench": (60, 30000, 10, 400), "config3": (60, 12000, 10, 200), "large": (240, 960000, 10, 500), "roof": (200, 2200000, 10, 1500), "omd": (300, 150000, 4, 40000)}
for name in (sys.argv[1:] or ["bench", "large", "roof"]):
    g = synth.make_ba_graph(*shapes[name], seed=7)
    ba = BatchBA(ctx, g)
    ba.profile_linearize(5)
    sw, lin, dims = ba.profile_linearize(20)
    p0, q0 = ba.estimates()
    ts = []
    for rep in range(4):
        ba.set_estimates(p0, q0)
        ctx.synchronize(); t = time.perf_counter()
        st = ba.optimize(max_iterations=5)
        ctx.synchronize(); ts.append((time.perf_counter() - t) * 1e3 / max(1, st.iterations))
    print(name, "edges", g.n_eb, "tiles", dims["tiles"], "max_slots", dims["max_slots"], "sweep %.4f lin %.4f" % (sw, lin), "ms/LM it %.3f" % min(ts), "its", st.iterations, "trials", st.total_trials, "chi2 %.12g" % st.final_chi2)
    ba.close()
