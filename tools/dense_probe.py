"""ms per LM iteration of the batch graph with the PCG and with the dense MFMA solver (bench graph by default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth
from vdo_slam_amd.ba import BatchBA, Context
shape = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else [60, 30000, 5, 800]
g = synth.make_ba_graph(*shape, seed=1)
ctx = Context(0)
versions = [a for a in os.environ.get("DENSE_PROBE_VERSIONS", "1,2,3,4,5").split(",") if a]
chunks = [a for a in os.environ.get("DENSE_PROBE_CHUNKS", "").split(",") if a]      # slots per workgroup of the assembly kernel (VDO_BA_DENSE_CHUNK), default sequence
for solver, ver, chunk in [(2, "", "")] + [(3, v, "") for v in versions] + [(3, "", c) for c in chunks]:
    os.environ.pop("VDO_BA_DENSE", None); os.environ.pop("VDO_BA_DENSE_CHUNK", None)
    if ver:
        os.environ["VDO_BA_DENSE"] = ver
    if chunk:
        os.environ["VDO_BA_DENSE_CHUNK"] = chunk
    ba = BatchBA(ctx, g)
    ba.optimize(max_iterations=1, gain_threshold=-1.0, solver=solver)
    ba.set_estimates(g.pose, g.point)
    t0 = time.perf_counter()
    st = ba.optimize(max_iterations=5, gain_threshold=-1.0, solver=solver)
    dt = (time.perf_counter() - t0) * 1e3
    print(f"solver {solver} VDO_BA_DENSE={ver or '-'} CHUNK={chunk or '-'}: {g.n_pose} poses ({6 * g.n_pose} unknowns), {st.iterations} its / {st.total_trials} trials, {dt / st.iterations:.3f} ms per LM iteration, final chi2 {st.final_chi2:.12g}")
    ba.close()
