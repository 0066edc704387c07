"""Debug probe: where a launch of k_pcg_chain<0> (one CG iteration of a pose chain: alpha, x / r update, z = M^-1 r partitioned over the chain's waves, r.z) spends its
time - shader-clock stamps of thread 0 of workgroup 0 (needs the -DPCG_PROF build: tools/build_variant.sh pcgprof "-DPCG_PROF" ba_solve, then
VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_pcgprof.so python tools/pcg_chain_phase_probe.py [bench|large|roof])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import synth, _capi as K
from vdo_slam_amd.ba import BatchBA, Context
shapes = {"bench": (60, 30000, 10, 400), "large": (240, 960000, 10, 500), "roof": (200, 2200000, 10, 1500)}
name = sys.argv[1] if len(sys.argv) > 1 else "large"
g = synth.make_ba_graph(*shapes[name], seed=7)
ctx = Context(0)
ba = BatchBA(ctx, g)
L = K.lib()
p0, q0 = ba.estimates()
ba.optimize(max_iterations=3)
ba.set_estimates(p0, q0)
out = (C.c_ulonglong * 20)()
L.vdo_debug_pcg_prof(None, 1)
st = ba.optimize(max_iterations=5)
L.vdo_debug_pcg_prof(out, 0)
names = ["p.q partials -> alpha", "x += alpha p, r -= alpha q, r -> LDS strip", "barrier at the head of the solve", "forward recurrences of the segments", "boundary carry (wave 0)",
         "corrections y = yhat + P y_in (+ far link)", "w = Dinv y (+ far link)", "backward recurrences", "boundary carry back (wave 0)", "corrections z = zhat + Q z_in", "z -> HBM, r.z block sum"]
n = max(1, out[19]); tot = sum(out[i] for i in range(11))
print(name, "launches timed", n, "LM iterations", st.iterations, "cycles per launch (workgroup 0)", tot / n, "= %.2f us at 2.4 GHz (the shader clock of clock64 is 100 MHz-based on some parts: compare the shares)" % (tot / n / 2400.0))
for i, nm in enumerate(names):
    print("  %-56s %9.0f  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / tot))
