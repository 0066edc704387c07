"""Debug probe: shader-clock cycles of k_schur_dense_tile (assembly of the explicit reduced-camera matrix) per phase, summed over its workgroups
(needs the -DDENSE_PROF build of ba_solve.hip: tools/build_variant.sh denseprof "-DDENSE_PROF" ba_solve, then
VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_denseprof.so python tools/dense_asm_probe.py [n_frames n_static n_objects dyn_tracks])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import synth, _capi as K
from vdo_slam_amd.ba import BatchBA, Context
shape = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else [60, 30000, 5, 800]
g = synth.make_ba_graph(*shape, seed=1)
ctx = Context(0)
ba = BatchBA(ctx, g)
L = K.lib()
ba.optimize(max_iterations=1, gain_threshold=-1.0, solver=3)
ba.set_estimates(g.pose, g.point)
L.vdo_debug_dense_prof(None, 1)
st = ba.optimize(max_iterations=4, gain_threshold=-1.0, solver=3)
out = (C.c_ulonglong * 16)()
L.vdo_debug_dense_prof(out, 0)
names = ["head (two rounds of requests, staging, barrier)", "factored blocks of the thread's incidences (make_f)", "zero u / q / touched + barrier", "pass A (B^T e) + barrier",
         "chain marking + barrier", "chain solves + barrier", "pass C (B w, segmented sums) + barrier", "atomics into S"]
nslot, nwg = max(1, out[8]), max(1, out[9])
tot = sum(out[i] for i in range(8))
print(f"{st.total_trials} launches: {nwg / st.total_trials:.0f} workgroups, {nslot / st.total_trials:.0f} slot passes per launch; {tot / nwg:.0f} cycles per workgroup, {tot / st.total_trials / 256 / 2.4e3:.1f} us per launch if spread over 256 CUs at 2.4 GHz")
for i, nm in enumerate(names):
    per = out[i] / (nwg if i < 2 else nslot)
    print("  %-58s %9.0f cycles per %s  %5.1f %%" % (nm, per, "workgroup" if i < 2 else "slot pass", 100.0 * out[i] / tot))
