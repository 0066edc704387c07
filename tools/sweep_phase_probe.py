"""Debug probe: shader-clock cycles of k_sweep_tile<true> per phase and wave (needs the -DSWEEP_PROF build:
tools/build_variant.sh prof "-DSWEEP_PROF", then VDO_HIP_LIB=$PWD/vdo_slam_amd/libvdo_hip_prof.so python tools/sweep_phase_probe.py [static landmarks])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vdo_slam_amd import synth, _capi as K
from vdo_slam_amd.ba import BatchBA, Context
g = synth.make_ba_graph(200, int(sys.argv[1]) if len(sys.argv) > 1 else 600000, 10, 1500, seed=7)
ctx = Context(0)
ba = BatchBA(ctx, g)
L = K.lib()
ba.profile_linearize(3)
out = (C.c_ulonglong * 16)()
L.vdo_debug_sweep_prof(None, 1)
ms_sweep, ms_lin, dims = ba.profile_linearize(10)
L.vdo_debug_sweep_prof(out, 0)
names = ["head: requests of the (next) tile issued", "staging: points + inverse poses -> LDS", "barrier", "EdgeSE3PointXYZ edges (incl. waiting for them)",
         "segmented scan -> slot accumulators", "ternary edges", "chi2 block sums (2 barriers)", "write-back issued"]
n = max(1, out[14])          # (wave, tile) pairs timed
tot = sum(out[i] for i in range(8))
print("tiles", dims["tiles"], "waves timed", out[15], "tiles per wave", out[14] / max(1, out[15]), "ms_sweep", ms_sweep, "cycles per wave and tile", tot / n)
for i, nm in enumerate(names):
    print("  %-52s %8.0f  %5.1f %%" % (nm, out[i] / n, 100.0 * out[i] / tot))
