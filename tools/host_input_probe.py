"""Where the time of System::TrackRGBD on host buffers goes: per-section host ms + wall per call."""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ
from vdo_slam_amd.pipeline import SECTIONS
from vdo_slam_amd.system import System, write_settings
n = 30
Ts = SQ.camera_poses(n); objs = SQ.survey_objects(leave_at=12, enter_at=18)
frames = [SQ.render_frame(k, Ts, objs, flow_sigma=0.3, invalid_depth=0.02, zero_flow=0.01) for k in range(n)]
W, H = synth.KITTI_W, synth.KITTI_H
for defer in (0, 1):
    with tempfile.TemporaryDirectory() as td:
        s = System(write_settings(os.path.join(td, "k.yaml"), W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, window=0, overlap=0))
        s.set_defer(defer)
        bufs = [(f["gray"], f["depth_raw"].copy(), f["flow"], f["mask"].copy(), np.array([[k, lab + 1] + [0.0] * 8 for lab in range(len(objs))], np.float32)) for k, f in enumerate(frames)]
        for k in range(5): s.track_rgbd(*bufs[k])
        ms0 = (C.c_double * 11)(); s._L.host_system_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]; s._L.host_system_timing(s._h, ms0)
        t0 = time.perf_counter()
        for k in range(5, n): s.track_rgbd(*bufs[k])
        s.flush()
        dt = (time.perf_counter() - t0) / (n - 5) * 1e3
        ms1 = (C.c_double * 11)(); s._L.host_system_timing(s._h, ms1)
        print(f"defer {defer}: {dt:.3f} ms per TrackRGBD;", {k_: round((b - a) / (n - 5), 3) for k_, a, b in zip(SECTIONS, ms0, ms1)})
        s.close()
# raw copies for reference
import torch
a = torch.from_numpy(frames[0]["flow"]); d = torch.empty_like(a, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): d.copy_(a)
torch.cuda.synchronize(); print("torch pageable H2D of the flow image (3.7 MB): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
