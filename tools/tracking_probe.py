"""Developer probe: wall time of the Tracking-side entry points (K11-K15) at KITTI sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import oracle_lib, frontend_ref as R
from vdo_slam_amd import synth, synth_frames as SF, tracking as TR
from vdo_slam_amd.ba import Context
from vdo_slam_amd.frontend import FrameImages
ctx = Context(0)
fr = SF.make_frame(seed=5)
depth = (SF.BF / np.maximum(fr["depth_raw"] / SF.DEPTH_MAP_FACTOR, 1e-9)).astype(np.float32)
depth[fr["depth_raw"] <= 0] = 0
W, H = 1242, 375
im = FrameImages(ctx, W, H); im.upload(depth, fr["flow"], fr["mask"])
last = FrameImages(ctx, W, H); last.upload(depth, fr["flow"], fr["mask"])
rng = np.random.default_rng(0)
o = oracle_lib.load()
ob = R.object_sample(o, fr["mask"], depth, fr["flow"], SF.TH_DEPTH_OBJ)
n_o = ob["label"].size
sx = rng.uniform(0, W, 1200).astype(np.float32); sy = rng.uniform(0, H, 1200).astype(np.float32)
ox = rng.uniform(0, W, 2500).astype(np.float32); oy = rng.uniform(0, H, 2500).astype(np.float32)
K4 = np.array(synth.KITTI_K, np.float32); I4 = np.eye(4, dtype=np.float32)
tm = np.arange(1200, dtype=np.int32)
def t(name, fn, n=50):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    print(f"{name:28s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms")
t("propagate_static 1200", lambda: TR.propagate_static(im, sx, sy))
t(f"propagate_object {n_o}", lambda: TR.propagate_object(im, ob["corr_x"], ob["corr_y"], 25.0))
t(f"scene_flow {n_o}", lambda: TR.scene_flow(ctx, (ob["corr_x"], ob["corr_y"], ob["depth"], ob["label"]), I4, (ob["key_x"], ob["key_y"], ob["depth"], ob["label"]), I4, K4, ob["label"]))
t("get3d_world 1200", lambda: TR.get3d_world(ctx, sx, sy, np.full(1200, 10, np.float32), K4, I4))
t("renew_static 1200/2500", lambda: TR.renew_static(im, tm, sx, sy, ox, oy, 1200))
t(f"update_mask {n_o}", lambda: TR.update_mask(im, last, ob["label"], ob["corr_x"], ob["corr_y"]))
