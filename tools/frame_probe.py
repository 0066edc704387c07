"""Per-stage wall time of the per-frame hot path (each stage bracketed by a stream sync), to see where
the frame time goes.  Usage: python tools/frame_probe.py [n_frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
from vdo_slam_amd import synth, synth_frames as SF  # noqa: E402
from vdo_slam_amd.ba import Context  # noqa: E402
from vdo_slam_amd.flow2 import Flow2Batch  # noqa: E402
from vdo_slam_amd.frontend import FrameImages, ORBextractor  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    ctx = Context(0, stream.cuda_stream)
    frames, cam, obj = bench.make_frame_inputs(1000)
    W, H = synth.KITTI_W, synth.KITTI_H
    dev = [dict(gray=torch.from_numpy(f["gray"]).cuda(), depth=torch.from_numpy(f["depth_raw"]).cuda(),
                flow=torch.from_numpy(f["flow"]).cuda(), mask=torch.from_numpy(f["mask"]).cuda()) for f in frames]
    orb = ORBextractor(ctx, W, H); fimg = FrameImages(ctx, W, H)
    cam_b = [Flow2Batch(ctx, [p]) for p in cam]; obj_b = [Flow2Batch(ctx, ps) for ps in obj]
    stages = ["upload+depth", "orb", "static_filter", "object_sample", "lm_cam", "lm_obj"]
    acc = dict.fromkeys(stages, 0.0)
    orb_dev = orb_tree = 0.0

    def timed(name, fn):
        t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] += time.perf_counter() - t
        return r

    for i in range(n + 5):
        if i == 5:
            acc = dict.fromkeys(stages, 0.0)
        k = i % len(dev); d = dev[k]
        timed("upload+depth", lambda: (fimg.upload_device(d["depth"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr()), fimg.depth_preprocess(SF.BF, SF.DEPTH_MAP_FACTOR)))
        kp = timed("orb", lambda: orb.extract_device(d["gray"].data_ptr(), W))
        if i >= 5:
            a, b = orb.last_timing(); orb_dev += a; orb_tree += b
        timed("static_filter", lambda: fimg.static_filter(kp["x"], kp["y"], SF.TH_DEPTH_BG))
        timed("object_sample", lambda: fimg.object_sample(SF.TH_DEPTH_OBJ))
        timed("lm_cam", lambda: cam_b[k].run())
        timed("lm_obj", lambda: obj_b[k].run())
    print(f"  orb: device stage + D2H {orb_dev / n:.3f} ms, host quadtree {orb_tree / n:.3f} ms")
    tot = sum(acc.values())
    for s in stages:
        print(f"{s:16s} {acc[s] / n * 1e3:8.3f} ms")
    print(f"{'total':16s} {tot / n * 1e3:8.3f} ms  ({n / tot:.1f} frames/s, stage-synchronous)")


if __name__ == "__main__":
    main()
