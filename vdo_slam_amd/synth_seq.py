"""Geometrically consistent synthetic RGB-D / flow / mask sequence (SURVEY.md §8d "Synthetic inputs"):
a static scene (ground plane, two side walls, a distant back wall) seen by a camera that drives forward with
a small yaw oscillation, plus K rigid objects (upright fronto-parallel panels, or boxes) that move on the ground: their
centres translate with constant velocity, boxes may also turn about their vertical axis with a constant yaw rate (<= 0.05
rad/frame), and an object may exist only for a range of frames [t0, t1) (one leaves, one enters: SURVEY 8d).
Depth, exact dense optical flow (frame t -> t+1) and instance masks are rendered analytically per pixel,
in the on-disk conventions of the reference (depth = disparity * DepthMapFactor, example/vdo_slam.cc:105-139).
The gray image is texture only (ORB needs corners; correspondences come from the flow)."""
import numpy as np

from .synth import KITTI_H, KITTI_K, KITTI_W, rotvec_to_R
from .synth_frames import BF, DEPTH_MAP_FACTOR, make_gray


def camera_poses(n_frames, step=0.8, yaw_amp=0.004):
    """T_wc of every frame (camera-to-world, y down, z forward); frame 0 is the identity."""
    Ts = [np.eye(4)]
    yaw = 0.0
    for k in range(1, n_frames + 1):
        yaw += yaw_amp * np.sin(0.3 * k)
        T = np.eye(4)
        T[:3, :3] = rotvec_to_R(np.array([0.0, yaw, 0.0]))
        T[:3, 3] = Ts[-1][:3, 3] + Ts[-1][:3, :3] @ np.array([0.0, 0.0, step])
        Ts.append(T)
    return Ts


def default_objects(n=3, box_depth=0.0):
    """(centre xyz at frame 0 [m], half width, half height, velocity per frame [m]); n = 3 or 5 objects.
    box_depth > 0: axis-aligned boxes of that half depth ("hd") instead of fronto-parallel panels."""
    # velocities close to the camera's 0.8 m/frame: the objects stay inside ThDepthObj for ~100 frames
    objs = [dict(c=np.array([-3.0, 0.9, 14.0]), hw=1.1, hh=0.75, v=np.array([0.0, 0.0, 0.9])),
            dict(c=np.array([2.5, 0.85, 10.0]), hw=1.0, hh=0.8, v=np.array([0.01, 0.0, 0.76])),
            dict(c=np.array([5.0, 0.9, 19.0]), hw=1.2, hh=0.75, v=np.array([-0.02, 0.0, 0.85]))] + ([] if n <= 3 else [
            dict(c=np.array([-6.0, 0.8, 9.0]), hw=0.9, hh=0.85, v=np.array([0.0, 0.0, 0.82])),
            dict(c=np.array([0.3, 0.95, 21.0]), hw=1.3, hh=0.7, v=np.array([0.015, 0.0, 0.7]))])
    if box_depth > 0:
        objs = [dict(ob, hd=box_depth) for ob in objs]
    return objs


def survey_objects(leave_at=60, enter_at=80, box_depth=0.9):
    """SURVEY.md 8d's object set: 5 boxes with labels 1..5 moving with constant twists - all but one also turn (yaw rate
    <= 0.05 rad/frame) - where object 2 exists only before frame `leave_at` and object 5 only from frame `enter_at` on."""
    objs = default_objects(5, box_depth=box_depth)
    for ob, (yaw0, rate) in zip(objs, [(0.10, 0.012), (-0.20, -0.02), (0.0, 0.0), (0.30, 0.035), (-0.10, 0.05)]):
        ob["yaw0"], ob["yaw_rate"] = yaw0, rate
    objs[1]["t1"] = leave_at
    objs[4]["t0"] = enter_at
    # object 5 starts so that it is in range when it appears (the camera has advanced 0.8 m/frame by then)
    objs[4]["c"] = objs[4]["c"] + np.array([0.0, 0.0, 0.1 * enter_at])
    return objs


def object_exists(ob, k):
    return ob.get("t0", 0) <= k < ob.get("t1", 1 << 30)


def _yaw_R(a):
    return rotvec_to_R(np.array([0.0, a, 0.0]))


def object_pose(ob, k):
    """(R, c): orientation and centre of the object at frame k (world frame)."""
    return _yaw_R(ob.get("yaw0", 0.0) + k * ob.get("yaw_rate", 0.0)), ob["c"] + k * ob["v"]


def render_frame(k, Ts, objects, w=KITTI_W, h=KITTI_H, K4=KITTI_K, flow_sigma=0.0, seed=0, invalid_depth=0.0, zero_flow=0.0, drop_masks=None):
    """Frame k: dict(gray u8, depth_raw f32 (disparity*256), flow f32 [h,w,2] (k -> k+1), mask i32, Tcw 4x4, Tcw_next).
    SURVEY.md 8d extras: flow_sigma px of Gaussian flow noise, a fraction invalid_depth of pixels with disparity 0, a fraction
    zero_flow of pixels with exactly-zero flow, drop_masks = {frame: labels whose instance mask is missing in that frame}."""
    fx, fy, cx, cy = K4
    T_wc, T_wc1 = Ts[k], Ts[k + 1]
    vv, uu = np.mgrid[0:h, 0:w].astype(np.float64)
    rays = np.stack([(uu - cx) / fx, (vv - cy) / fy, np.ones_like(uu)], -1)            # camera frame, z = 1  ->  parameter = depth
    d = rays @ T_wc[:3, :3].T
    o = T_wc[:3, 3]
    s = np.full((h, w), np.inf)
    label = np.zeros((h, w), np.int32)

    def hit(cond, sv):
        nonlocal s
        upd = cond & (sv > 0.5) & (sv < s)
        s = np.where(upd, sv, s)
        return upd

    with np.errstate(divide="ignore", invalid="ignore"):
        hit(d[..., 1] > 1e-9, (1.65 - o[1]) / d[..., 1])                               # ground y = 1.65 (camera height)
        for xw in (-9.0, 9.0):                                                         # side walls
            sv = (xw - o[0]) / d[..., 0]
            yy = o[1] + sv * d[..., 1]
            hit((np.abs(d[..., 0]) > 1e-9) & (yy > -6.0) & (yy < 1.65), sv)
        hit(d[..., 2] > 1e-9, (400.0 - o[2]) / d[..., 2])                              # back wall far beyond ThDepthBG
        for j, ob in enumerate(objects):
            if not object_exists(ob, k):
                continue
            Ro, c = object_pose(ob, k)
            hd = ob.get("hd", 0.0)
            if hd <= 0:                                                                # panel z = const, facing the camera
                assert not ob.get("yaw_rate") and not ob.get("yaw0"), "only boxes turn"
                sv = (c[2] - o[2]) / d[..., 2]
                px = o[0] + sv * d[..., 0]; py = o[1] + sv * d[..., 1]
                upd = hit((d[..., 2] > 1e-9) & (np.abs(px - c[0]) < ob["hw"]) & (np.abs(py - c[1]) < ob["hh"]), sv)
            else:                                                                      # oriented box (slab method in the box frame): front, sides and top are seen
                ol = (o - c) @ Ro; dl = d @ Ro                                         # (the ray parameter = camera depth is frame independent)
                half = np.array([ob["hw"], ob["hh"], hd])
                t1 = (-half - ol) / dl; t2 = (half - ol) / dl
                tn = np.nanmax(np.minimum(t1, t2), axis=-1); tf = np.nanmin(np.maximum(t1, t2), axis=-1)
                upd = hit((tn <= tf) & (tf > 0), tn)
            label = np.where(upd, j + 1, label)
    depth = s                                                                          # z-depth in the camera frame (ray z = 1)
    Xw = o + depth[..., None] * d
    Xn = Xw.copy()
    for j, ob in enumerate(objects):
        sel = label == j + 1
        if sel.any():
            H = object_motion(ob, k)
            Xn[sel] = Xw[sel] @ H[:3, :3].T + H[:3, 3]
    T_c1w = np.linalg.inv(T_wc1)
    Xc1 = Xn @ T_c1w[:3, :3].T + T_c1w[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        un = fx * Xc1[..., 0] / Xc1[..., 2] + cx; vn = fy * Xc1[..., 1] / Xc1[..., 2] + cy
    flow = np.stack([un - uu, vn - vv], -1)
    if flow_sigma:
        flow = flow + np.random.default_rng(seed + 77 * k).normal(0, flow_sigma, flow.shape)
    flow = np.nan_to_num(flow, nan=0.0, posinf=0.0, neginf=0.0).astype(np.float32)
    valid = np.isfinite(depth) & (depth < 300)
    disp = np.where(valid, np.rint(DEPTH_MAP_FACTOR * BF / np.where(valid, depth, 1.0)), 0.0)
    if invalid_depth or zero_flow:
        r = np.random.default_rng(seed + 31 * k + 5).random((2, h, w))
        if invalid_depth:
            disp = np.where(r[0] < invalid_depth, 0.0, disp)
        if zero_flow:
            flow[r[1] < zero_flow] = 0.0
    if drop_masks and k in drop_masks:
        label = np.where(np.isin(label, list(drop_masks[k])), 0, label).astype(np.int32)
    return dict(gray=make_gray(seed + 1000 + k, w, h), depth_raw=np.ascontiguousarray(disp.astype(np.float32)), flow=np.ascontiguousarray(flow),
                mask=np.ascontiguousarray(label), Tcw=np.linalg.inv(T_wc), Tcw_next=np.linalg.inv(T_wc1), depth_true=depth)


def object_motion(ob, k=0):
    """World-frame rigid motion H of an object from frame k to k+1: X' = R_d (X - c_k) + c_{k+1}, R_d = yaw by the yaw rate
    (a pure translation by v when the object does not turn)."""
    H = np.eye(4)
    rate = ob.get("yaw_rate", 0.0)
    if rate:
        _, c0 = object_pose(ob, k)
        Rd = _yaw_R(rate)
        H[:3, :3] = Rd
        H[:3, 3] = (c0 + ob["v"]) - Rd @ c0
    else:
        H[:3, 3] = ob["v"]
    return H


# ---- the sequence bench.py times, the parity leg checks and tests/test_bench_sequence_gpu.py replays (one definition) -------------------------
# SURVEY.md 8d "synthetic inputs": flow noise N(0, 0.3^2) px, 2 % invalid depth, 1 % exactly-zero flow, 5 moving boxes (4 turning), one
# instance mask missing for two frames (exercises UpdateMask), one object leaving and one entering
BENCH_FLOW_SIGMA, BENCH_INVALID_DEPTH, BENCH_ZERO_FLOW, BENCH_N_OBJECTS, BENCH_BOX_DEPTH = 0.3, 0.02, 0.01, 5, 0.9
BENCH_MAX_FRAMES = 160          # the objects stay in view that long
KITTI0000_FRAMES = 153          # example/vdo_slam.cc:95-96


def bench_events(warmup, steps):
    """Where the 8d events fall: SURVEY's frames (mask dropped at 30 and 31, object 2 leaves at 60, object 5 enters at 80) for a
    KITTI-0000-length run, pulled inside the timed window [warmup, warmup+steps) whatever `steps` is."""
    drop_at = min(30, warmup + 3)
    leave_at = min(60, warmup + max(2, (2 * steps) // 5))
    enter_at = min(80, warmup + max(4, (3 * steps) // 5))
    return {drop_at: {1}, drop_at + 1: {1}}, leave_at, enter_at


def bench_spec(warmup, steps, seed=0):
    """Everything that defines the bench sequence of a (warmup, steps) run: a plain dict (picklable, hashable through repr)."""
    drop, leave_at, enter_at = bench_events(warmup, steps)
    return dict(n_seq=min(warmup + steps, BENCH_MAX_FRAMES), drop_masks={k: sorted(v) for k, v in drop.items()}, leave_at=leave_at, enter_at=enter_at, seed=int(seed))


def bench_objects(spec):
    return survey_objects(leave_at=spec["leave_at"], enter_at=spec["enter_at"], box_depth=BENCH_BOX_DEPTH)


def render_bench_frame(spec, k):
    Ts = camera_poses(spec["n_seq"])
    return render_frame(k, Ts, bench_objects(spec), flow_sigma=BENCH_FLOW_SIGMA, seed=spec["seed"], invalid_depth=BENCH_INVALID_DEPTH, zero_flow=BENCH_ZERO_FLOW,
                        drop_masks={k_: set(v) for k_, v in spec["drop_masks"].items()})


FRAME_KEYS = ("gray", "depth_raw", "flow", "mask", "Tcw", "Tcw_next")


def _render_worker(spec_json, out_dir, k0, stride, n=None):
    """child process: frames k0, k0 + stride, ... (below n) of the sequence into out_dir/f{k}.npz (numpy only)"""
    import json
    import os
    spec = json.loads(spec_json)
    spec["drop_masks"] = {int(k): v for k, v in spec["drop_masks"].items()}
    for k in range(int(k0), min(spec["n_seq"], int(n)) if n else spec["n_seq"], int(stride)):
        fr = render_bench_frame(spec, k)
        tmp = os.path.join(out_dir, f".f{k}.tmp.npz")
        np.savez(tmp, **{q: fr[q] for q in FRAME_KEYS})
        os.replace(tmp, os.path.join(out_dir, f"f{k}.npz"))


def load_bench_frame(out_dir, k):
    import os
    z = np.load(os.path.join(out_dir, f"f{k}.npz"))
    return {q: z[q] for q in FRAME_KEYS}


def render_bench_sequence(spec, out_dir, workers=None, timeout_s=900, n=None):
    """All frames of the sequence as files out_dir/f{k}.npz, rendered by `workers` numpy-only child processes started with their own command line
    (a frame is ~1 s of numpy on one core; no fork of a process that holds the HIP runtime, no re-import of the caller's __main__).  Returns the
    frames as a list of dicts."""
    import json
    import os
    import subprocess
    import sys
    n = spec["n_seq"] if n is None else min(int(n), spec["n_seq"])      # (n: only the first n frames of the sequence)
    if workers is None:
        workers = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                workers = min(workers, max(1, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
    workers = max(1, min(int(workers), 16, n))
    os.makedirs(out_dir, exist_ok=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sj = json.dumps(spec)
    code = "import sys; sys.path.insert(0, sys.argv[1]); from vdo_slam_amd.synth_seq import _render_worker; _render_worker(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6])"
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", code, root, sj, out_dir, str(w), str(workers), str(n)], env=env, stdout=subprocess.DEVNULL) for w in range(workers)]
    try:
        for pr in procs:
            if pr.wait(timeout=timeout_s) != 0:
                raise RuntimeError("render_bench_sequence: a render worker failed")
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return [load_bench_frame(out_dir, k) for k in range(n)]
