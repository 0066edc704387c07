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
