"""Synthetic inputs for the hot path: batch dynamic-SLAM factor graphs in
the SoA layout of ``include/vdo_slam_hip.h`` (``vdo_ba_graph``), built the way
``Optimizer::FullBatchOptimization`` builds them (reference src/Optimizer.cc:1232-1768):

* one ``VertexSE3`` per camera frame, prior on the first (info 1e5), odometry ``EdgeSE3``
  between consecutive cameras (info I/σ²_cam);
* static landmarks: one ``VertexPointXYZ`` per track, one ``EdgeSE3PointXYZ`` per observation;
* per (object, frame>=1) one motion ``VertexSE3`` initialised to identity, smoothness
  ``EdgeSE3`` (identity measurement) between consecutive motions of an object from frame 3 on;
* dynamic tracks: one ``VertexPointXYZ`` per observation, ``EdgeSE3PointXYZ`` to the camera and
  a ``LandmarkMotionTernaryEdge`` to the previous observation and the object's motion.

Edges are emitted frame by frame (camera-major), which is also the order the reference inserts
them in.  All arrays are numpy, fp64 / int32, C-contiguous.
"""
from __future__ import annotations

import dataclasses
import numpy as np

# constants of the full-batch builder (src/Optimizer.cc:1330-1352, 1370)
SIGMA2_CAM = float(np.float32(0.001))
SIGMA2_3D_STA = 80.0
SIGMA2_OBJ_SMO = float(np.float32(0.001))
SIGMA2_OBJ = 100.0
SIGMA2_3D_DYN = 80.0
HUBER_DELTA = float(np.float32(0.0001))
PRIOR_INFO = 100000.0


@dataclasses.dataclass
class BAGraph:
    pose: np.ndarray      # [P,12]
    point: np.ndarray     # [L,3]
    eb_pose: np.ndarray
    eb_point: np.ndarray
    eb_z: np.ndarray      # [3,Eb]
    eb_w: np.ndarray
    et_p1: np.ndarray
    et_p2: np.ndarray
    et_pose: np.ndarray
    et_z: np.ndarray      # [3,Et]
    et_w: np.ndarray
    ep_i: np.ndarray
    ep_j: np.ndarray
    ep_z: np.ndarray      # [Ep,12]
    ep_info: np.ndarray   # [Ep,36]
    pr_pose: np.ndarray
    pr_z: np.ndarray
    pr_info: np.ndarray
    huber_eb: float = HUBER_DELTA
    huber_et: float = HUBER_DELTA
    huber_ep: float = HUBER_DELTA
    # ground truth (not part of the C struct)
    pose_gt: np.ndarray | None = None
    point_gt: np.ndarray | None = None
    n_cam: int = 0

    @property
    def n_pose(self): return self.pose.shape[0]
    @property
    def n_point(self): return self.point.shape[0]
    @property
    def n_eb(self): return self.eb_pose.shape[0]
    @property
    def n_et(self): return self.et_p1.shape[0]
    @property
    def n_ep(self): return self.ep_i.shape[0]
    @property
    def n_prior(self): return self.pr_pose.shape[0]

    def sweep_bytes(self) -> int:
        """Algorithmic HBM bytes of one linearisation sweep."""
        return (208 * self.n_eb + 452 * self.n_et + 640 * (self.n_ep + self.n_prior)
                + 432 * self.n_pose + 96 * self.n_point)


# ----------------------------------------------------------------------------- SE(3) helpers
def quat_to_R(q: np.ndarray) -> np.ndarray:
    """q[...,4] = (x,y,z,w) -> R[...,3,3] (same formula as Eigen toRotationMatrix)."""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - (ty * y + tz * z); R[..., 0, 1] = ty * x - tz * w; R[..., 0, 2] = tz * x + ty * w
    R[..., 1, 0] = ty * x + tz * w; R[..., 1, 1] = 1 - (tx * x + tz * z); R[..., 1, 2] = tz * y - tx * w
    R[..., 2, 0] = tz * x - ty * w; R[..., 2, 1] = tz * y + tx * w; R[..., 2, 2] = 1 - (tx * x + ty * y)
    return R


def rotvec_to_R(rv: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(rv, axis=-1, keepdims=True)
    half = 0.5 * th
    k = np.where(th > 1e-12, np.sin(half) / np.maximum(th, 1e-300), 0.5)
    q = np.concatenate([rv * k, np.cos(half)], axis=-1)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    return quat_to_R(q)


def iso(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """pack (R[...,3,3], t[...,3]) -> [...,12]"""
    return np.concatenate([R.reshape(R.shape[:-2] + (9,)), t], axis=-1)


def iso_R(T): return T[..., :9].reshape(T.shape[:-1] + (3, 3))
def iso_t(T): return T[..., 9:12]


def iso_mul(A, B):
    RA, RB = iso_R(A), iso_R(B)
    return iso(RA @ RB, (RA @ iso_t(B)[..., None])[..., 0] + iso_t(A))


def iso_inv(A):
    Rt = np.swapaxes(iso_R(A), -1, -2)
    return iso(Rt, -(Rt @ iso_t(A)[..., None])[..., 0])


def iso_apply(A, p):
    return (iso_R(A) @ p[..., None])[..., 0] + iso_t(A)


IDENT12 = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)


def _track_lengths(rng, n, mean_extra=3.0, min_len=3):
    # L ~ min_len + Geom, mean 6
    return min_len + rng.geometric(1.0 / (1.0 + mean_extra), size=n) - 1


def make_ba_graph(n_frames: int = 40, n_static: int = 2000, n_objects: int = 3,
                  dyn_tracks_per_object: int = 150, seed: int = 1,
                  outlier_frac: float = 0.05, meas_sigma: float = 0.05,
                  init_sigma_t: float = 0.02, init_sigma_r: float = 0.005, long_dyn_tracks: int = 0) -> BAGraph:
    """KITTI-shaped synthetic dynamic-SLAM factor graph (SURVEY.md §8d "Batch graphs").  long_dyn_tracks: that many dynamic tracks run over ALL frames
    (an object point followed through the whole sequence: the longest chain the batch solver's tiles have to hold)."""
    rng = np.random.default_rng(seed)
    F = n_frames
    # ---- cameras: forward 0.8 m/frame, sinusoidal yaw <= 0.01 rad/frame
    yaw_rate = 0.01 * np.sin(np.arange(F) * 0.15)
    yaw = np.cumsum(yaw_rate)
    cam_R = rotvec_to_R(np.stack([np.zeros(F), yaw, np.zeros(F)], -1))
    step = (cam_R @ np.array([0, 0, 0.8])[None, :, None])[..., 0]
    cam_t = np.cumsum(step, axis=0) - step[0]
    cam_gt = iso(cam_R, cam_t)
    # ---- initial camera estimates: truth (+) N(0,(0.02 m, 0.005 rad)^2); frame 0 exact (origin)
    dR = rotvec_to_R(rng.normal(0, init_sigma_r, (F, 3)))
    dt = rng.normal(0, init_sigma_t, (F, 3))
    dR[0] = np.eye(3); dt[0] = 0
    cam_init = iso_mul(cam_gt, iso(dR, dt))

    # ---- static landmarks
    Ls = n_static
    s_len = np.minimum(_track_lengths(rng, Ls), F)
    s_start = rng.integers(0, np.maximum(F - s_len + 1, 1))
    mid = np.minimum(s_start + s_len // 2, F - 1)
    pc = np.stack([rng.uniform(-20, 20, Ls), rng.uniform(-3, 3, Ls), rng.uniform(5, 40, Ls)], -1)
    Xs = iso_apply(cam_gt[mid], pc)
    # observations (frame-major order)
    reps = s_len
    lm_idx = np.repeat(np.arange(Ls), reps)
    frame = np.repeat(s_start, reps) + (np.arange(reps.sum()) - np.repeat(np.cumsum(reps) - reps, reps))
    order = np.lexsort((lm_idx, frame))
    lm_idx, frame = lm_idx[order], frame[order]

    def observe(cam_frames, Xw):
        z = iso_apply(iso_inv(cam_gt[cam_frames]), Xw)
        noise = rng.normal(0, meas_sigma, z.shape)
        outl = rng.random(z.shape[0]) < outlier_frac
        noise[outl] *= 20.0
        return (z + noise).astype(np.float32).astype(np.float64)   # Map stores fp32

    zs = observe(frame, Xs[lm_idx])
    # initial static point = first observation back-projected with the initial pose
    first = np.full(Ls, -1, dtype=np.int64)
    # frame-major order => first occurrence of each landmark is its first frame
    uniq, first_pos = np.unique(lm_idx, return_index=True)
    first[uniq] = first_pos
    Xs_init = iso_apply(cam_init[frame[first]], zs[first]).astype(np.float32).astype(np.float64)

    # ---- objects
    K = n_objects
    P_cam = F
    n_mot = K * (F - 1)

    def mot_vertex(k, f):  # motion of object k between frame f-1 and f (f>=1)
        return P_cam + (f - 1) * K + k

    obj_pose = np.zeros((K, F, 12))
    H_gt = np.zeros((K, F, 12))
    H_gt[:] = IDENT12
    for k in range(K):
        T0 = iso(rotvec_to_R(np.array([0, rng.uniform(-0.3, 0.3), 0])),
                 np.array([rng.uniform(-6, 6), rng.uniform(-0.5, 0.5), rng.uniform(10, 25)]))
        speed = rng.uniform(0.3, 1.2)
        yawr = rng.uniform(-0.05, 0.05)
        delta = iso(rotvec_to_R(np.array([0, yawr, 0])), np.array([0, 0, speed]))
        obj_pose[k, 0] = T0
        for f in range(1, F):
            obj_pose[k, f] = iso_mul(obj_pose[k, f - 1], delta)
            H_gt[k, f] = iso_mul(obj_pose[k, f], iso_inv(obj_pose[k, f - 1]))
    Td = dyn_tracks_per_object * K
    d_obj = np.repeat(np.arange(K), dyn_tracks_per_object)
    d_len = np.minimum(_track_lengths(rng, Td), F)
    d_start = rng.integers(0, np.maximum(F - d_len + 1, 1))
    if long_dyn_tracks:
        d_len[:long_dyn_tracks] = F; d_start[:long_dyn_tracks] = 0
    body = np.stack([rng.uniform(-1, 1, Td), rng.uniform(-0.8, 0.8, Td), rng.uniform(-2, 2, Td)], -1)
    reps = d_len
    tr_idx = np.repeat(np.arange(Td), reps)
    pos_in_track = np.arange(reps.sum()) - np.repeat(np.cumsum(reps) - reps, reps)
    dframe = np.repeat(d_start, reps) + pos_in_track
    order = np.lexsort((tr_idx, dframe))
    tr_idx, dframe, pos_in_track = tr_idx[order], dframe[order], pos_in_track[order]
    Xd = iso_apply(obj_pose[d_obj[tr_idx], dframe], body[tr_idx]) if Td else np.zeros((0, 3))
    zd = observe(dframe, Xd) if Td else np.zeros((0, 3))
    n_dyn_pts = tr_idx.shape[0]
    dyn_pt_id = Ls + np.arange(n_dyn_pts)
    Xd_init = iso_apply(cam_init[dframe], zd).astype(np.float32).astype(np.float64) if Td else np.zeros((0, 3))
    # previous observation of the same track: index in frame-major arrays
    key = tr_idx.astype(np.int64) * (F + 1) + pos_in_track
    sorter = np.argsort(key)
    prev_key = key - 1
    loc = np.searchsorted(key[sorter], prev_key)
    loc = np.clip(loc, 0, max(n_dyn_pts - 1, 0))
    has_prev = pos_in_track > 0
    prev_idx = sorter[loc] if n_dyn_pts else loc

    # ---- assemble vertices
    P = P_cam + n_mot
    pose = np.empty((P, 12))
    pose[:P_cam] = cam_init
    pose[P_cam:] = IDENT12                       # motions start at identity (Optimizer.cc:1581)
    pose_gt = np.empty((P, 12))
    pose_gt[:P_cam] = cam_gt
    for f in range(1, F):
        for k in range(K):
            pose_gt[mot_vertex(k, f)] = H_gt[k, f]
    point = np.concatenate([Xs_init, Xd_init], 0)
    point_gt = np.concatenate([Xs, Xd], 0)

    # ---- binary edges, camera-major: merge static + dynamic observations per frame
    eb_pose = np.concatenate([frame, dframe]).astype(np.int32)
    eb_point = np.concatenate([lm_idx, dyn_pt_id]).astype(np.int32)
    eb_zz = np.concatenate([zs, zd], 0)
    eb_w = np.concatenate([np.full(frame.shape[0], 1.0 / SIGMA2_3D_STA), np.full(n_dyn_pts, 1.0 / SIGMA2_3D_DYN)])
    o = np.argsort(eb_pose, kind="stable")
    eb_pose, eb_point, eb_zz, eb_w = eb_pose[o], eb_point[o], eb_zz[o], eb_w[o]

    # ---- ternary edges
    sel = np.nonzero(has_prev)[0]
    et_p1 = dyn_pt_id[prev_idx[sel]].astype(np.int32)
    et_p2 = dyn_pt_id[sel].astype(np.int32)
    et_pose = (P_cam + (dframe[sel] - 1) * K + d_obj[tr_idx[sel]]).astype(np.int32)
    et_z = np.zeros((3, sel.shape[0]))
    et_w = np.full(sel.shape[0], 1.0 / SIGMA2_OBJ)

    # ---- pose-pose edges: odometry + motion smoothness
    ep_i, ep_j, ep_z, ep_info = [], [], [], []
    odo = iso_mul(iso_inv(cam_init[:-1]), cam_init[1:])
    I6 = np.eye(6)
    for f in range(1, F):
        ep_i.append(f - 1); ep_j.append(f); ep_z.append(odo[f - 1]); ep_info.append((I6 / SIGMA2_CAM).ravel())
        if f >= 3:
            for k in range(K):
                ep_i.append(mot_vertex(k, f - 1)); ep_j.append(mot_vertex(k, f))
                ep_z.append(IDENT12); ep_info.append((I6 / SIGMA2_OBJ_SMO).ravel())
    ep_i = np.asarray(ep_i, dtype=np.int32); ep_j = np.asarray(ep_j, dtype=np.int32)
    ep_z = np.asarray(ep_z, dtype=np.float64).reshape(-1, 12)
    ep_info = np.asarray(ep_info, dtype=np.float64).reshape(-1, 36)

    g = BAGraph(
        pose=np.ascontiguousarray(pose), point=np.ascontiguousarray(point),
        eb_pose=eb_pose, eb_point=eb_point, eb_z=np.ascontiguousarray(eb_zz.T), eb_w=np.ascontiguousarray(eb_w),
        et_p1=et_p1, et_p2=et_p2, et_pose=et_pose, et_z=np.ascontiguousarray(et_z), et_w=et_w,
        ep_i=ep_i, ep_j=ep_j, ep_z=ep_z, ep_info=ep_info,
        pr_pose=np.zeros(1, dtype=np.int32), pr_z=cam_init[:1].copy(),
        pr_info=(I6 * PRIOR_INFO).reshape(1, 36).copy(),
        pose_gt=pose_gt, point_gt=point_gt, n_cam=P_cam)
    return g


# ------------------------------------------------------------------------ per-frame problems
KITTI_K = (721.5377, 721.5377, 609.5593, 172.854)   # example/kitti-0000-0013.yaml:8-16
KITTI_W, KITTI_H = 1242, 375


@dataclasses.dataclass
class Flow2Problem:
    """Inputs of Optimizer::PoseOptimizationFlow2Cam / PoseOptimizationFlow2 (SoA, fp64 holding
    fp32-representable values, as the reference converts cv::Mat float -> double)."""
    obs: np.ndarray        # [n,2]
    flow: np.ndarray       # [n,2]
    depth: np.ndarray      # [n]
    K: tuple
    Twl: np.ndarray        # [4,4]
    T0: np.ndarray         # [4,4]
    info_flow: float = 0.1
    info_prior: float = 0.3
    huber_delta: float = float(np.sqrt(np.float32(0.04)))     # const float deltaMono = sqrt(rp_thres)
    chi2_gate: float = float(np.float32(0.04))
    max_iterations: int = 100
    ref_quirks: int = 1
    T_true: np.ndarray | None = None

    @property
    def n(self): return self.obs.shape[0]


def _mat4(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def make_flow2_problem(n: int = 1200, seed: int = 1, is_object: bool = False, outlier_frac: float = 0.1,
                       flow_sigma: float = 0.3, init_sigma_t: float = 0.05, init_sigma_r: float = 0.004) -> Flow2Problem:
    """KITTI-shaped correspondences between the last and the current frame."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = KITTI_K
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    # last-frame pose T_lw (camera from world) and its inverse
    R_lw = rotvec_to_R(rng.normal(0, 0.05, 3))
    t_lw = rng.normal(0, 2.0, 3)
    T_lw = f32(_mat4(R_lw, t_lw))
    Rwl = T_lw[:3, :3].T
    Twl = _mat4(f32(Rwl), f32(-(T_lw[:3, :3].T @ T_lw[:3, 3])))     # float arithmetic as in the reference
    if is_object:
        u = rng.uniform(500, 700, n); v = rng.uniform(150, 260, n); z = rng.uniform(8, 20, n)
    else:
        u = rng.uniform(5, KITTI_W - 5, n); v = rng.uniform(5, KITTI_H - 5, n); z = rng.uniform(4, 40, n)
    obs = f32(np.stack([u, v], 1)); depth = f32(z)
    Xc = np.stack([(obs[:, 0] - cx) * depth / fx, (obs[:, 1] - cy) * depth / fy, depth], 1)
    Xw = Xc @ Twl[:3, :3].T + Twl[:3, 3]
    # true transform applied to world points in this frame: camera motion (and object motion)
    dR = rotvec_to_R(np.array([0.0, rng.uniform(-0.01, 0.01), 0.0]))
    dT = _mat4(dR, np.array([rng.normal(0, 0.02), rng.normal(0, 0.01), -0.8]))
    if is_object:
        dT = dT @ _mat4(rotvec_to_R(np.array([0, rng.uniform(-0.03, 0.03), 0])), np.array([rng.normal(0, 0.1), 0, rng.uniform(0.2, 0.8)]))
    T_true = dT @ T_lw
    Xn = Xw @ T_true[:3, :3].T + T_true[:3, 3]
    proj = np.stack([Xn[:, 0] / Xn[:, 2] * fx + cx, Xn[:, 1] / Xn[:, 2] * fy + cy], 1)
    flow = proj - obs + rng.normal(0, flow_sigma, (n, 2))
    outl = rng.random(n) < outlier_frac
    flow[outl] += rng.normal(0, 5.0, (int(outl.sum()), 2))
    T0 = T_true.copy()
    T0[:3, :3] = rotvec_to_R(rng.normal(0, init_sigma_r, 3)) @ T0[:3, :3]
    T0[:3, 3] += rng.normal(0, init_sigma_t, 3)
    return Flow2Problem(obs=obs, flow=f32(flow), depth=depth, K=KITTI_K, Twl=Twl, T0=f32(T0),
                        info_prior=0.5 if is_object else 0.3, max_iterations=200 if is_object else 100,
                        T_true=T_true)


def with_hub_points(g: "BAGraph", n_hubs: int, seed: int = 0, sigma: float = 0.05) -> "BAGraph":
    """g with `n_hubs` of its static points (never an end of a ternary edge) observed from EVERY camera: one more EdgeSE3PointXYZ from each camera that does not
    see them yet (fp32-representable measurements, like the graphs the reference builds).  Beyond 256 cameras such a point no longer fits a tile of the batch
    solver: a hub landmark (csrc/ba_hub.hip)."""
    import dataclasses
    rng = np.random.default_rng(seed)
    dyn = np.zeros(g.n_point, bool); dyn[g.et_p1] = True; dyn[g.et_p2] = True
    sta = np.nonzero(~dyn)[0]
    pick = sta[:: max(1, len(sta) // n_hubs)][:n_hubs]
    seen = set(zip(g.eb_pose.tolist(), g.eb_point.tolist()))
    ep, el, ez = [], [], []
    for l in pick:
        for c in range(g.n_cam):
            if (int(c), int(l)) in seen:
                continue
            R = g.pose[c, :9].reshape(3, 3); t = g.pose[c, 9:]
            ep.append(c); el.append(l); ez.append(np.float32(R.T @ (g.point[l] - t) + rng.normal(0, sigma, 3)).astype(np.float64))
    return dataclasses.replace(g, eb_pose=np.concatenate([g.eb_pose, np.array(ep, np.int32)]), eb_point=np.concatenate([g.eb_point, np.array(el, np.int32)]),
                               eb_z=np.ascontiguousarray(np.concatenate([g.eb_z, np.array(ez).T], 1)), eb_w=np.concatenate([g.eb_w, np.full(len(ep), g.eb_w[0])]))
