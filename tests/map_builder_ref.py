"""TEST INFRASTRUCTURE (not part of the product package).  Synthetic ``Map`` (reference include/Map.h:35-84) in flat-array form: per-frame static /
dynamic features with depths and world points, tracklets, camera poses and rigid motions — the
INPUT of ``Optimizer::FullBatchOptimization`` / ``PartialBatchOptimization``.  Used to test the
C++ host classes (vdo_slam_amd/host) end to end."""
from __future__ import annotations

import ctypes as C

import numpy as np

from vdo_slam_amd import _capi as K
from vdo_slam_amd import synth
from vdo_slam_amd.synth import KITTI_K, iso, iso_apply, iso_inv, iso_mul, rotvec_to_R


def _T44(T12):
    M = np.zeros(T12.shape[:-1] + (4, 4), np.float32)
    M[..., :3, :3] = T12[..., :9].reshape(T12.shape[:-1] + (3, 3))
    M[..., :3, 3] = T12[..., 9:]
    M[..., 3, 3] = 1
    return M


def make_map(n_frames=10, n_static=300, n_objects=2, dyn_tracks_per_object=30, seed=1, meas_sigma=0.05, short_track_frac=0.15):
    rng = np.random.default_rng(seed)
    F, K = n_frames, n_objects
    fx, fy, cx, cy = KITTI_K
    yaw = np.cumsum(0.01 * np.sin(np.arange(F) * 0.15))
    R = rotvec_to_R(np.stack([np.zeros(F), yaw, np.zeros(F)], -1))
    step = (R @ np.array([0, 0, 0.8])[None, :, None])[..., 0]
    cam_gt = iso(R, np.cumsum(step, 0) - step[0])
    dR = rotvec_to_R(rng.normal(0, 0.005, (F, 3))); dt = rng.normal(0, 0.02, (F, 3)); dR[0] = np.eye(3); dt[0] = 0
    cam_init = iso_mul(cam_gt, iso(dR, dt))
    cam44 = _T44(cam_init)                                      # float32 Map poses (T_wc)
    cam_init32 = iso(cam44[:, :3, :3].astype(np.float64), cam44[:, :3, 3].astype(np.float64))

    def observe(f, Xw):
        Xc = iso_apply(iso_inv(cam_gt[f]), Xw) + rng.normal(0, meas_sigma, 3)
        z = np.float32(Xc[2])
        u = np.float32(fx * Xc[0] / Xc[2] + cx); v = np.float32(fy * Xc[1] / Xc[2] + cy)
        # Get3DinWorld with the current (initial) pose, fp32
        xc = np.array([(u - np.float32(cx)) * z * (np.float32(1) / np.float32(fx)), (v - np.float32(cy)) * z * (np.float32(1) / np.float32(fy)), z], np.float32)
        xw = (cam44[f, :3, :3] @ xc + cam44[f, :3, 3]).astype(np.float32)
        return (u, v), z, xw

    feats = [dict(sta_uv=[], sta_d=[], sta_xw=[], dyn_uv=[], dyn_d=[], dyn_xw=[]) for _ in range(F)]
    tr_sta, tr_dyn, obj_of_dyn = [], [], []
    # static tracks (some shorter than 3 frames -> must be ignored by the builder)
    for _ in range(n_static):
        L = int(2 if rng.random() < short_track_frac else min(3 + rng.geometric(0.25) - 1, F))
        s = int(rng.integers(0, F - L + 1))
        mid = min(s + L // 2, F - 1)
        Xw = iso_apply(cam_gt[mid], np.array([rng.uniform(-20, 20), rng.uniform(-3, 3), rng.uniform(5, 40)]))
        tr = []
        for f in range(s, s + L):
            uv, z, xw = observe(f, Xw)
            tr.append((f, len(feats[f]["sta_uv"])))
            feats[f]["sta_uv"].append(uv); feats[f]["sta_d"].append(z); feats[f]["sta_xw"].append(xw)
        tr_sta.append(tr)
    # objects: constant body twist
    obj_pose = np.zeros((K, F, 12))
    for k in range(K):
        obj_pose[k, 0] = iso(rotvec_to_R(np.array([0, rng.uniform(-0.3, 0.3), 0])), np.array([rng.uniform(-6, 6), rng.uniform(-0.5, 0.5), rng.uniform(10, 25)]))
        delta = iso(rotvec_to_R(np.array([0, rng.uniform(-0.05, 0.05), 0])), np.array([0, 0, rng.uniform(0.3, 1.2)]))
        for f in range(1, F):
            obj_pose[k, f] = iso_mul(obj_pose[k, f - 1], delta)
    for k in range(K):
        for _ in range(dyn_tracks_per_object):
            L = int(2 if rng.random() < short_track_frac else min(3 + rng.geometric(0.25) - 1, F))
            s = int(rng.integers(0, F - L + 1))
            body = np.array([rng.uniform(-1, 1), rng.uniform(-0.8, 0.8), rng.uniform(-2, 2)])
            tr = []
            for f in range(s, s + L):
                uv, z, xw = observe(f, iso_apply(obj_pose[k, f], body))
                tr.append((f, len(feats[f]["dyn_uv"])))
                feats[f]["dyn_uv"].append(uv); feats[f]["dyn_d"].append(z); feats[f]["dyn_xw"].append(xw)
            tr_dyn.append(tr); obj_of_dyn.append(k + 1)
    # rigid motions: [0] camera motion T_wc(i)^-1 T_wc(i+1) (fp32, from the initial poses), [1..K] objects
    rm, rml = [], []
    for i in range(F - 1):
        camm = (np.linalg.inv(cam44[i].astype(np.float64)) @ cam44[i + 1].astype(np.float64)).astype(np.float32)
        mots = [camm]
        for k in range(K):
            H = iso_mul(obj_pose[k, i + 1], iso_inv(obj_pose[k, i]))
            mots.append(_T44(H))
        rm.append(np.stack(mots)); rml.append(np.arange(K + 1, dtype=np.int32))
    return dict(n_frames=F, K=np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32), cam_pose=cam44, cam_gt=cam_gt,
                feats=feats, tr_sta=tr_sta, tr_dyn=tr_dyn, obj_of_dyn=np.array(obj_of_dyn, np.int32), rigid_motion=rm, rm_label=rml)


def _quat_roundtrip(R, positive_w):
    """Eigen Quaterniond(R) -> normalise (-> w>=0) -> rotation matrix (what toSE3Quat/Isometry3 produce)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t
        x = (R[2, 1] - R[1, 2]) * t; y = (R[0, 2] - R[2, 0]) * t; z = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]: i = 1
        if R[2, 2] > R[i, i]: i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        c = [0.0, 0.0, 0.0]; c[i] = 0.5 * t; t = 0.5 / t
        w = (R[k, j] - R[j, k]) * t; c[j] = (R[j, i] + R[i, j]) * t; c[k] = (R[k, i] + R[i, k]) * t
        x, y, z = c
    if positive_w and w < 0: x, y, z, w = -x, -y, -z, -w
    q = np.array([x, y, z, w]); q = q / np.sqrt((q * q).sum())
    return synth.quat_to_R(q)


def _iso12(T44):
    T = T44.astype(np.float64)
    return np.concatenate([_quat_roundtrip(T[:3, :3], True).ravel(), T[:3, 3]])


def map_to_graph(m, partial_window=None):
    """Python restatement of the reference graph builder (src/Optimizer.cc:1259-1766, or the
    partial-batch variant :44-637 when ``partial_window`` is given) -> synth.BAGraph.
    TEST-SIDE code: feeds the oracle with the graph the C++ host class must build."""
    F = m["n_frames"]
    fx, fy, cx, cy = [np.float32(v) for v in (m["K"][0, 0], m["K"][1, 1], m["K"][0, 2], m["K"][1, 2])]
    full = partial_window is None
    sig = dict(cam=np.float32(0.001), s3=np.float32(80), smo=np.float32(0.001), obj=np.float32(100), d3=np.float32(80)) if full else \
        dict(cam=np.float32(0.0001), s3=np.float32(16))

    def xc(uv, z):
        z = np.float32(z)
        return np.array([(np.float32(uv[0]) - cx) * z * (np.float32(1) / fx), (np.float32(uv[1]) - cy) * z * (np.float32(1) / fy), z], np.float32).astype(np.float64)

    pose, point = [], []
    eb, et, ep, pr = [], [], [], []
    labS = [np.full(len(f["sta_uv"]), -1) for f in m["feats"]]; posS = [a.copy() for a in labS]; mkS = [a.copy() for a in labS]
    labD = [np.full(len(f["dyn_uv"]), -1) for f in m["feats"]]; posD = [a.copy() for a in labD]; mkD = [a.copy() for a in labD]
    for t, tr in enumerate(m["tr_sta"]):
        if len(tr) < 3: continue
        for k, (f, j) in enumerate(tr): labS[f][j] = t; posS[f][j] = k
    for t, tr in enumerate(m["tr_dyn"]):
        if len(tr) < 3: continue
        for k, (f, j) in enumerate(tr): labD[f][j] = t; posD[f][j] = k
    I6 = np.eye(6)
    start = 0 if full else F - partial_window
    vid = [[-1] * len(l) for l in m["rm_label"]]
    pre = -1
    cam_idx = []
    ident = np.eye(4, dtype=np.float32)
    for i in range(start, F):
        cam = len(pose); pose.append(_iso12(m["cam_pose"][i]))
        cam_idx.append(cam)
        if i == start and (full or F == partial_window):
            pr.append((cam, _iso12(m["cam_pose"][i]), I6 * (100000.0 if full else 1.0 / 0.0000001)))
        if i != start:
            if full: vid[i - 1][0] = cam
            ep.append((pre, cam, _iso12(m["rigid_motion"][i - 1][0]), I6 / float(sig["cam"])))
        fe = m["feats"][i]
        for j in range(len(labS[i])):
            if labS[i][j] == -1: continue
            tr = m["tr_sta"][labS[i][j]]; ps = posS[i][j]
            if ps == 0:
                pt = len(point); point.append(fe["sta_xw"][j].astype(np.float64))
            else:
                pf, pj = tr[ps - 1]
                pt = mkS[pf][pj] if pf >= start else -1
            if pt < 0: continue
            eb.append((cam, pt, xc(fe["sta_uv"][j], fe["sta_d"][j]), 1.0 / float(sig["s3"])))
            mkS[i][j] = pt
        if full:
            if i == 0:
                for j in range(len(labD[i])):
                    if labD[i][j] == -1: continue
                    pt = len(point); point.append(fe["dyn_xw"][j].astype(np.float64))
                    eb.append((cam, pt, xc(fe["dyn_uv"][j], fe["dyn_d"][j]), 1.0 / float(sig["d3"])))
                    mkD[i][j] = pt
            else:
                nm = len(m["rigid_motion"][i - 1])
                uniq = [-1] * (nm - 1)
                for j in range(1, nm):
                    mv = len(pose); pose.append(_iso12(ident))
                    if i > 2:
                        trace = -1
                        for k, lab in enumerate(m["rm_label"][i - 2]):
                            if lab == m["rm_label"][i - 1][j]: trace = k; break
                        if trace != -1 and vid[i - 2][trace] >= 0:
                            ep.append((vid[i - 2][trace], mv, _iso12(ident), I6 / float(sig["smo"])))
                    uniq[j - 1] = mv; vid[i - 1][j] = mv
                for j in range(len(labD[i])):
                    if labD[i][j] == -1: continue
                    t = labD[i][j]; tr = m["tr_dyn"][t]; ps = posD[i][j]
                    objpos = -1
                    for k in range(1, len(m["rm_label"][i - 1])):
                        if m["rm_label"][i - 1][k] == m["obj_of_dyn"][t]: objpos = uniq[k - 1]; break
                    if objpos == -1 and ps != 0: continue
                    pt = len(point); point.append(fe["dyn_xw"][j].astype(np.float64))
                    eb.append((cam, pt, xc(fe["dyn_uv"][j], fe["dyn_d"][j]), 1.0 / float(sig["d3"])))
                    if ps != 0:
                        pf, pj = tr[ps - 1]
                        if mkD[pf][pj] >= 0: et.append((mkD[pf][pj], pt, objpos, 1.0 / float(sig["obj"])))
                    mkD[i][j] = pt
        pre = cam
    A = lambda l, dt: np.asarray(l, dtype=dt)
    g = synth.BAGraph(
        pose=A(pose, np.float64).reshape(-1, 12), point=A(point, np.float64).reshape(-1, 3),
        eb_pose=A([e[0] for e in eb], np.int32), eb_point=A([e[1] for e in eb], np.int32),
        eb_z=np.ascontiguousarray(A([e[2] for e in eb], np.float64).reshape(-1, 3).T), eb_w=A([e[3] for e in eb], np.float64),
        et_p1=A([e[0] for e in et], np.int32), et_p2=A([e[1] for e in et], np.int32), et_pose=A([e[2] for e in et], np.int32),
        et_z=np.zeros((3, len(et))), et_w=A([e[3] for e in et], np.float64),
        ep_i=A([e[0] for e in ep], np.int32), ep_j=A([e[1] for e in ep], np.int32),
        ep_z=A([e[2] for e in ep], np.float64).reshape(-1, 12), ep_info=A([e[3].ravel() for e in ep], np.float64).reshape(-1, 36),
        pr_pose=A([e[0] for e in pr], np.int32), pr_z=A([e[1] for e in pr], np.float64).reshape(-1, 12),
        pr_info=A([e[2].ravel() for e in pr], np.float64).reshape(-1, 36), n_cam=F - start)
    return g, dict(vid=vid, mkS=mkS, mkD=mkD, start=start, cam_idx=cam_idx)


# ---- the flat form both the product's host hook (vdo_slam_amd/host/host_capi.cc host_batch_optimization) and the reference-side hook
# (oracle/ref/ref_g2o_entry.cc ref_batch_optimization) take: they fill a VDO_SLAM::Map from it and call the batch optimiser ------------------
class HostMapFlat(C.Structure):
    _fields_ = [("n_frames", C.c_int), ("K", K.c_float_p), ("cam_pose", K.c_float_p),
                ("sta_cnt", K.c_int32_p), ("sta_uv", K.c_float_p), ("sta_d", K.c_float_p), ("sta_xw", K.c_float_p),
                ("n_tr_sta", C.c_int), ("tr_sta_len", K.c_int32_p), ("tr_sta_pairs", K.c_int32_p),
                ("dyn_cnt", K.c_int32_p), ("dyn_uv", K.c_float_p), ("dyn_d", K.c_float_p), ("dyn_xw", K.c_float_p),
                ("n_tr_dyn", C.c_int), ("tr_dyn_len", K.c_int32_p), ("tr_dyn_pairs", K.c_int32_p), ("obj_of_dyn", K.c_int32_p),
                ("rm_cnt", K.c_int32_p), ("rm", K.c_float_p), ("rm_label", K.c_int32_p)]



def _f(a): return np.ascontiguousarray(a, dtype=np.float32)
def _i(a): return np.ascontiguousarray(a, dtype=np.int32)


def flatten_map(m):
    keep = []
    def fp(a): a = _f(a); keep.append(a); return a.ctypes.data_as(K.c_float_p)
    def ip(a): a = _i(a); keep.append(a); return a.ctypes.data_as(K.c_int32_p)
    fe = m["feats"]
    cat = lambda key, w: np.concatenate([np.asarray(f[key], np.float32).reshape(-1, w) for f in fe]) if sum(len(f[key]) for f in fe) else np.zeros((0, w), np.float32)
    s = HostMapFlat()
    s.n_frames = m["n_frames"]; s.K = fp(m["K"]); s.cam_pose = fp(m["cam_pose"])
    s.sta_cnt = ip([len(f["sta_uv"]) for f in fe]); s.sta_uv = fp(cat("sta_uv", 2)); s.sta_d = fp(cat("sta_d", 1)); s.sta_xw = fp(cat("sta_xw", 3))
    s.n_tr_sta = len(m["tr_sta"]); s.tr_sta_len = ip([len(t) for t in m["tr_sta"]]); s.tr_sta_pairs = ip([p for t in m["tr_sta"] for p in t])
    s.dyn_cnt = ip([len(f["dyn_uv"]) for f in fe]); s.dyn_uv = fp(cat("dyn_uv", 2)); s.dyn_d = fp(cat("dyn_d", 1)); s.dyn_xw = fp(cat("dyn_xw", 3))
    s.n_tr_dyn = len(m["tr_dyn"]); s.tr_dyn_len = ip([len(t) for t in m["tr_dyn"]]); s.tr_dyn_pairs = ip([p for t in m["tr_dyn"] for p in t])
    s.obj_of_dyn = ip(m["obj_of_dyn"])
    s.rm_cnt = ip([len(r) for r in m["rigid_motion"]]); s.rm = fp(np.concatenate(m["rigid_motion"])); s.rm_label = ip(np.concatenate(m["rm_label"]))
    return s, keep
