"""The C++ host classes (vdo_slam_amd/host: ORBextractor, Frame, Optimizer with the reference's
signatures) driven through libvdo_host.so and compared with the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import frontend_ref as R
from vdo_slam_amd import _capi as K
from tests import map_builder_ref as SM
from vdo_slam_amd import synth, synth_frames as SF

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HostMapFlat = SM.HostMapFlat


@pytest.fixture(scope="module")
def host():
    L = K.load_host_lib()
    L.host_batch_optimization.argtypes = [C.POINTER(HostMapFlat), C.c_int, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p, C.POINTER(K.LMStatsC)]
    return L


def _f(a): return np.ascontiguousarray(a, dtype=np.float32)
def _i(a): return np.ascontiguousarray(a, dtype=np.int32)


_flatten = SM.flatten_map


@pytest.mark.parametrize("window", [0, 8])
def test_optimizer_batch_matches_oracle_on_the_reference_built_graph(host, oracle, window):
    """Optimizer::FullBatchOptimization / PartialBatchOptimization (C++ host class, GPU solve) vs the
    oracle LM on the graph built by the Python restatement of the reference builder."""
    m = SM.make_map(n_frames=8 if window else 12, n_static=400, n_objects=2, dyn_tracks_per_object=40, seed=5)
    s, keep = _flatten(m)
    F = m["n_frames"]
    n_sta = sum(len(f["sta_uv"]) for f in m["feats"]); n_dyn = sum(len(f["dyn_uv"]) for f in m["feats"])
    n_rm = sum(len(r) for r in m["rigid_motion"])
    cam_out = np.zeros((F, 4, 4), np.float32); rm_out = np.zeros((n_rm, 4, 4), np.float32)
    sta_out = np.zeros((n_sta, 3), np.float32); dyn_out = np.zeros((max(n_dyn, 1), 3), np.float32)
    st = K.LMStatsC()
    assert host.host_batch_optimization(C.byref(s), window, _p(cam_out), _p(rm_out), _p(sta_out), _p(dyn_out), C.byref(st)) == 0
    # oracle on the same graph
    g, info = SM.map_to_graph(m, partial_window=window or None)
    gc, keep2 = K.graph_to_c(g)
    opt = K.LMOptionsC(100 if window else 300, 1e-3 if window else 1e-4, 0, 0, 0.0, 0)
    st_o = K.LMStatsC()
    pose_o = np.zeros_like(g.pose); point_o = np.zeros_like(g.point)
    assert oracle.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), K._dp(pose_o), K._dp(point_o), C.byref(st_o)) == 0
    assert st.iterations == st_o.iterations and st.total_trials == st_o.total_trials
    assert abs(st.final_chi2 - st_o.final_chi2) <= 1e-6 * st_o.final_chi2
    # refined camera poses (float32 in the Map): frame `start` is the gauge, the others must match
    start = info["start"]
    for i in range(start + (0 if window else 1), F):
        ref = pose_o[info["cam_idx"][i - start]]
        np.testing.assert_allclose(cam_out[i][:3, :3].ravel(), ref[:9], atol=2e-6)
        np.testing.assert_allclose(cam_out[i][:3, 3], ref[9:], rtol=1e-4, atol=1e-5)
    # refined static points
    off = np.cumsum([0] + [len(f["sta_uv"]) for f in m["feats"]])
    checked = 0
    for i in range(start, F):
        for j, mk in enumerate(info["mkS"][i]):
            if mk >= 0:
                np.testing.assert_allclose(sta_out[off[i] + j], point_o[mk], rtol=1e-4, atol=1e-4)
                checked += 1
    assert checked > 100


@pytest.mark.parametrize("window,shape", [(0, dict(n_frames=12, n_static=400, n_objects=2, dyn_tracks_per_object=40, seed=5)),
                                          (8, dict(n_frames=8, n_static=400, n_objects=2, dyn_tracks_per_object=40, seed=5)),
                                          (0, dict(n_frames=20, n_static=900, n_objects=3, dyn_tracks_per_object=60, seed=9))])
def test_optimizer_batch_equals_the_reference_source(host, window, shape):
    """Round 5, no oracle in between: the PRODUCT's Optimizer::FullBatchOptimization / PartialBatchOptimization (graph built by the C++ host class, Levenberg on
    the GPU: tile sweep, Schur complement, chain-preconditioned PCG) against the reference's OWN src/Optimizer.cc:42-1230 / :1232-2175 with its vendored g2o
    (BlockSolverX + LinearSolverCSparse on the whole system), compiled verbatim into oracle/_ref/libref_full.so (tests/test_ref_g2o.py) - on the same Map, filled
    from the same flat arrays: refined camera poses, object motions, static and dynamic points as both write them back into the Map (CV_32F).  Bar: 5e-6
    (float storage 6e-8 relative + the two solvers' 1e-6; the north star asks 1e-4)."""
    from tests import oracle_lib
    from tests.ref_track import Quiet
    ref = oracle_lib.load_ref_full()
    if ref is None:
        pytest.skip("parity unpinned: oracle/_ref/libref_full.so absent")
    ref.ref_batch_optimization.argtypes = [C.c_void_p, C.c_int, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p]
    m = SM.make_map(**shape)
    F = m["n_frames"]
    n_sta = sum(len(f["sta_uv"]) for f in m["feats"]); n_dyn = sum(len(f["dyn_uv"]) for f in m["feats"]); n_rm = sum(len(r) for r in m["rigid_motion"])
    outs = {}
    for who in ("product", "reference"):
        s, keep = _flatten(m)
        cam = np.zeros((F, 4, 4), np.float32); rm = np.zeros((n_rm, 4, 4), np.float32)
        sta = np.zeros((n_sta, 3), np.float32); dyn = np.zeros((max(n_dyn, 1), 3), np.float32)
        if who == "product":
            st = K.LMStatsC()
            assert host.host_batch_optimization(C.byref(s), window, _p(cam), _p(rm), _p(sta), _p(dyn), C.byref(st)) == 0
            assert st.iterations >= 2
        else:
            with Quiet():                                       # (the reference saves its .g2o dumps into the current directory)
                assert ref.ref_batch_optimization(C.byref(s), window, _p(cam), _p(rm), _p(sta), _p(dyn)) == 0
        outs[who] = (cam, rm, sta, dyn)
    (ca, ra, sa, da), (cb, rb, sb, db) = outs["product"], outs["reference"]
    assert np.abs(cb).max() > 0 and np.abs(sb).max() > 0
    scale_t = max(1.0, float(np.abs(cb[:, :3, 3]).max()))
    assert np.abs(ca[:, :3, :3] - cb[:, :3, :3]).max() <= 5e-6 and np.abs(ca[:, :3, 3] - cb[:, :3, 3]).max() <= 5e-6 * scale_t, (np.abs(ca - cb).max(),)
    assert np.abs(ra[:, :3, :3] - rb[:, :3, :3]).max() <= 5e-6 and np.abs(ra[:, :3, 3] - rb[:, :3, 3]).max() <= 5e-6 * max(1.0, float(np.abs(rb[:, :3, 3]).max())), (np.abs(ra - rb).max(),)
    assert np.abs(sa - sb).max() <= 5e-6 * max(1.0, float(np.abs(sb).max())), np.abs(sa - sb).max()
    assert np.abs(da - db).max() <= 5e-6 * max(1.0, float(np.abs(db).max())), np.abs(da - db).max()


def _p(a):
    return a.ctypes.data_as(K.c_float_p)


def test_frame_constructor_matches_oracle(host, oracle):
    host.host_frame.argtypes = [K.c_uint8_p, K.c_float_p, K.c_float_p, K.c_int32_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                K.c_float_p, K.c_float_p, K.c_int32_p, C.c_int, C.POINTER(C.c_int), K.c_float_p, K.c_float_p,
                                C.POINTER(C.c_int), K.c_float_p, K.c_int32_p, C.c_int]
    fr = SF.make_frame(seed=12)
    depth = fr["depth_raw"].copy()
    cap, capo = 4096, 40000
    kx = np.zeros(cap, np.float32); ky = np.zeros(cap, np.float32); ko = np.zeros(cap, np.int32)
    sc = np.zeros((cap, 2), np.float32); sd = np.zeros(cap, np.float32); ok = np.zeros((capo, 2), np.float32); ol = np.zeros(capo, np.int32)
    ns, no = C.c_int(), C.c_int()
    n = host.host_frame(R._u8(fr["gray"]), _p(depth), _p(fr["flow"]), R._ip(fr["mask"]), 1242, 375, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ,
                        _p(kx), _p(ky), R._ip(ko), cap, C.byref(ns), _p(sc), _p(sd), C.byref(no), _p(ok), R._ip(ol), capo)
    d_ref = fr["depth_raw"].copy()
    oracle.vdo_oracle_depth_preprocess(R._fp(d_ref), d_ref.size, SF.BF, SF.DEPTH_MAP_FACTOR)
    assert np.array_equal(depth, d_ref)                      # GrabImageRGBD mutates the depth in place
    ref = R.extract(oracle, fr["gray"])
    assert n == ref["x"].size
    assert np.array_equal(kx[:n], ref["x"]) and np.array_equal(ky[:n], ref["y"]) and np.array_equal(ko[:n], ref["octave"])
    sf = R.static_filter(oracle, ref["x"], ref["y"], ref["octave"], fr["mask"], d_ref, fr["flow"], SF.TH_DEPTH_BG)
    assert ns.value == sf["keep_idx"].size
    assert np.array_equal(sc[:ns.value, 0], sf["corr_x"]) and np.array_equal(sc[:ns.value, 1], sf["corr_y"]) and np.array_equal(sd[:ns.value], sf["depth"])
    ob = R.object_sample(oracle, fr["mask"], d_ref, fr["flow"], SF.TH_DEPTH_OBJ)
    assert no.value == ob["label"].size
    assert np.array_equal(ok[:no.value, 0], ob["key_x"]) and np.array_equal(ol[:no.value], ob["label"])


def test_pose_optimization_flow2cam_class_matches_oracle(host, oracle):
    from tests.test_oracle_flow2 import run_oracle
    host.host_pose_optimization_flow2cam.argtypes = [C.c_int, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p, K.c_int32_p, K.c_float_p]
    prob = synth.make_flow2_problem(900, seed=31)
    # the class receives the LAST frame pose T_lw (float) and derives Twl itself (Optimizer.cc:2414-2420)
    Twl = prob.Twl
    Tlw = np.linalg.inv(Twl).astype(np.float32)
    from tests.pipeline_ref import inv_rigid_f32
    inv = inv_rigid_f32(Tlw)                               # Converter::toInvMatrix (cv::gemm with a transposed operand: double accumulation, one rounding)
    prob.Twl = inv.astype(np.float64)
    T, flow, inl, ninl, st = run_oracle(oracle, prob)
    n = prob.n
    Tout = np.zeros((4, 4), np.float32); match = np.zeros(n, np.int32); cur = np.zeros((n, 2), np.float32)
    got = host.host_pose_optimization_flow2cam(n, _p(_f(prob.obs)), _p(_f(prob.flow)), _p(_f(prob.depth)), _p(Tlw), _p(_f(prob.T0)), _p(Tout), R._ip(match), _p(cur))
    assert got == ninl
    assert np.array_equal(match >= 0, inl.astype(bool))
    np.testing.assert_allclose(Tout, T.astype(np.float32), rtol=1e-4, atol=1e-5)
    exp = (prob.obs + flow).astype(np.float32)
    np.testing.assert_allclose(cur[inl.astype(bool)], exp[inl.astype(bool)], atol=1e-4)


def test_non_joint_statics_of_the_optimizer_class_match_the_oracle(host, oracle):
    """Optimizer::PoseOptimizationNew / PoseOptimizationObjMot (include/Optimizer.h:25,27): the host statics marshal Frame members
    (UnprojectStereoStat / Object, P = K * Tcw, Init = Tcw^-1 * mInitModel) into vdo_pose_optimize; result == the oracle's
    unary-edge LM on the same marshalled problem."""
    from tests.test_oracle_pose_only import run_oracle
    from vdo_slam_amd import pose_only as PO
    fx, fy, cx, cy = synth.KITTI_K
    f32 = np.float32
    rng = np.random.default_rng(8)

    from tests.pipeline_ref import inv_rigid_f32 as inv32     # Converter::toInvMatrix

    def unproject(xy, d, Tcw):                           # Frame::UnprojectStereo*: Rwl * x3Dc + twl (float fast path of cv::gemm), twl = -Rlw^T tlw (generic path)
        x3 = np.stack([(xy[:, 0] - f32(cx)) * d * (f32(1) / f32(fx)), (xy[:, 1] - f32(cy)) * d * (f32(1) / f32(fy)), d], 1).astype(f32)
        twl = inv32(Tcw)[:3, 3]
        Rwl = Tcw[:3, :3].T.astype(f32)
        out = np.zeros((xy.shape[0], 3), f32)
        for i in range(3):
            t = f32(Rwl[i, 0]) * x3[:, 0]
            t = (t + f32(Rwl[i, 1]) * x3[:, 1]).astype(f32)
            t = (t + f32(Rwl[i, 2]) * x3[:, 2]).astype(f32)
            out[:, i] = (t + twl[i]).astype(f32)
        return out

    n = 700
    Tl = synth._mat4(synth.rotvec_to_R(rng.normal(0, 0.02, 3)), rng.normal(0, 1.0, 3)).astype(f32)
    last_xy = np.c_[rng.uniform(50, 1190, n), rng.uniform(30, 340, n)].astype(f32)
    depth = rng.uniform(5, 35, n).astype(f32)
    Xw = unproject(last_xy, depth, Tl)
    # ---- camera: true current pose = small motion on top of the last pose
    dT = synth._mat4(synth.rotvec_to_R(np.array([0.0, 0.006, 0.0])), np.array([0.02, -0.01, -0.8]))
    Tc_true = dT @ Tl.astype(np.float64)
    Xc = Xw.astype(np.float64) @ Tc_true[:3, :3].T + Tc_true[:3, 3]
    cur_xy = np.c_[fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy] + rng.normal(0, 0.05, (n, 2))
    cur_xy[rng.random(n) < 0.1] += rng.normal(0, 4.0, 2)
    cur_xy = cur_xy.astype(f32)
    Tinit = (synth._mat4(synth.rotvec_to_R(rng.normal(0, 0.003, 3)), rng.normal(0, 0.03, 3)) @ Tc_true).astype(f32)
    host.host_pose_optimization_new.argtypes = [C.c_int] + [K.c_float_p] * 6 + [K.c_int32_p]
    Tout = np.zeros((4, 4), f32); match = np.zeros(n, np.int32)
    got = host.host_pose_optimization_new(n, _p(last_xy), _p(depth), _p(cur_xy), _p(Tl), _p(Tinit), _p(Tout), R._ip(match))
    prob = PO.PoseProblem(kind=0, obs=cur_xy.astype(np.float64), Xw=Xw.astype(np.float64), K=tuple(float(f32(v)) for v in synth.KITTI_K), P=np.zeros((3, 4)),
                          T0=Tinit.astype(np.float64), huber_delta=float(np.sqrt(f32(0.01))), max_iterations=100)
    T, inl, ninl, st = run_oracle(oracle, prob)
    assert got == ninl and 0.6 * n < ninl < n
    assert np.array_equal(match >= 0, inl.astype(bool))
    np.testing.assert_allclose(Tout, T.astype(f32), rtol=1e-4, atol=1e-5)
    # ---- object: points moved by a world-frame motion H, seen from the current camera
    Hm = synth._mat4(synth.rotvec_to_R(np.array([0.0, 0.02, 0.0])), np.array([0.1, 0.0, 0.6]))
    Tcur = Tc_true.astype(f32)
    Xn = Xw.astype(np.float64) @ Hm[:3, :3].T + Hm[:3, 3]
    Xc = Xn @ Tcur[:3, :3].astype(np.float64).T + Tcur[:3, 3].astype(np.float64)
    obj_xy = (np.c_[fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy] + rng.normal(0, 0.03, (n, 2))).astype(f32)
    init_model = (Tcur.astype(np.float64) @ synth._mat4(np.eye(3), np.array([0.05, 0.0, 0.5]))).astype(f32)      # mInitModel = Tcw * H0
    host.host_pose_optimization_objmot.argtypes = [C.c_int] + [K.c_float_p] * 7 + [K.c_int32_p, K.c_int32_p]
    Hout = np.zeros((4, 4), f32); flag = np.zeros(n, np.int32); lab = np.zeros(n, np.int32)
    got = host.host_pose_optimization_objmot(n, _p(last_xy), _p(depth), _p(obj_xy), _p(Tl), _p(Tcur), _p(init_model), _p(Hout), R._ip(flag), R._ip(lab))
    KK = np.array([[f32(fx), 0, f32(cx), 0], [0, f32(fy), f32(cy), 0], [0, 0, 1, 0]], np.float64)
    from tests.pipeline_ref import matmul4_f32
    Init = matmul4_f32(inv32(Tcur), init_model)                                                        # cv::Mat product (cv::gemm's float fast path)
    probo = PO.PoseProblem(kind=1, obs=obj_xy.astype(np.float64), Xw=Xw.astype(np.float64), K=tuple(float(f32(v)) for v in synth.KITTI_K),
                           P=KK @ Tcur.astype(np.float64), T0=Init.astype(np.float64), huber_delta=0.0, max_iterations=200)
    T, inl, ninl, st = run_oracle(oracle, probo)
    assert got == ninl and ninl > 0.8 * n
    assert np.array_equal(flag.astype(bool), inl.astype(bool)) and np.array_equal(lab == -1, ~inl.astype(bool))
    np.testing.assert_allclose(Hout, T.astype(f32), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(Hout[:3, 3], Hm[:3, 3], atol=0.02)
