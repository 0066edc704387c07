"""oracle/_ref pin of the OPTIMISER arithmetic: the oracle's restatements against the REFERENCE'S OWN src/Optimizer.cc,
src/Converter.cc and vendored g2o (dependencies/g2o/g2o: every source of its CMake target), compiled verbatim from /root/reference into
oracle/_ref/libref_full.so (oracle/ref/Makefile) against shim/Eigen - a small dense-algebra library with Eigen's interface written for this repository -
and shim/cs.h + minics.cpp (CSparse's interface).  Entry points: oracle/ref/ref_g2o_entry.cc (object construction and copying only).

Pinned by this: g2o's edges and vertices (error functions, Jacobians, the (+) operators, SE3Quat), Huber with its float dsqr, constructQuadraticForm and the
upper-triangular block placement, BlockSolver (incl. the BlockSolver_6_3 / 2-DoF aliasing F3 through the reference's real memory layout), Levenberg's control
flow and stop rules, the terminate action, LinearSolverDense / LinearSolverCSparse's call sequences, Optimizer.cc's graph builders, thresholds, outlier loops
and write-backs, Converter.  NOT pinned (restated on both sides, from the published algorithms): what Eigen and CSparse do INSIDE - the order of the
floating-point operations of a product or a factorisation, Eigen's pivoted LDLT, SuiteSparse's AMD ordering (a plain minimum degree here)."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib
from vdo_slam_amd import _capi as K
from vdo_slam_amd import synth

dp = K.c_double_p


def _d(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def ref():
    L = oracle_lib.load_ref_full()
    if L is None:
        pytest.skip("parity unpinned: oracle/_ref/libref_full.so absent and no reference checkout to build it from")
    L.ref_se3_exp.argtypes = [dp, dp]
    L.ref_se3quat_oplus.argtypes = [dp, dp, dp]
    L.ref_se3quat_roundtrip.argtypes = [dp, dp]
    L.ref_iso_oplus.argtypes = [dp, dp, dp]
    L.ref_iso_to_mqt.argtypes = [dp, dp]
    L.ref_edge_se3_jac.argtypes = [dp] * 6
    L.ref_edge_prior_jac.argtypes = [dp] * 4
    L.ref_edge_eb_jac.argtypes = [dp] * 6
    L.ref_edge_et_jac.argtypes = [dp] * 8
    L.ref_edge_unary_jac.argtypes = [C.c_int] + [dp] * 7
    L.ref_edge_flow2_jac.argtypes = [dp, dp, C.c_double] + [dp] * 9
    L.ref_huber.argtypes = [C.c_double, C.c_double, dp]
    L.ref_ba_optimize.argtypes = [C.POINTER(K.BAGraphC), C.POINTER(K.LMOptionsC), dp, dp, C.POINTER(K.LMStatsC)]
    L.ref_ba_linearize.argtypes = [C.POINTER(K.BAGraphC), C.POINTER(K.BASystemC)]
    return L


@pytest.fixture(scope="module")
def ora(oracle):
    oracle.vdo_oracle_edge_flow2_jac.argtypes = [dp, dp, C.c_double] + [dp] * 7
    oracle.vdo_oracle_huber.argtypes = [C.c_double, C.c_double, dp]
    oracle.vdo_oracle_se3quat_oplus.argtypes = [dp, dp, dp]
    return oracle


def _rand_iso(rng, rot=1.0, trans=5.0):
    R = synth.rotvec_to_R(rng.normal(0, rot, 3))
    return np.concatenate([R.ravel(), rng.normal(0, trans, 3)])


def _T16(T12):
    M = np.eye(4); M[:3, :3] = T12[:9].reshape(3, 3); M[:3, 3] = T12[9:]
    return M.ravel().copy()


N_EDGE = 10000           # random inputs per edge class (>= 1e4)


def _worst(a, b):
    """largest difference in units of the last place of the larger magnitude"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b) / (scale * np.finfo(np.float64).eps))) if a.size else 0.0


def test_se3quat_exp_and_oplus_equal_the_reference(ref, ora):
    """g2o::SE3Quat::exp (both branches), VertexSE3Expmap::oplusImpl = exp(update) * estimate, the SE3Quat(R, t) round trip of Converter::toSE3Quat"""
    rng = np.random.default_rng(1)
    for k in range(N_EDGE):
        scale = (1e-7, 1e-3, 0.05, 1.0)[k % 4]                   # theta < 1e-5 takes the small-angle branch (se3quat.h:242-248)
        u = rng.normal(0, scale, 6)
        a = np.zeros(16); b = np.zeros(16)
        ref.ref_se3_exp(_d(u), _d(a)); ora.vdo_oracle_se3_exp(_d(u), _d(b))
        assert np.array_equal(a, b), (k, u)
        T = _T16(_rand_iso(rng))
        ref.ref_se3quat_roundtrip(_d(T), _d(a)); ora.vdo_oracle_se3quat_oplus(_d(T), None, _d(b))
        assert np.array_equal(a, b), (k, "toSE3Quat round trip")
        ref.ref_se3quat_oplus(_d(T), _d(u), _d(a)); ora.vdo_oracle_se3quat_oplus(_d(T), _d(u), _d(b))
        assert np.array_equal(a, b), (k, "oplus")


def test_vertex_se3_oplus_and_mqt_equal_the_reference(ref, ora):
    """VertexSE3::oplusImpl (T * fromVectorMQT(d), w = sqrt(1 - |q|^2), identity beyond the unit ball) and toVectorMQT (Quaternion(R) branches, w >= 0)"""
    rng = np.random.default_rng(2)
    for k in range(N_EDGE):
        T = _rand_iso(rng, rot=(0.1, 1.0, 3.0)[k % 3])
        d = rng.normal(0, (1e-3, 0.1, 0.7)[k % 3], 6)            # the last one leaves the unit ball now and then
        a = np.zeros(12); b = np.zeros(12)
        ref.ref_iso_oplus(_d(T), _d(d), _d(a)); ora.vdo_oracle_iso_oplus(_d(T), _d(d), _d(b))
        assert np.array_equal(a, b), (k, d)
        e1 = np.zeros(6); e2 = np.zeros(6)
        ref.ref_iso_to_mqt(_d(T), _d(e1)); ora.vdo_oracle_iso_to_mqt(_d(T), _d(e2))
        assert np.array_equal(e1, e2), k


def test_edge_se3_and_prior_equal_the_reference(ref, ora):
    """EdgeSE3 / EdgeSE3Prior: error = toVectorMQT(Z^-1 Xi^-1 Xj), 6x6 Jacobians through dq/dR in its four cases (isometry3d_gradients.h, dquat2mat.cpp)"""
    rng = np.random.default_rng(3)
    worst = 0.0
    for k in range(N_EDGE):
        Xi, Xj = _rand_iso(rng, rot=(0.05, 1.0, 2.5)[k % 3]), _rand_iso(rng, rot=(0.05, 1.0, 2.5)[(k // 3) % 3])
        Z = _rand_iso(rng)
        e1, e2 = np.zeros(6), np.zeros(6); Ji1, Ji2, Jj1, Jj2 = (np.zeros(36) for _ in range(4))
        ref.ref_edge_se3_jac(_d(Z), _d(Xi), _d(Xj), _d(e1), _d(Ji1), _d(Jj1)); ora.vdo_oracle_edge_se3_jac(_d(Z), _d(Xi), _d(Xj), _d(e2), _d(Ji2), _d(Jj2))
        assert np.array_equal(e1, e2), k
        worst = max(worst, _worst(Ji1, Ji2), _worst(Jj1, Jj2))
        np.testing.assert_allclose(Ji1, Ji2, rtol=0, atol=4e-15 * max(1.0, np.abs(Ji2).max()))
        np.testing.assert_allclose(Jj1, Jj2, rtol=0, atol=4e-15 * max(1.0, np.abs(Jj2).max()))
        J1, J2 = np.zeros(36), np.zeros(36)
        ref.ref_edge_prior_jac(_d(Z), _d(Xi), _d(e1), _d(J1)); ora.vdo_oracle_edge_prior_jac(_d(Z), _d(Xi), _d(e2), _d(J2))
        assert np.array_equal(e1, e2), k
        np.testing.assert_allclose(J1, J2, rtol=0, atol=4e-15 * max(1.0, np.abs(J2).max()))


def test_edge_se3_pointxyz_and_ternary_equal_the_reference(ref, ora):
    """EdgeSE3PointXYZ (+ CacheSE3Offset with the identity offset) and LandmarkMotionTernaryEdge incl. its factor-1 rotation columns (F4)"""
    rng = np.random.default_rng(4)
    for k in range(N_EDGE):
        X = _rand_iso(rng); p = rng.normal(0, 10, 3); z = rng.normal(0, 10, 3)
        e1, e2 = np.zeros(3), np.zeros(3); A1, A2 = np.zeros(18), np.zeros(18); B1, B2 = np.zeros(9), np.zeros(9)
        ref.ref_edge_eb_jac(_d(X), _d(p), _d(z), _d(e1), _d(A1), _d(B1)); ora.vdo_oracle_edge_eb_jac(_d(X), _d(p), _d(z), _d(e2), _d(A2), _d(B2))
        assert np.array_equal(e1, e2) and np.array_equal(A1, A2) and np.array_equal(B1, B2), k
        p2 = rng.normal(0, 10, 3)
        Jp1a, Jp1b, Jp2a, Jp2b = (np.zeros(9) for _ in range(4)); Jha, Jhb = np.zeros(18), np.zeros(18)
        ref.ref_edge_et_jac(_d(X), _d(p), _d(p2), _d(z), _d(e1), _d(Jp1a), _d(Jp2a), _d(Jha)); ora.vdo_oracle_edge_et_jac(_d(X), _d(p), _d(p2), _d(z), _d(e2), _d(Jp1b), _d(Jp2b), _d(Jhb))
        assert np.array_equal(e1, e2) and np.array_equal(Jp1a, Jp1b) and np.array_equal(Jp2a, Jp2b) and np.array_equal(Jha, Jhb), k


def test_per_frame_edges_equal_the_reference(ref, ora):
    """EdgeSE3ProjectFlow2 + EdgeFlowPrior (joint optimisers) and EdgeSE3ProjectXYZOnlyPose / OnlyObjMotion (non-joint)"""
    from vdo_slam_amd.pose_only import PoseProblemC
    rng = np.random.default_rng(5)
    K4 = np.array(synth.KITTI_K, np.float64)
    for k in range(N_EDGE):
        T = _T16(_rand_iso(rng, rot=0.2, trans=1.0)); Twl = _T16(_rand_iso(rng, rot=0.2, trans=1.0))
        obs = rng.uniform(0, 1200, 2) * np.array([1.0, 0.3]); depth = float(rng.uniform(2, 60)); fe = rng.normal(0, 5, 2); fm = rng.normal(0, 5, 2)
        e1, e2, ep1, ep2 = (np.zeros(2) for _ in range(4)); Jf = np.zeros(4); Jpr = np.zeros(4); J1, J2 = np.zeros(12), np.zeros(12)
        ref.ref_edge_flow2_jac(_d(K4), _d(Twl), depth, _d(obs), _d(fe), _d(fm), _d(T), _d(e1), _d(Jf), _d(J1), _d(ep1), _d(Jpr))
        ora.vdo_oracle_edge_flow2_jac(_d(K4), _d(Twl), depth, _d(obs), _d(fe), _d(fm), _d(T), _d(e2), _d(J2), _d(ep2))
        assert np.array_equal(e1, e2) and np.array_equal(J1, J2) and np.array_equal(ep1, ep2), k
        assert np.array_equal(Jf, [1, 0, 0, 1]) and np.array_equal(Jpr, [1, 0, 0, 1])
        Xw = rng.normal(0, 5, 3) + np.array([0, 0, 20.0]); ob = rng.uniform(0, 1000, 2)
        for kind in (0, 1):
            P = (np.array([[K4[0], 0, K4[2], 0], [0, K4[1], K4[3], 0], [0, 0, 1, 0]]) @ _T16(_rand_iso(rng, rot=0.1, trans=1.0)).reshape(4, 4)).ravel().copy()
            q = PoseProblemC(); q.n = 1; q.kind = kind
            for i in range(4): q.K[i] = K4[i]
            for i in range(12): q.P[i] = P[i]
            q.huber_delta = 0.1; q.chi2_gate = 0.01; q.max_iterations = 1
            ora.vdo_oracle_edge_unary_jac(C.byref(q), _d(T), _d(Xw), _d(ob), _d(e2), _d(J2))
            ref.ref_edge_unary_jac(kind, _d(K4), _d(P), _d(T), _d(Xw), _d(ob), _d(e1), _d(J1))
            assert np.array_equal(e1, e2) and np.array_equal(J1, J2), (k, kind)


def test_huber_equals_the_reference(ref, ora):
    """RobustKernelHuber::robustify with its FLOAT member dsqr (robust_kernel_impl.h:84): the deltas of src/Optimizer.cc"""
    rng = np.random.default_rng(6)
    for delta in (1e-4, float(np.float32(np.sqrt(np.float32(0.04)))), float(np.float32(np.sqrt(np.float32(0.01)))), 1.0):
        for e2 in np.concatenate([rng.uniform(0, 4 * delta * delta, 2000), [delta * delta, float(np.float32(delta * delta)), 0.0], rng.uniform(0, 100, 500)]):
            a = np.zeros(3); b = np.zeros(2)
            ref.ref_huber(delta, float(e2), _d(a)); ora.vdo_oracle_huber(delta, float(e2), _d(b))
            assert np.array_equal(a[:2], b), (delta, e2)


# ---- whole systems and whole optimisations -----------------------------------------------------------------------------------------------------
def _graph(frames=8, points=400, objects=2, dyn=60, seed=3, **kw):
    return synth.make_ba_graph(frames, points, objects, dyn, seed=seed, **kw)


def test_buildsystem_equals_the_reference(ref, oracle):
    """BlockSolver::buildSystem over a batch graph (computeActiveErrors, robustify, constructQuadraticForm of unary / binary / multi edges, block placement):
    every block of H and b and both chi2 of the oracle's linearisation against g2o's own"""
    g = _graph(frames=10, points=1500, objects=3, dyn=150, seed=11)
    gc, keep = K.graph_to_c(g)
    so, sr = K.BASystem(g), K.BASystem(g)
    assert oracle.vdo_oracle_ba_linearize(C.byref(gc), C.byref(so.c)) == 0
    assert ref.ref_ba_linearize(C.byref(gc), C.byref(sr.c)) == 0
    assert g.eb_pose.size > 5000 and g.et_p1.size > 300
    assert abs(so.chi2 - sr.chi2) <= 1e-12 * sr.chi2 and abs(so.robust_chi2 - sr.robust_chi2) <= 1e-12 * sr.robust_chi2
    for name in ("Hpp", "bp", "Hll", "bl", "Hpl_eb", "Hll_et", "Hlp1_et", "Hlp2_et", "Hpp_ep"):
        a, b = getattr(so, name), getattr(sr, name)
        scale = max(np.abs(b).max(), 1e-300)
        assert np.abs(a - b).max() <= 1e-12 * scale, (name, np.abs(a - b).max() / scale)


@pytest.mark.parametrize("seed,kw", [(1, dict()), (2, dict(outlier_frac=0.1)), (3, dict(init_sigma_t=0.1, init_sigma_r=0.02))])
def test_batch_lm_equals_the_reference(ref, oracle, seed, kw):
    """optimize(): the oracle's Levenberg against g2o's own OptimizationAlgorithmLevenberg + BlockSolverX + LinearSolverCSparse on the same graph -
    same number of outer iterations, same trials per iteration (g2o's batch statistics), chi2 trace and final estimates"""
    g = _graph(frames=8, points=300, objects=2, dyn=40, seed=seed, **kw)
    gc, keep = K.graph_to_c(g)
    opt = K.LMOptionsC(60, 1e-4, 0, 0, 0.0, 0)
    so, sr = K.LMStatsC(), K.LMStatsC()
    po, pr = np.zeros_like(g.pose), np.zeros_like(g.pose); qo, qr = np.zeros_like(g.point), np.zeros_like(g.point)
    assert oracle.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), _d(po), _d(qo), C.byref(so)) == 0
    assert ref.ref_ba_optimize(C.byref(gc), C.byref(opt), _d(pr), _d(qr), C.byref(sr)) == 0
    assert so.iterations == sr.iterations and so.iterations >= 3, (so.iterations, sr.iterations)
    assert list(so.trials_trace[:so.iterations]) == list(sr.trials_trace[:sr.iterations])
    np.testing.assert_allclose(np.array(so.chi2_trace[:so.iterations]), np.array(sr.chi2_trace[:sr.iterations]), rtol=1e-7)
    assert abs(so.final_lambda - sr.final_lambda) <= 1e-6 * sr.final_lambda
    np.testing.assert_allclose(po, pr, rtol=0, atol=1e-7 * max(1.0, np.abs(pr).max()))        # north star: 1e-4 relative on poses
    np.testing.assert_allclose(qo, qr, rtol=0, atol=1e-6 * max(1.0, np.abs(qr).max()))


# ---- the statics of src/Optimizer.cc themselves -------------------------------------------------------------------------------------------------
def _fp(a):
    return a.ctypes.data_as(K.c_float_p)


def _bind_statics(ref):
    fp, ip = K.c_float_p, K.c_int32_p
    ref.ref_pose_optimization_flow2cam.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp, fp, ip, fp]
    ref.ref_pose_optimization_flow2.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp, fp, fp, ip, ip, fp]
    ref.ref_pose_optimization_new.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp, fp, ip]
    ref.ref_pose_optimization_objmot.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp, ip, ip]
    ref.ref_batch_optimization.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp]
    ref.vdo_ref_set_gaussian_scale.argtypes = [C.c_double]


def _flow2_case(seed, n, is_object, **kw):
    """a joint problem as the reference's static sees it: float key points / flow / depth, the LAST pose as a CV_32F matrix (the static derives Twl itself,
    src/Optimizer.cc:2414-2420), the initial estimate as a CV_32F matrix"""
    from tests.pipeline_ref import inv_rigid_f32
    prob = synth.make_flow2_problem(n, seed=seed, is_object=is_object, **kw)
    Tlw = np.linalg.inv(prob.Twl).astype(np.float32)
    prob.Twl = inv_rigid_f32(Tlw).astype(np.float64)           # Rwl = Rlw.t(), twl = -Rlw.t() * tlw through cv::Mat
    prob.K = tuple(float(np.float32(v)) for v in prob.K)       # Frame::fx .. cy are floats
    return prob, Tlw


@pytest.mark.parametrize("is_object", [False, True])
def test_joint_statics_equal_the_reference(ref, oracle, is_object):
    """Optimizer::PoseOptimizationFlow2Cam / PoseOptimizationFlow2 - the reference's own functions on Frames filled from flat arrays - against the oracle's
    restatement on the same numbers: pose / motion as the CV_32F matrix they return (np.array_equal), inlier sets, refined key points, outlier labels.
    40 problems each: 6 .. 1500 correspondences, clean and heavily contaminated, near and far initial estimates (the F3 aliasing makes some of them run
    100+ Levenberg iterations), plus the degenerate sizes the reference handles before it builds a graph."""
    from tests.test_oracle_flow2 import run_oracle
    _bind_statics(ref)
    K4 = np.array(synth.KITTI_K, np.float32)
    rng = np.random.default_rng(17)
    exact = 0
    for case in range(40):
        n = int((6, 40, 300, 900, 1500)[case % 5])
        kw = dict(outlier_frac=(0.0, 0.1, 0.35)[case % 3], flow_sigma=(0.0, 0.3, 1.0)[(case // 3) % 3], init_sigma_t=(0.05, 0.5)[(case // 9) % 2], init_sigma_r=(0.004, 0.03)[(case // 9) % 2])
        prob, Tlw = _flow2_case(100 + case, n, is_object, **kw)
        T, flow, inl, ninl, st = run_oracle(oracle, prob)
        last_xy = prob.obs.astype(np.float32); fl = prob.flow.astype(np.float32); dep = prob.depth.astype(np.float32)
        T0 = prob.T0.astype(np.float32)
        cur = np.zeros((n, 2), np.float32); Tout = np.zeros((4, 4), np.float32)
        if not is_object:
            match = np.zeros(n, np.int32)
            got = ref.ref_pose_optimization_flow2cam(n, _fp(K4), _fp(last_xy), _fp(fl), _fp(dep), _fp(Tlw), _fp(T0), _fp(Tout), match.ctypes.data_as(K.c_int32_p), _fp(cur))
            got_inl = match >= 0
        else:
            flag = np.zeros(n, np.int32); lab = np.zeros(n, np.int32)
            Tcur = np.eye(4, dtype=np.float32)                   # (only mInitModel of the current frame is read)
            got = ref.ref_pose_optimization_flow2(n, _fp(K4), _fp(last_xy), _fp(fl), _fp(dep), _fp(Tlw), _fp(Tcur), _fp(T0), _fp(Tout), flag.ctypes.data_as(K.c_int32_p),
                                                  lab.ctypes.data_as(K.c_int32_p), _fp(cur))
            got_inl = flag.astype(bool)
            assert np.array_equal(lab == -1, ~got_inl), case
        assert got == ninl, (case, got, ninl)
        assert np.array_equal(got_inl, inl.astype(bool)), case
        assert np.array_equal(Tout, T.astype(np.float32)), (case, np.abs(Tout - T).max(), st.iterations)
        exact += 1
        exp = (last_xy.astype(np.float64) + flow).astype(np.float32)        # pt.x + flow_new(0): a float plus a double, rounded once (src/Optimizer.cc:2529-2530)
        assert np.array_equal(cur[got_inl], exp[got_inl]), case
    assert exact == 40
    # fewer than three correspondences: nothing is optimised (src/Optimizer.cc:2449-2450, :2872-2873)
    prob, Tlw = _flow2_case(5, 2, is_object)
    last_xy = prob.obs.astype(np.float32); fl = prob.flow.astype(np.float32); dep = prob.depth.astype(np.float32); T0 = prob.T0.astype(np.float32)
    cur = np.zeros((2, 2), np.float32); Tout = np.zeros((4, 4), np.float32); m2 = np.zeros(2, np.int32); l2 = np.zeros(2, np.int32)
    if not is_object:
        assert ref.ref_pose_optimization_flow2cam(2, _fp(K4), _fp(last_xy), _fp(fl), _fp(dep), _fp(Tlw), _fp(T0), _fp(Tout), m2.ctypes.data_as(K.c_int32_p), _fp(cur)) == 0
        assert np.array_equal(Tout, T0)
    else:
        ref.ref_pose_optimization_flow2(2, _fp(K4), _fp(last_xy), _fp(fl), _fp(dep), _fp(Tlw), _fp(np.eye(4, dtype=np.float32)), _fp(T0), _fp(Tout), m2.ctypes.data_as(K.c_int32_p),
                                        l2.ctypes.data_as(K.c_int32_p), _fp(cur))
        assert np.array_equal(Tout, np.eye(4, dtype=np.float32))


@pytest.mark.parametrize("window", [0, 8])
def test_batch_statics_equal_the_reference(ref, oracle, window):
    """Optimizer::FullBatchOptimization / PartialBatchOptimization - the reference's own graph builders (src/Optimizer.cc:1259-1766, :42-637), its optimiser
    set-up and its write-back into the Map - on a Map filled from flat arrays, against the oracle's LM on the graph that the Python restatement of the
    builder (tests/map_builder_ref.py - what the product's host classes are tested with on the GPU) makes of the same Map: refined camera poses, object
    motions and points."""
    from tests import map_builder_ref as SM
    from tests.ref_track import Quiet
    _bind_statics(ref)
    m = SM.make_map(n_frames=8 if window else 12, n_static=400, n_objects=2, dyn_tracks_per_object=40, seed=5)
    s, keep = SM.flatten_map(m)
    F = m["n_frames"]
    n_sta = sum(len(f["sta_uv"]) for f in m["feats"]); n_dyn = sum(len(f["dyn_uv"]) for f in m["feats"]); n_rm = sum(len(r) for r in m["rigid_motion"])
    cam_out = np.zeros((F, 4, 4), np.float32); rm_out = np.zeros((n_rm, 4, 4), np.float32)
    sta_out = np.zeros((n_sta, 3), np.float32); dyn_out = np.zeros((max(n_dyn, 1), 3), np.float32)
    with Quiet():                                               # (the reference saves its .g2o dumps into the current directory)
        assert ref.ref_batch_optimization(C.byref(s), window, _fp(cam_out), _fp(rm_out), _fp(sta_out), _fp(dyn_out)) == 0
    g, info = SM.map_to_graph(m, partial_window=window or None)
    gc, keep2 = K.graph_to_c(g)
    opt = K.LMOptionsC(100 if window else 300, 1e-3 if window else 1e-4, 0, 0, 0.0, 0)
    st_o = K.LMStatsC()
    pose_o = np.zeros_like(g.pose); point_o = np.zeros_like(g.point)
    assert oracle.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), _d(pose_o), _d(point_o), C.byref(st_o)) == 0
    start = info["start"]
    worst = 0.0
    for i in range(start, F):
        o = pose_o[info["cam_idx"][i - start]]
        worst = max(worst, np.abs(cam_out[i][:3, :3].ravel() - o[:9]).max(), np.abs(cam_out[i][:3, 3] - o[9:]).max() / max(1.0, np.abs(o[9:]).max()))
    assert worst <= 2e-6, worst                                  # (float32 storage: 6e-8 relative; north star: 1e-4)
    off = np.cumsum([0] + [len(f["sta_uv"]) for f in m["feats"]])
    checked = 0
    for i in range(start, F):
        for j, mk in enumerate(info["mkS"][i]):
            if mk >= 0:
                np.testing.assert_allclose(sta_out[off[i] + j], point_o[mk], rtol=2e-6, atol=2e-6)
                checked += 1
    assert checked > 100
    if not window:                                               # object motions (full batch only): vmRigidMotion_RF[i][j], j >= 1
        ro = np.cumsum([0] + [len(r) for r in m["rigid_motion"]])
        nmot = 0
        for i in range(F - 1):
            for j in range(1, len(m["rigid_motion"][i])):
                v = info["vid"][i][j]
                if v < 0:
                    continue
                H = rm_out[ro[i] + j]
                np.testing.assert_allclose(H[:3, :3].ravel(), pose_o[v][:9], rtol=0, atol=2e-6)
                np.testing.assert_allclose(H[:3, 3], pose_o[v][9:], rtol=2e-6, atol=2e-6)
                nmot += 1
        assert nmot >= 10


def test_non_joint_statics_equal_the_reference(ref, oracle):
    """Optimizer::PoseOptimizationNew / PoseOptimizationObjMot (src/Optimizer.cc:2177-2331, :2544-2753; unreachable from Track(), which forces bJoint) - the
    reference's own functions against the oracle's unary-edge LM on the problem marshalled the way those functions marshal it (UnprojectStereoStat / Object
    through cv::Mat arithmetic, P = K * Tcw, Init = Tcw^-1 * mInitModel).  The reference back-projects with addnoise = 1 here - test noise from a cv::RNG
    seeded with time(NULL): the shim's generator is silenced for the comparison (vdo_ref_set_gaussian_scale(0))."""
    from tests.test_oracle_pose_only import run_oracle
    from tests.pipeline_ref import inv_rigid_f32 as inv32, matmul4_f32
    from vdo_slam_amd import pose_only as PO
    _bind_statics(ref)
    ref.vdo_ref_set_gaussian_scale(0.0)
    try:
        fx, fy, cx, cy = synth.KITTI_K
        f32 = np.float32
        K4 = np.array(synth.KITTI_K, f32)

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

        for seed in range(8, 14):
            rng = np.random.default_rng(seed)
            n = int((700, 60, 250)[seed % 3])
            Tl = synth._mat4(synth.rotvec_to_R(rng.normal(0, 0.02, 3)), rng.normal(0, 1.0, 3)).astype(f32)
            last_xy = np.c_[rng.uniform(50, 1190, n), rng.uniform(30, 340, n)].astype(f32)
            depth = rng.uniform(5, 35, n).astype(f32)
            Xw = unproject(last_xy, depth, Tl)
            dT = synth._mat4(synth.rotvec_to_R(np.array([0.0, 0.006, 0.0])), np.array([0.02, -0.01, -0.8]))
            Tc_true = dT @ Tl.astype(np.float64)
            Xc = Xw.astype(np.float64) @ Tc_true[:3, :3].T + Tc_true[:3, 3]
            cur_xy = np.c_[fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy] + rng.normal(0, 0.05, (n, 2))
            cur_xy[rng.random(n) < 0.1] += rng.normal(0, 4.0, 2)
            cur_xy = cur_xy.astype(f32)
            Tinit = (synth._mat4(synth.rotvec_to_R(rng.normal(0, 0.003, 3)), rng.normal(0, 0.03, 3)) @ Tc_true).astype(f32)
            Tout = np.zeros((4, 4), f32); match = np.zeros(n, np.int32)
            got = ref.ref_pose_optimization_new(n, _fp(K4), _fp(last_xy), _fp(depth), _fp(cur_xy), _fp(Tl), _fp(Tinit), _fp(Tout), match.ctypes.data_as(K.c_int32_p))
            prob = PO.PoseProblem(kind=0, obs=cur_xy.astype(np.float64), Xw=Xw.astype(np.float64), K=tuple(float(f32(v)) for v in synth.KITTI_K), P=np.zeros((3, 4)),
                                  T0=Tinit.astype(np.float64), huber_delta=float(np.sqrt(f32(0.01))), max_iterations=100)
            T, inl, ninl, st = run_oracle(oracle, prob)
            assert got == ninl and 0.5 * n < ninl < n, (seed, got, ninl)
            assert np.array_equal(match >= 0, inl.astype(bool)), seed
            assert np.array_equal(Tout, T.astype(f32)), (seed, np.abs(Tout - T).max())
            # object: points moved by a world-frame motion H, seen from the current camera
            Hm = synth._mat4(synth.rotvec_to_R(np.array([0.0, 0.02, 0.0])), np.array([0.1, 0.0, 0.6]))
            Tcur = Tc_true.astype(f32)
            Xn = Xw.astype(np.float64) @ Hm[:3, :3].T + Hm[:3, 3]
            Xc = Xn @ Tcur[:3, :3].astype(np.float64).T + Tcur[:3, 3].astype(np.float64)
            obj_xy = (np.c_[fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy] + rng.normal(0, 0.03, (n, 2))).astype(f32)
            init_model = (Tcur.astype(np.float64) @ synth._mat4(np.eye(3), np.array([0.05, 0.0, 0.5]))).astype(f32)      # mInitModel = Tcw * H0
            Hout = np.zeros((4, 4), f32); flag = np.zeros(n, np.int32); lab = np.zeros(n, np.int32)
            got = ref.ref_pose_optimization_objmot(n, _fp(K4), _fp(last_xy), _fp(depth), _fp(obj_xy), None, _fp(Tl), _fp(Tcur), _fp(init_model), _fp(Hout),
                                                   flag.ctypes.data_as(K.c_int32_p), lab.ctypes.data_as(K.c_int32_p))
            KK = np.array([[f32(fx), 0, f32(cx), 0], [0, f32(fy), f32(cy), 0], [0, 0, 1, 0]], np.float64)
            Init = matmul4_f32(inv32(Tcur), init_model)                                                        # cv::Mat product (cv::gemm's float fast path)
            probo = PO.PoseProblem(kind=1, obs=obj_xy.astype(np.float64), Xw=Xw.astype(np.float64), K=tuple(float(f32(v)) for v in synth.KITTI_K),
                                   P=KK @ Tcur.astype(np.float64), T0=Init.astype(np.float64), huber_delta=0.0, max_iterations=200)
            T, inl, ninl, st = run_oracle(oracle, probo)
            assert got == ninl and ninl > 0.8 * n, (seed, got, ninl)
            assert np.array_equal(flag.astype(bool), inl.astype(bool)) and np.array_equal(lab == -1, ~inl.astype(bool)), seed
            assert np.array_equal(Hout, T.astype(f32)), (seed, np.abs(Hout - T).max())
    finally:
        ref.vdo_ref_set_gaussian_scale(1.0)
