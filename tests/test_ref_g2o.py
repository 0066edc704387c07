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
