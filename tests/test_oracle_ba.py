"""KATs that pin the CPU oracle of the batch BA path (the reference has no tests —
SURVEY.md §4/§8c — so the oracle is pinned by self-consistency):
  * analytic vs numeric Jacobians of every edge type (incl. the F4 factor-1/2 quirk),
  * normal equations + sparse Cholesky vs scipy.sparse (SuperLU) on the same system,
  * LM converges towards ground truth on a synthetic dynamic-SLAM graph.
"""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from vdo_slam_amd import _capi as K
from vdo_slam_amd import synth


def _p(a):
    return a.ctypes.data_as(K.c_double_p)


def _rand_iso(rng, ang=0.7, tr=3.0):
    return synth.iso(synth.rotvec_to_R(rng.normal(0, ang, 3)), rng.normal(0, tr, 3))


def _oplus(oracle, T, d):
    out = np.zeros(12)
    oracle.vdo_oracle_iso_oplus(_p(np.ascontiguousarray(T)), _p(np.ascontiguousarray(d)), _p(out))
    return out


def _num_jac(f, x0_list, oplus_list, dims, h=1e-6):
    e0 = f(*x0_list)
    cols = []
    for k, (x0, op, dim) in enumerate(zip(x0_list, oplus_list, dims)):
        J = np.zeros((e0.size, dim))
        for j in range(dim):
            d = np.zeros(dim); d[j] = h
            xp = list(x0_list); xm = list(x0_list)
            xp[k] = op(x0, d); xm[k] = op(x0, -d)
            J[:, j] = (f(*xp) - f(*xm)) / (2 * h)
        cols.append(J)
    return e0, cols


def test_edge_se3_jacobians(oracle):
    rng = np.random.default_rng(0)
    for trial in range(20):
        ang = [0.2, 1.5, 3.0][trial % 3]
        Z, Xi, Xj = _rand_iso(rng, ang), _rand_iso(rng, ang), _rand_iso(rng, ang)

        def f(xi, xj):
            e = np.zeros(6)
            oracle.vdo_oracle_edge_se3_jac(_p(Z), _p(np.ascontiguousarray(xi)), _p(np.ascontiguousarray(xj)), _p(e), None, None)
            return e
        e = np.zeros(6); Ji = np.zeros(36); Jj = np.zeros(36)
        oracle.vdo_oracle_edge_se3_jac(_p(Z), _p(Xi), _p(Xj), _p(e), _p(Ji), _p(Jj))
        op = lambda T, d: _oplus(oracle, T, d)
        e0, (Ni, Nj) = _num_jac(f, [Xi, Xj], [op, op], [6, 6])
        assert np.allclose(e, e0)
        np.testing.assert_allclose(Ji.reshape(6, 6), Ni, atol=2e-6)
        np.testing.assert_allclose(Jj.reshape(6, 6), Nj, atol=2e-6)


def test_edge_prior_jacobian(oracle):
    rng = np.random.default_rng(1)
    for _ in range(10):
        Z, X = _rand_iso(rng), _rand_iso(rng)

        def f(x):
            e = np.zeros(6)
            oracle.vdo_oracle_edge_prior_jac(_p(Z), _p(np.ascontiguousarray(x)), _p(e), None)
            return e
        e = np.zeros(6); J = np.zeros(36)
        oracle.vdo_oracle_edge_prior_jac(_p(Z), _p(X), _p(e), _p(J))
        _, (N,) = _num_jac(f, [X], [lambda T, d: _oplus(oracle, T, d)], [6])
        np.testing.assert_allclose(J.reshape(6, 6), N, atol=2e-6)


def test_edge_binary_jacobian(oracle):
    rng = np.random.default_rng(2)
    for _ in range(10):
        X = _rand_iso(rng); p = rng.normal(0, 5, 3); z = rng.normal(0, 5, 3)

        def f(x, pp):
            e = np.zeros(3)
            oracle.vdo_oracle_edge_eb_jac(_p(np.ascontiguousarray(x)), _p(np.ascontiguousarray(pp)), _p(z), _p(e), None, None)
            return e
        e = np.zeros(3); Jp = np.zeros(18); Jl = np.zeros(9)
        oracle.vdo_oracle_edge_eb_jac(_p(X), _p(p), _p(z), _p(e), _p(Jp), _p(Jl))
        _, (Np, Nl) = _num_jac(f, [X, p], [lambda T, d: _oplus(oracle, T, d), lambda a, d: a + d], [6, 3])
        np.testing.assert_allclose(Jp.reshape(3, 6), Np, atol=2e-6)
        np.testing.assert_allclose(Jl.reshape(3, 3), Nl, atol=2e-6)


def test_edge_ternary_jacobian_has_reference_quirk_F4(oracle):
    """J wrt the motion vertex: translation columns exact, rotation columns are exactly
    HALF the true derivative (reference omits the factor 2, types_dyn_slam3d.cpp:73-78)."""
    rng = np.random.default_rng(3)
    for _ in range(10):
        H = _rand_iso(rng, 0.3, 1.0); p1 = rng.normal(0, 5, 3); p2 = rng.normal(0, 5, 3); z = np.zeros(3)

        def f(h, a, b):
            e = np.zeros(3)
            oracle.vdo_oracle_edge_et_jac(_p(np.ascontiguousarray(h)), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(z), _p(e), None, None, None)
            return e
        e = np.zeros(3); J1 = np.zeros(9); J2 = np.zeros(9); Jh = np.zeros(18)
        oracle.vdo_oracle_edge_et_jac(_p(H), _p(p1), _p(p2), _p(z), _p(e), _p(J1), _p(J2), _p(Jh))
        add = lambda a, d: a + d
        _, (Nh, N1, N2) = _num_jac(f, [H, p1, p2], [lambda T, d: _oplus(oracle, T, d), add, add], [6, 3, 3])
        np.testing.assert_allclose(J1.reshape(3, 3), N1, atol=2e-6)
        np.testing.assert_allclose(J2.reshape(3, 3), N2, atol=2e-6)
        Jh = Jh.reshape(3, 6)
        np.testing.assert_allclose(Jh[:, :3], Nh[:, :3], atol=2e-6)
        np.testing.assert_allclose(2.0 * Jh[:, 3:], Nh[:, 3:], atol=4e-6)


def test_se3_exp_is_rotation_and_matches_rodrigues(oracle):
    rng = np.random.default_rng(4)
    for _ in range(10):
        u = rng.normal(0, 0.5, 6)
        T = np.zeros(16)
        oracle.vdo_oracle_se3_exp(_p(u), _p(T))
        T = T.reshape(4, 4)
        R = T[:3, :3]
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
        np.testing.assert_allclose(R, synth.rotvec_to_R(u[:3]), atol=1e-12)


@pytest.fixture(scope="module")
def small_graph():
    return synth.make_ba_graph(n_frames=12, n_static=300, n_objects=2, dyn_tracks_per_object=40, seed=3)


def test_normal_equations_and_cholesky_vs_scipy(oracle, small_graph):
    g = small_graph
    gc, keep = K.graph_to_c(g)
    n = 3 * g.n_point + 6 * g.n_pose
    nnz = oracle.vdo_oracle_ba_normal_equations(C.byref(gc), None, None, None, 0, None)
    rows = np.zeros(nnz, np.int32); cols = np.zeros(nnz, np.int32); vals = np.zeros(nnz); rhs = np.zeros(n)
    oracle.vdo_oracle_ba_normal_equations(C.byref(gc), rows.ctypes.data_as(K.c_int32_p), cols.ctypes.data_as(K.c_int32_p), _p(vals), nnz, _p(rhs))
    U = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    H = U + sp.triu(U, 1).T
    lam = 1e-3
    x_ref = spla.spsolve((H + lam * sp.identity(n)).tocsc(), rhs)
    x = np.zeros(n)
    assert oracle.vdo_oracle_ba_solve(C.byref(gc), lam, _p(x)) == 0
    np.testing.assert_allclose(x, x_ref, rtol=1e-7, atol=1e-9 * np.abs(x_ref).max())
    # H must be symmetric PSD-ish: diagonal blocks symmetric
    S = BASystem = K.BASystem(g)
    oracle.vdo_oracle_ba_linearize(C.byref(gc), C.byref(S.c))
    Hpp = S.Hpp.reshape(-1, 6, 6)
    np.testing.assert_allclose(Hpp, np.swapaxes(Hpp, 1, 2), rtol=1e-12, atol=1e-9)
    assert S.robust_chi2 <= S.chi2 + 1e-9


def test_lm_reduces_error_towards_ground_truth(oracle, small_graph):
    g = small_graph
    gc, keep = K.graph_to_c(g)
    opt = K.LMOptionsC(60, 1e-4, 0, 0, 0.0, 0)
    st = K.LMStatsC()
    pose = np.zeros_like(g.pose); point = np.zeros_like(g.point)
    assert oracle.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), _p(pose), _p(point), C.byref(st)) == 0
    assert st.iterations >= 2
    assert st.final_chi2 < st.initial_chi2
    nc = g.n_cam
    err0 = np.linalg.norm(g.pose[:nc, 9:] - g.pose_gt[:nc, 9:], axis=1).mean()
    err1 = np.linalg.norm(pose[:nc, 9:] - g.pose_gt[:nc, 9:], axis=1).mean()
    assert err1 < err0
    # object motions start at identity and must move towards the true motion
    m0 = np.linalg.norm(g.pose[nc:, 9:] - g.pose_gt[nc:, 9:], axis=1).mean()
    m1 = np.linalg.norm(pose[nc:, 9:] - g.pose_gt[nc:, 9:], axis=1).mean()
    assert m1 < 0.5 * m0


def test_single_point_schur_block_has_a_closed_form_in_the_weight_and_the_camera_frame_point(oracle):
    """The identity the GPU preconditioner kernel rests on (vdo_slam_amd/csrc/ba_solve.hip k_precond_tile): for an EdgeSE3PointXYZ the
    pose-landmark block is  B = -we [I ; 2[c]x] R^T  (c: the point in the pose's frame, we: Huber-weighted information), so for a landmark
    with the scalar block  Hll = h I  the Schur term  B (h + lambda)^-1 B^T  does not depend on R:
        g we^2 [[I, -2[c]x], [2[c]x, 4 (|c|^2 I - c c^T)]] .
    Checked on the ORACLE's explicit 6x3 blocks of a random graph (we, R and c are read back from each block itself)."""
    g = synth.make_ba_graph(6, 120, 1, 8, seed=9)
    gc, keep = K.graph_to_c(g)
    S = K.BASystem(g)
    assert oracle.vdo_oracle_ba_linearize(C.byref(gc), C.byref(S.c)) == 0
    lam = 0.37
    worst, checked = 0.0, 0
    for e in range(0, g.n_eb, 7):
        B = S.Hpl_eb[:, e].reshape(6, 3)
        we = np.linalg.norm(B[0])
        if we < 1e-9:
            continue
        Rt = -B[:3] / we                                       # top block = -we R^T
        np.testing.assert_allclose(Rt @ Rt.T, np.eye(3), atol=1e-9)
        Cx = -(B[3:] @ Rt.T) / (2.0 * we)                     # bottom block = -2 we [c]x R^T
        c = np.array([Cx[2, 1], Cx[0, 2], Cx[1, 0]])
        np.testing.assert_allclose(Cx, np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]]), atol=1e-9 * max(1.0, np.abs(c).max()))
        l = g.eb_point[e]
        h = S.Hll[l].reshape(3, 3)
        if np.abs(h - h[0, 0] * np.eye(3)).max() > 1e-9 * h[0, 0]:      # a point of a dynamic track: not the scalar case
            continue
        gsc = 1.0 / (h[0, 0] + lam)
        explicit = gsc * B @ B.T
        s = gsc * we * we
        K2 = 2.0 * Cx
        closed = s * np.block([[np.eye(3), K2.T], [K2, 4.0 * (c @ c * np.eye(3) - np.outer(c, c))]])
        worst = max(worst, np.abs(explicit - closed).max() / np.abs(explicit).max())
        checked += 1
    assert checked >= 40 and worst < 1e-12, (checked, worst)
