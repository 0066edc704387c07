"""KATs for the RANSAC initialiser oracle (oracle/p3p_oracle.cpp): quartic roots vs numpy, the P3P quartic
coefficients vs the geometry they were derived from, pose recovery on exact data, RANSAC on contaminated data."""
import ctypes as C

import numpy as np
import pytest

from vdo_slam_amd import _capi as K
from vdo_slam_amd.synth import KITTI_K, rotvec_to_R

dp = K.c_double_p


def _bind(o):
    o.vdo_oracle_quartic.argtypes = [C.c_double] * 5 + [dp]
    o.vdo_oracle_p3p.argtypes = [dp, dp, dp, dp]
    o.vdo_oracle_cbrt_exact.argtypes = [C.c_double]; o.vdo_oracle_cbrt_exact.restype = C.c_double
    o.vdo_oracle_cubic3_largest_root.argtypes = [C.c_double, C.c_double]; o.vdo_oracle_cubic3_largest_root.restype = C.c_double
    o.vdo_oracle_ransac_subsets.argtypes = [C.c_int, C.c_int, K.c_int32_p]
    o.vdo_oracle_p3p_ransac.argtypes = [C.c_int, dp, dp, dp, C.c_int, C.c_double, C.c_double, dp, K.c_uint8_p, K.c_int32_p, K.c_int32_p]
    o.vdo_oracle_epnp.restype = C.c_double
    o.vdo_oracle_epnp.argtypes = [C.c_int, dp, dp, dp, dp]
    o.vdo_oracle_pnp_ransac_refit.argtypes = [C.c_int, dp, dp, dp, C.c_int, C.c_double, C.c_double, C.c_int, dp, K.c_uint8_p, K.c_int32_p, K.c_int32_p]
    return o


def test_libm_free_cube_root_and_cubic(oracle):
    """The +,-,*,/,sqrt-only routines the oracle and the GPU share (bit-identical RANSAC poses): within 2 ulp of cbrt over
    the whole exponent range incl. subnormals, signs and specials; the Newton-from-the-right root equals the trigonometric
    largest root of a three-real-root cubic."""
    o = _bind(oracle)
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-10, 10, 2000), 10.0 ** rng.uniform(-320, 300, 2000) * rng.choice([-1, 1], 2000),
                         [1.0, 8.0, 27.0, -64.0, 1e-310, -4e-320, 2.0 ** -1074, 1.7e308]])
    for x in xs:
        got, ref = o.vdo_oracle_cbrt_exact(float(x)), np.cbrt(x)
        assert abs(got - ref) <= 2 * np.spacing(abs(ref)), (x, got, ref)
    assert o.vdo_oracle_cbrt_exact(0.0) == 0.0 and np.isinf(o.vdo_oracle_cbrt_exact(np.inf)) and np.isnan(o.vdo_oracle_cbrt_exact(np.nan))
    assert o.vdo_oracle_cbrt_exact(27.0) == 3.0 and o.vdo_oracle_cbrt_exact(-8.0) == -2.0
    for _ in range(2000):
        r = np.sort(rng.uniform(-5, 5, 2) * 10.0 ** rng.uniform(-3, 3))
        r3 = -(r[0] + r[1])                                  # depressed: the roots sum to zero
        roots = np.array([r[0], r[1], r3])
        P = roots[0] * roots[1] + roots[0] * roots[2] + roots[1] * roots[2]
        Q = -roots.prod()
        if not (Q * Q / 4 + P * P * P / 27 < 0):
            continue
        t = o.vdo_oracle_cubic3_largest_root(float(P), float(Q))
        big = roots.max()
        gap = big - np.sort(roots)[1]
        assert abs(t - big) <= 1e-9 * abs(big) * max(1.0, abs(big) / max(gap, 1e-300) * 1e-3), (roots, t)


def test_quartic_roots_match_numpy(oracle):
    o = _bind(oracle)
    rng = np.random.default_rng(0)
    for trial in range(300):
        if trial % 3 == 0:                       # 4 real roots
            r = rng.uniform(-3, 3, 4); co = np.poly(r)
        elif trial % 3 == 1:                     # 2 real + complex pair
            z = complex(rng.uniform(-2, 2), rng.uniform(0.2, 2)); co = np.real(np.poly([rng.uniform(-3, 3), rng.uniform(-3, 3), z, z.conjugate()]))
        else:
            co = rng.normal(0, 1, 5); co[0] = rng.uniform(0.5, 2)
        co = co * rng.uniform(0.1, 10)
        out = np.zeros(4)
        n = o.vdo_oracle_quartic(*[float(c) for c in co], K._dp(out))
        ref = np.roots(co)
        ref = np.sort(ref[np.abs(ref.imag) < 1e-9].real)
        got = np.sort(out[:n])
        assert n == ref.size, (trial, co, got, ref)
        np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-8)


def _scene(rng, n, outlier_frac=0.0, pix_sigma=0.0):
    fx, fy, cx, cy = KITTI_K
    R = rotvec_to_R(rng.normal(0, 0.2, 3)); t = rng.normal(0, 1.0, 3)
    Xc = np.c_[rng.uniform(-15, 15, n), rng.uniform(-3, 3, n), rng.uniform(4, 40, n)]
    Xw = (Xc - t) @ R                                     # Xc = R Xw + t
    uv = np.c_[fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy] + rng.normal(0, pix_sigma, (n, 2)) if pix_sigma else np.c_[fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy]
    out = rng.random(n) < outlier_frac
    uv[out] += rng.uniform(-60, 60, (int(out.sum()), 2))
    return np.ascontiguousarray(Xw), np.ascontiguousarray(uv), R, t, out


def test_p3p_contains_the_true_pose(oracle):
    o = _bind(oracle)
    rng = np.random.default_rng(1)
    fx, fy, cx, cy = KITTI_K
    hits = 0
    for trial in range(200):
        Xw, uv, R, t, _ = _scene(rng, 3)
        f = np.c_[(uv[:, 0] - cx) / fx, (uv[:, 1] - cy) / fy, np.ones(3)]
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        Ro = np.zeros((4, 9)); to = np.zeros((4, 3))
        n = o.vdo_oracle_p3p(K._dp(np.ascontiguousarray(f)), K._dp(Xw), K._dp(Ro), K._dp(to))
        assert 1 <= n <= 4
        errs = [max(np.abs(Ro[s].reshape(3, 3) - R).max(), np.abs(to[s] - t).max()) for s in range(n)]
        for s in range(n):                                # every returned pose is a rotation that maps the 3 points onto their rays
            Rs = Ro[s].reshape(3, 3)
            np.testing.assert_allclose(Rs @ Rs.T, np.eye(3), atol=1e-9)
            assert np.linalg.det(Rs) > 0
            Xc = Xw @ Rs.T + to[s]
            np.testing.assert_allclose(Xc / np.linalg.norm(Xc, axis=1, keepdims=True), f, atol=2e-5)   # ill-conditioned triangles lose digits in the quartic
        hits += min(errs) < 1e-6
    assert hits >= 198                                    # (near-degenerate triangles may lose the root to rounding)


def test_subsets_are_distinct_and_deterministic(oracle):
    o = _bind(oracle)
    a = np.zeros((500, 4), np.int32); b = np.zeros((500, 4), np.int32)
    o.vdo_oracle_ransac_subsets(1200, 500, a.ctypes.data_as(K.c_int32_p)); o.vdo_oracle_ransac_subsets(1200, 500, b.ctypes.data_as(K.c_int32_p))
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() < 1200
    assert all(len(set(r)) == 4 for r in a.tolist())
    c = np.zeros((50, 4), np.int32)
    o.vdo_oracle_ransac_subsets(5, 50, c.ctypes.data_as(K.c_int32_p))
    assert all(len(set(r)) == 4 for r in c.tolist())


@pytest.mark.parametrize("n,outl", [(1200, 0.3), (300, 0.5), (60, 0.1), (4, 0.0)])
def test_ransac_finds_the_pose_among_outliers(oracle, n, outl):
    o = _bind(oracle)
    rng = np.random.default_rng(n)
    Xw, uv, R, t, is_out = _scene(rng, n, outl, pix_sigma=0.1)
    T = np.zeros(16); inl = np.zeros(n, np.uint8); its = C.c_int32(); bi = C.c_int32()
    K4 = np.array(KITTI_K, np.float64)
    good = o.vdo_oracle_p3p_ransac(n, K._dp(Xw), K._dp(uv), K._dp(K4), 500, 0.4, 0.98, K._dp(T), inl.ctypes.data_as(K.c_uint8_p), C.byref(its), C.byref(bi))
    T = T.reshape(4, 4)
    assert good == inl.sum() and good >= 4
    assert its.value <= 500 and 0 <= bi.value < its.value
    if n >= 60:
        assert good > 0.5 * (~is_out).sum()
        assert not inl[is_out].any() or inl[is_out].sum() <= 2
        assert np.abs(T[:3, :3] - R).max() < 5e-3 and np.abs(T[:3, 3] - t).max() < 0.1
        assert its.value < 500                             # the confidence rule stopped early


def test_ransac_with_too_few_points(oracle):
    o = _bind(oracle)
    T = np.zeros(16)
    X = np.zeros((3, 3)); uv = np.zeros((3, 2)); K4 = np.array(KITTI_K, np.float64)
    assert o.vdo_oracle_p3p_ransac(3, K._dp(X), K._dp(uv), K._dp(K4), 500, 0.4, 0.98, K._dp(T), None, None, None) == 0
    assert np.array_equal(T.reshape(4, 4), np.eye(4))


def test_epnp_recovers_the_pose_and_refines_the_ransac_model(oracle):
    """EPnP (4 control points, oracle/epnp_oracle.hpp): exact data -> the exact pose (any n >= 6); noisy data -> the least-squares
    pose, closer to the truth than a single P3P hypothesis; and inside the RANSAC wrapper the refit replaces the winning
    hypothesis while the inlier set stays the RANSAC one.  Coplanar points go through like any others (round 5; a liberty of rounds 2-4 left them to
    the hypothesis): exact pixels -> the exact pose."""
    o = _bind(oracle)
    rng = np.random.default_rng(11)
    K4 = np.array(KITTI_K, np.float64)
    for n in (6, 40, 700):
        Xw, uv, R, t, _ = _scene(rng, n)
        T = np.zeros(16)
        err = o.vdo_oracle_epnp(n, K._dp(Xw), K._dp(uv), K._dp(K4), K._dp(T))
        T = T.reshape(4, 4)
        assert err < 1e-8 and np.abs(T[:3, :3] - R).max() < 1e-10 and np.abs(T[:3, 3] - t).max() < 1e-9, (n, err)
        assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12
    # noisy + outliers: RANSAC hypothesis vs refit
    Xw, uv, R, t, outl = _scene(rng, 900, 0.3, pix_sigma=0.15)
    res = {}
    for refit in (0, 1):
        T = np.zeros(16); inl = np.zeros(900, np.uint8); its = C.c_int32(); bi = C.c_int32()
        good = o.vdo_oracle_pnp_ransac_refit(900, K._dp(Xw), K._dp(uv), K._dp(K4), 500, 0.4, 0.98, refit, K._dp(T), inl.ctypes.data_as(K.c_uint8_p), C.byref(its), C.byref(bi))
        res[refit] = (good, T.reshape(4, 4).copy(), inl.copy(), its.value, bi.value)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][2], res[1][2]) and res[0][3:] == res[1][3:]       # same consensus
    e0 = np.abs(res[0][1][:3, 3] - t).max(); e1 = np.abs(res[1][1][:3, 3] - t).max()
    assert e1 < e0 and e1 < 0.01, (e0, e1)                 # the refit uses all ~600 inliers instead of 3 points
    assert not np.array_equal(res[0][1], res[1][1])
    # coplanar points: the pseudo-inverse of the control-point matrix gives every point a zero fourth barycentric coordinate (cvInvert(CV_SVD)) and EPnP goes on
    Xp = Xw.copy(); Xp[:, 2] = 12.0
    Xc = Xp @ R.T + t
    uvp = np.c_[KITTI_K[0] * Xc[:, 0] / Xc[:, 2] + KITTI_K[2], KITTI_K[1] * Xc[:, 1] / Xc[:, 2] + KITTI_K[3]]
    T = np.zeros(16)
    err = o.vdo_oracle_epnp(900, K._dp(np.ascontiguousarray(Xp)), K._dp(np.ascontiguousarray(uvp)), K._dp(K4), K._dp(T))
    T = T.reshape(4, 4)
    assert 0 <= err < 1e-6 and np.abs(T[:3, :3] - R).max() < 1e-6 and np.abs(T[:3, 3] - t).max() < 1e-5, err


def test_p3p_solution_set_is_complete_against_a_numerical_solver(oracle):
    """Second opinion on the minimal solver that shares no algebra with it: the three law-of-cosines equations
    d_i^2 + d_j^2 - 2 d_i d_j cos(theta_ij) = |P_i - P_j|^2 are solved for the depths by Newton from many starting points; the
    set of positive solutions found that way must be the set of depths |R P_i + t| the oracle's P3P returns (Grunert's quartic,
    resolvent cubic, absolute orientation) - no root lost, none invented."""
    o = _bind(oracle)
    rng = np.random.default_rng(21)
    fx, fy, cx, cy = KITTI_K
    checked = multi = 0
    for trial in range(120):
        Xw, uv, R, t, _ = _scene(rng, 3)
        f = np.c_[(uv[:, 0] - cx) / fx, (uv[:, 1] - cy) / fy, np.ones(3)]
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        cosv = np.array([f[0] @ f[1], f[0] @ f[2], f[1] @ f[2]])
        dist2 = np.array([((Xw[0] - Xw[1]) ** 2).sum(), ((Xw[0] - Xw[2]) ** 2).sum(), ((Xw[1] - Xw[2]) ** 2).sum()])
        pairs = ((0, 1), (0, 2), (1, 2))

        def F(d):
            return np.array([d[i] ** 2 + d[j] ** 2 - 2 * d[i] * d[j] * c - q for (i, j), c, q in zip(pairs, cosv, dist2)])

        def J(d):
            Jm = np.zeros((3, 3))
            for r, ((i, j), c) in enumerate(zip(pairs, cosv)):
                Jm[r, i] = 2 * d[i] - 2 * d[j] * c; Jm[r, j] = 2 * d[j] - 2 * d[i] * c
            return Jm
        found = []
        scale = np.sqrt(dist2.max())
        for _ in range(400):
            d = rng.uniform(0.05, 60.0, 3)
            ok = False
            for _it in range(60):
                try:
                    step = np.linalg.solve(J(d), F(d))
                except np.linalg.LinAlgError:
                    break
                d = d - step
                if np.abs(step).max() < 1e-13 * max(1.0, np.abs(d).max()):
                    ok = True
                    break
            if ok and d.min() > 1e-6 and np.abs(F(d)).max() < 1e-9 * scale ** 2 and not any(np.abs(d - e).max() < 1e-6 * max(1.0, np.abs(e).max()) for e in found):
                found.append(d)
        # ill-conditioned triangles (nearly double roots) are left out: Newton and the quartic both lose digits there
        if any(abs(np.linalg.det(J(d))) < 1e-3 * scale ** 3 for d in found):
            continue
        Ro = np.zeros((4, 9)); to = np.zeros((4, 3))
        n = o.vdo_oracle_p3p(K._dp(np.ascontiguousarray(f)), K._dp(Xw), K._dp(Ro), K._dp(to))
        got = [np.linalg.norm(Xw @ Ro[s].reshape(3, 3).T + to[s], axis=1) for s in range(n)]
        assert len(got) == len(found), (trial, got, found)
        for d in found:
            assert min(np.abs(d - g).max() for g in got) < 1e-5 * max(1.0, d.max()), (trial, d, got)
        checked += 1; multi += len(found) > 1
    assert checked >= 60 and multi >= 10
