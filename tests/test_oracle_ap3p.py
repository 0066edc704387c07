"""AP3P (Ke & Roumeliotis) as restated in oracle/ap3p_oracle.cpp - the minimal solver the reference's cv::solvePnPRansac(..., SOLVEPNP_AP3P)
calls name (src/Tracking.cc:1652-1657, 1755-1760) - against the geometry and against Grunert's P3P (oracle/p3p_oracle.cpp, the one the
product runs): every solution is a rotation that maps the three world points onto their bearings, the true pose is among them, the two
solvers return the same SET of poses, and the reference's RANSAC with either solver inside finds the same model on clean-inlier data.
OpenCV itself is not in the image: the ORDER of AP3P's solutions and their last bits are unpinned (header of ap3p_oracle.cpp)."""
import ctypes as C

import numpy as np

from vdo_slam_amd import _capi as K
from vdo_slam_amd.synth import KITTI_K, rotvec_to_R

dp = K.c_double_p


def _bind(o):
    o.vdo_oracle_ap3p.argtypes = [dp, dp, dp, dp]
    o.vdo_oracle_p3p.argtypes = [dp, dp, dp, dp]
    o.vdo_oracle_ap3p_quartic.argtypes = [dp, dp]
    sig = [C.c_int, dp, dp, dp, C.c_int, C.c_double, C.c_double, dp, K.c_uint8_p, K.c_int32_p, K.c_int32_p]
    o.vdo_oracle_ap3p_ransac.argtypes = sig
    o.vdo_oracle_p3p_ransac.argtypes = sig
    o.vdo_oracle_ap3p_lf_ransac.argtypes = sig
    o.vdo_oracle_ap3p_lf.argtypes = [dp, dp, dp, dp]
    o.vdo_oracle_ap3p_quartic_lf.argtypes = [dp, dp]
    return o


def _scene(rng, n):
    R = rotvec_to_R(rng.normal(size=3) * 0.3)
    t = rng.normal(size=3) * np.array([0.5, 0.3, 1.0])
    Xc = np.stack([rng.uniform(-6, 6, n), rng.uniform(-2, 2, n), rng.uniform(4, 30, n)], 1)
    Xw = (Xc - t) @ R                      # x_cam = R x_world + t
    return R, t, Xw, Xc


def _poses(fn, f, Xw):
    Ro = np.zeros((4, 9)); to = np.zeros((4, 3))
    n = fn(K._dp(np.ascontiguousarray(f)), K._dp(np.ascontiguousarray(Xw)), K._dp(Ro), K._dp(to))
    return [(Ro[i].reshape(3, 3).copy(), to[i].copy()) for i in range(n)]


def test_ap3p_quartic_roots(oracle):
    """Ferrari in complex arithmetic + two Newton steps: the real roots of quartics with four, two and no real roots (numpy's companion-matrix
    roots as the reference; a complex pair's entry carries its real part - as in OpenCV, whose caller then drops |cos| > 1 or meets no solution)."""
    o = _bind(oracle)
    rng = np.random.default_rng(2)
    for trial in range(300):
        r = rng.uniform(-0.95, 0.95, 4)
        co = np.poly(r) * rng.uniform(0.5, 3.0)
        out = np.zeros(4)
        o.vdo_oracle_ap3p_quartic(K._dp(np.ascontiguousarray(co)), K._dp(out))
        gap = np.abs(r[:, None] - r[None, :])[np.triu_indices(4, 1)].min()
        if gap < 1e-3:
            continue
        assert np.abs(np.sort(out) - np.sort(r)).max() <= 1e-9 / gap, (r, out)


def test_ap3p_solutions_are_poses_and_equal_grunerts(oracle):
    """Every solution puts the three world points on their lines of sight.  The ones from REAL roots of the quartic are rotations, the true pose is
    among them, and those with all three points in front of the camera are exactly Grunert's solutions (which keeps positive depths only).  A complex
    pair of roots comes back from solveQuartic as its real part, twice: two equal, slightly non-orthogonal "solutions" - ap3p.cpp of 3.4 does not
    test for that (only |cos| <= 1), the fourth point of a RANSAC sample sorts them out; kept, counted here."""
    o = _bind(oracle)
    rng = np.random.default_rng(5)
    n_sol, n_spurious, n_behind = [], 0, 0
    for trial in range(400):
        R, t, Xw, Xc = _scene(rng, 3)
        f = Xc / np.linalg.norm(Xc, axis=1, keepdims=True)
        A = _poses(o.vdo_oracle_ap3p, f, Xw)
        G = _poses(o.vdo_oracle_p3p, f, Xw)
        assert 1 <= len(A) <= 4
        n_sol.append(len(A))
        proper = []
        for Rs, ts in A:
            Y = Xw @ Rs.T + ts
            Yn = Y / np.linalg.norm(Y, axis=1, keepdims=True)
            assert np.abs(np.abs((Yn * f).sum(1)) - 1).max() < 1e-8, trial        # on the lines of sight, whichever root it came from
            if np.abs(Rs @ Rs.T - np.eye(3)).max() < 1e-7:
                assert abs(np.linalg.det(Rs) - 1) < 1e-7
                proper.append((Rs, ts))
            else:
                n_spurious += 1
        assert len(A) - len(proper) in (0, 2)                                     # (a complex pair gives its real part twice)
        assert min(max(np.abs(Rs - R).max(), np.abs(ts - t).max()) for Rs, ts in proper) < 1e-6, trial       # the true pose is there
        front = [(Rs, ts) for Rs, ts in proper if (((Xw @ Rs.T + ts) * f).sum(1) > 0).all()]
        n_behind += len(proper) - len(front)
        dist = lambda P, Q: max(np.abs(P[0] - Q[0]).max(), np.abs(P[1] - Q[1]).max() / max(1.0, np.abs(Q[1]).max()))
        for g in G:
            assert min(dist(a, g) for a in front) < 1e-5, (trial, len(A), len(G))
        for a in front:
            assert min(dist(a, g) for g in G) < 1e-5, (trial, len(A), len(G))
    assert max(n_sol) >= 4 and n_spurious > 0 and n_behind > 0                    # (all three kinds occur in 400 scenes)


def test_ransac_with_ap3p_finds_the_model_grunert_finds(oracle):
    """The reference's RANSAC (same subsets, same vote, same budget rule) around either solver: on data whose inliers are exact the winning
    hypothesis has the same inlier set; with 0.1 px noise the sets agree up to the points within 1e-6 px^2 of the gate... counted, not asserted
    equal - the rounding of the two solvers differs, which is all DESIGN.md claims about them."""
    o = _bind(oracle)
    rng = np.random.default_rng(9)
    K4 = np.array([KITTI_K[0], KITTI_K[1], KITTI_K[2], KITTI_K[3]], np.float64) if len(KITTI_K) == 4 else np.array([KITTI_K[0, 0], KITTI_K[1, 1], KITTI_K[0, 2], KITTI_K[1, 2]], np.float64)
    for noise in (0.0, 0.1):
        R, t, Xw, Xc = _scene(rng, 400)
        uv = np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], 1)
        uv += rng.normal(size=uv.shape) * noise
        bad = rng.choice(400, 120, replace=False)
        uv[bad] += rng.uniform(3, 40, (120, 2)) * rng.choice([-1, 1], (120, 2))
        res = []
        for fn in (o.vdo_oracle_ap3p_ransac, o.vdo_oracle_p3p_ransac):
            T = np.zeros(16); inl = np.zeros(400, np.uint8); its = C.c_int32(0); bi = C.c_int32(0)
            good = fn(400, K._dp(np.ascontiguousarray(Xw)), K._dp(np.ascontiguousarray(uv)), K._dp(K4), 500, 0.4, 0.98, K._dp(T), inl.ctypes.data_as(K.c_uint8_p), C.byref(its), C.byref(bi))
            res.append((good, T.reshape(4, 4).copy(), inl.copy(), its.value, bi.value))
        (ga, Ta, ia, ita, bia), (gg, Tg, ig, itg, big) = res
        assert ga >= 4 and gg >= 4
        if noise == 0.0:
            assert bia == big and ita == itg and np.array_equal(ia, ig) and ga == 280
            assert np.abs(Ta - Tg).max() < 1e-6 and np.abs(Ta[:3, :3] - R).max() < 1e-6 and np.abs(Ta[:3, 3] - t).max() < 1e-5
        else:
            assert abs(ga - gg) <= 0.05 * gg and np.abs(Ta[:3, :3] - Tg[:3, :3]).max() < 5e-3


def test_libm_free_form_equals_the_complex_form(oracle):
    """The product runs Ferrari's formulas in real arithmetic with +, -, *, / and sqrt only (csrc/ransac.hip ap3p_quartic; its twin in the oracle is
    vdo_oracle_ap3p_lf*: same operations, same bits on the GPU - tests/test_ransac_gpu.py).  Here the twin against the std::complex / libm form above,
    which shares nothing with it but the algebra: the four root slots (real parts, OpenCV's order) of the quartics of 3 000 scenes, every solution of
    every scene, and whole RANSAC runs - same winning hypothesis, same number of hypotheses examined, same inlier set - on 120 seeded problems with
    0 .. 0.5 px of noise and 10 .. 50 % outliers."""
    o = _bind(oracle)
    rng = np.random.default_rng(23)
    worst = 0.0
    for trial in range(3000):
        R, t, Xw, Xc = _scene(rng, 3)
        f = Xc / np.linalg.norm(Xc, axis=1, keepdims=True)
        A = _poses(o.vdo_oracle_ap3p, f, Xw)
        B = _poses(o.vdo_oracle_ap3p_lf, f, Xw)
        assert len(A) == len(B), trial
        for (Ra, ta), (Rb, tb) in zip(A, B):                     # same order
            d = max(np.abs(Ra - Rb).max(), np.abs(ta - tb).max() / max(1.0, np.abs(ta).max()))
            worst = max(worst, d)
            assert d < 1e-7, (trial, d)
    for trial in range(500):
        r = rng.uniform(-0.95, 0.95, 4)
        if trial % 3 == 1:                                       # a complex pair
            a, b = rng.uniform(-0.9, 0.9), rng.uniform(0.05, 0.8)
            co = np.real(np.poly([r[0], r[1], a + 1j * b, a - 1j * b]))
        else:
            co = np.poly(r)
        co = np.ascontiguousarray(co * rng.uniform(0.5, 3.0))
        x1, x2 = np.zeros(4), np.zeros(4)
        o.vdo_oracle_ap3p_quartic(K._dp(co), K._dp(x1)); o.vdo_oracle_ap3p_quartic_lf(K._dp(co), K._dp(x2))
        assert np.abs(x1 - x2).max() < 1e-7 * max(1.0, np.abs(x1).max()), (trial, x1, x2)
    K4 = np.array(KITTI_K[:4], np.float64)
    same = 0
    for trial in range(120):
        n = int(rng.integers(40, 900)); noise = (0.0, 0.1, 0.3, 0.5)[trial % 4]; frac = (0.1, 0.3, 0.5)[trial % 3]
        R, t, Xw, Xc = _scene(rng, n)
        uv = np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], 1) + rng.normal(size=(n, 2)) * noise
        bad = rng.choice(n, int(frac * n), replace=False)
        uv[bad] += rng.uniform(3, 40, (bad.size, 2)) * rng.choice([-1, 1], (bad.size, 2))
        res = []
        for fn in (o.vdo_oracle_ap3p_ransac, o.vdo_oracle_ap3p_lf_ransac):
            T = np.zeros(16); inl = np.zeros(n, np.uint8); its = C.c_int32(0); bi = C.c_int32(0)
            good = fn(n, K._dp(np.ascontiguousarray(Xw)), K._dp(np.ascontiguousarray(uv)), K._dp(K4), 500, 0.4, 0.98, K._dp(T), inl.ctypes.data_as(K.c_uint8_p), C.byref(its), C.byref(bi))
            res.append((good, T.reshape(4, 4).copy(), inl.copy(), its.value, bi.value))
        (ga, Ta, ia, ita, bia), (gb, Tb, ib, itb, bib) = res
        if (ga, ita, bia) == (gb, itb, bib) and np.array_equal(ia, ib):
            same += 1
            assert np.abs(Ta - Tb).max() < 1e-7 * max(1.0, np.abs(Ta).max())
        else:                                                    # a point within rounding of the 0.4 px gate decided differently: the models must still agree
            assert abs(ga - gb) <= 2 and np.abs(Ta[:3, :3] - Tb[:3, :3]).max() < 1e-3, (trial, ga, gb, bia, bib)
    assert same >= 117, same
