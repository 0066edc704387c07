"""GPU RANSAC initialiser (vdo_slam_amd/csrc/ransac.hip) against the oracle's sequential run: same winning
hypothesis, same number of iterations examined, same inlier set, and the SAME BITS in the pose (the minimal solver - AP3P in the layout of
OpenCV 3.4's ap3p.cpp since round 5, Grunert's P3P on request - is written with IEEE-exact operations only on both sides)."""
import ctypes as C
import time

import numpy as np
import pytest

from tests.test_oracle_p3p import _bind, _scene
from vdo_slam_amd import _capi as K
from vdo_slam_amd.ba import Context
from vdo_slam_amd.ransac import pnp_ransac_batch
from vdo_slam_amd.synth import KITTI_K

pytestmark = pytest.mark.gpu


def _oracle(o, Xw, uv, refit=0, solver="ap3p"):
    refit = int(refit) | (2 if solver == "grunert" else 0)           # (bit 1: Grunert, as vdo_pnp_problem.refit)
    n = Xw.shape[0]
    T = np.zeros(16); inl = np.zeros(max(n, 1), np.uint8); its = C.c_int32(); bi = C.c_int32()
    K4 = np.array(KITTI_K, np.float64)
    good = o.vdo_oracle_pnp_ransac_refit(n, K._dp(Xw), K._dp(uv), K._dp(K4), 500, 0.4, 0.98, refit, K._dp(T), inl.ctypes.data_as(K.c_uint8_p), C.byref(its), C.byref(bi))
    return dict(T=T.reshape(4, 4), n_inliers=good, iterations_run=its.value, best_iteration=bi.value, inliers=inl[:n])


@pytest.mark.parametrize("solver", ["ap3p", "grunert"])
@pytest.mark.parametrize("refit", [0, 1])
def test_batch_matches_the_sequential_oracle(oracle, refit, solver):
    """refit = 0: the pose is the same bit pattern on both sides.  refit = 1: + OpenCV's final EPnP re-estimation on the inliers -
    host code of the C-ABI (csrc/epnp_refit.hpp) against the oracle's INDEPENDENT restatement (oracle/epnp_oracle.hpp: SVD-based,
    different arithmetic): consensus and inliers identical, pose to 1e-9."""
    o = _bind(oracle)
    ctx = Context(0)
    rng = np.random.default_rng(5)
    cases = [(1200, 0.3), (800, 0.5), (400, 0.2), (150, 0.7), (60, 0.0), (4, 0.0), (3, 0.0), (0, 0.0)]
    probs = []
    for n, outl in cases:
        if n:
            Xw, uv, R, t, _ = _scene(rng, n, outl, pix_sigma=0.1)
        else:
            Xw, uv = np.zeros((0, 3)), np.zeros((0, 2))
        probs.append((Xw, uv))
    got = pnp_ransac_batch(ctx, probs, KITTI_K, refit=refit, solver=solver)
    for (n, outl), g, (Xw, uv) in zip(cases, got, probs):
        e = _oracle(o, Xw, uv, refit, solver)
        assert g["n_inliers"] == e["n_inliers"] and g["iterations_run"] == e["iterations_run"] and g["best_iteration"] == e["best_iteration"], (n, g, e)
        assert np.array_equal(g["inliers"], e["inliers"])
        if refit and e["n_inliers"] >= 6:
            assert np.abs(g["T"] - e["T"]).max() <= 1e-9 * max(1.0, np.abs(e["T"]).max()), (n, np.abs(g["T"] - e["T"]).max())
        elif refit:
            # 4 - 5 inliers: 2n < 11 equations, the null space of EPnP's M has several dimensions whatever the data and the answer is decided by
            # how each side truncates its pseudo-inverses (tests/test_epnp_independent.py) - the consensus above is what is pinned
            assert np.isfinite(g["T"]).all()
        else:
            assert np.array_equal(g["T"], e["T"]), (n, np.abs(g["T"] - e["T"]).max())
    assert got[0]["n_inliers"] > 700 and got[0]["iterations_run"] < 500
    assert got[6]["n_inliers"] == 0 and np.array_equal(got[6]["T"], np.eye(4))


def test_pure_outliers_run_the_full_budget(oracle):
    """No consistent pose (what the bench's chained random frames produce): all 500 hypotheses are examined."""
    o = _bind(oracle)
    ctx = Context(0)
    rng = np.random.default_rng(9)
    Xw = np.c_[rng.uniform(-15, 15, 1200), rng.uniform(-3, 3, 1200), rng.uniform(4, 40, 1200)]
    uv = np.c_[rng.uniform(0, 1242, 1200), rng.uniform(0, 375, 1200)]
    g = pnp_ransac_batch(ctx, [(Xw, uv)], KITTI_K)[0]
    e = _oracle(o, Xw, uv)
    assert g["iterations_run"] == e["iterations_run"] == 500
    assert g["n_inliers"] == e["n_inliers"] and g["best_iteration"] == e["best_iteration"] and np.array_equal(g["inliers"], e["inliers"])
    assert np.array_equal(g["T"], e["T"])
    t0 = time.perf_counter()
    for _ in range(20):
        pnp_ransac_batch(ctx, [(Xw, uv)], KITTI_K)
    print("ransac 1200 pts x 500 hyp: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))


def test_ap3p_on_200_seeded_problems(oracle):
    """the product's AP3P RANSAC against the oracle's (vdo_oracle_pnp_ransac_refit, the libm-free twin) on 208 seeded problems - 20 .. 1 500
    correspondences, 0 / 0.1 / 0.3 / 0.5 px of noise on the inliers, 0 .. 60 % outliers, refit 0 and 1: winning hypothesis, hypotheses examined, inlier
    count and inlier set EQUAL on every one; the pose bit-identical without the refit, to 1e-9 with it (the two EPnP restatements share nothing)."""
    o = _bind(oracle)
    ctx = Context(0)
    rng = np.random.default_rng(2025)
    cases = []
    for k in range(208):
        n = int((20, 60, 150, 400, 800, 1500)[k % 6]); sigma = (0.0, 0.1, 0.3, 0.5)[k % 4]; outl = (0.0, 0.1, 0.3, 0.6)[(k // 4) % 4]
        Xw, uv, R, t, _ = _scene(rng, n, outl, pix_sigma=sigma)
        cases.append((Xw, uv))
    n_refit_checked = 0
    for refit in (0, 1):
        for lo in range(0, len(cases), 8):                        # batches of 8, as a frame's camera + objects come
            batch = cases[lo:lo + 8]
            got = pnp_ransac_batch(ctx, batch, KITTI_K, refit=refit)
            for g, (Xw, uv) in zip(got, batch):
                e = _oracle(o, Xw, uv, refit)
                assert (g["n_inliers"], g["iterations_run"], g["best_iteration"]) == (e["n_inliers"], e["iterations_run"], e["best_iteration"]), (Xw.shape[0], g["n_inliers"], e["n_inliers"])
                assert np.array_equal(g["inliers"], e["inliers"])
                if not refit:
                    assert np.array_equal(g["T"], e["T"])
                elif e["n_inliers"] >= 12:
                    n_refit_checked += 1
                    assert np.abs(g["T"] - e["T"]).max() <= 1e-9 * max(1.0, np.abs(e["T"]).max()), (Xw.shape[0], e["n_inliers"], np.abs(g["T"] - e["T"]).max())
    assert n_refit_checked >= 150


def test_product_ransac_against_the_independent_ap3p_oracle(oracle):
    """The oracle above shares its minimal solver's arithmetic with the product on purpose (bit-identical poses).  This one shares only the algebra:
    the reference's RANSAC around AP3P as OpenCV 3.4 evaluates it (oracle/ap3p_oracle.cpp vdo_oracle_ap3p_ransac: Ferrari's formulas in std::complex
    with libm's pow / cbrt / sqrt), and - with solver "grunert" - nothing at all (other derivation, other quartic).  On data whose inliers are exact the
    0.4 px gate has no borderline points, so the product's run must end on the same winning hypothesis, the same number of hypotheses examined, the
    same inlier set and the same pose to rounding; with 0.1 px noise the consensus may differ by the points at the gate (reported, bounded)."""
    o = oracle
    sig = [C.c_int, K.c_double_p, K.c_double_p, K.c_double_p, C.c_int, C.c_double, C.c_double, K.c_double_p, K.c_uint8_p, K.c_int32_p, K.c_int32_p]
    o.vdo_oracle_ap3p_ransac.argtypes = sig
    ctx = Context(0)
    rng = np.random.default_rng(17)
    K4 = np.array(KITTI_K, np.float64)
    probs, exact = [], []
    for n, outl, sigma in [(900, 0.3, 0.0), (500, 0.5, 0.0), (200, 0.1, 0.0), (40, 0.0, 0.0), (700, 0.3, 0.1), (300, 0.2, 0.1)]:
        Xw, uv, R, t, _ = _scene(rng, n, outl, pix_sigma=sigma)
        probs.append((Xw, uv)); exact.append(sigma == 0.0)
    for solver in ("ap3p", "grunert"):
        got = pnp_ransac_batch(ctx, probs, KITTI_K, refit=0, solver=solver)
        for (Xw, uv), g, ex in zip(probs, got, exact):
            n = Xw.shape[0]
            T = np.zeros(16); inl = np.zeros(n, np.uint8); its = C.c_int32(); bi = C.c_int32()
            good = o.vdo_oracle_ap3p_ransac(n, K._dp(Xw), K._dp(uv), K._dp(K4), 500, 0.4, 0.98, K._dp(T), inl.ctypes.data_as(K.c_uint8_p), C.byref(its), C.byref(bi))
            T = T.reshape(4, 4)
            if ex:
                assert (g["n_inliers"], g["iterations_run"], g["best_iteration"]) == (good, its.value, bi.value), (solver, n, g["n_inliers"], good, g["best_iteration"], bi.value)
                assert np.array_equal(g["inliers"], inl)
                assert np.abs(g["T"] - T).max() <= 1e-7 * max(1.0, np.abs(T).max())
            else:
                print("noisy (%s): product %d inliers (hypothesis %d), complex-arithmetic AP3P oracle %d (hypothesis %d)" % (solver, g["n_inliers"], g["best_iteration"], good, bi.value))
                assert abs(g["n_inliers"] - good) <= 0.05 * good and np.abs(g["T"][:3, :3] - T[:3, :3]).max() < 5e-3
