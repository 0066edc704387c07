"""The product's EPnP refit (host code of the C-ABI, vdo_slam_amd/csrc/epnp_refit.hpp) against the oracle's INDEPENDENT
restatement (oracle/epnp_oracle.hpp: OpenCV's numerical tools - SVD everywhere - where the product uses running sums, QL / Jacobi
eigen-solvers, normal equations and Horn's quaternion).  Runs on the CPU: the product header is plain C++ and is compiled here
behind a C entry point (tests/helpers/product_host_epnp.cpp).  The two share the algorithm, not the arithmetic: 1e-9, not bits."""
import numpy as np
import pytest

from tests.test_oracle_p3p import _bind, _scene
from vdo_slam_amd import _capi as K
from vdo_slam_amd.synth import KITTI_K



@pytest.fixture(scope="module")
def product_epnp():
    from tests import oracle_lib
    return oracle_lib.load_product_epnp()


def _both(P, o, Xw, uv):
    K4 = np.array(KITTI_K, np.float64)
    Ta, Tb = np.zeros(16), np.zeros(16)
    n = Xw.shape[0]
    ea = P.product_host_epnp(n, K._dp(Xw), K._dp(uv), K._dp(K4), K._dp(Ta))
    eb = o.vdo_oracle_epnp(n, K._dp(Xw), K._dp(uv), K._dp(K4), K._dp(Tb))
    return ea, Ta.reshape(4, 4), eb, Tb.reshape(4, 4)


def test_product_epnp_agrees_with_the_independent_oracle(oracle, product_epnp):
    o = _bind(oracle)
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(300):
        n = int(rng.choice([4, 5, 6, 12, 40, 150, 600, 1200]))
        sig = float(rng.choice([0.0, 0.1, 0.3, 1.0]))
        Xw, uv, R, t, _ = _scene(rng, n, 0.0, pix_sigma=sig)
        Xw = np.ascontiguousarray(Xw.astype(np.float32).astype(np.float64))          # (the reference's 3-D points are CV_32F)
        ea, Ta, eb, Tb = _both(product_epnp, o, Xw, uv)
        assert (ea < 0) == (eb < 0), (trial, n, sig, ea, eb)
        if ea < 0:
            continue
        if n <= 5:
            # 4-5 points: 2n < 11 equations, the null space of M has more than one dimension whatever the data and the answer is decided
            # by how each side truncates its pseudo-inverses (1e-7 of the largest singular value vs 2 eps of their sum) - only sanity here
            assert np.isfinite(Ta).all() and np.isfinite(Tb).all()
            continue
        d = np.abs(Ta - Tb).max() / max(1.0, np.abs(Tb).max())
        worst = max(worst, d)
        assert d < 1e-9, (trial, n, sig, d, ea, eb)
        assert abs(ea - eb) <= 1e-9 * max(1.0, eb)
    assert worst > 0.0            # different arithmetic: agreement is evidence, not identity


def test_control_point_signs_follow_the_same_convention(oracle, product_epnp):
    """With noisy data EPnP's pose depends on which side of the centroid each control point lies (the sign of the principal
    directions).  Scatter matrices whose principal directions come out of the decomposition with every sign pattern: same pose."""
    o = _bind(oracle)
    rng = np.random.default_rng(8)
    for trial in range(60):
        Xw, uv, R, t, _ = _scene(rng, 30, 0.0, pix_sigma=0.5)
        Q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        Xw = np.ascontiguousarray((Xw - Xw.mean(0)) @ Q * rng.uniform(0.3, 3, 3) + Xw.mean(0))    # arbitrary orientation / anisotropy of the cloud
        Xc = Xw @ R.T + t
        if Xc[:, 2].min() < 1.0:
            continue
        uv = np.ascontiguousarray(np.c_[KITTI_K[0] * Xc[:, 0] / Xc[:, 2] + KITTI_K[2], KITTI_K[1] * Xc[:, 1] / Xc[:, 2] + KITTI_K[3]] + rng.normal(0, 0.5, (30, 2)))
        ea, Ta, eb, Tb = _both(product_epnp, o, Xw, uv)
        assert np.abs(Ta - Tb).max() < 1e-8 * max(1.0, np.abs(Tb).max()), (trial, ea, eb)


def _planar_patch(rng, n, flat, sig):
    """n points on a patch of relative thickness `flat` (the visible face of a box: 0 = an exact plane before the float rounding of the 3-D points), any
    orientation, 8-25 m away, seen through a small random pose with `sig` px of pixel noise."""
    from scipy.spatial.transform import Rotation as Rot
    K4 = np.array(KITTI_K, np.float64)
    ext = rng.uniform(0.5, 2.0)
    loc = np.c_[rng.uniform(-ext, ext, n), rng.uniform(-ext / 2, ext / 2, n), rng.normal(0, flat * ext, n) if flat > 0 else np.zeros(n)]
    Q = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    Xc = loc @ Q.T + np.array([rng.uniform(-5, 5), rng.uniform(-1, 1), rng.uniform(8, 25)])
    R = Rot.from_rotvec(rng.normal(0, 0.05, 3)).as_matrix(); t = rng.normal(0, 0.3, 3)
    Xw = np.ascontiguousarray(((Xc - t) @ R).astype(np.float32).astype(np.float64))          # (the reference's 3-D points are CV_32F)
    Xc = Xw @ R.T + t
    uv = np.c_[K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]] + rng.normal(0, sig, (n, 2))
    return Xw, np.ascontiguousarray(uv)


def test_near_planar_and_coplanar_sets_agree(oracle, product_epnp):
    """VERDICT r4 #6.  Rounds 2-4: on NEAR-PLANAR inlier sets (the visible face of a box) the two restatements differed by up to 1.5 in the pose, and
    (near-)coplanar sets were left to the RANSAC hypothesis on both sides - a liberty.  Both now do what OpenCV 3.4's epnp.cpp does with its own tools (one-sided
    Jacobi SVD: pseudo-inverse of the control-point matrix and of the beta systems with the 2 eps sum(w) threshold, R = U V^T with the third ROW negated when
    det < 0): 1 200 patches of relative thickness 0 (exact planes) .. 0.1, 6 .. 400 points, 0 .. 0.3 px: poses within 1e-6 (measured: 6e-8), INCLUDING the
    ones where OpenCV's row flip returns a rotation that is not the nearest one (mean reprojection error of several pixels - what the reference receives)."""
    o = _bind(oracle)
    rng = np.random.default_rng(11)
    worst, n_flipped = 0.0, 0
    for trial in range(1200):
        n = int(rng.choice([6, 12, 30, 100, 400]))
        flat = float(rng.choice([0.0, 1e-7, 1e-6, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 1e-1]))
        sig = float(rng.choice([0.0, 0.1, 0.3]))
        Xw, uv = _planar_patch(rng, n, flat, sig)
        ea, Ta, eb, Tb = _both(product_epnp, o, Xw, uv)
        assert np.isfinite(Ta).all() and np.isfinite(Tb).all() and ea >= 0 and eb >= 0, (trial, n, flat, sig)
        d = np.abs(Ta - Tb).max() / max(1.0, np.abs(Tb).max())
        worst = max(worst, d)
        assert d <= 1e-6, (trial, n, flat, sig, d, ea, eb)
        assert abs(ea - eb) <= 1e-6 * max(1.0, eb)
        n_flipped += bool(sig > 0 and eb > 10 * sig + 1.0)
    assert n_flipped >= 20          # the row-flip artefact is in the sample (and equal on both sides)
    print("near-planar EPnP: worst product-vs-oracle %.1e, %d of 1200 with the row-flip artefact" % (worst, n_flipped))


def test_exactly_coplanar_points_go_through(oracle, product_epnp):
    """Points in one world plane, exact or noisy pixels: the fourth control point falls onto the centroid, the pseudo-inverse gives every point a zero fourth
    barycentric coordinate, M^T M gets three exactly-zero eigenvalues beside its null vector (the product switches to OpenCV's Jacobi SVD for the degenerate
    eigenspace) - and with exact pixels the true pose comes out."""
    o = _bind(oracle)
    rng = np.random.default_rng(2)
    for trial in range(60):
        n = int(rng.choice([6, 40, 200])); sig = float(rng.choice([0.0, 0.3]))
        Xw, uv, R, t, _ = _scene(rng, n, 0.0)
        Xw[:, 2] = 12.0
        Xc = Xw @ R.T + t
        uv = np.ascontiguousarray(np.c_[KITTI_K[0] * Xc[:, 0] / Xc[:, 2] + KITTI_K[2], KITTI_K[1] * Xc[:, 1] / Xc[:, 2] + KITTI_K[3]] + rng.normal(0, sig, (n, 2)))
        ea, Ta, eb, Tb = _both(product_epnp, o, np.ascontiguousarray(Xw), uv)
        assert ea >= 0 and eb >= 0
        assert np.abs(Ta - Tb).max() <= 1e-6 * max(1.0, np.abs(Tb).max()), (trial, n, sig, np.abs(Ta - Tb).max())
        if sig == 0.0 and n >= 40:
            assert eb < 1e-6 and np.abs(Tb[:3, :3] - R).max() < 1e-6 and np.abs(Tb[:3, 3] - t).max() < 1e-5, (trial, eb)


@pytest.mark.parametrize("scenario", ["exact", "low_noise", "twelve_objects"])
def test_track_sequences_do_not_depend_on_whose_epnp_seeds_the_lm(oracle, scenario):
    """End to end: the oracle-composed Track() with its own EPnP and with the product's host EPnP (CPU build) as the refit of every
    RANSAC model - same counts in every frame, same poses and motions - on sequences where the LM is not chaotic in its seed (exact
    or low-noise flow).  On the noisy 5-object sequences a float ulp of the seed can change an object's LM path; those tests inject
    the product's seed instead (tests/pipeline_ref.py)."""
    from tests.pipeline_ref import OraclePipeline
    from vdo_slam_amd import synth_seq as SQ
    n, objs, kw = {"exact": (6, SQ.default_objects(), {}),
                   "low_noise": (6, SQ.default_objects(), dict(flow_sigma=0.1)),
                   "twelve_objects": (4, [dict(c=np.array([-6.6 + 1.2 * j, 0.9, 9.0 + 1.5 * (j % 3)]), hw=0.45, hh=0.6, v=np.array([0.0, 0.0, 0.7 + 0.015 * j])) for j in range(12)],
                                      dict(flow_sigma=0.05))}[scenario]
    Ts = SQ.camera_poses(n)
    a, b = OraclePipeline(oracle, build_lm=True), OraclePipeline(oracle, build_lm=True, seed_refit="product")
    for k in range(n):
        fr = SQ.render_frame(k, Ts, objs, **kw)
        ca, cb = a.step(fr), b.step(fr)
        assert ca == cb, (k, {q: (ca[q], cb[q]) for q in ca if ca[q] != cb[q]})
        np.testing.assert_allclose(a.Tl, b.Tl, rtol=0, atol=1e-6)
        assert len(a.motions) == len(b.motions)
        for ma, mb in zip(a.motions, b.motions):
            np.testing.assert_allclose(ma["H"], mb["H"], rtol=0, atol=1e-5)
