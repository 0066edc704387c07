"""KATs for the per-frame joint pose+flow oracle (PoseOptimizationFlow2Cam / Flow2)."""
import ctypes as C

import numpy as np
import pytest

from vdo_slam_amd import _capi as K
from vdo_slam_amd import synth


def run_oracle(oracle, prob):
    pc, keep = K.flow2_to_c(prob)
    T = np.zeros(16); flow = np.zeros((prob.n, 2)); inl = np.zeros(prob.n, np.uint8)
    st = K.LMStatsC()
    ninl = oracle.vdo_oracle_flow2_optimize(C.byref(pc), K._dp(T), K._dp(flow), inl.ctypes.data_as(K.c_uint8_p), C.byref(st))
    return T.reshape(4, 4), flow, inl, ninl, st


def pose_err(T, Tt):
    dR = T[:3, :3] @ Tt[:3, :3].T
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))
    return ang, np.linalg.norm(T[:3, 3] - Tt[:3, 3])


@pytest.mark.parametrize("is_object,n", [(False, 1200), (True, 300)])
def test_intended_lm_recovers_pose(oracle, is_object, n):
    """ref_quirks=0 (the mathematically intended 2x2 Schur step) converges to the true pose."""
    prob = synth.make_flow2_problem(n, seed=4, is_object=is_object)
    prob.ref_quirks = 0
    T, flow, inl, ninl, st = run_oracle(oracle, prob)
    a0, t0 = pose_err(prob.T0, prob.T_true)
    a1, t1 = pose_err(T, prob.T_true)
    assert st.final_chi2 < st.initial_chi2
    assert t1 < 0.5 * t0 and a1 < max(0.5 * a0, 2e-4)
    assert ninl > 0.6 * n


@pytest.mark.parametrize("is_object,n", [(False, 1200), (True, 300), (False, 40)])
def test_reference_quirk_mode_is_stable(oracle, is_object, n):
    """ref_quirks=1 reproduces the BlockSolver_6_3/2-DoF mismatch (F3): steps are not true LM
    steps but the gain test keeps chi2 non-increasing and the result stays near the truth."""
    prob = synth.make_flow2_problem(n, seed=5, is_object=is_object)
    T, flow, inl, ninl, st = run_oracle(oracle, prob)
    assert st.iterations >= 1
    assert st.final_chi2 <= st.initial_chi2 * (1 + 1e-12)
    a1, t1 = pose_err(T, prob.T_true)
    assert t1 < 0.5 and a1 < 0.05
    R = T[:3, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)


def test_fewer_than_three_matches_returns_identity(oracle):
    prob = synth.make_flow2_problem(2, seed=1)
    T, flow, inl, ninl, st = run_oracle(oracle, prob)
    assert ninl == 0 and np.array_equal(T, np.eye(4))


def test_f3_lm_is_chaotic_in_the_seed(oracle):
    """The claim behind OraclePipeline(seed_refit="product") and DESIGN.md §2, as a test on the ORACLE alone (no product code): every
    object problem the oracle-composed Track() builds on 9 frames of the noisy 5-object sequence (0.3 px flow noise, 2 % invalid depth)
    is solved from its seed and from the seed with every element of [R | t] moved by ONE float ulp - the size of the disagreement
    between two correct EPnP implementations after the CV_32F cast.
      * ref_quirks = 1 (the reference's LM: 2-DoF flow vertices aliased onto BlockSolver_6_3's 3x3 blocks): most problems
        keep the perturbation at its size, but the weakly constrained ones (small, distant objects: 100+ Levenberg iterations that never
        settle) end somewhere else - beyond the north star's 1e-4 on the pose, with another iteration count.  So 1e-4 parity of such
        an object motion with the real reference is out of reach for ANY implementation of cv::solvePnPRansac's refit, and a
        frame-by-frame equality test has to start both sides from the same float.
      * ref_quirks = 0 (the intended 2x2 Schur step) contracts the same perturbation below 1e-7 on every problem: the sensitivity
        is a property of the aliased system, not of the data."""
    import dataclasses
    import tests.pipeline_ref as PR
    from vdo_slam_amd import synth_seq as SQ
    n_frames = 9
    Ts = SQ.camera_poses(n_frames)
    objs = SQ.default_objects(5, box_depth=0.9)
    probs = []

    class Recorder(PR.OraclePipeline):
        def _lm(self, kx, ky, fx, fy, d, T0, info_prior, max_it):
            import tests.test_oracle_flow2 as me
            orig = me.run_oracle

            def rec(o, prob):
                if prob.max_iterations == 200:
                    probs.append(prob)
                return orig(o, prob)
            me.run_oracle = rec
            try:
                return super()._lm(kx, ky, fx, fy, d, T0, info_prior, max_it)
            finally:
                me.run_oracle = orig

    ref = Recorder(oracle, build_lm=True)
    ref.ransac_flags = 2              # (the object problems this demonstration was recorded on: the inlier sets of the RANSAC around Grunert's P3P; with AP3P -
                                      #  a few borderline inliers differ - the worst problem of these nine frames moves 3e-4 instead of 2e-2: the property is the LM's)
    for k in range(n_frames):
        ref.step(SQ.render_frame(k, Ts, objs, flow_sigma=0.3, invalid_depth=0.02, zero_flow=0.01))
    assert len(probs) >= 25

    def moved(prob, quirks):
        out = []
        for ulp in (0, 1):
            T0 = prob.T0.astype(np.float32)
            if ulp:
                T0[:3, :] = np.nextafter(T0[:3, :], np.float32(np.inf))
            out.append(run_oracle(oracle, dataclasses.replace(prob, T0=T0.astype(np.float64), ref_quirks=quirks)))
        (Ta, _, _, _, sa), (Tb, _, _, _, sb) = out
        return float(np.abs(Ta - Tb).max()), sa.iterations, sb.iterations

    q = [moved(p, 1) for p in probs]
    i = [moved(p, 0) for p in probs]
    dq = np.array([m[0] for m in q]); di = np.array([m[0] for m in i])
    worst = int(np.argmax(dq))
    print(f"one-ulp seed perturbation over {len(probs)} object problems: F3 LM median {np.median(dq):.2e} max {dq.max():.2e} "
          f"(n = {probs[worst].n}, {q[worst][1]} vs {q[worst][2]} iterations); intended LM median {np.median(di):.2e} max {di.max():.2e}")
    assert dq.max() > 1e-3 and q[worst][1] >= 100              # some problem leaves the 1e-4 band: the long-running, never-settling kind
    assert q[worst][1] != q[worst][2]                          # ... along another Levenberg trajectory
    assert np.median(dq) < 1e-8                                # the typical object problem is well behaved
    assert di.max() < 1e-7 and all(a == b for _, a, b in i)    # the intended LM contracts the same perturbation everywhere
