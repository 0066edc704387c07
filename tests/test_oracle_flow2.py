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
