"""The HIP path against the COMMITTED golden vectors of tests/golden/ - outputs of the reference's own Optimizer.cc + g2o (compiled verbatim here into
oracle/_ref/libref_full.so; tools/pin_reference/make_outputs_ref_full.py wrote them, tests/golden/PINNED_BY.txt says from what).  Needs neither /root/reference
nor any oracle: product against reference, on the GPU box.  The north star's bar: inlier sets / counts equal, poses and motions within 1e-4 relative."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "pin_reference"))
import make_inputs as MI  # noqa: E402

from vdo_slam_amd import _capi as K, synth  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
f32 = np.float32


def _need(*names):
    paths = [os.path.join(GOLD, n) for n in names]
    if not all(os.path.exists(p) for p in paths):
        pytest.skip("parity unpinned: golden vector(s) %s absent - tools/pin_reference/make_outputs_ref_full.py writes them" % ", ".join(names))
    return paths


@pytest.mark.parametrize("case", range(len(MI.FLOW2_CASES)))
def test_per_frame_lm_against_the_reference_outputs(case):
    """Optimizer::PoseOptimizationFlow2Cam (case 0) / PoseOptimizationFlow2 (case 1) of the reference on tests/golden/inputs/flow2_case*.bin: the kernel's inlier
    classification EQUALS the reference's, the pose agrees as a CV_32F matrix to 1e-6 (bar 1e-4), the refined key points of the inliers to 2e-4 px."""
    from tests.pipeline_ref import inv_rigid_f32
    from vdo_slam_amd.ba import Context
    from vdo_slam_amd.flow2 import Flow2Batch
    (p_out,) = _need(f"flow2_case{case}.out")
    q = MI.read_flow2(os.path.join(GOLD, "inputs", f"flow2_case{case}.bin"))
    good, pose, inl, keys = MI.read_flow2_out(p_out, q["n"])
    prob = synth.Flow2Problem(obs=q["key"].astype(np.float64), flow=q["flow"].astype(np.float64), depth=q["depth"].astype(np.float64), K=tuple(float(v) for v in q["K4"]),
                              Twl=inv_rigid_f32(q["Tcw_last"]).astype(np.float64), T0=q["T0"].astype(np.float64), info_prior=0.5 if q["is_object"] else 0.3,
                              max_iterations=200 if q["is_object"] else 100)
    prob.huber_delta = float(np.sqrt(f32(0.04))); prob.chi2_gate = float(f32(0.04)); prob.info_flow = 0.1; prob.ref_quirks = 1
    ctx = Context(0)
    b = Flow2Batch(ctx, [prob])
    b.run()
    (res,) = b.fetch()
    b.close(); ctx.close()
    assert res["n_inliers"] == good and np.array_equal(np.asarray(res["inliers"]).astype(np.int32), inl), "per-frame LM: inlier classification differs from the reference's"
    np.testing.assert_allclose(res["T"].astype(f32), pose, rtol=0, atol=1e-6 * max(1.0, float(np.abs(pose[:3, 3]).max())))
    sel = inl.astype(bool)
    np.testing.assert_allclose((q["key"].astype(np.float64) + res["flow"])[sel].astype(f32), keys[sel], rtol=0, atol=2e-4)


def test_full_batch_optimization_against_the_reference_outputs():
    """Optimizer::FullBatchOptimization of the product's host class (graph built in C++, Levenberg on the GPU) on the Map of tests/golden/inputs/batch_map_case0.bin
    against what the reference's own FullBatchOptimization wrote back into that Map: refined camera poses, object motions, static and dynamic points."""
    from tests import map_builder_ref as SM
    p_out, p_sta, p_dyn = _need(os.path.join("batch_case0", "batch_refined.bin"), os.path.join("batch_case0", "static_points_refined.f32"), os.path.join("batch_case0", "dynamic_points_refined.f32"))
    cams, mots = MI.read_batch_out(p_out)
    L = K.load_host_lib()
    L.host_batch_optimization.argtypes = [C.POINTER(SM.HostMapFlat), C.c_int, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p, C.POINTER(K.LMStatsC)]
    m = MI.golden_map()
    s, keep = SM.flatten_map(m)
    F = m["n_frames"]
    n_sta = sum(len(fe["sta_uv"]) for fe in m["feats"]); n_dyn = sum(len(fe["dyn_uv"]) for fe in m["feats"]); n_rm = sum(len(r) for r in m["rigid_motion"])
    cam = np.zeros((F, 4, 4), f32); rm = np.zeros((n_rm, 4, 4), f32); sta = np.zeros((n_sta, 3), f32); dyn = np.zeros((max(n_dyn, 1), 3), f32)
    st = K.LMStatsC()
    fp = lambda a: a.ctypes.data_as(K.c_float_p)      # noqa: E731
    assert L.host_batch_optimization(C.byref(s), 0, fp(cam), fp(rm), fp(sta), fp(dyn), C.byref(st)) == 0
    tol = 1e-4                                          # (measured: a few 1e-6)
    for i in range(F):
        np.testing.assert_allclose(cam[i], cams[i], rtol=0, atol=tol * max(1.0, float(np.abs(cams[i][:3, 3]).max())), err_msg=f"refined camera pose {i}")
    off = 0
    for i in range(F - 1):
        for j in range(len(mots[i])):
            np.testing.assert_allclose(rm[off + j], mots[i][j], rtol=0, atol=tol * max(1.0, float(np.abs(mots[i][j][:3, 3]).max())), err_msg=f"refined object motion {i}/{j}")
        off += len(mots[i])
    gs = np.fromfile(p_sta, f32).reshape(-1, 3); gd = np.fromfile(p_dyn, f32).reshape(-1, 3)
    assert gs.shape == sta.shape and gd.shape[0] == n_dyn
    np.testing.assert_allclose(sta, gs, rtol=tol, atol=tol)
    np.testing.assert_allclose(dyn[:n_dyn], gd, rtol=tol, atol=tol)
    print("batch golden: worst camera %.1e, motion %.1e, static point %.1e" % (np.abs(cam - cams).max(), max(np.abs(rm[:len(np.concatenate(mots))] - np.concatenate(mots)).max(), 0.0), np.abs(sta - gs).max()))
