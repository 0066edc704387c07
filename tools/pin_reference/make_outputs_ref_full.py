"""Golden OUTPUTS of the optimiser cases from the reference's own sources compiled here (oracle/_ref/libref_full.so = src/Optimizer.cc, src/Converter.cc and the 48
sources of the reference's vendored g2o, verbatim from /root/reference, against the mini-Eigen / mini-CSparse of oracle/ref/shim - oracle/ref/Makefile):

    tests/golden/flow2_case*.out                 Optimizer::PoseOptimizationFlow2Cam / PoseOptimizationFlow2 on tests/golden/inputs/flow2_case*.bin
    tests/golden/batch_case0/batch_refined.bin   Optimizer::FullBatchOptimization on the Map of tests/golden/inputs/batch_map_case0.bin
    tests/golden/PINNED_BY.txt                   what produced them

in the layouts of tools/pin_reference/pin_dump.cc (tests/golden/README.md), so that tests/test_golden.py and the -m gpu test tests/test_golden_gpu.py compare the
oracle and the HIP path with the reference's results wherever the repository is checked out - /root/reference is only needed to run THIS script.  The cases that
need OpenCV itself (ORB front-end, FAST, blur, solvePnPRansac) stay absent: tools/pin_reference/run.sh on a host with OpenCV 3.4.0 writes those.

usage: python tools/pin_reference/make_outputs_ref_full.py      (needs /root/reference; builds oracle/_ref/libref_full.so if absent)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_inputs as MI  # noqa: E402
from tests import oracle_lib, map_builder_ref as SM  # noqa: E402
from tests.ref_track import Quiet  # noqa: E402
from vdo_slam_amd import _capi as K  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
f32 = np.float32


def _fp(a):
    return a.ctypes.data_as(K.c_float_p)


def main():
    ref = oracle_lib.load_ref_full()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_full.so is absent and /root/reference is not there to build it from")
    fp, ip = K.c_float_p, K.c_int32_p
    ref.ref_pose_optimization_flow2cam.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp, fp, ip, fp]
    ref.ref_pose_optimization_flow2.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp, fp, fp, ip, ip, fp]
    ref.ref_batch_optimization.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp]
    for case in range(len(MI.FLOW2_CASES)):
        q = MI.read_flow2(os.path.join(GOLD, "inputs", f"flow2_case{case}.bin"))
        n = q["n"]
        K4 = np.ascontiguousarray(q["K4"], f32); key = np.ascontiguousarray(q["key"], f32); flow = np.ascontiguousarray(q["flow"], f32); dep = np.ascontiguousarray(q["depth"], f32)
        Tlw = np.ascontiguousarray(q["Tcw_last"], f32); T0 = np.ascontiguousarray(q["T0"], f32)
        cur = np.zeros((n, 2), f32); Tout = np.zeros((4, 4), f32)
        with Quiet():
            if q["is_object"]:
                flag = np.zeros(n, np.int32); lab = np.zeros(n, np.int32)
                good = ref.ref_pose_optimization_flow2(n, _fp(K4), _fp(key), _fp(flow), _fp(dep), _fp(Tlw), _fp(np.eye(4, dtype=f32)), _fp(T0), _fp(Tout), flag.ctypes.data_as(ip), lab.ctypes.data_as(ip), _fp(cur))
                inl = (flag != 0).astype(np.int32)
            else:
                match = np.zeros(n, np.int32)
                good = ref.ref_pose_optimization_flow2cam(n, _fp(K4), _fp(key), _fp(flow), _fp(dep), _fp(Tlw), _fp(T0), _fp(Tout), match.ctypes.data_as(ip), _fp(cur))
                inl = (match >= 0).astype(np.int32)
        assert good == int(inl.sum()) and good > 0.5 * n, (case, good, int(inl.sum()))
        with open(os.path.join(GOLD, f"flow2_case{case}.out"), "wb") as f:
            np.int32(good).tofile(f); Tout.tofile(f); inl.tofile(f); cur.tofile(f)
        print(f"flow2_case{case}: n {n} object {q['is_object']} inliers {good}")
    m = MI.golden_map()
    s, keep = SM.flatten_map(m)
    F = m["n_frames"]
    n_sta = sum(len(fe["sta_uv"]) for fe in m["feats"]); n_dyn = sum(len(fe["dyn_uv"]) for fe in m["feats"]); n_rm = sum(len(r) for r in m["rigid_motion"])
    cam = np.zeros((F, 4, 4), f32); rm = np.zeros((n_rm, 4, 4), f32); sta = np.zeros((n_sta, 3), f32); dyn = np.zeros((max(n_dyn, 1), 3), f32)
    with Quiet():
        assert ref.ref_batch_optimization(C.byref(s), 0, _fp(cam), _fp(rm), _fp(sta), _fp(dyn)) == 0
    os.makedirs(os.path.join(GOLD, "batch_case0"), exist_ok=True)
    with open(os.path.join(GOLD, "batch_case0", "batch_refined.bin"), "wb") as f:
        np.int32(F).tofile(f); cam.tofile(f)
        off = 0
        for i in range(F - 1):
            nm = len(m["rigid_motion"][i])
            np.int32(nm).tofile(f); rm[off:off + nm].tofile(f); off += nm
    sta.tofile(os.path.join(GOLD, "batch_case0", "static_points_refined.f32")); dyn[:n_dyn].tofile(os.path.join(GOLD, "batch_case0", "dynamic_points_refined.f32"))
    print("batch_case0:", F, "frames,", n_rm, "motions,", n_sta, "static /", n_dyn, "dynamic point observations")
    with open(os.path.join(GOLD, "PINNED_BY.txt"), "w") as f:
        f.write("flow2_case*.out, batch_case0/*: written by tools/pin_reference/make_outputs_ref_full.py from oracle/_ref/libref_full.so - the reference's own src/Optimizer.cc,\n"
                "src/Converter.cc and the 48 sources of its vendored g2o (dependencies/g2o/g2o), compiled VERBATIM from the checkout of halajun/VDO_SLAM under /root/reference\n"
                "(oracle/ref/Makefile: g++ -O2 -DNDEBUG -ffp-contract=off) against oracle/ref/shim/Eigen (a mini-Eigen written for the surface g2o uses) and oracle/ref/minics.cpp\n"
                "(CSparse's interface restated).  First-party code and g2o are the reference's; the dense-algebra kernels under them (LDLT, LLT, quaternion, sparse Cholesky) are\n"
                "restatements of Eigen 3 / CSparse, NOT those libraries.  The cases that need OpenCV 3.4.0 (orb_*, fast_*, cvtcolor_*, fastatan2_*, pnp_case*.out) are absent:\n"
                "tools/pin_reference/run.sh on a host with the real libraries writes those (and may overwrite these).\n")


if __name__ == "__main__":
    main()
