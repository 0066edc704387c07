"""Loader for the CPU oracle (oracle/libvdo_oracle.so) — TEST INFRASTRUCTURE ONLY.

Builds it with `make -C oracle` when missing/outdated.  Shares the ctypes structure
classes with the product binding because the C layouts are identical by construction.
"""
import ctypes as C
import os
import subprocess

from vdo_slam_amd import _capi as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libvdo_oracle.so")

_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB)
    dp, ip = K.c_double_p, K.c_int32_p
    L.vdo_oracle_ba_linearize.argtypes = [C.POINTER(K.BAGraphC), C.POINTER(K.BASystemC)]
    L.vdo_oracle_ba_optimize.argtypes = [C.POINTER(K.BAGraphC), C.POINTER(K.LMOptionsC), dp, dp, C.POINTER(K.LMStatsC)]
    L.vdo_oracle_ba_normal_equations.argtypes = [C.POINTER(K.BAGraphC), ip, ip, dp, C.c_int64, dp]
    L.vdo_oracle_ba_normal_equations.restype = C.c_int64
    L.vdo_oracle_ba_solve.argtypes = [C.POINTER(K.BAGraphC), C.c_double, dp]
    L.vdo_oracle_se3_exp.argtypes = [dp, dp]
    L.vdo_oracle_iso_oplus.argtypes = [dp, dp, dp]
    L.vdo_oracle_iso_to_mqt.argtypes = [dp, dp]
    L.vdo_oracle_edge_se3_jac.argtypes = [dp] * 6
    L.vdo_oracle_edge_prior_jac.argtypes = [dp] * 4
    L.vdo_oracle_edge_eb_jac.argtypes = [dp] * 6
    L.vdo_oracle_edge_et_jac.argtypes = [dp] * 8
    L.vdo_oracle_flow2_optimize.argtypes = [C.POINTER(K.Flow2ProblemC), dp, dp, K.c_uint8_p, C.POINTER(K.LMStatsC)]
    from vdo_slam_amd.frontend import OrbParamsC
    fp, i32p, u8p = K.c_float_p, K.c_int32_p, K.c_uint8_p
    L.vdo_oracle_depth_preprocess.argtypes = [fp, C.c_int64, C.c_float, C.c_float]
    L.vdo_oracle_rgb2gray.argtypes = [u8p, C.c_int64, C.c_int, C.c_int, u8p]
    L.vdo_oracle_orb_level_sizes.argtypes = [C.POINTER(OrbParamsC), C.c_int, C.c_int, i32p, i32p, i32p]
    L.vdo_oracle_orb_pyramid.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(OrbParamsC), u8p]
    L.vdo_oracle_orb_fast_level.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(OrbParamsC), C.c_int, fp, fp, fp, C.c_int]
    L.vdo_oracle_orb_extract.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(OrbParamsC), fp, fp, fp, fp, i32p, fp, C.c_int]
    L.vdo_oracle_gaussian_blur7.argtypes = [u8p, C.c_int, C.c_int, u8p]
    L.vdo_oracle_frame_static_filter.argtypes = [C.c_int, fp, fp, i32p, i32p, fp, fp, C.c_int, C.c_int, C.c_float, i32p, fp, fp, fp, fp, fp]
    L.vdo_oracle_frame_object_sample.argtypes = [i32p, fp, fp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, fp, i32p]
    from vdo_slam_amd.pose_only import PoseProblemC
    L.vdo_oracle_pose_optimize.argtypes = [C.POINTER(PoseProblemC), dp, K.c_uint8_p, C.POINTER(K.LMStatsC)]
    L.vdo_oracle_edge_unary_jac.argtypes = [C.POINTER(PoseProblemC), dp, dp, dp, dp, dp]
    L.vdo_oracle_propagate_static.argtypes = [C.c_int, fp, fp, fp, C.c_int, C.c_int, fp]
    L.vdo_oracle_propagate_object.argtypes = [C.c_int, fp, fp, fp, i32p, C.c_int, C.c_int, C.c_float, fp, i32p]
    L.vdo_oracle_scene_flow.argtypes = [C.c_int, fp, fp, fp, i32p, fp, fp, fp, fp, i32p, fp, fp, fp, i32p]
    L.vdo_oracle_get3d_world.argtypes = [C.c_int, fp, fp, fp, fp, fp, fp]
    L.vdo_oracle_renew_static.argtypes = [C.c_int, i32p, fp, fp, C.c_int, fp, fp, i32p, fp, fp, C.c_int, C.c_int, C.c_int,
                                          fp, fp, fp, fp, fp, fp, i32p, fp]
    L.vdo_oracle_mask_at.argtypes = [C.c_int, fp, fp, i32p, C.c_int, C.c_int, i32p]
    L.vdo_oracle_mask_warp.argtypes = [i32p, fp, C.c_int, C.c_int, C.c_int32, i32p]
    u8 = K.c_uint8_p
    L.vdo_oracle_dyn_obj_tracking.argtypes = [C.c_int, i32p, i32p, fp, fp, fp, fp, i32p, C.c_int, i32p, i32p, u8, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_float, C.c_float, C.c_float, C.c_int, i32p, i32p, i32p, i32p, i32p]
    L.vdo_oracle_renew_object.argtypes = [C.c_int, i32p, i32p, u8, i32p, i32p, fp, fp, i32p, C.c_int, fp, fp, fp, i32p, fp, fp, fp, fp,
                                          i32p, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, i32p, fp, fp, fp, fp, i32p, i32p]
    L.vdo_oracle_update_mask.argtypes = [C.c_int, i32p, fp, fp, i32p, fp, C.c_int, C.c_int, i32p]
    L.vdo_oracle_build_tracks.argtypes = [C.c_int, i32p, i32p, i32p, C.c_int, C.c_int, i32p, i32p, i32p, i32p]
    _lib = L
    return L


_prod_epnp = None


def load_product_epnp():
    """The PRODUCT's host-side EPnP refit (vdo_slam_amd/csrc/epnp_refit.hpp - plain C++, no GPU) compiled behind a C entry point
    (tests/helpers/product_host_epnp.cpp).  Two uses: tests/test_epnp_independent.py compares it with the oracle's independent EPnP;
    the sequence tests hand it to OraclePipeline(seed_refit="product") so that both sides start every LM from the same float seed
    (see tests/pipeline_ref.py for why that is needed)."""
    global _prod_epnp
    if _prod_epnp is not None:
        return _prod_epnp
    here = os.path.join(ROOT, "tests", "helpers")
    src, out = os.path.join(here, "product_host_epnp.cpp"), os.path.join(here, "libproduct_host_epnp.so")
    hdr = os.path.join(ROOT, "vdo_slam_amd", "csrc", "epnp_refit.hpp")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", out, src], check=True)
    L = C.CDLL(out)
    L.product_host_epnp.restype = C.c_double
    L.product_host_epnp.argtypes = [C.c_int] + [K.c_double_p] * 4
    _prod_epnp = L
    return L


REF_DIR = os.path.join(ORACLE_DIR, "ref")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libref_orb.so")
REFERENCE_ROOT = os.environ.get("VDO_REFERENCE_ROOT", "/root/reference")
_ref_orb = None


def load_ref_orb():
    """oracle/_ref/libref_orb.so = the REFERENCE's own src/ORBextractor.cc compiled verbatim against the mini-cv shim (oracle/ref/).
    (Re)built when the reference checkout is present (the build container); on the GPU box only the prebuilt library travels.
    Returns None when neither exists - callers skip with "parity unpinned"."""
    global _ref_orb
    if _ref_orb is not None:
        return _ref_orb
    load()   # libvdo_oracle.so first: the shim's primitives live there
    if os.path.exists(os.path.join(REFERENCE_ROOT, "src", "ORBextractor.cc")):
        subprocess.run(["make", "-C", REF_DIR, "-s", f"REF={REFERENCE_ROOT}"], check=True)
    if not os.path.exists(REF_LIB):
        return None
    from vdo_slam_amd.frontend import OrbParamsC
    L = C.CDLL(REF_LIB)
    fp, i32p, u8p = K.c_float_p, K.c_int32_p, K.c_uint8_p
    head = [u8p, C.c_int, C.c_int, C.POINTER(OrbParamsC)]
    L.vdo_ref_orb_extract.argtypes = head + [fp, fp, fp, fp, i32p, fp, C.c_int]
    L.vdo_ref_orb_extract_desc.argtypes = head + [fp, fp, fp, fp, i32p, fp, C.c_int, u8p]
    L.vdo_ref_orb_level_facts.argtypes = [C.c_int, C.c_int, C.POINTER(OrbParamsC), i32p, i32p, i32p, i32p]
    L.vdo_ref_orb_pyramid.argtypes = head + [u8p]
    L.vdo_ref_ic_angle.argtypes = [u8p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.vdo_ref_ic_angle.restype = C.c_float
    L.vdo_ref_orb_descriptor.argtypes = [u8p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, u8p]
    _ref_orb = L
    return L


REF_TRACK_LIB = os.path.join(ORACLE_DIR, "_ref", "libref_track.so")
REF_FULL_LIB = os.path.join(ORACLE_DIR, "_ref", "libref_full.so")
_ref_track = None
_ref_full = None


def _bind_ref_system(L):
    vp = C.c_void_p
    L.vdo_ref_system_create.restype = vp
    L.vdo_ref_system_create.argtypes = [C.c_char_p]
    L.vdo_ref_system_destroy.argtypes = [vp]
    L.vdo_ref_system_track.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, vp]
    L.vdo_ref_system_counts.argtypes = [vp, vp]
    L.vdo_ref_system_frame_state.argtypes = [vp, C.c_int, vp, C.c_int]
    L.vdo_ref_system_tracks.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.vdo_ref_system_timing.argtypes = [vp, vp]
    L.vdo_ref_set_time.argtypes = [C.c_long]
    if hasattr(L, "vdo_ref_system_map_export"):
        L.vdo_ref_system_map_export.restype = C.c_long
        L.vdo_ref_system_map_export.argtypes = [vp, C.c_int, vp, C.c_long]


def load_ref_full():
    """oracle/_ref/libref_full.so = the WHOLE reference library: the front-end sources of libref_track.so plus src/Optimizer.cc, src/Converter.cc and the
    vendored g2o (dependencies/g2o/g2o: every source of its CMake target), all compiled verbatim - against the mini-cv shim, shim/Eigen (a small
    dense-algebra library with Eigen's interface) and shim/cs.h + minics.cpp (CSparse's interface).  None when it cannot be had."""
    global _ref_full
    if _ref_full is not None:
        return _ref_full
    load()
    if os.path.exists(os.path.join(REFERENCE_ROOT, "src", "Optimizer.cc")):
        subprocess.run(["make", "-C", REF_DIR, "-s", "-j8", f"REF={REFERENCE_ROOT}"], check=True)
    if not os.path.exists(REF_FULL_LIB):
        return None
    L = C.CDLL(REF_FULL_LIB)
    _bind_ref_system(L)
    _ref_full = L
    return L


def load_ref_track():
    """oracle/_ref/libref_track.so = the REFERENCE's own src/System.cc + Tracking.cc + Frame.cc + Map.cc + ORBextractor.cc compiled verbatim against
    the mini-cv shim (oracle/ref/), OpenCV primitives and the g2o behind the Optimizer statics supplied by the oracle.  (Re)built when the reference
    checkout is present; on the GPU box only the prebuilt library travels.  None when neither exists."""
    global _ref_track
    if _ref_track is not None:
        return _ref_track
    load()
    if os.path.exists(os.path.join(REFERENCE_ROOT, "src", "Tracking.cc")):
        subprocess.run(["make", "-C", REF_DIR, "-s", f"REF={REFERENCE_ROOT}"], check=True)
    if not os.path.exists(REF_TRACK_LIB):
        return None
    L = C.CDLL(REF_TRACK_LIB)
    _bind_ref_system(L)
    _ref_track = L
    return L
