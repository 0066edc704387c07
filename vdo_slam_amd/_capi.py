"""ctypes binding of the C-ABI in ``include/vdo_slam_hip.h`` (libvdo_hip.so).

The library is the product: hand-written HIP kernels for gfx950 behind ``extern "C"`` entry
points.  There is NO CPU fallback — importing :func:`lib` raises if the shared object is
missing, and every call fails loudly when no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VDO_HIP_LIB") or os.path.join(_HERE, "libvdo_hip.so")     # (override: the phase-profiler build of tools/build_profiled_flow2.sh)

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_float_p = C.POINTER(C.c_float)
c_uint8_p = C.POINTER(C.c_uint8)

VDO_LM_MAX_TRACE = 512


class BAGraphC(C.Structure):
    _fields_ = [
        ("n_pose", C.c_int32), ("n_point", C.c_int32), ("n_eb", C.c_int32),
        ("n_et", C.c_int32), ("n_ep", C.c_int32), ("n_prior", C.c_int32),
        ("pose", c_double_p), ("point", c_double_p),
        ("eb_pose", c_int32_p), ("eb_point", c_int32_p), ("eb_z", c_double_p), ("eb_w", c_double_p),
        ("et_p1", c_int32_p), ("et_p2", c_int32_p), ("et_pose", c_int32_p), ("et_z", c_double_p), ("et_w", c_double_p),
        ("ep_i", c_int32_p), ("ep_j", c_int32_p), ("ep_z", c_double_p), ("ep_info", c_double_p),
        ("pr_pose", c_int32_p), ("pr_z", c_double_p), ("pr_info", c_double_p),
        ("huber_eb", C.c_double), ("huber_et", C.c_double), ("huber_ep", C.c_double),
    ]


class BASystemC(C.Structure):
    _fields_ = [
        ("Hpp", c_double_p), ("bp", c_double_p), ("Hll", c_double_p), ("bl", c_double_p),
        ("Hpl_eb", c_double_p), ("Hll_et", c_double_p), ("Hlp1_et", c_double_p), ("Hlp2_et", c_double_p),
        ("Hpp_ep", c_double_p), ("chi2", C.c_double), ("robust_chi2", C.c_double),
    ]


class LMOptionsC(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("gain_threshold", C.c_double), ("verbose", C.c_int32),
        ("solver", C.c_int32), ("pcg_tolerance", C.c_double), ("pcg_max_iterations", C.c_int32),
    ]


class LMStatsC(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("total_trials", C.c_int32), ("stop_reason", C.c_int32),
        ("initial_chi2", C.c_double), ("final_chi2", C.c_double), ("final_lambda", C.c_double),
        ("chi2_trace", C.c_double * VDO_LM_MAX_TRACE), ("trials_trace", C.c_int32 * VDO_LM_MAX_TRACE),
        ("ms_total", C.c_double), ("ms_linearize", C.c_double), ("ms_solve", C.c_double),
    ]


class Flow2ProblemC(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("obs", c_double_p), ("flow", c_double_p), ("depth", c_double_p),
        ("K", C.c_double * 4), ("Twl", C.c_double * 16), ("T0", C.c_double * 16),
        ("info_flow", C.c_double), ("info_prior", C.c_double), ("huber_delta", C.c_double),
        ("chi2_gate", C.c_double), ("max_iterations", C.c_int32), ("ref_quirks", C.c_int32),
    ]


class Flow2ResultC(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16), ("n_inliers", C.c_int32), ("iterations", C.c_int32), ("trials", C.c_int32),
        ("stop_reason", C.c_int32), ("initial_chi2", C.c_double), ("final_chi2", C.c_double), ("final_lambda", C.c_double),
    ]


def flow2_to_c(p):
    """Build the C struct for a :class:`vdo_slam_amd.synth.Flow2Problem`; returns (struct, keepalive)."""
    obs = np.ascontiguousarray(p.obs, dtype=np.float64); flow = np.ascontiguousarray(p.flow, dtype=np.float64)
    depth = np.ascontiguousarray(p.depth, dtype=np.float64)
    s = Flow2ProblemC()
    s.n = p.n; s.obs = _dp(obs); s.flow = _dp(flow); s.depth = _dp(depth)
    s.K = (C.c_double * 4)(*p.K)
    s.Twl = (C.c_double * 16)(*np.asarray(p.Twl, dtype=np.float64).ravel())
    s.T0 = (C.c_double * 16)(*np.asarray(p.T0, dtype=np.float64).ravel())
    s.info_flow = p.info_flow; s.info_prior = p.info_prior; s.huber_delta = p.huber_delta
    s.chi2_gate = p.chi2_gate; s.max_iterations = p.max_iterations; s.ref_quirks = p.ref_quirks
    return s, [obs, flow, depth]


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_int32_p) if a is not None else None


def graph_to_c(g) -> tuple[BAGraphC, list]:
    """Build the C struct for a :class:`vdo_slam_amd.synth.BAGraph`; returns (struct, keepalive)."""
    keep = []

    def d(a):
        a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); return _dp(a)

    def i(a):
        a = np.ascontiguousarray(a, dtype=np.int32); keep.append(a); return _ip(a)

    s = BAGraphC(
        g.n_pose, g.n_point, g.n_eb, g.n_et, g.n_ep, g.n_prior,
        d(g.pose), d(g.point),
        i(g.eb_pose), i(g.eb_point), d(g.eb_z), d(g.eb_w),
        i(g.et_p1), i(g.et_p2), i(g.et_pose), d(g.et_z), d(g.et_w),
        i(g.ep_i), i(g.ep_j), d(g.ep_z), d(g.ep_info),
        i(g.pr_pose), d(g.pr_z), d(g.pr_info),
        float(g.huber_eb), float(g.huber_et), float(g.huber_ep))
    return s, keep


class BASystem:
    """numpy-backed ``vdo_ba_system`` (one linearisation in block form)."""

    def __init__(self, g):
        self.Hpp = np.zeros((g.n_pose, 36)); self.bp = np.zeros((g.n_pose, 6))
        self.Hll = np.zeros((g.n_point, 9)); self.bl = np.zeros((g.n_point, 3))
        self.Hpl_eb = np.zeros((18, g.n_eb)); self.Hll_et = np.zeros((9, g.n_et))
        self.Hlp1_et = np.zeros((18, g.n_et)); self.Hlp2_et = np.zeros((18, g.n_et))
        self.Hpp_ep = np.zeros((g.n_ep, 36))
        self.c = BASystemC(_dp(self.Hpp), _dp(self.bp), _dp(self.Hll), _dp(self.bl), _dp(self.Hpl_eb),
                           _dp(self.Hll_et), _dp(self.Hlp1_et), _dp(self.Hlp2_et), _dp(self.Hpp_ep), 0.0, 0.0)

    @property
    def chi2(self): return self.c.chi2
    @property
    def robust_chi2(self): return self.c.robust_chi2


_lib = None


def lib() -> C.CDLL:
    """Load libvdo_hip.so (built in-tree by ``__graft_entry__.build()``).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def load_host_lib() -> C.CDLL:
    """libvdo_host.so (C++ classes with the reference's signatures, links libvdo_hip.so)."""
    path = os.path.join(_HERE, "libvdo_host.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run __graft_entry__.build()")
    lib()                                          # same HIP runtime, libvdo_hip resolved first
    return C.CDLL(path)


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so; if libvdo_hip pulled in
    /opt/rocm's copy first, a later ``import torch`` would load a SECOND runtime that finds no device ("No HIP
    GPUs are available").  Loading torch's copy first (without importing torch) makes the dynamic linker resolve
    libvdo_hip's DT_NEEDED libamdhip64.so.N to it by SONAME, whichever of the two the process imports first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                                     # torch already brought its runtime in
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _declare(L):
    vp = C.c_void_p
    L.vdo_version.restype = C.c_int
    L.vdo_last_error.restype = C.c_char_p
    L.vdo_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.vdo_ctx_destroy.argtypes = [vp]
    L.vdo_ctx_synchronize.argtypes = [vp]
    L.vdo_ba_create.argtypes = [vp, C.POINTER(BAGraphC), C.POINTER(vp)]
    L.vdo_ba_destroy.argtypes = [vp]
    L.vdo_ba_linearize.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.vdo_ba_download_system.argtypes = [vp, C.POINTER(BASystemC)]
    L.vdo_ba_optimize.argtypes = [vp, C.POINTER(LMOptionsC), C.POINTER(LMStatsC)]
    L.vdo_ba_get_estimates.argtypes = [vp, c_double_p, c_double_p]
    L.vdo_ba_set_estimates.argtypes = [vp, c_double_p, c_double_p]
    pp_d = C.POINTER(c_double_p)
    pp_u8 = C.POINTER(c_uint8_p)
    L.vdo_flow2_batch_create.argtypes = [vp, C.c_int, C.POINTER(Flow2ProblemC), C.POINTER(vp)]
    L.vdo_flow2_batch_run.argtypes = [vp]
    L.vdo_flow2_batch_fetch.argtypes = [vp, C.POINTER(Flow2ResultC), pp_d, pp_u8]
    L.vdo_flow2_batch_destroy.argtypes = [vp]
    L.vdo_flow2_optimize.argtypes = [vp, C.POINTER(Flow2ProblemC), C.POINTER(Flow2ResultC), c_double_p, c_uint8_p]
    for f in ("vdo_ctx_create", "vdo_ctx_destroy", "vdo_ctx_synchronize", "vdo_ba_create", "vdo_ba_destroy",
              "vdo_ba_linearize", "vdo_ba_download_system", "vdo_ba_optimize", "vdo_ba_get_estimates",
              "vdo_ba_set_estimates", "vdo_flow2_batch_create", "vdo_flow2_batch_run", "vdo_flow2_batch_fetch",
              "vdo_flow2_batch_destroy", "vdo_flow2_optimize"):
        getattr(L, f).restype = C.c_int


class VdoError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        msg = lib().vdo_last_error()
        raise VdoError(f"libvdo_hip error {rc}: {msg.decode() if msg else ''}")
