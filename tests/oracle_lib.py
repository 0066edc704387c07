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
    _lib = L
    return L
