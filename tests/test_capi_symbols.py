"""CPU-side checks of the product library: it loads without a GPU, exports every symbol
include/vdo_slam_hip.h declares, and fails loudly (no CPU fallback) when no device exists."""
import ctypes as C
import os
import re
import subprocess

import pytest

from vdo_slam_amd import _capi as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(K.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "vdo_slam_amd", "csrc"), "-j8"], check=True)
    return K.lib()


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "vdo_slam_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(vdo_[a-z0-9_]+)\s*\(", hdr)) - {"vdo_allreduce_fn"})


def test_every_declared_symbol_is_exported(lib):
    syms = _declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vdo_slam_hip.h but not exported"


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    lib.vdo_last_error.restype = C.c_char_p
    rc = lib.vdo_ctx_create(0, None, C.byref(h))
    assert rc == -2          # VDO_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.vdo_last_error()
