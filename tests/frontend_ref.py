"""Python-side helpers that call the front-end ORACLE (tests only)."""
import ctypes as C

import numpy as np

from vdo_slam_amd import _capi as K
from vdo_slam_amd.frontend import OrbParamsC


def _fp(a): return a.ctypes.data_as(K.c_float_p)
def _ip(a): return a.ctypes.data_as(K.c_int32_p)
def _u8(a): return a.ctypes.data_as(K.c_uint8_p)


PARAMS = OrbParamsC(2500, 1.2, 8, 20, 7)


def level_sizes(o, w, h, prm=PARAMS):
    ws = np.zeros(prm.n_levels, np.int32); hs = np.zeros(prm.n_levels, np.int32); nf = np.zeros(prm.n_levels, np.int32)
    o.vdo_oracle_orb_level_sizes(C.byref(prm), w, h, _ip(ws), _ip(hs), _ip(nf))
    return ws, hs, nf


def pyramid(o, gray, prm=PARAMS):
    h, w = gray.shape
    ws, hs, _ = level_sizes(o, w, h, prm)
    total = int(((ws + 38) * (hs + 38)).sum())
    buf = np.zeros(total, np.uint8)
    o.vdo_oracle_orb_pyramid(_u8(gray), w, h, C.byref(prm), _u8(buf))
    out, off = [], 0
    for a, b in zip(ws, hs):
        n = (a + 38) * (b + 38)
        out.append(buf[off:off + n].reshape(b + 38, a + 38)); off += n
    return out


def fast_level(o, gray, level, prm=PARAMS, cap=200000):
    h, w = gray.shape
    x, y, r = (np.zeros(cap, np.float32) for _ in range(3))
    n = o.vdo_oracle_orb_fast_level(_u8(gray), w, h, C.byref(prm), level, _fp(x), _fp(y), _fp(r), cap)
    return x[:n], y[:n], r[:n]


def extract(o, gray, prm=PARAMS, cap=8192):
    h, w = gray.shape
    a = [np.zeros(cap, np.float32) for _ in range(5)]
    octv = np.zeros(cap, np.int32)
    n = o.vdo_oracle_orb_extract(_u8(gray), w, h, C.byref(prm), _fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(a[3]), _ip(octv), _fp(a[4]), cap)
    assert n >= 0
    return dict(x=a[0][:n], y=a[1][:n], response=a[2][:n], angle=a[3][:n], octave=octv[:n], size=a[4][:n])


def extract_desc(o, gray, prm=PARAMS, cap=8192):
    """extract() + the 32-byte rotated-BRIEF rows (oracle of K8)."""
    h, w = gray.shape
    a = [np.zeros(cap, np.float32) for _ in range(5)]
    octv = np.zeros(cap, np.int32)
    desc = np.zeros((cap, 32), np.uint8)
    o.vdo_oracle_orb_extract_desc.argtypes = [K.c_uint8_p, C.c_int, C.c_int, C.POINTER(OrbParamsC)] + [K.c_float_p] * 4 + [K.c_int32_p, K.c_float_p, C.c_int, K.c_uint8_p]
    n = o.vdo_oracle_orb_extract_desc(_u8(gray), w, h, C.byref(prm), _fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(a[3]), _ip(octv), _fp(a[4]), cap, _u8(desc))
    assert n >= 0
    return dict(x=a[0][:n], y=a[1][:n], response=a[2][:n], angle=a[3][:n], octave=octv[:n], size=a[4][:n], desc=desc[:n])


def blur7(o, img):
    img = np.ascontiguousarray(img)
    out = np.zeros_like(img)
    o.vdo_oracle_gaussian_blur7(_u8(img), img.shape[1], img.shape[0], _u8(out))
    return out


def static_filter(o, kx, ky, koct, mask, depth, flow, th):
    n = kx.size
    h, w = mask.shape
    idx = np.zeros(n, np.int32); f = [np.zeros(n, np.float32) for _ in range(5)]
    m = o.vdo_oracle_frame_static_filter(n, _fp(kx), _fp(ky), _ip(koct), _ip(mask), _fp(depth), _fp(flow), w, h, th, _ip(idx), *[_fp(a) for a in f])
    return dict(keep_idx=idx[:m], corr_x=f[0][:m], corr_y=f[1][:m], flow_x=f[2][:m], flow_y=f[3][:m], depth=f[4][:m])


def object_sample(o, mask, depth, flow, th, step=4):
    h, w = mask.shape
    cap = ((w + step - 1) // step) * ((h + step - 1) // step)
    f = [np.zeros(cap, np.float32) for _ in range(7)]; lab = np.zeros(cap, np.int32)
    m = o.vdo_oracle_frame_object_sample(_ip(mask), _fp(depth), _fp(flow), w, h, th, step, cap, *[_fp(a) for a in f], _ip(lab))
    names = ("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y", "depth")
    out = {k: a[:m] for k, a in zip(names, f)}
    out["label"] = lab[:m]
    return out


# ---- oracle/_ref: the reference's own ORBextractor.cc (tests/oracle_lib.load_ref_orb) -----------------------------------------
def ref_extract(ref, gray, prm=PARAMS, cap=8192, desc=False):
    """ORBextractor::operator() of the reference-compiled library; desc=True runs its stages one by one + computeDescriptors."""
    h, w = gray.shape
    a = [np.zeros(cap, np.float32) for _ in range(5)]
    octv = np.zeros(cap, np.int32)
    d = np.zeros((cap, 32), np.uint8)
    args = [_u8(gray), w, h, C.byref(prm), _fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(a[3]), _ip(octv), _fp(a[4]), cap]
    n = ref.vdo_ref_orb_extract_desc(*args, _u8(d)) if desc else ref.vdo_ref_orb_extract(*args)
    assert n >= 0
    out = dict(x=a[0][:n], y=a[1][:n], response=a[2][:n], angle=a[3][:n], octave=octv[:n], size=a[4][:n])
    if desc:
        out["desc"] = d[:n]
    return out


def ref_level_facts(ref, w, h, prm=PARAMS):
    ws, hs, nf = (np.zeros(prm.n_levels, np.int32) for _ in range(3))
    um = np.zeros(16, np.int32)
    ref.vdo_ref_orb_level_facts(w, h, C.byref(prm), _ip(ws), _ip(hs), _ip(nf), _ip(um))
    return ws, hs, nf, um


def ref_pyramid(ref, gray, prm=PARAMS):
    h, w = gray.shape
    ws, hs, _, _ = ref_level_facts(ref, w, h, prm)
    buf = np.zeros(int(((ws + 38) * (hs + 38)).sum()), np.uint8)
    ref.vdo_ref_orb_pyramid(_u8(gray), w, h, C.byref(prm), _u8(buf))
    out, off = [], 0
    for a, b in zip(ws, hs):
        n = (a + 38) * (b + 38)
        out.append(buf[off:off + n].reshape(b + 38, a + 38)); off += n
    return out
