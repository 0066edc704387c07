"""Host-side mirrors of ``ORBextractor`` and the data-parallel part of ``Frame::Frame``
(thin ctypes wrappers; all compute in libvdo_hip.so)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K


class OrbParamsC(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th", C.c_int32), ("min_th", C.c_int32)]


class KeypointsC(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n", C.c_int32), ("x", K.c_float_p), ("y", K.c_float_p),
                ("response", K.c_float_p), ("angle", K.c_float_p), ("size", K.c_float_p), ("octave", K.c_int32_p)]


def _fp(a): return a.ctypes.data_as(K.c_float_p)
def _ip(a): return a.ctypes.data_as(K.c_int32_p)
def _u8(a): return a.ctypes.data_as(K.c_uint8_p)


_declared = False


def _lib():
    global _declared
    L = K.lib()
    if not _declared:
        vp = C.c_void_p
        ip = C.POINTER(C.c_int)
        L.vdo_orb_create.argtypes = [vp, C.POINTER(OrbParamsC), C.c_int, C.c_int, C.POINTER(vp)]
        L.vdo_orb_destroy.argtypes = [vp]
        L.vdo_orb_extract.argtypes = [vp, K.c_uint8_p, C.c_int, C.c_int, C.POINTER(KeypointsC)]
        L.vdo_orb_extract_desc.argtypes = [vp, K.c_uint8_p, C.c_int, C.c_int, C.POINTER(KeypointsC), K.c_uint8_p]
        L.vdo_orb_descriptors.argtypes = [vp, K.c_uint8_p, C.c_int]
        L.vdo_orb_level_info.argtypes = [vp, C.c_int, ip, ip, ip, ip]
        L.vdo_orb_get_pyramid.argtypes = [vp, C.c_int, K.c_uint8_p]
        L.vdo_orb_get_blurred.argtypes = [vp, C.c_int, K.c_uint8_p]
        L.vdo_orb_get_candidates.argtypes = [vp, C.c_int, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p, C.c_int, ip]
        L.vdo_depth_preprocess.argtypes = [vp, K.c_float_p, C.c_int64, C.c_float, C.c_float, C.c_int]
        L.vdo_rgb2gray.argtypes = [vp, K.c_uint8_p, C.c_int64, C.c_int, C.c_int, K.c_uint8_p]
        L.vdo_frame_images_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.vdo_frame_images_upload.argtypes = [vp, K.c_float_p, K.c_float_p, K.c_int32_p]
        L.vdo_frame_images_destroy.argtypes = [vp]
        L.vdo_frame_static_filter.argtypes = [vp, C.c_int, K.c_float_p, K.c_float_p, C.c_float, K.c_int32_p] + [K.c_float_p] * 5 + [ip]
        L.vdo_frame_object_sample.argtypes = [vp, C.c_float, C.c_int, C.c_int] + [K.c_float_p] * 7 + [K.c_int32_p, ip]
        _declared = True
    return L


class ORBextractor:
    """``ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)`` (include/ORBextractor.h:39-40)."""

    def __init__(self, ctx, width, height, nfeatures=2500, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.ctx = ctx
        self.params = OrbParamsC(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.w, self.h, self.nlevels = width, height, nlevels
        self._h = C.c_void_p()
        K.check(_lib().vdo_orb_create(ctx._h, C.byref(self.params), width, height, C.byref(self._h)))

    def __call__(self, gray: np.ndarray, capacity=None, descriptors=False):
        """Keypoints; with ``descriptors=True`` also ``desc`` [n, 32] uint8 (rotated BRIEF, K8)."""
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        cap = capacity or (self.params.n_features + 256)
        a = {k: np.zeros(cap, np.float32) for k in ("x", "y", "response", "angle", "size")}
        octave = np.zeros(cap, np.int32)
        kp = KeypointsC(cap, 0, _fp(a["x"]), _fp(a["y"]), _fp(a["response"]), _fp(a["angle"]), _fp(a["size"]), _ip(octave))
        if descriptors:
            desc = np.zeros((cap, 32), np.uint8)
            K.check(_lib().vdo_orb_extract_desc(self._h, _u8(gray), gray.shape[1], 0, C.byref(kp), _u8(desc)))
        else:
            K.check(_lib().vdo_orb_extract(self._h, _u8(gray), gray.shape[1], 0, C.byref(kp)))
        n = kp.n
        out = {k: v[:n].copy() for k, v in a.items()}
        out["octave"] = octave[:n].copy()
        if descriptors:
            out["desc"] = desc[:n].copy()
        return out

    def descriptors(self, n):
        """Descriptors of the ``n`` keypoints the last extraction returned (separate call, vdo_orb_descriptors)."""
        desc = np.zeros((max(n, 1), 32), np.uint8)
        K.check(_lib().vdo_orb_descriptors(self._h, _u8(desc), max(n, 1)))
        return desc[:n]

    def extract_device(self, gray_ptr: int, stride: int, capacity=None):
        """Like ``__call__`` but the gray image is already resident in HBM (raw device pointer).
        The returned arrays are views of buffers owned by this extractor: valid until the next call."""
        cap = capacity or (self.params.n_features + 256)
        st = getattr(self, "_dev_out", None)
        if st is None or st[0] != cap:
            a = {k: np.zeros(cap, np.float32) for k in ("x", "y", "response", "angle", "size")}
            octave = np.zeros(cap, np.int32)
            kp = KeypointsC(cap, 0, _fp(a["x"]), _fp(a["y"]), _fp(a["response"]), _fp(a["angle"]), _fp(a["size"]), _ip(octave))
            st = self._dev_out = (cap, a, octave, kp, C.byref(kp), _lib().vdo_orb_extract)
        _, a, octave, kp, kp_ref, fn = st
        K.check(fn(self._h, C.cast(C.c_void_p(gray_ptr), K.c_uint8_p), stride, 1, kp_ref))
        n = kp.n
        out = {k: v[:n] for k, v in a.items()}
        out["octave"] = octave[:n]
        return out

    def last_timing(self):
        """(ms device stage incl. D2H of the candidates, ms host quadtree) of the last extraction."""
        ms = (C.c_double * 2)()
        L = _lib()
        L.vdo_orb_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        K.check(L.vdo_orb_last_timing(self._h, ms))
        return ms[0], ms[1]

    def pyramid_launches(self):
        """Kernel launches per pyramid: 2 / 1 (cascaded in LDS) or the number of levels."""
        L = _lib()
        L.vdo_orb_pyramid_launches.argtypes = [C.c_void_p]
        return int(L.vdo_orb_pyramid_launches(self._h))

    def level_info(self, level):
        w, h, nf, nc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        K.check(_lib().vdo_orb_level_info(self._h, level, C.byref(w), C.byref(h), C.byref(nf), C.byref(nc)))
        return w.value, h.value, nf.value, nc.value

    def pyramid(self, level):
        w, h, _, _ = self.level_info(level)
        out = np.zeros((h + 38, w + 38), np.uint8)
        K.check(_lib().vdo_orb_get_pyramid(self._h, level, _u8(out)))
        return out

    def blurred(self, level):
        w, h, _, _ = self.level_info(level)
        out = np.zeros((h, w), np.uint8)
        K.check(_lib().vdo_orb_get_blurred(self._h, level, _u8(out)))
        return out

    def candidates(self, level):
        _, _, _, nc = self.level_info(level)
        cap = max(nc, 1)
        x, y, r, a = (np.zeros(cap, np.float32) for _ in range(4))
        n = C.c_int()
        K.check(_lib().vdo_orb_get_candidates(self._h, level, _fp(x), _fp(y), _fp(r), _fp(a), cap, C.byref(n)))
        return x[:n.value], y[:n.value], r[:n.value], a[:n.value]

    def close(self):
        if self._h:
            _lib().vdo_orb_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass


def depth_preprocess(ctx, depth: np.ndarray, bf: float, factor: float) -> np.ndarray:
    d = np.ascontiguousarray(depth, dtype=np.float32).copy()
    K.check(_lib().vdo_depth_preprocess(ctx._h, _fp(d), d.size, bf, factor, 0))
    return d


def rgb2gray(ctx, rgb: np.ndarray, rgb_order=True) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.zeros(rgb.shape[:2], np.uint8)
    K.check(_lib().vdo_rgb2gray(ctx._h, _u8(rgb), out.size, rgb.shape[2], 1 if rgb_order else 0, _u8(out)))
    return out


class FrameImages:
    """Depth / flow / mask of one frame resident in HBM + the Frame::Frame kernels over them."""

    def __init__(self, ctx, width, height):
        self.ctx, self.w, self.h = ctx, width, height
        self._h = C.c_void_p()
        K.check(_lib().vdo_frame_images_create(ctx._h, width, height, C.byref(self._h)))

    def upload(self, depth, flow, mask):
        depth = np.ascontiguousarray(depth, dtype=np.float32); flow = np.ascontiguousarray(flow, dtype=np.float32)
        mask = np.ascontiguousarray(mask, dtype=np.int32)
        K.check(_lib().vdo_frame_images_upload(self._h, _fp(depth), _fp(flow), _ip(mask)))

    def upload_device(self, depth_ptr: int, flow_ptr: int, mask_ptr: int):
        """Bind device-resident inputs (raw pointers, e.g. ``tensor.data_ptr()``): D2D, stream-ordered."""
        L = _lib()
        L.vdo_frame_images_upload_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        K.check(L.vdo_frame_images_upload_device(self._h, C.c_void_p(depth_ptr), C.c_void_p(flow_ptr), C.c_void_p(mask_ptr)))

    def depth_preprocess(self, bf, factor):
        L = _lib()
        L.vdo_frame_images_depth_preprocess.argtypes = [C.c_void_p, C.c_float, C.c_float]
        K.check(L.vdo_frame_images_depth_preprocess(self._h, bf, factor))

    def static_filter(self, kx, ky, th_depth):
        """Outputs are views of buffers owned by this object: valid until the next call."""
        kx = np.ascontiguousarray(kx, dtype=np.float32); ky = np.ascontiguousarray(ky, dtype=np.float32)
        n = kx.size
        st = getattr(self, "_sf", None)
        if st is None or st[0] < n:
            cap = max(n, 4096)
            idx = np.zeros(cap, np.int32); f = [np.zeros(cap, np.float32) for _ in range(5)]
            m = C.c_int()
            st = self._sf = (cap, idx, f, m, [_ip(idx)] + [_fp(a) for a in f] + [C.byref(m)], _lib().vdo_frame_static_filter)
        _, idx, f, m, args, fn = st
        K.check(fn(self._h, n, _fp(kx), _fp(ky), th_depth, *args))
        m = m.value
        return dict(keep_idx=idx[:m], corr_x=f[0][:m], corr_y=f[1][:m], flow_x=f[2][:m], flow_y=f[3][:m], depth=f[4][:m])

    def object_sample(self, th_depth_obj, step=4):
        """Outputs are views of buffers owned by this object: valid until the next call."""
        cap = ((self.w + step - 1) // step) * ((self.h + step - 1) // step)
        st = getattr(self, "_os", None)
        if st is None or st[0] != cap:
            f = [np.zeros(cap, np.float32) for _ in range(7)]
            lab = np.zeros(cap, np.int32)
            m = C.c_int()
            st = self._os = (cap, f, lab, m, [_fp(a) for a in f] + [_ip(lab), C.byref(m)], _lib().vdo_frame_object_sample)
        _, f, lab, m, args, fn = st
        K.check(fn(self._h, th_depth_obj, step, cap, *args))
        m = m.value
        names = ("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y", "depth")
        out = {k: a[:m] for k, a in zip(names, f)}
        out["label"] = lab[:m]
        return out

    def filters(self, kx, ky, th_depth, th_depth_obj, step=4, sampled=False):
        """K9 + K10 in one call / one synchronisation: (static_filter dict, object_sample dict)."""
        kx = np.ascontiguousarray(kx, dtype=np.float32); ky = np.ascontiguousarray(ky, dtype=np.float32)
        n = kx.size
        idx = np.zeros(max(n, 1), np.int32); sf = [np.zeros(max(n, 1), np.float32) for _ in range(5)]
        cap = ((self.w + step - 1) // step) * ((self.h + step - 1) // step)
        f = [np.zeros(cap, np.float32) for _ in range(7)]
        lab = np.zeros(cap, np.int32)
        ms, mo = C.c_int(), C.c_int()
        L = _lib()
        fp, ip = K.c_float_p, K.c_int32_p
        L.vdo_frame_filters.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_float, C.c_int, ip, fp, fp, fp, fp, fp, C.POINTER(C.c_int),
                                        C.c_float, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp, fp, ip, C.POINTER(C.c_int)]
        K.check(L.vdo_frame_filters(self._h, n, _fp(kx), _fp(ky), th_depth, int(bool(sampled)), _ip(idx), *[_fp(a) for a in sf], C.byref(ms),
                                    th_depth_obj, step, cap, *[_fp(a) for a in f], _ip(lab), C.byref(mo)))
        m = ms.value
        st = dict(keep_idx=idx[:m], corr_x=sf[0][:m], corr_y=sf[1][:m], flow_x=sf[2][:m], flow_y=sf[3][:m], depth=sf[4][:m])
        m = mo.value
        names = ("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y", "depth")
        ob = {k: a[:m] for k, a in zip(names, f)}
        ob["label"] = lab[:m]
        return st, ob

    def close(self):
        if self._h:
            _lib().vdo_frame_images_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass
