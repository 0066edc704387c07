"""Tracking-side gathers (K11-K15) over HBM-resident images: thin ctypes veneer over the C-ABI
(include/vdo_slam_hip.h).  Mirrors the per-point loops of Tracking::GrabImageRGBD / GetSceneFlowObj /
RenewFrameInfo / UpdateMask (reference src/Tracking.cc:259-305, 1278-1364, 2666-2790, 3015-3065)."""
import ctypes as C

import numpy as np

from . import _capi as K


def _f(a): return np.ascontiguousarray(a, dtype=np.float32)
def _i(a): return np.ascontiguousarray(a, dtype=np.int32)
def _fp(a): return a.ctypes.data_as(K.c_float_p)
def _ip(a): return a.ctypes.data_as(K.c_int32_p)


_bound = False


def _lib():
    global _bound
    L = K.lib()
    if not _bound:
        fp, ip, vp = K.c_float_p, K.c_int32_p, C.c_void_p
        L.vdo_propagate_static.argtypes = [vp, C.c_int, fp, fp, fp]
        L.vdo_propagate_object.argtypes = [vp, C.c_int, fp, fp, C.c_float, fp, ip]
        L.vdo_get3d_world.argtypes = [vp, C.c_int, fp, fp, fp, fp, fp, fp]
        L.vdo_scene_flow.argtypes = [vp, C.c_int, fp, fp, fp, ip, fp, fp, fp, fp, ip, fp, fp, fp, ip]
        L.vdo_renew_static.argtypes = [vp, C.c_int, ip, fp, fp, C.c_int, fp, fp, C.c_int, fp, fp, fp, fp, fp, fp, ip, fp, C.POINTER(C.c_int)]
        L.vdo_mask_at.argtypes = [vp, C.c_int, fp, fp, ip]
        L.vdo_mask_warp.argtypes = [vp, vp, C.c_int32]
        L.vdo_frame_images_download_mask.argtypes = [vp, ip]
        _bound = True
    return L


def propagate_static(images, kx, ky):
    kx, ky = _f(kx), _f(ky)
    out = np.zeros(kx.size, np.float32)
    K.check(_lib().vdo_propagate_static(images._h, kx.size, _fp(kx), _fp(ky), _fp(out)))
    return out


def propagate_object(images, kx, ky, th_depth_obj):
    kx, ky = _f(kx), _f(ky)
    d = np.zeros(kx.size, np.float32); lab = np.zeros(kx.size, np.int32)
    K.check(_lib().vdo_propagate_object(images._h, kx.size, _fp(kx), _fp(ky), th_depth_obj, _fp(d), _ip(lab)))
    return d, lab


def get3d_world(ctx, kx, ky, depth, K4, Twc):
    kx, ky, depth, K4, Twc = _f(kx), _f(ky), _f(depth), _f(K4), _f(Twc)
    out = np.zeros((kx.size, 3), np.float32)
    K.check(_lib().vdo_get3d_world(ctx._h, kx.size, _fp(kx), _fp(ky), _fp(depth), _fp(K4), _fp(Twc), _fp(out)))
    return out


def scene_flow(ctx, cur, Tcw_cur, last, Tcw_last, K4, obj_label):
    """cur/last = (x, y, depth, label) arrays.  Returns (flow3d [n,3], obj_label with -1 where invalid)."""
    cx, cy, cd = (_f(a) for a in cur[:3]); cl = _i(cur[3])
    lx, ly, ld = (_f(a) for a in last[:3]); ll = _i(last[3])
    Tc, Tl, K4 = _f(Tcw_cur), _f(Tcw_last), _f(K4)
    ol = _i(obj_label).copy()
    out = np.zeros((cx.size, 3), np.float32)
    K.check(_lib().vdo_scene_flow(ctx._h, cx.size, _fp(cx), _fp(cy), _fp(cd), _ip(cl), _fp(Tc), _fp(lx), _fp(ly), _fp(ld), _ip(ll), _fp(Tl), _fp(K4), _fp(out), _ip(ol)))
    return out, ol


def renew_static(images, tm_sta, stat_x, stat_y, orb_x, orb_y, max_num_sta):
    tm, sx, sy, ox, oy = _i(tm_sta), _f(stat_x), _f(stat_y), _f(orb_x), _f(orb_y)
    cap = max_num_sta + 2
    f = [np.zeros(cap, np.float32) for _ in range(6)]
    ids = np.zeros(cap, np.int32); d = np.zeros(cap, np.float32)
    n = C.c_int()
    K.check(_lib().vdo_renew_static(images._h, tm.size, _ip(tm), _fp(sx), _fp(sy), ox.size, _fp(ox), _fp(oy), max_num_sta,
                                    *[_fp(a) for a in f], _ip(ids), _fp(d), C.byref(n)))
    n = n.value
    names = ("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y")
    out = {k: a[:n] for k, a in zip(names, f)}
    out["inlier_id"] = ids[:n]; out["depth"] = d[:n]
    return out


def mask_at(images, cx, cy):
    cx, cy = _f(cx), _f(cy)
    out = np.zeros(cx.size, np.int32)
    K.check(_lib().vdo_mask_at(images._h, cx.size, _fp(cx), _fp(cy), _ip(out)))
    return out


def mask_warp(cur_images, last_images, label):
    K.check(_lib().vdo_mask_warp(cur_images._h, last_images._h, int(label)))


def download_mask(images):
    out = np.zeros((images.h, images.w), np.int32)
    K.check(_lib().vdo_frame_images_download_mask(images._h, _ip(out)))
    return out
