"""numpy-facing wrappers over the oracle's Tracking-side functions (oracle/tracking_oracle.cpp)."""
import numpy as np

from vdo_slam_amd import _capi as K


def _f(a): return np.ascontiguousarray(a, dtype=np.float32)
def _i(a): return np.ascontiguousarray(a, dtype=np.int32)
def _fp(a): return a.ctypes.data_as(K.c_float_p)
def _ip(a): return a.ctypes.data_as(K.c_int32_p)


def propagate_static(o, kx, ky, depth):
    kx, ky, depth = _f(kx), _f(ky), _f(depth)
    h, w = depth.shape
    out = np.zeros(kx.size, np.float32)
    o.vdo_oracle_propagate_static(kx.size, _fp(kx), _fp(ky), _fp(depth), w, h, _fp(out))
    return out


def propagate_object(o, kx, ky, depth, mask, th):
    kx, ky, depth, mask = _f(kx), _f(ky), _f(depth), _i(mask)
    h, w = depth.shape
    d = np.zeros(kx.size, np.float32); lab = np.zeros(kx.size, np.int32)
    o.vdo_oracle_propagate_object(kx.size, _fp(kx), _fp(ky), _fp(depth), _ip(mask), w, h, th, _fp(d), _ip(lab))
    return d, lab


def get3d_world(o, kx, ky, d, K4, Twc):
    kx, ky, d, K4, Twc = _f(kx), _f(ky), _f(d), _f(K4), _f(Twc)
    out = np.zeros((kx.size, 3), np.float32)
    o.vdo_oracle_get3d_world(kx.size, _fp(kx), _fp(ky), _fp(d), _fp(K4), _fp(Twc), _fp(out))
    return out


def scene_flow(o, cur, Tcw_cur, last, Tcw_last, K4, obj_label):
    cx, cy, cd = (_f(a) for a in cur[:3]); cl = _i(cur[3])
    lx, ly, ld = (_f(a) for a in last[:3]); ll = _i(last[3])
    Tc, Tl, K4 = _f(Tcw_cur), _f(Tcw_last), _f(K4)
    ol = _i(obj_label).copy()
    out = np.zeros((cx.size, 3), np.float32)
    o.vdo_oracle_scene_flow(cx.size, _fp(cx), _fp(cy), _fp(cd), _ip(cl), _fp(Tc), _fp(lx), _fp(ly), _fp(ld), _ip(ll), _fp(Tl), _fp(K4), _fp(out), _ip(ol))
    return out, ol


def renew_static(o, tm_sta, stat_x, stat_y, orb_x, orb_y, mask, depth, flow, max_num_sta):
    tm, sx, sy, ox, oy = _i(tm_sta), _f(stat_x), _f(stat_y), _f(orb_x), _f(orb_y)
    mask, depth, flow = _i(mask), _f(depth), _f(flow)
    h, w = mask.shape
    cap = max_num_sta + 2
    f = [np.zeros(cap, np.float32) for _ in range(6)]
    ids = np.zeros(cap, np.int32); d = np.zeros(cap, np.float32)
    n = o.vdo_oracle_renew_static(tm.size, _ip(tm), _fp(sx), _fp(sy), ox.size, _fp(ox), _fp(oy), _ip(mask), _fp(depth), _fp(flow), w, h, max_num_sta,
                                  *[_fp(a) for a in f], _ip(ids), _fp(d))
    names = ("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y")
    out = {k: a[:n] for k, a in zip(names, f)}
    out["inlier_id"] = ids[:n]; out["depth"] = d[:n]
    return out


def mask_at(o, cx, cy, mask):
    cx, cy, mask = _f(cx), _f(cy), _i(mask)
    h, w = mask.shape
    out = np.zeros(cx.size, np.int32)
    o.vdo_oracle_mask_at(cx.size, _fp(cx), _fp(cy), _ip(mask), w, h, _ip(out))
    return out


def mask_warp(o, mask_last, flow_last, lab, mask_cur):
    mask_last, flow_last = _i(mask_last), _f(flow_last)
    out = _i(mask_cur).copy()
    h, w = mask_last.shape
    o.vdo_oracle_mask_warp(_ip(mask_last), _fp(flow_last), w, h, int(lab), _ip(out))
    return out
