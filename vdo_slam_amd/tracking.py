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


def renew_static(images, tm_sta, stat_x, stat_y, orb_x, orb_y, max_num_sta, world=None):
    """RenewFrameInfo (static).  ``world`` = (K4, Twc): the same call also returns the 3-D points ``xyz`` of the new set."""
    tm, sx, sy, ox, oy = _i(tm_sta), _f(stat_x), _f(stat_y), _f(orb_x), _f(orb_y)
    cap = max_num_sta + 2
    f = [np.zeros(cap, np.float32) for _ in range(6)]
    ids = np.zeros(cap, np.int32); d = np.zeros(cap, np.float32)
    n = C.c_int()
    if world is None:
        K.check(_lib().vdo_renew_static(images._h, tm.size, _ip(tm), _fp(sx), _fp(sy), ox.size, _fp(ox), _fp(oy), max_num_sta,
                                        *[_fp(a) for a in f], _ip(ids), _fp(d), C.byref(n)))
    else:
        K4, Twc = _f(world[0]), _f(world[1])
        xyz = np.zeros((cap, 3), np.float32)
        L = K.lib()
        fp, ip = K.c_float_p, K.c_int32_p
        L.vdo_renew_static_world.argtypes = [C.c_void_p, C.c_int, ip, fp, fp, C.c_int, fp, fp, C.c_int, fp, fp, fp, fp, fp, fp, fp, fp, ip, fp, fp, C.POINTER(C.c_int)]
        K.check(L.vdo_renew_static_world(images._h, tm.size, _ip(tm), _fp(sx), _fp(sy), ox.size, _fp(ox), _fp(oy), max_num_sta, _fp(K4), _fp(Twc),
                                         *[_fp(a) for a in f], _ip(ids), _fp(d), _fp(xyz), C.byref(n)))
    n = n.value
    names = ("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y")
    out = {k: a[:n] for k, a in zip(names, f)}
    out["inlier_id"] = ids[:n]; out["depth"] = d[:n]
    if world is not None:
        out["xyz"] = xyz[:n]
    return out


def mask_at(images, cx, cy):
    cx, cy = _f(cx), _f(cy)
    out = np.zeros(cx.size, np.int32)
    K.check(_lib().vdo_mask_at(images._h, cx.size, _fp(cx), _fp(cy), _ip(out)))
    return out


def mask_warp(cur_images, last_images, label):
    K.check(_lib().vdo_mask_warp(cur_images._h, last_images._h, int(label)))


class DynObjParamsC(C.Structure):
    _fields_ = [("img_w", C.c_int32), ("img_h", C.c_int32), ("shrink_row", C.c_int32), ("shrink_col", C.c_int32),
                ("sf_mg_thres", C.c_float), ("sf_ds_thres", C.c_float), ("th_depth_obj", C.c_float), ("f_id", C.c_int32)]


def _u8p(a): return a.ctypes.data_as(K.c_uint8_p)


def dyn_obj_tracking(prm: DynObjParamsC, sem_label, obj_label, key_x, key_y, depth, flow3d, last_sem_label,
                     last_sem_pos, last_mod_label, last_obj_stat, max_id):
    """DynObjTracking.  Returns dict(obj_label, objects=[index arrays], sem=[...], mod=[...], max_id)."""
    sem, ol = _i(sem_label), _i(obj_label).copy()
    kx, ky, d, fl, ls = _f(key_x), _f(key_y), _f(depth), _f(flow3d), _i(last_sem_label)
    lsp, lml = _i(last_sem_pos), _i(last_mod_label)
    lst = np.ascontiguousarray(last_obj_stat, dtype=np.uint8)
    n = sem.size
    nl = max(1, np.unique(sem).size)
    off = np.zeros(nl + 1, np.int32); idx = np.zeros(max(n, 1), np.int32); osem = np.zeros(nl, np.int32); omod = np.zeros(nl, np.int32)
    mid = C.c_int32(max_id); nobj = C.c_int()
    L = K.lib()
    L.vdo_dyn_obj_tracking.argtypes = [C.POINTER(DynObjParamsC), C.c_int, K.c_int32_p, K.c_int32_p, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p,
                                       K.c_int32_p, C.c_int, K.c_int32_p, K.c_int32_p, K.c_uint8_p, C.POINTER(C.c_int32),
                                       K.c_int32_p, K.c_int32_p, K.c_int32_p, K.c_int32_p, C.POINTER(C.c_int)]
    K.check(L.vdo_dyn_obj_tracking(C.byref(prm), n, _ip(sem), _ip(ol), _fp(kx), _fp(ky), _fp(d), _fp(fl), _ip(ls), lsp.size, _ip(lsp), _ip(lml), _u8p(lst),
                                   C.byref(mid), _ip(off), _ip(idx), _ip(osem), _ip(omod), C.byref(nobj)))
    k = nobj.value
    return dict(obj_label=ol, objects=[idx[off[a]:off[a + 1]].copy() for a in range(k)], sem=osem[:k].copy(), mod=omod[:k].copy(), max_id=mid.value)


def renew_object(images, inl_sets, obj_stat, sem_pos, mod_label, cur_x, cur_y, cur_obj_label, tmp, max_num_obj, cap=None, world=None):
    """RenewFrameInfo (objects).  ``inl_sets``: list of index arrays; ``tmp``: dict(x,y,depth,label,flow_x,flow_y,corr_x,corr_y);
    ``world`` = (K4, Twc): the same call also returns the 3-D points ``xyz`` of the new set."""
    off = np.zeros(len(inl_sets) + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in inl_sets])
    idx = _i(np.concatenate([np.asarray(s, np.int32) for s in inl_sets])) if len(inl_sets) and off[-1] else np.zeros(1, np.int32)
    st = np.ascontiguousarray(obj_stat, dtype=np.uint8)
    sp, ml = _i(sem_pos), _i(mod_label)
    cx, cy, col = _f(cur_x), _f(cur_y), _i(cur_obj_label)
    t = {k: (_i(v) if k == "label" else _f(v)) for k, v in tmp.items()}
    n_tmp = t["x"].size
    cap = cap or (int(off[-1]) + n_tmp + 8)
    f = [np.zeros(cap, np.float32) for _ in range(7)]       # key_x key_y depth flow_x flow_y corr_x corr_y
    sem = np.zeros(cap, np.int32); inl = np.zeros(cap, np.int32); ol = np.zeros(cap, np.int32)
    n = C.c_int()
    L = K.lib()
    fp, ip = K.c_float_p, K.c_int32_p
    L.vdo_renew_object.argtypes = [C.c_void_p, C.c_int, ip, ip, K.c_uint8_p, ip, ip, fp, fp, ip, C.c_int, fp, fp, fp, ip, fp, fp, fp, fp, C.c_int, C.c_int,
                                   fp, fp, fp, ip, fp, fp, fp, fp, ip, ip, C.POINTER(C.c_int)]
    if world is None:
        K.check(L.vdo_renew_object(images._h, len(inl_sets), _ip(off), _ip(idx), _u8p(st), _ip(sp), _ip(ml), _fp(cx), _fp(cy), _ip(col),
                                   n_tmp, _fp(t["x"]), _fp(t["y"]), _fp(t["depth"]), _ip(t["label"]), _fp(t["flow_x"]), _fp(t["flow_y"]), _fp(t["corr_x"]), _fp(t["corr_y"]),
                                   max_num_obj, cap, _fp(f[0]), _fp(f[1]), _fp(f[2]), _ip(sem), _fp(f[3]), _fp(f[4]), _fp(f[5]), _fp(f[6]), _ip(inl), _ip(ol), C.byref(n)))
    else:
        K4, Twc = _f(world[0]), _f(world[1])
        xyz = np.zeros((cap, 3), np.float32)
        L.vdo_renew_object_world.argtypes = [C.c_void_p, C.c_int, ip, ip, K.c_uint8_p, ip, ip, fp, fp, ip, C.c_int, fp, fp, fp, ip, fp, fp, fp, fp, C.c_int, C.c_int, fp, fp,
                                             fp, fp, fp, ip, fp, fp, fp, fp, ip, ip, fp, C.POINTER(C.c_int)]
        K.check(L.vdo_renew_object_world(images._h, len(inl_sets), _ip(off), _ip(idx), _u8p(st), _ip(sp), _ip(ml), _fp(cx), _fp(cy), _ip(col),
                                         n_tmp, _fp(t["x"]), _fp(t["y"]), _fp(t["depth"]), _ip(t["label"]), _fp(t["flow_x"]), _fp(t["flow_y"]), _fp(t["corr_x"]), _fp(t["corr_y"]),
                                         max_num_obj, cap, _fp(K4), _fp(Twc), _fp(f[0]), _fp(f[1]), _fp(f[2]), _ip(sem), _fp(f[3]), _fp(f[4]), _fp(f[5]), _fp(f[6]), _ip(inl), _ip(ol),
                                         _fp(xyz), C.byref(n)))
    m = n.value
    names = ("key_x", "key_y", "depth", "flow_x", "flow_y", "corr_x", "corr_y")
    out = {k: a[:m] for k, a in zip(names, f)}
    out.update(sem=sem[:m], inlier_id=inl[:m], obj_label=ol[:m])
    if world is not None:
        out["xyz"] = xyz[:m]
    return out


def update_mask(cur_images, last_images, last_sem_label, last_corr_x, last_corr_y):
    """UpdateMask: ``cur_images``' mask is updated in HBM; returns the number of recovered labels."""
    sl, cx, cy = _i(last_sem_label), _f(last_corr_x), _f(last_corr_y)
    n = C.c_int()
    L = K.lib()
    L.vdo_update_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int, K.c_int32_p, K.c_float_p, K.c_float_p, C.POINTER(C.c_int)]
    K.check(L.vdo_update_mask(cur_images._h, last_images._h, sl.size, _ip(sl), _fp(cx), _fp(cy), C.byref(n)))
    return n.value


def object_chain(cur_images, last_images, last_sem_label, last_corr_x, last_corr_y, th_depth_obj, Tcw_cur, last_x, last_y, last_d, Tcw_last, K4):
    """UpdateMask -> object propagation -> scene flow in one call (vdo_object_chain).
    Returns (n_recovered, depth, sem_label, flow3d [n,3], obj_label)."""
    sl, cx, cy = _i(last_sem_label), _f(last_corr_x), _f(last_corr_y)
    lx, ly, ld = _f(last_x), _f(last_y), _f(last_d)
    Tc, Tl, K4 = _f(Tcw_cur), _f(Tcw_last), _f(K4)
    n = sl.size
    rec = C.c_int()
    d = np.zeros(n, np.float32); sem = np.zeros(n, np.int32); fl = np.zeros((n, 3), np.float32); ol = np.zeros(n, np.int32)
    L = K.lib()
    L.vdo_object_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, K.c_int32_p, K.c_float_p, K.c_float_p, C.c_float, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p,
                                   K.c_float_p, K.c_float_p, C.POINTER(C.c_int), K.c_float_p, K.c_int32_p, K.c_float_p, K.c_int32_p]
    K.check(L.vdo_object_chain(cur_images._h, last_images._h, n, _ip(sl), _fp(cx), _fp(cy), th_depth_obj, _fp(Tc), _fp(lx), _fp(ly), _fp(ld), _fp(Tl), _fp(K4),
                               C.byref(rec), _fp(d), _ip(sem), _fp(fl), _ip(ol)))
    return rec.value, d, sem, fl, ol


def object_chain_prestage(ctx, last_sem_label, last_corr_x, last_corr_y, last_x, last_y, last_d):
    """vdo_object_chain_prestage: the last frame's half of object_chain's inputs sent to the device ahead (asynchronously, on the context's stream)."""
    sl, cx, cy = _i(last_sem_label), _f(last_corr_x), _f(last_corr_y)
    lx, ly, ld = _f(last_x), _f(last_y), _f(last_d)
    L = K.lib()
    L.vdo_object_chain_prestage.argtypes = [C.c_void_p, C.c_int, K.c_int32_p, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p, K.c_float_p]
    K.check(L.vdo_object_chain_prestage(ctx._h, sl.size, _ip(sl), _fp(cx), _fp(cy), _fp(lx), _fp(ly), _fp(ld)))


class TrackBuilder:
    """Incremental GetStaticTrack / GetDynamicTrackNew (host only: usable without a GPU)."""

    def __init__(self, with_object_label=False):
        L = K.lib()
        L.vdo_tracks_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.vdo_tracks_destroy.argtypes = [C.c_void_p]
        L.vdo_tracks_add_frame.argtypes = [C.c_void_p, C.c_int, K.c_int32_p, K.c_int32_p]
        L.vdo_tracks_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.vdo_tracks_get.argtypes = [C.c_void_p, K.c_int32_p, K.c_int32_p, K.c_int32_p, K.c_int32_p]
        self.with_label = with_object_label
        self._h = C.c_void_p()
        K.check(L.vdo_tracks_create(int(with_object_label), C.byref(self._h)))

    def add_frame(self, asso, feat_label=None):
        a = _i(asso)
        lab = _i(feat_label) if feat_label is not None else None
        K.check(K.lib().vdo_tracks_add_frame(self._h, a.size, _ip(a), _ip(lab) if lab is not None else None))

    def get(self):
        nt, npairs = C.c_int(), C.c_int64()
        K.check(K.lib().vdo_tracks_size(self._h, C.byref(nt), C.byref(npairs)))
        off = np.zeros(nt.value + 1, np.int32); fr = np.zeros(max(npairs.value, 1), np.int32); ft = np.zeros(max(npairs.value, 1), np.int32)
        oid = np.zeros(max(nt.value, 1), np.int32)
        K.check(K.lib().vdo_tracks_get(self._h, _ip(off), _ip(fr), _ip(ft), _ip(oid)))
        return off, fr[:npairs.value], ft[:npairs.value], (oid[:nt.value] if self.with_label else None)

    def close(self):
        if self._h:
            K.lib().vdo_tracks_destroy(self._h); self._h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass


def download_mask(images):
    out = np.zeros((images.h, images.w), np.int32)
    K.check(_lib().vdo_frame_images_download_mask(images._h, _ip(out)))
    return out
