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


def dyn_obj_tracking(o, prm, sem_label, obj_label, key_x, key_y, depth, flow3d, last_sem_label, last_sem_pos, last_mod_label, last_obj_stat, max_id):
    import ctypes as C
    sem, ol = _i(sem_label), _i(obj_label).copy()
    kx, ky, d, fl, ls = _f(key_x), _f(key_y), _f(depth), _f(flow3d), _i(last_sem_label)
    lsp, lml = _i(last_sem_pos), _i(last_mod_label)
    lst = np.ascontiguousarray(last_obj_stat, dtype=np.uint8)
    n = sem.size
    nl = max(1, np.unique(sem).size)
    off = np.zeros(nl + 1, np.int32); idx = np.zeros(max(n, 1), np.int32); osem = np.zeros(nl, np.int32); omod = np.zeros(nl, np.int32)
    mid = np.array([max_id], np.int32)
    k = o.vdo_oracle_dyn_obj_tracking(n, _ip(sem), _ip(ol), _fp(kx), _fp(ky), _fp(d), _fp(fl), _ip(ls), lsp.size, _ip(lsp), _ip(lml),
                                      lst.ctypes.data_as(K.c_uint8_p), prm.img_w, prm.img_h, prm.shrink_row, prm.shrink_col,
                                      prm.sf_mg_thres, prm.sf_ds_thres, prm.th_depth_obj, prm.f_id, _ip(mid), _ip(off), _ip(idx), _ip(osem), _ip(omod))
    return dict(obj_label=ol, objects=[idx[off[a]:off[a + 1]].copy() for a in range(k)], sem=osem[:k].copy(), mod=omod[:k].copy(), max_id=int(mid[0]))


def renew_object(o, inl_sets, obj_stat, sem_pos, mod_label, cur_x, cur_y, cur_obj_label, tmp, mask, depth, flow, max_num_obj, cap=None):
    off = np.zeros(len(inl_sets) + 1, np.int32)
    off[1:] = np.cumsum([len(s) for s in inl_sets])
    idx = _i(np.concatenate([np.asarray(s, np.int32) for s in inl_sets])) if len(inl_sets) and off[-1] else np.zeros(1, np.int32)
    st = np.ascontiguousarray(obj_stat, dtype=np.uint8)
    sp, ml = _i(sem_pos), _i(mod_label)
    cx, cy, col = _f(cur_x), _f(cur_y), _i(cur_obj_label)
    t = {k: (_i(v) if k == "label" else _f(v)) for k, v in tmp.items()}
    mask, depth, flow = _i(mask), _f(depth), _f(flow)
    h, w = mask.shape
    n_tmp = t["x"].size
    cap = cap or (int(off[-1]) + n_tmp + 8)
    f = [np.zeros(cap, np.float32) for _ in range(7)]
    sem = np.zeros(cap, np.int32); inl = np.zeros(cap, np.int32); ol = np.zeros(cap, np.int32)
    m = o.vdo_oracle_renew_object(len(inl_sets), _ip(off), _ip(idx), st.ctypes.data_as(K.c_uint8_p), _ip(sp), _ip(ml), _fp(cx), _fp(cy), _ip(col),
                                  n_tmp, _fp(t["x"]), _fp(t["y"]), _fp(t["depth"]), _ip(t["label"]), _fp(t["flow_x"]), _fp(t["flow_y"]), _fp(t["corr_x"]), _fp(t["corr_y"]),
                                  _ip(mask), _fp(depth), _fp(flow), w, h, max_num_obj, cap,
                                  _fp(f[0]), _fp(f[1]), _fp(f[2]), _ip(sem), _fp(f[3]), _fp(f[4]), _fp(f[5]), _fp(f[6]), _ip(inl), _ip(ol))
    assert m >= 0
    names = ("key_x", "key_y", "depth", "flow_x", "flow_y", "corr_x", "corr_y")
    out = {k: a[:m] for k, a in zip(names, f)}
    out.update(sem=sem[:m], inlier_id=inl[:m], obj_label=ol[:m])
    return out


def update_mask(o, last_sem_label, last_corr_x, last_corr_y, mask_last, flow_last, mask_cur):
    sl, cx, cy = _i(last_sem_label), _f(last_corr_x), _f(last_corr_y)
    ml, fl = _i(mask_last), _f(flow_last)
    out = _i(mask_cur).copy()
    h, w = ml.shape
    rec = o.vdo_oracle_update_mask(sl.size, _ip(sl), _fp(cx), _fp(cy), _ip(ml), _fp(fl), w, h, _ip(out))
    return out, rec


def build_tracks(o, assos, labels=None):
    off = np.zeros(len(assos) + 1, np.int32)
    off[1:] = np.cumsum([len(a) for a in assos])
    flat = _i(np.concatenate(assos)) if off[-1] else np.zeros(1, np.int32)
    lab = _i(np.concatenate(labels)) if labels is not None and off[-1] else None
    cap_t = int(off[-1]) + 1; cap_p = 2 * int(off[-1]) + 2
    toff = np.zeros(cap_t + 1, np.int32); pf = np.zeros(cap_p, np.int32); pt = np.zeros(cap_p, np.int32); oid = np.zeros(cap_t, np.int32)
    nt = o.vdo_oracle_build_tracks(len(assos), _ip(off), _ip(flat), _ip(lab) if lab is not None else None, cap_t, cap_p, _ip(toff), _ip(pf), _ip(pt), _ip(oid))
    assert nt >= 0
    np_ = toff[nt]
    return toff[:nt + 1].copy(), pf[:np_].copy(), pt[:np_].copy(), (oid[:nt].copy() if labels is not None else None)


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
