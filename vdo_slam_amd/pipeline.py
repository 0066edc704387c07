"""ctypes handle on the C++ ``FramePipeline`` (vdo_slam_amd/host/FramePipeline.{h,cc}): the per-frame
sequence of Tracking::GrabImageRGBD + Track over the C-ABI.  Used by bench.py and the tests."""
import ctypes as C
import os

from . import _capi as K

HOST_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvdo_host.so")


class PipelineParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("K4", C.c_float * 4), ("bf", C.c_float), ("depth_map_factor", C.c_float),
                ("th_depth_bg", C.c_float), ("th_depth_obj", C.c_float), ("max_track_bg", C.c_int), ("max_track_obj", C.c_int),
                ("sf_mg_thres", C.c_float), ("sf_ds_thres", C.c_float), ("n_features", C.c_int), ("n_levels", C.c_int), ("ini_th", C.c_int),
                ("min_th", C.c_int), ("scale_factor", C.c_float), ("build_lm", C.c_int), ("defer_objects", C.c_int), ("use_sample_feature", C.c_int), ("sample_seed", C.c_int), ("pnp_refit", C.c_int), ("window_size", C.c_int), ("overlap_size", C.c_int)]


class FrameCounts(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n_orb", "n_static_new", "n_object_samples", "n_static_tracked", "n_object_tracked", "n_objects",
                                       "n_recovered_masks", "n_static_tracks", "n_dynamic_tracks", "n_ransac_cam", "n_motion_model_cam", "n_ransac_obj", "n_cam_inliers", "cam_lm_iterations", "n_mm_inliers_obj", "n_motion_model_obj")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


SECTIONS = ("k1_k11_ransac_cam", "orb", "k9_k10", "wait_cam_lm", "k13_dynobj", "renew_static", "wait_obj_lm", "renew_object", "tracklets", "ransac_obj", "k15_k11_objects")


def kitti_params(width, height, K4, bf, depth_map_factor, th_bg, th_obj, build_lm=0, defer_objects=0, window_size=0, overlap_size=0, use_sample_feature=0, sample_seed=1,
                 sf_mg_thres=0.12, sf_ds_thres=0.3, n_features=2500, pnp_refit=1):
    """example/kitti-0000-0013.yaml: MaxTrackPointBG 1200, MaxTrackPointOBJ 800, SFMgThres 0.12, SFDsThres 0.3, ORB 2500/1.2/8/20/7."""
    return PipelineParams(width, height, (C.c_float * 4)(*K4), bf, depth_map_factor, th_bg, th_obj, 1200, 800, sf_mg_thres, sf_ds_thres, n_features, 8, 20, 7, 1.2, int(build_lm), int(defer_objects),
                          int(use_sample_feature), int(sample_seed), int(pnp_refit), int(window_size), int(overlap_size))


class FramePipeline:
    def __init__(self, ctx, ctx_lm, params: PipelineParams, ctx_obj=None, ctx_worker=None, ctx_orb=None):
        L = self._L = K.load_host_lib()
        L.host_pipeline_create.restype = C.c_void_p
        L.host_pipeline_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PipelineParams), C.c_void_p, C.c_void_p, C.c_void_p]
        L.host_pipeline_flush.argtypes = [C.c_void_p, C.POINTER(FrameCounts)]
        L.host_pipeline_step.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.POINTER(FrameCounts)]
        L.host_pipeline_destroy.argtypes = [C.c_void_p]
        L.host_pipeline_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._keep = (ctx, ctx_lm, ctx_obj, ctx_worker, ctx_orb)
        self._h = L.host_pipeline_create(ctx._h, ctx_lm._h, C.byref(params), ctx_obj._h if ctx_obj is not None else None,
                                         ctx_worker._h if ctx_worker is not None else None, ctx_orb._h if ctx_orb is not None else None)
        if not self._h:
            raise K.VdoError("FramePipeline could not be created")
        for c in self._keep:
            if c is not None:
                c._retain()
        self.counts = FrameCounts()

    def step(self, d_gray: int, d_depth_raw: int, d_flow: int, d_mask: int, cam_batch=None, obj_batch=None, n_cam_pts=0, n_obj_problems=0):
        """Device pointers of the raw inputs + the frame's (resident) pose problems.  Returns the counts of the frame."""
        rc = self._L.host_pipeline_step(self._h, d_gray, d_depth_raw, d_flow, d_mask, cam_batch._h if cam_batch else None,
                                        obj_batch._h if obj_batch else None, n_cam_pts, n_obj_problems, C.byref(self.counts))
        if rc != 0:
            raise K.VdoError("FramePipeline.Step failed: " + (K.lib().vdo_last_error() or b"").decode())
        return self.counts.as_dict()

    def flush(self):
        """Deferred mode: ends the pending object stage of the last frame; returns the counts (object fields of that frame)."""
        if self._L.host_pipeline_flush(self._h, C.byref(self.counts)) != 0:
            raise K.VdoError("FramePipeline.Flush failed: " + (K.lib().vdo_last_error() or b"").decode())
        return self.counts.as_dict()

    def tracks(self, dynamic=False):
        """The tracklets Track() has built so far: (off, frame, feat, obj) - track t = pairs [off[t], off[t+1]) of (frame, feature index);
        obj = object id per track (dynamic tracklets; None for static)."""
        import numpy as np
        L = self._L
        L.host_pipeline_tracks.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), K.c_int32_p, K.c_int32_p, K.c_int32_p, K.c_int32_p]
        sz = (C.c_int64 * 2)()
        if L.host_pipeline_tracks(self._h, int(dynamic), sz, None, None, None, None) != 0:
            raise K.VdoError("FramePipeline.GetTracks failed")
        nt, npairs = int(sz[0]), int(sz[1])
        off = np.zeros(nt + 1, np.int32); fr = np.zeros(max(npairs, 1), np.int32); ft = np.zeros(max(npairs, 1), np.int32); ob = np.zeros(max(nt, 1), np.int32)
        ip = lambda a: a.ctypes.data_as(K.c_int32_p)
        if L.host_pipeline_tracks(self._h, int(dynamic), sz, ip(off), ip(fr), ip(ft), ip(ob)) != 0:
            raise K.VdoError("FramePipeline.GetTracks failed")
        return off, fr[:npairs], ft[:npairs], (ob[:nt] if dynamic else None)

    # ---- Map: Track() -> Map -> Optimizer::FullBatchOptimization
    def attach_map(self):
        """From now on every frame appends its features / poses / motions to a VDO_SLAM::Map ("Save Graph Structure")."""
        self._L.host_pipeline_attach_map.restype = C.c_void_p
        self._L.host_pipeline_attach_map.argtypes = [C.c_void_p]
        self._map = self._L.host_pipeline_attach_map(self._h)

    def keep_graph(self):
        """Like attach_map() but without a Map: the frames are appended to the pipeline's flat GraphStore only."""
        self._L.host_pipeline_keep_graph.argtypes = [C.c_void_p]
        self._L.host_pipeline_keep_graph(self._h)

    def full_batch_store(self):
        """Optimizer::FullBatchOptimization built straight from the GraphStore (no Map); returns the LM statistics."""
        st = K.LMStatsC()
        self._L.host_pipeline_full_batch.argtypes = [C.c_void_p, C.POINTER(K.LMStatsC)]
        if self._L.host_pipeline_full_batch(self._h, C.byref(st)) != 0:
            raise K.VdoError("FullBatchOptimization (store) failed")
        return st

    def store_poses(self, refined=False):
        """Camera poses T_wc [frames, 4, 4] of the GraphStore (refined: after the full batch)."""
        import numpy as np
        dims = (C.c_int64 * 5)()
        self._L.host_pipeline_store_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self._L.host_pipeline_store_dims(self._h, dims)
        out = np.zeros((max(1, dims[0]), 4, 4), np.float32)
        self._L.host_pipeline_store_poses.argtypes = [C.c_void_p, C.c_int, K.c_float_p]
        self._L.host_pipeline_store_poses(self._h, int(refined), out.ctypes.data_as(K.c_float_p))
        return out[:dims[0]]

    def partial_batches(self):
        """PartialBatchOptimization runs so far (window_size / overlap_size of the parameters)."""
        self._L.host_pipeline_partial_batches.argtypes = [C.c_void_p]
        return int(self._L.host_pipeline_partial_batches(self._h))

    def finalize_map(self):
        self._L.host_pipeline_finalize_map.argtypes = [C.c_void_p]
        if self._L.host_pipeline_finalize_map(self._h) != 0:
            raise K.VdoError("FramePipeline.FinalizeMap failed")

    def export_map(self, K4, refined=False):
        """The Map as a dict of flat arrays (the layout tests/map_builder_ref.py uses: cam_pose, feats, tr_sta, tr_dyn, obj_of_dyn, rigid_motion, rm_label)."""
        import numpy as np
        L = self._L
        dims = (C.c_int * 8)()
        L.host_map_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.host_map_dims(self._map, dims)
        F, ns, nd, ts, tsp, td, tdp, nrm = list(dims)
        f32 = lambda *sh: np.zeros(sh if all(sh) else tuple(max(1, q) for q in sh), np.float32)
        i32 = lambda n: np.zeros(max(1, n), np.int32)
        cam = f32(F, 4, 4); sta_cnt = i32(F); sta_uv = f32(ns, 2); sta_d = f32(ns); sta_xw = f32(ns, 3); tsl = i32(ts); tspairs = i32(2 * tsp)
        dyn_cnt = i32(F); dyn_uv = f32(nd, 2); dyn_d = f32(nd); dyn_xw = f32(nd, 3); tdl = i32(td); tdpairs = i32(2 * tdp); oid = i32(td)
        rm_cnt = i32(F); rm = f32(nrm, 4, 4); rml = i32(nrm)
        fp = lambda a: a.ctypes.data_as(K.c_float_p)
        ip = lambda a: a.ctypes.data_as(K.c_int32_p)
        L.host_map_export.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 17
        L.host_map_export(self._map, int(refined), fp(cam), ip(sta_cnt), fp(sta_uv), fp(sta_d), fp(sta_xw), ip(tsl), ip(tspairs),
                          ip(dyn_cnt), fp(dyn_uv), fp(dyn_d), fp(dyn_xw), ip(tdl), ip(tdpairs), ip(oid), ip(rm_cnt), fp(rm), ip(rml))
        so = np.concatenate([[0], np.cumsum(sta_cnt[:F])]); do = np.concatenate([[0], np.cumsum(dyn_cnt[:F])])
        feats = [dict(sta_uv=sta_uv[so[i]:so[i + 1]], sta_d=sta_d[so[i]:so[i + 1]], sta_xw=sta_xw[so[i]:so[i + 1]],
                      dyn_uv=dyn_uv[do[i]:do[i + 1]], dyn_d=dyn_d[do[i]:do[i + 1]], dyn_xw=dyn_xw[do[i]:do[i + 1]]) for i in range(F)]
        def tracks(lens, pairs, n):
            out, o = [], 0
            for t in range(n):
                out.append([(int(pairs[2 * (o + k)]), int(pairs[2 * (o + k) + 1])) for k in range(lens[t])]); o += lens[t]
            return out
        ro = np.concatenate([[0], np.cumsum(rm_cnt[:max(F - 1, 0)])])
        Km = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], np.float32)
        return dict(n_frames=F, K=Km, cam_pose=cam[:F], feats=feats, tr_sta=tracks(tsl, tspairs, ts), tr_dyn=tracks(tdl, tdpairs, td),
                    obj_of_dyn=oid[:td].copy(), rigid_motion=[rm[ro[i]:ro[i + 1]] for i in range(F - 1)], rm_label=[rml[ro[i]:ro[i + 1]] for i in range(F - 1)])

    def full_batch(self, K4):
        """Optimizer::FullBatchOptimization (C++ graph builder, GPU solve) on the attached Map; returns the LM statistics."""
        import numpy as np
        Km = np.array([K4[0], 0, K4[2], 0, K4[1], K4[3], 0, 0, 1], np.float32)
        st = K.LMStatsC()
        self._L.host_map_full_batch.argtypes = [C.c_void_p, K.c_float_p, C.POINTER(K.LMStatsC)]
        if self._L.host_map_full_batch(self._map, Km.ctypes.data_as(K.c_float_p), C.byref(st)) != 0:
            raise K.VdoError("FullBatchOptimization failed")
        return st

    def pose(self):
        """Tcw (4x4 float32) of the last frame."""
        import numpy as np
        T = np.zeros(16, np.float32)
        self._L.host_pipeline_pose.argtypes = [C.c_void_p, K.c_float_p]
        self._L.host_pipeline_pose(self._h, T.ctypes.data_as(K.c_float_p))
        return T.reshape(4, 4)

    def motions(self, cap=16):
        """Tracked objects of the last frame (build_lm mode): list of dict(mod_label, sem_label, n_inliers, H 4x4)."""
        import numpy as np
        ml = np.zeros(cap, np.int32); sl = np.zeros(cap, np.int32); ni = np.zeros(cap, np.int32); H = np.zeros((cap, 16), np.float32)
        self._L.host_pipeline_motions.argtypes = [C.c_void_p, C.c_int, K.c_int32_p, K.c_int32_p, K.c_int32_p, K.c_float_p]
        n = self._L.host_pipeline_motions(self._h, cap, ml.ctypes.data_as(K.c_int32_p), sl.ctypes.data_as(K.c_int32_p), ni.ctypes.data_as(K.c_int32_p), H.ctypes.data_as(K.c_float_p))
        return [dict(mod_label=int(ml[a]), sem_label=int(sl[a]), n_inliers=int(ni[a]), H=H[a].reshape(4, 4).copy()) for a in range(min(n, cap))]

    def section_ms(self):
        ms = (C.c_double * 11)()
        self._L.host_pipeline_timing(self._h, ms)
        return dict(zip(SECTIONS, ms))

    EVENTS = ("inputs_k1_k11", "cam_fetched", "obj_chain_k15_k11_k13", "dynobj", "obj_lm_built", "obj_lm_launched", "orb_device", "orb_done", "k9_filters",
              "static_done", "static_joined", "cam_stage_done", "obj_lm_fetched", "obj_renewed", "obj_done", "step_end")

    def events_ms(self, reset=True):
        """VDO_PIPE_EVENTS=1: mean time (ms after the start of its Step) at which each milestone of a frame was reached; -1 = never."""
        ms = (C.c_double * 16)()
        self._L.host_pipeline_events.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        self._L.host_pipeline_events(self._h, ms, int(reset))
        return dict(zip(self.EVENTS, (round(v, 4) for v in ms)))

    def close(self):
        if self._h:
            self._L.host_pipeline_destroy(self._h); self._h = None
            if getattr(self, "_map", None):
                self._L.host_map_destroy.argtypes = [C.c_void_p]
                self._L.host_map_destroy(self._map); self._map = None
            for c in self._keep:
                if c is not None:
                    c._release()

    def __del__(self):
        try: self.close()
        except Exception: pass
