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
                ("min_th", C.c_int), ("scale_factor", C.c_float), ("build_lm", C.c_int), ("defer_objects", C.c_int)]


class FrameCounts(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n_orb", "n_static_new", "n_object_samples", "n_static_tracked", "n_object_tracked", "n_objects",
                                       "n_recovered_masks", "n_static_tracks", "n_dynamic_tracks", "n_ransac_cam", "n_motion_model_cam", "n_ransac_obj", "n_cam_inliers", "cam_lm_iterations")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


SECTIONS = ("k1_k11_ransac_cam", "orb", "k9_k10", "wait_cam_lm", "k13_dynobj", "renew_static", "wait_obj_lm", "renew_object", "tracklets", "ransac_obj", "k15_k11_objects")


def kitti_params(width, height, K4, bf, depth_map_factor, th_bg, th_obj, build_lm=0, defer_objects=0):
    """example/kitti-0000-0013.yaml: MaxTrackPointBG 1200, MaxTrackPointOBJ 800, SFMgThres 0.12, SFDsThres 0.3, ORB 2500/1.2/8/20/7."""
    return PipelineParams(width, height, (C.c_float * 4)(*K4), bf, depth_map_factor, th_bg, th_obj, 1200, 800, 0.12, 0.3, 2500, 8, 20, 7, 1.2, int(build_lm), int(defer_objects))


class FramePipeline:
    def __init__(self, ctx, ctx_lm, params: PipelineParams, ctx_obj=None, ctx_worker=None):
        L = self._L = K.load_host_lib()
        L.host_pipeline_create.restype = C.c_void_p
        L.host_pipeline_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PipelineParams), C.c_void_p, C.c_void_p]
        L.host_pipeline_flush.argtypes = [C.c_void_p, C.POINTER(FrameCounts)]
        L.host_pipeline_step.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.POINTER(FrameCounts)]
        L.host_pipeline_destroy.argtypes = [C.c_void_p]
        L.host_pipeline_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._keep = (ctx, ctx_lm, ctx_obj, ctx_worker)
        self._h = L.host_pipeline_create(ctx._h, ctx_lm._h, C.byref(params), ctx_obj._h if ctx_obj is not None else None,
                                         ctx_worker._h if ctx_worker is not None else None)
        if not self._h:
            raise K.VdoError("FramePipeline could not be created")
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

    def close(self):
        if self._h:
            self._L.host_pipeline_destroy(self._h); self._h = None

    def __del__(self):
        try: self.close()
        except Exception: pass
