"""ctypes handle on the C++ ``System`` (vdo_slam_amd/host/System.{h,cc}): the reference's entry API
``System(settings, RGBD)`` / ``TrackRGBD(im, depth, flow, mask, Tcw_gt, objPose_gt, t, imTraj, nImage)``
(include/System.h:42-53) on HOST images.  Used by bench.py (`value_host_inputs`) and the tests."""
import ctypes as C

import numpy as np

from . import _capi as K

SETTINGS = """%YAML:1.0
Camera.fx: {fx}
Camera.fy: {fy}
Camera.cx: {cx}
Camera.cy: {cy}
Camera.k1: 0.0
Camera.width: {w}
Camera.height: {h}
Camera.fps: 10.0
Camera.bf: {bf}
Camera.RGB: 1
ChooseData: {choose_data}
DepthMapFactor: {dmf}
ThDepthBG: {thbg}
ThDepthOBJ: {thobj}
MaxTrackPointBG: 1200 # 1200 1500 2000
MaxTrackPointOBJ: 800
SFMgThres: {sf_mg_thres} # 0.05
SFDsThres: 0.3
WINDOW_SIZE: {window}
OVERLAP_SIZE: {overlap}
UseSampleFeature: {use_sample_feature}
ORBextractor.nFeatures: {n_features}
ORBextractor.scaleFactor: 1.2
ORBextractor.nLevels: 8
ORBextractor.iniThFAST: 20
ORBextractor.minThFAST: 7
"""


def write_settings(path, w, h, K4, bf, dmf, thbg, thobj, window=20, overlap=4, choose_data=2, use_sample_feature=0, sf_mg_thres=0.12, n_features=2500):
    """A settings file in the reference's flat YAML dialect (example/kitti-0000-0013.yaml)."""
    fx, fy, cx, cy = K4
    with open(path, "w") as f:
        f.write(SETTINGS.format(fx=fx, fy=fy, cx=cx, cy=cy, w=w, h=h, bf=bf, dmf=dmf, thbg=thbg, thobj=thobj, window=window, overlap=overlap, choose_data=choose_data,
                                use_sample_feature=use_sample_feature, sf_mg_thres=sf_mg_thres, n_features=n_features))
    return str(path)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class System:
    def __init__(self, settings_path):
        L = self._L = K.load_host_lib()
        L.host_system_create.restype = C.c_void_p
        L.host_system_create.argtypes = [C.c_char_p]
        L.host_system_destroy.argtypes = [C.c_void_p]
        L.host_system_track.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.host_system_motions.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.host_system_refined_poses.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.host_system_save.argtypes = [C.c_void_p, C.c_char_p]
        self._h = L.host_system_create(str(settings_path).encode())
        if not self._h:
            raise K.VdoError("System could not be created (settings file / HIP device)")
        self._T = np.zeros(16, np.float32)

    def track_rgbd(self, im, depth, flow, mask, obj_rows=None, n_images=1 << 30):
        """One TrackRGBD call.  im: [h, w] or [h, w, 3|4] uint8; depth [h, w] float32 (raw on entry, METRES on return, as the
        reference mutates it); flow [h, w, 2] float32; mask [h, w] int32 (recovered labels are written back); obj_rows:
        ground-truth rows [n, L] float32 (row[1] = label) or None.  Returns Tcw 4x4 float32, or None (rejected / failed frame)."""
        h, w = im.shape[:2]
        ch = 1 if im.ndim == 2 else im.shape[2]
        rows = None if obj_rows is None or len(obj_rows) == 0 else np.ascontiguousarray(obj_rows, np.float32)
        rc = self._L.host_system_track(self._h, _ptr(im), ch, _ptr(depth), _ptr(flow), _ptr(mask), w, h, _ptr(rows), 0 if rows is None else rows.shape[0],
                                       0 if rows is None else rows.shape[1], int(n_images), _ptr(self._T))
        return None if rc != 0 else self._T.reshape(4, 4).copy()

    def set_defer(self, on=True):
        """Throughput mode: the object stage of a frame ends inside the next track_rgbd call (same results one frame later)."""
        self._L.host_system_set_defer.argtypes = [C.c_void_p, C.c_int]
        self._L.host_system_set_defer(self._h, int(bool(on)))

    def flush(self):
        self._L.host_system_flush.argtypes = [C.c_void_p]
        if self._L.host_system_flush(self._h) != 0:
            raise K.VdoError("System flush failed")

    def motions(self, cap=32):
        sl = np.zeros(cap, np.int32); Hm = np.zeros((cap, 16), np.float32)
        n = self._L.host_system_motions(self._h, cap, _ptr(sl), _ptr(Hm))
        return [(int(sl[a]), Hm[a].reshape(4, 4).copy()) for a in range(min(n, cap))]

    def refined_poses(self, n):
        rf = np.zeros((n, 16), np.float32)
        m = self._L.host_system_refined_poses(self._h, n, _ptr(rf))
        return rf[:min(m, n)].reshape(-1, 4, 4)

    def tracks(self, dynamic=False):
        """(off, frame, feat, obj): tracklet t = pairs [off[t], off[t+1]) of (frame, feature index); obj = object id per dynamic tracklet"""
        L = self._L
        L.host_system_tracks.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), K.c_int32_p, K.c_int32_p, K.c_int32_p, K.c_int32_p]
        sz = (C.c_int64 * 2)()
        if L.host_system_tracks(self._h, int(dynamic), sz, None, None, None, None) != 0:
            raise K.VdoError("System.tracks failed")
        nt, npairs = int(sz[0]), int(sz[1])
        off = np.zeros(nt + 1, np.int32); fr = np.zeros(max(npairs, 1), np.int32); ft = np.zeros(max(npairs, 1), np.int32); ob = np.zeros(max(nt, 1), np.int32)
        ip = lambda a: a.ctypes.data_as(K.c_int32_p)
        if L.host_system_tracks(self._h, int(dynamic), sz, ip(off), ip(fr), ip(ft), ip(ob)) != 0:
            raise K.VdoError("System.tracks failed")
        return off, fr[:npairs], ft[:npairs], (ob[:nt] if dynamic else None)

    def frame_state(self, what, rows):
        """Tracking::SyncFrameState() + a flat copy (host_system_frame_state): what = 0 static set [10][n], 1 object set [12][n], 2 per object [n][19],
        3 samples [8][n], 4 scalars [17]; returns (n, flat float32 array)"""
        L = self._L
        L.host_system_frame_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        n = L.host_system_frame_state(self._h, what, None, 0)
        if n < 0:
            raise K.VdoError("System.frame_state failed")
        buf = np.zeros(max(rows * n, 1), np.float32)
        if L.host_system_frame_state(self._h, what, _ptr(buf), buf.size) != n:
            raise K.VdoError("System.frame_state failed")
        return n, buf[:rows * n]

    def map_export(self, what):
        """flat copy of the Map in the reference's format (host_system_map_export): 0 vmCameraPose [F][16], 1 vmCameraPose_RF, 2 vmRigidMotion [n][16], 3 vmRigidMotion_RF,
        4 vnRMLabel [n], 5 vp3DPointSta [n][3], 6 vp3DPointDyn [n][3], 7 motions per frame [F-1]"""
        L = self._L
        L.host_system_map_export.restype = C.c_long
        L.host_system_map_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
        n = L.host_system_map_export(self._h, what, None, 0)
        if n < 0:
            raise K.VdoError("System.map_export failed")
        buf = np.zeros(max(n, 1), np.float32)
        if L.host_system_map_export(self._h, what, _ptr(buf), n) != n:
            raise K.VdoError("System.map_export failed")
        return buf[:n]

    def save(self, path):
        self._L.host_system_save(self._h, str(path).encode())

    def close(self):
        if self._h:
            self._L.host_system_destroy(self._h); self._h = None

    def __del__(self):
        try: self.close()
        except Exception: pass
