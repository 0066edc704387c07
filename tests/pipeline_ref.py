"""The per-frame sequence of Tracking::GrabImageRGBD + Track composed from the ORACLE's functions
(sequential CPU restatements), chained frame to frame exactly like the product's C++ FramePipeline
(vdo_slam_amd/host/FramePipeline.cc).  Used as the checker of the pipeline test and as bench.py's CPU baseline."""
import numpy as np

from tests import frontend_ref as R
from tests import tracking_ref as T
from vdo_slam_amd import synth, synth_frames as SF
from vdo_slam_amd.tracking import DynObjParamsC
from vdo_slam_amd import _capi as K


class OraclePipeline:
    def __init__(self, oracle, max_bg=1200, max_obj=800, sf_mg=0.12, sf_ds=0.3):
        self.o = oracle
        self.K4 = np.array(synth.KITTI_K, np.float32)
        self.max_bg, self.max_obj, self.sf_mg, self.sf_ds = max_bg, max_obj, sf_mg, sf_ds
        self.last = None
        self.Tl = np.eye(4, dtype=np.float32)
        self.max_id = 1
        self.f_id = 0
        self.assos_s, self.assos_d, self.labs_d = [], [], []
        self.stage_s = {"depth": 0.0, "orb": 0.0, "frame": 0.0, "tracking_k11_k15": 0.0, "ransac_init": 0.0}
        self.vel = np.eye(4, dtype=np.float32)
        import ctypes as C
        dp = K.c_double_p
        oracle.vdo_oracle_p3p_ransac.argtypes = [C.c_int, dp, dp, dp, C.c_int, C.c_double, C.c_double, dp, K.c_uint8_p, K.c_int32_p, K.c_int32_p]

    def _ransac(self, X, uv):
        n = X.shape[0]
        if n < 4:
            return 0
        X = np.ascontiguousarray(X, np.float64); uv = np.ascontiguousarray(uv, np.float64)
        T = np.zeros(16)
        return self.o.vdo_oracle_p3p_ransac(n, K._dp(X), K._dp(uv), K._dp(self.K4.astype(np.float64)), 500, 0.4, 0.98, K._dp(T), None, None, None)

    def step(self, fr, Tc=None, inl=None, timer=None):
        """fr: dict(gray, depth_raw, flow, mask).  Tc: camera pose of this frame (float32 4x4, default: previous);
        inl: inlier flags of the camera optimisation (cycled over the static set, default all)."""
        import time
        o = self.o
        tick = time.perf_counter
        t = tick()
        d = fr["depth_raw"].copy()
        o.vdo_oracle_depth_preprocess(R._fp(d), d.size, SF.BF, SF.DEPTH_MAP_FACTOR)
        self.stage_s["depth"] += tick() - t; t = tick()
        mask = fr["mask"]
        last = self.last
        rec = 0
        if last is not None:                                                            # K15, K11
            mask, rec = T.update_mask(o, last["ob"]["label"], last["ob"]["corr_x"], last["ob"]["corr_y"], last["mask"], last["flow"], mask)
            T.propagate_static(o, last["st"]["corr_x"], last["st"]["corr_y"], d)
            od, osem = T.propagate_object(o, last["ob"]["corr_x"], last["ob"]["corr_y"], d, mask, SF.TH_DEPTH_OBJ)
        self.stage_s["tracking_k11_k15"] += tick() - t; t = tick()
        n_rc = n_mm = n_ro = 0
        if last is not None and last["st"]["corr_x"].size >= 4:                        # GetInitModelCam
            ls = last["st"]
            n_rc = self._ransac(ls["xyz"], np.c_[ls["corr_x"], ls["corr_y"]])
            MM = (self.vel.astype(np.float32) @ self.Tl.astype(np.float32)).astype(np.float32)
            Xc = ls["xyz"].astype(np.float32) @ MM[:3, :3].T + MM[:3, 3]
            u = self.K4[0] * Xc[:, 0] / Xc[:, 2] + self.K4[2]; v = self.K4[1] * Xc[:, 1] / Xc[:, 2] + self.K4[3]
            n_mm = int((np.sqrt((ls["corr_x"] - u) ** 2 + (ls["corr_y"] - v) ** 2) < 0.4).sum())
        self.stage_s["ransac_init"] += tick() - t; t = tick()
        kp = R.extract(o, fr["gray"])
        self.stage_s["orb"] += tick() - t; t = tick()
        st = R.static_filter(o, kp["x"], kp["y"], kp["octave"], mask, d, fr["flow"], SF.TH_DEPTH_BG)
        ob = R.object_sample(o, mask, d, fr["flow"], SF.TH_DEPTH_OBJ)
        self.stage_s["frame"] += tick() - t; t = tick()
        Tc = self.Tl if Tc is None else np.asarray(Tc, np.float32)
        counts = dict(n_orb=int(kp["x"].size), n_static_new=int(st["keep_idx"].size), n_object_samples=int(ob["label"].size), n_recovered_masks=int(rec), n_objects=0)
        if last is not None:
            lo = last["ob"]
            fl, olab = T.scene_flow(o, (lo["corr_x"], lo["corr_y"], od, osem), Tc, (lo["key_x"], lo["key_y"], lo["depth"], lo["label"]), self.Tl, self.K4,
                                    np.full(od.size, -2, np.int32))
            h, w = fr["mask"].shape
            prm = DynObjParamsC(w, h, 25, 50, self.sf_mg, self.sf_ds, SF.TH_DEPTH_OBJ, self.f_id)
            dyn = T.dyn_obj_tracking(o, prm, osem, olab, lo["corr_x"], lo["corr_y"], od, fl, lo["label"], last["sem_pos"], last["mod"],
                                     np.ones(len(last["mod"]), np.uint8), self.max_id)
            self.max_id = dyn["max_id"]
            counts["n_objects"] = len(dyn["objects"])
            self.stage_s["tracking_k11_k15"] += tick() - t; t = tick()
            for ids in dyn["objects"]:                                                  # GetInitModelObj
                n_ro += self._ransac(lo["xyz"][ids], np.c_[lo["corr_x"][ids], lo["corr_y"][ids]])
            self.stage_s["ransac_init"] += tick() - t; t = tick()
            ns = last["st"]["corr_x"].size
            if inl is None or inl.size == 0:
                tm = np.arange(ns, dtype=np.int32)
            else:
                tm = np.where(inl[np.arange(ns) % inl.size] != 0, np.arange(ns), -1).astype(np.int32)
            rs = T.renew_static(o, tm, last["st"]["corr_x"], last["st"]["corr_y"], kp["x"], kp["y"], mask, d, fr["flow"], self.max_bg)
            Twc = np.eye(4, dtype=np.float32)
            Twc[:3, :3] = Tc[:3, :3].T
            Twc[:3, 3] = -(Tc[:3, :3].T @ Tc[:3, 3])
            xyz_s = T.get3d_world(o, rs["key_x"], rs["key_y"], rs["depth"], self.K4, Twc)
            tmp = dict(x=ob["key_x"], y=ob["key_y"], depth=ob["depth"], label=ob["label"], flow_x=ob["flow_x"], flow_y=ob["flow_y"], corr_x=ob["corr_x"], corr_y=ob["corr_y"])
            ro = T.renew_object(o, dyn["objects"], np.ones(len(dyn["objects"]), np.uint8), dyn["sem"], dyn["mod"], lo["corr_x"], lo["corr_y"], dyn["obj_label"], tmp,
                                mask, d, fr["flow"], self.max_obj)
            xyz_o = T.get3d_world(o, ro["key_x"], ro["key_y"], ro["depth"], self.K4, Twc)
            self.assos_s.append(rs["inlier_id"]); self.assos_d.append(ro["inlier_id"]); self.labs_d.append(ro["obj_label"])
            ts = T.build_tracks(o, self.assos_s); td = T.build_tracks(o, self.assos_d, self.labs_d)   # the reference rebuilds from frame 0
            counts["n_static_tracks"], counts["n_dynamic_tracks"] = ts[0].size - 1, td[0].size - 1
            st_n = dict(corr_x=rs["corr_x"], corr_y=rs["corr_y"], xyz=xyz_s)
            ob_n = dict(key_x=ro["key_x"], key_y=ro["key_y"], corr_x=ro["corr_x"], corr_y=ro["corr_y"], depth=ro["depth"], label=ro["sem"], xyz=xyz_o)
            sem_pos, mod = dyn["sem"], dyn["mod"]
            self.result = dict(static=rs, objects=ro)
        else:
            I4 = np.eye(4, dtype=np.float32)                                           # Initialization(): Get3DinCamera
            sx, sy = kp["x"][st["keep_idx"]], kp["y"][st["keep_idx"]]
            st_n = dict(corr_x=st["corr_x"], corr_y=st["corr_y"], xyz=T.get3d_world(o, sx, sy, st["depth"], self.K4, I4))
            ob_n = dict(key_x=ob["key_x"], key_y=ob["key_y"], corr_x=ob["corr_x"], corr_y=ob["corr_y"], depth=ob["depth"], label=ob["label"],
                        xyz=T.get3d_world(o, ob["key_x"], ob["key_y"], ob["depth"], self.K4, I4))
            sem_pos, mod = np.zeros(0, np.int32), np.zeros(0, np.int32)
            counts["n_static_tracks"] = counts["n_dynamic_tracks"] = 0
        counts["n_static_tracked"], counts["n_object_tracked"] = int(st_n["corr_x"].size), int(ob_n["corr_x"].size)
        counts["n_ransac_cam"], counts["n_motion_model_cam"], counts["n_ransac_obj"] = int(n_rc), int(n_mm), int(n_ro)
        Twl = np.eye(4, dtype=np.float32); Twl[:3, :3] = self.Tl[:3, :3].T; Twl[:3, 3] = -(self.Tl[:3, :3].T @ self.Tl[:3, 3])
        self.vel = (Tc @ Twl).astype(np.float32)                                       # mVelocity
        self.stage_s["tracking_k11_k15"] += tick() - t
        self.last = dict(st=st_n, ob=ob_n, mask=mask, flow=fr["flow"], sem_pos=sem_pos, mod=mod)
        self.Tl = Tc
        self.f_id += 1
        return counts
