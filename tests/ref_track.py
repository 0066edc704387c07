"""Driver of oracle/_ref/libref_track.so - the reference's own System / Tracking / Frame / Map / ORBextractor sources compiled verbatim (oracle/ref/) -
for the tests: one System::TrackRGBD call per frame and flat views of what Track() leaves behind.  TEST INFRASTRUCTURE."""
import ctypes as C
import os

import numpy as np

from tests import oracle_lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_SCRATCH = None


def scratch_dir():
    """where the reference may drop its files: GetStaticTrack / GetDynamicTrackNew write track_distribution*.txt (src/Tracking.cc:2294-2303, :2410-2419) and the
    batch optimisers dynamic_slam_graph_{before,after}_opt.g2o / local_ba_*.g2o (src/Optimizer.cc:806-808, :1934-1936) into the CURRENT directory"""
    global _SCRATCH
    if _SCRATCH is None:
        import atexit
        import shutil
        import tempfile
        _SCRATCH = tempfile.mkdtemp(prefix="vdo_ref_cwd_")
        atexit.register(shutil.rmtree, _SCRATCH, ignore_errors=True)
    return _SCRATCH


class Quiet:
    """The reference prints a few hundred lines per frame to stdout: sent to /dev/null for the duration of a call; the call runs with a scratch directory
    as its cwd (the reference writes its dumps there)."""
    def __enter__(self):
        import sys
        self._cwd = os.getcwd()
        os.chdir(scratch_dir())
        self._on = not os.environ.get("VDO_REF_VERBOSE")      # (debug: let the reference talk)
        if not self._on:
            return self
        sys.stdout.flush()
        self._saved = os.dup(1)
        self._null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self._null, 1)
        return self

    def __exit__(self, *a):
        os.chdir(self._cwd)
        if not self._on:
            return
        os.dup2(self._saved, 1)
        os.close(self._null); os.close(self._saved)


class RefSystem:
    COUNTS = ("n_keys", "n_static", "n_object", "n_objects", "n_cam_subset", "cam_lm_iterations", "f_id", "max_id", "n_samples", "full_batch_calls", "partial_batch_calls",
              "n_static_tracks", "n_dynamic_tracks")

    def __init__(self, settings_path, full=False):
        """full: libref_full.so (the reference's real Optimizer.cc + g2o behind Track()) instead of libref_track.so (the oracle's optimisers behind it)"""
        self.L = oracle_lib.load_ref_full() if full else oracle_lib.load_ref_track()
        if self.L is None:
            raise RuntimeError("oracle/_ref/libref_full.so is absent" if full else "oracle/_ref/libref_track.so is absent")
        with Quiet():
            self.h = self.L.vdo_ref_system_create(str(settings_path).encode())

    def track(self, fr, k, n_images, labels=(1, 2, 3, 4, 5, 6, 7, 8), timestamp=0.0, fake_time=None, Tcw_gt=None):
        """fr: dict(gray u8 [h, w] or [h, w, c], depth_raw f32, flow f32 [h, w, 2], mask i32).  Returns (Tcw 4x4 f32, converted depth, mask as the call left it)."""
        gray = np.ascontiguousarray(fr["gray"])
        ch = 1 if gray.ndim == 2 else gray.shape[2]
        depth = np.ascontiguousarray(fr["depth_raw"], np.float32).copy(); mask = np.ascontiguousarray(fr["mask"], np.int32).copy()
        flow = np.ascontiguousarray(fr["flow"], np.float32)
        h, w = mask.shape
        rows = np.array([[k, lab, 0, 0, 0, 0, 0, 0, 0, 0] for lab in labels], np.float32).reshape(-1, 10)
        T = np.zeros(16, np.float32)
        gt = np.eye(4, dtype=np.float32) if Tcw_gt is None else np.ascontiguousarray(Tcw_gt, np.float32)
        self.L.vdo_ref_set_time(-1 if fake_time is None else int(fake_time))
        with Quiet():
            rc = self.L.vdo_ref_system_track(self.h, _p(gray), ch, _p(depth), _p(flow), _p(mask), w, h, _p(gt), _p(rows) if len(rows) else None, len(rows), 10, float(timestamp), int(n_images), _p(T))
        # The reference keeps SHALLOW headers on the caller's images for one more frame (mImGrayLast / mDepthMapLast / mFlowMapLast / mSegMapLast,
        # src/Tracking.cc:641-644 - UpdateMask reads them): the buffers of the last two calls stay alive here whatever the caller drops
        self._alive = (getattr(self, "_alive", ()) + ((gray, depth, flow, mask),))[-2:]
        if rc != 0:
            raise RuntimeError("System::TrackRGBD returned an empty pose")
        return T.reshape(4, 4), depth, mask

    def counts(self):
        c = np.zeros(13, np.int32)
        self.L.vdo_ref_system_counts(self.h, _p(c))
        return dict(zip(self.COUNTS, (int(v) for v in c)))

    def state(self, what, rows):
        n = self.L.vdo_ref_system_frame_state(self.h, what, None, 0)
        assert n >= 0
        buf = np.zeros(max(rows * n, 1), np.float32)
        assert self.L.vdo_ref_system_frame_state(self.h, what, _p(buf), buf.size) == n
        return n, buf[:rows * n]

    def tracks(self, dynamic=False):
        sz = np.zeros(2, np.int64)
        self.L.vdo_ref_system_tracks(self.h, int(dynamic), _p(sz), None, None, None, None)
        nt, npairs = int(sz[0]), int(sz[1])
        off = np.zeros(nt + 1, np.int32); fr = np.zeros(max(npairs, 1), np.int32); ft = np.zeros(max(npairs, 1), np.int32); ob = np.zeros(max(nt, 1), np.int32)
        self.L.vdo_ref_system_tracks(self.h, int(dynamic), _p(sz), _p(off), _p(fr), _p(ft), _p(ob))
        return off, fr[:npairs], ft[:npairs], (ob[:nt] if dynamic else None)

    def map_export(self, what):
        """flat copy of the Map (vdo_ref_system_map_export): 0 vmCameraPose, 1 vmCameraPose_RF, 2 vmRigidMotion, 3 vmRigidMotion_RF, 4 vnRMLabel, 5 vp3DPointSta, 6 vp3DPointDyn, 7 motions per frame"""
        n = self.L.vdo_ref_system_map_export(self.h, what, None, 0)
        assert n >= 0
        buf = np.zeros(max(n, 1), np.float32)
        assert self.L.vdo_ref_system_map_export(self.h, what, _p(buf), n) == n
        return buf[:n]

    def timing_ms(self):
        t = np.zeros(5, np.float32)
        self.L.vdo_ref_system_timing(self.h, _p(t))
        return dict(zip(("mask_update", "camera_estimate", "object_tracking", "object_estimate", "map_update"), (float(v) for v in t)))

    def close(self):
        if self.h:
            self.L.vdo_ref_system_destroy(self.h); self.h = None


# ---- the same through a child process (GPU tests: the reference's single-threaded code with its never-initialised reads stays out of a process that
# carries the HIP runtime and the product's helper threads) -----------------------------------------------------------------------------------
def worker_main(settings, frames_npz, out_npz, n_images, labels, full=False):
    z = np.load(frames_npz)
    n = int(z["n"])
    rs = RefSystem(settings, full=full)
    out = {"n": n}
    for k in range(n):
        fr = {q: z[f"{q}_{k}"] for q in ("gray", "depth_raw", "flow", "mask")}
        T, depth, mask = rs.track(fr, k, n_images=n_images, labels=labels)
        out[f"T_{k}"] = T; out[f"depth_{k}"] = depth; out[f"mask_{k}"] = mask
        for what, rows in ((0, 10), (1, 12), (2, 19), (3, 8), (4, 17)):
            cnt, a = rs.state(what, rows)
            out[f"s{what}_{k}"] = a.copy(); out[f"n{what}_{k}"] = cnt
        c = rs.counts()
        out[f"counts_{k}"] = np.array([c[q] for q in RefSystem.COUNTS], np.int32)
    for which, name in ((0, "sta"), (1, "dyn")):
        off, fr_, ft_, ob_ = rs.tracks(bool(which))
        out[f"tr_{name}_off"] = off; out[f"tr_{name}_frame"] = fr_; out[f"tr_{name}_feat"] = ft_
        if ob_ is not None:
            out[f"tr_{name}_obj"] = ob_
    rs.close()
    np.savez(out_npz, **out)


def run_sequence_in_subprocess(settings, frames, tmp_dir, n_images=1 << 30, labels=(1, 2, 3, 4, 5, 6, 7, 8), full=False):
    """frames: list of dict(gray, depth_raw, flow, mask); full: libref_full.so - the reference's REAL Optimizer.cc + g2o behind Track().  Returns the npz the worker wrote (per frame: T_k, depth_k, mask_k, s{what}_k / n{what}_k in
    the layouts of RefSystem.state, counts_k; tracklets at the end)."""
    import subprocess
    import sys
    fin = os.path.join(str(tmp_dir), "ref_frames.npz"); fout = os.path.join(str(tmp_dir), "ref_out.npz")
    d = {"n": len(frames)}
    for k, fr in enumerate(frames):
        for q in ("gray", "depth_raw", "flow", "mask"):
            d[f"{q}_{k}"] = np.ascontiguousarray(fr[q])
    np.savez(fin, **d)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"import sys; sys.path.insert(0, {root!r}); from tests.ref_track import worker_main; worker_main({str(settings)!r}, {fin!r}, {fout!r}, {int(n_images)}, {tuple(labels)!r}, {bool(full)!r})"
    subprocess.run([sys.executable, "-c", code], check=True, stdout=subprocess.DEVNULL, cwd=root)
    return np.load(fout)


def brackets_worker_main(settings, frames_npz, n_images=1 << 30):
    """child process of bench.py's five-bracket leg: the reference's own all_timing (src/Tracking.cc) over the stored frames, as one JSON line"""
    import json
    import time
    z = np.load(frames_npz)
    n = int(z["n"])
    rs = RefSystem(settings)
    acc = np.zeros(5); nfr = 0
    t0 = time.perf_counter()
    for k in range(n):
        fr = {q: z[f"{q}_{k}"] for q in ("gray", "depth_raw", "flow", "mask")}
        rs.track(fr, k, n_images=n_images)
        if k >= 1:
            tm = rs.timing_ms()
            acc += np.array([tm[q] for q in ("mask_update", "camera_estimate", "object_tracking", "object_estimate", "map_update")]); nfr += 1
    dt = time.perf_counter() - t0
    rs.close()
    print(json.dumps({"ms": [float(v) / max(nfr, 1) for v in acc], "tracked_frames": nfr, "frames_per_s": n / dt}))


# ---- long sequences: frames read one by one from a directory of f{k}.npz files (vdo_slam_amd/synth_seq.render_bench_sequence), depth / mask kept as digests ----
def _digest(a):
    import hashlib
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


MAP_PARTS = ("cam_pose", "cam_pose_rf", "rigid_motion", "rigid_motion_rf", "rm_label", "points_sta", "points_dyn", "rm_count")


def dir_worker_main(settings, frames_dir, n, out_npz, n_images, labels, full=True, keep_images=False, export_map=False):
    from vdo_slam_amd.synth_seq import load_bench_frame
    rs = RefSystem(settings, full=full)
    out = {"n": n}
    import time
    t0 = time.perf_counter()
    for k in range(n):
        fr = load_bench_frame(frames_dir, k)
        T, depth, mask = rs.track(fr, k, n_images=n_images, labels=labels)
        out[f"T_{k}"] = T; out[f"depth_sha_{k}"] = _digest(depth); out[f"mask_sha_{k}"] = _digest(mask)
        if keep_images:
            out[f"depth_{k}"] = depth; out[f"mask_{k}"] = mask
        for what, rows in ((0, 10), (1, 12), (2, 19), (3, 8), (4, 17)):
            cnt, a = rs.state(what, rows)
            out[f"s{what}_{k}"] = a.copy(); out[f"n{what}_{k}"] = cnt
        c = rs.counts()
        out[f"counts_{k}"] = np.array([c[q] for q in RefSystem.COUNTS], np.int32)
    out["seconds"] = time.perf_counter() - t0
    if export_map:                                   # what the windowed / final batch optimisation left in the Map
        for what, name in enumerate(MAP_PARTS):
            out["map_" + name] = rs.map_export(what)
    for which, name in ((0, "sta"), (1, "dyn")):
        off, fr_, ft_, ob_ = rs.tracks(bool(which))
        out[f"tr_{name}_off"] = off; out[f"tr_{name}_frame"] = fr_; out[f"tr_{name}_feat"] = ft_
        if ob_ is not None:
            out[f"tr_{name}_obj"] = ob_
    rs.close()
    tmp = out_npz + ".tmp.npz"
    np.savez(tmp, **out)
    os.replace(tmp, out_npz)


def start_sequence_from_dir(settings, frames_dir, n, out_npz, n_images=1 << 30, labels=(1, 2, 3, 4, 5), full=True, keep_images=False, export_map=False):
    """The reference (child process, CPU) over the n frames of frames_dir; returns the Popen - the caller overlaps it with GPU work and then calls finish_sequence."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); from tests.ref_track import dir_worker_main; "
            f"dir_worker_main({str(settings)!r}, {str(frames_dir)!r}, {int(n)}, {str(out_npz)!r}, {int(n_images)}, {tuple(labels)!r}, {bool(full)!r}, {bool(keep_images)!r}, {bool(export_map)!r})")
    return subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.DEVNULL, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="1"))


def finish_sequence(proc, out_npz, timeout_s=600):
    if proc.wait(timeout=timeout_s) != 0:
        raise RuntimeError("the reference's TrackRGBD sequence (oracle/_ref) failed in its child process")
    return np.load(out_npz)
