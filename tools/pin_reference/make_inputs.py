"""Writes the committed inputs of the pin harness into tests/golden/inputs/ (seeded, from this repository's own generators) and
knows the raw layouts of both directions (tests/test_golden.py imports the readers).  Run from the repository root:
    python tools/pin_reference/make_inputs.py
Little-endian; int32 / float32 / float64 as stated in tests/golden/README.md."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
IN_DIR = os.path.join(ROOT, "tests", "golden", "inputs")
ORB_W, ORB_H = 640, 200
PNP_CASES = ((600, 0.3, 7), (150, 0.6, 8))                  # (points, outlier fraction, seed)
FLOW2_CASES = ((600, 0, 3), (300, 1, 4))                    # (correspondences, is_object, seed)


def write_pnp(path, K4, X, uv):
    with open(path, "wb") as f:
        np.int32(X.shape[0]).tofile(f); np.asarray(K4, np.float32).tofile(f)
        np.ascontiguousarray(X, np.float32).tofile(f); np.ascontiguousarray(uv, np.float32).tofile(f)


def read_pnp(path):
    b = open(path, "rb").read()
    n = int(np.frombuffer(b, np.int32, 1)[0]); o = 4
    K4 = np.frombuffer(b, np.float32, 4, o); o += 16
    X = np.frombuffer(b, np.float32, 3 * n, o).reshape(n, 3); o += 12 * n
    uv = np.frombuffer(b, np.float32, 2 * n, o).reshape(n, 2)
    return K4, X, uv


def read_pnp_out(path):
    b = open(path, "rb").read()
    R = np.frombuffer(b, np.float64, 9).reshape(3, 3); t = np.frombuffer(b, np.float64, 3, 72)
    m = int(np.frombuffer(b, np.int32, 1, 96)[0])
    return R, t, np.frombuffer(b, np.int32, m, 100)


def write_flow2(path, is_object, K4, Tcw_last, T0, key, depth, flow):
    with open(path, "wb") as f:
        np.array([key.shape[0], is_object], np.int32).tofile(f); np.asarray(K4, np.float32).tofile(f)
        np.ascontiguousarray(Tcw_last, np.float32).tofile(f); np.ascontiguousarray(T0, np.float32).tofile(f)
        np.ascontiguousarray(key, np.float32).tofile(f); np.ascontiguousarray(depth, np.float32).tofile(f); np.ascontiguousarray(flow, np.float32).tofile(f)


def read_flow2(path):
    b = open(path, "rb").read()
    n, is_object = (int(v) for v in np.frombuffer(b, np.int32, 2)); o = 8
    K4 = np.frombuffer(b, np.float32, 4, o); o += 16
    Tl = np.frombuffer(b, np.float32, 16, o).reshape(4, 4); o += 64
    T0 = np.frombuffer(b, np.float32, 16, o).reshape(4, 4); o += 64
    key = np.frombuffer(b, np.float32, 2 * n, o).reshape(n, 2); o += 8 * n
    depth = np.frombuffer(b, np.float32, n, o); o += 4 * n
    flow = np.frombuffer(b, np.float32, 2 * n, o).reshape(n, 2)
    return dict(n=n, is_object=is_object, K4=K4, Tcw_last=Tl, T0=T0, key=key, depth=depth, flow=flow)


def read_flow2_out(path, n):
    b = open(path, "rb").read()
    good = int(np.frombuffer(b, np.int32, 1)[0])
    pose = np.frombuffer(b, np.float32, 16, 4).reshape(4, 4)
    inl = np.frombuffer(b, np.int32, n, 68)
    keys = np.frombuffer(b, np.float32, 2 * n, 68 + 4 * n).reshape(n, 2)
    return good, pose, inl, keys


def write_map(path, m):
    """m: the dict of tests/map_builder_ref.make_map."""
    with open(path, "wb") as f:
        F = m["n_frames"]
        np.int32(F).tofile(f); np.ascontiguousarray(m["K"], np.float32).tofile(f)
        for i in range(F):
            np.ascontiguousarray(m["cam_pose"][i], np.float32).tofile(f)
            for pre in ("sta", "dyn"):
                fe = m["feats"][i]
                k = len(fe[pre + "_uv"])
                np.int32(k).tofile(f)
                np.asarray(fe[pre + "_uv"], np.float32).reshape(k, 2).tofile(f); np.asarray(fe[pre + "_d"], np.float32).reshape(k).tofile(f)
                np.asarray(fe[pre + "_xw"], np.float32).reshape(k, 3).tofile(f)
        for which in ("tr_sta", "tr_dyn"):
            np.int32(len(m[which])).tofile(f)
            for tr in m[which]:
                np.int32(len(tr)).tofile(f); np.asarray(tr, np.int32).reshape(len(tr), 2).tofile(f)
            if which == "tr_dyn":
                np.asarray(m["obj_of_dyn"], np.int32).tofile(f)
        for i in range(F - 1):
            nm = len(m["rigid_motion"][i])
            np.int32(nm).tofile(f); np.ascontiguousarray(m["rigid_motion"][i], np.float32).tofile(f); np.asarray(m["rm_label"][i], np.int32).tofile(f)


def read_batch_out(path):
    b = open(path, "rb").read()
    F = int(np.frombuffer(b, np.int32, 1)[0]); o = 4
    cams = np.frombuffer(b, np.float32, 16 * F, o).reshape(F, 4, 4); o += 64 * F
    mots = []
    for _ in range(F - 1):
        n = int(np.frombuffer(b, np.int32, 1, o)[0]); o += 4
        mots.append(np.frombuffer(b, np.float32, 16 * n, o).reshape(n, 4, 4)); o += 64 * n
    return cams, mots


def golden_map():
    from tests.map_builder_ref import make_map
    return make_map(n_frames=8, n_static=150, n_objects=2, dyn_tracks_per_object=20, seed=5)


def main():
    from tests.test_oracle_p3p import _scene
    from vdo_slam_amd import synth, synth_frames as SF
    os.makedirs(IN_DIR, exist_ok=True)
    SF.make_gray(11, ORB_W, ORB_H).tofile(os.path.join(IN_DIR, f"orb_gray_{ORB_W}x{ORB_H}.u8"))
    for c, (n, outl, seed) in enumerate(PNP_CASES):
        Xw, uv, R, t, _ = _scene(np.random.default_rng(seed), n, outl, pix_sigma=0.1)
        write_pnp(os.path.join(IN_DIR, f"pnp_case{c}.bin"), synth.KITTI_K, Xw, uv)
    for c, (n, is_object, seed) in enumerate(FLOW2_CASES):
        p = synth.make_flow2_problem(n, seed=seed, is_object=bool(is_object))
        Twl = p.Twl.astype(np.float32)
        Tcw_last = np.eye(4, dtype=np.float32); Tcw_last[:3, :3] = Twl[:3, :3].T; Tcw_last[:3, 3] = -(Twl[:3, :3].T @ Twl[:3, 3])
        write_flow2(os.path.join(IN_DIR, f"flow2_case{c}.bin"), is_object, p.K, Tcw_last, p.T0, p.obs, p.depth, p.flow)
    write_map(os.path.join(IN_DIR, "batch_map_case0.bin"), golden_map())
    print("inputs written to", IN_DIR)


if __name__ == "__main__":
    main()
