"""The per-frame record of tests/ref_track.dir_worker_main (what tests/bench_parity.compare reads) taken from the oracle-composed Track() (tests/pipeline_ref.py):
lets the CPU suite compare the oracle pipeline with the whole reference over a long sequence with the same comparison the GPU test applies to the product."""
import numpy as np

from tests.ref_track import _digest

f32 = np.float32


def record_frame(out, k, ora, exp, depth_sha=None):
    L = ora.last
    out[f"T_{k}"] = ora.Tl.copy()
    out[f"mask_sha_{k}"] = _digest(L["mask"])
    if depth_sha is not None:
        out[f"depth_sha_{k}"] = depth_sha
    st = np.stack([np.asarray(L["st"][q], f32) for q in ("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y", "depth")] + list(np.asarray(L["st"]["xyz"], f32).reshape(-1, 3).T))
    out[f"s0_{k}"] = st.ravel(); out[f"n0_{k}"] = st.shape[1]
    n = L["ob"]["key_x"].size
    lab2 = np.asarray(ora.result["objects"]["obj_label"], f32) if k > 0 else None
    ob = np.stack([np.asarray(L["ob"][q], f32) for q in ("key_x", "key_y", "corr_x", "corr_y", "flow_x", "flow_y", "depth")] + list(np.asarray(L["ob"]["xyz"], f32).reshape(-1, 3).T) +
                  [np.asarray(L["ob"]["label"], f32), lab2 if lab2 is not None else np.zeros(n, f32)])
    out[f"s1_{k}"] = ob.ravel(); out[f"n1_{k}"] = n
    no = len(L["sem_pos"])
    po = np.zeros((no, 19), f32)
    for a in range(no):
        po[a, 0] = L["sem_pos"][a]; po[a, 1] = L["mod"][a]; po[a, 2] = 1.0 if L["stat"][a] else 0.0
        po[a, 3:] = (np.asarray(L["H"][a], f32) if L["stat"][a] else np.eye(4, dtype=f32)).ravel()
    out[f"s2_{k}"] = po.ravel(); out[f"n2_{k}"] = no
    out[f"n3_{k}"] = exp["n_object_samples"]
    sc = np.zeros(17, f32); sc[0] = ora.max_id; sc[1:] = ora.Tl.ravel()
    out[f"s4_{k}"] = sc; out[f"n4_{k}"] = 1
