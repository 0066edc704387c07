"""RANSAC initialiser of the per-frame pose problems (cv::solvePnPRansac(AP3P) as GetInitModelCam/Obj call it,
reference src/Tracking.cc:1614-1849): ctypes veneer over vdo_pnp_ransac_batch."""
import ctypes as C

import numpy as np

from . import _capi as K


class PnpProblemC(C.Structure):
    _fields_ = [("n", C.c_int32), ("X", K.c_double_p), ("uv", K.c_double_p), ("K", C.c_double * 4),
                ("max_iterations", C.c_int32), ("reproj_threshold", C.c_double), ("confidence", C.c_double), ("refit", C.c_int32)]


class PnpResultC(C.Structure):
    _fields_ = [("T", C.c_double * 16), ("n_inliers", C.c_int32), ("iterations_run", C.c_int32), ("best_iteration", C.c_int32)]


def pnp_ransac_batch(ctx, problems, K4, max_iterations=500, thr=0.4, confidence=0.98, refit=False, solver="ap3p"):
    """problems: list of (X [n,3], uv [n,2]).  refit: OpenCV's final EPnP re-estimation of the winning model on its inliers.  solver: "ap3p" (what the
    reference's calls name, the default) or "grunert" (rounds 1-4).  Returns a list of dict(T, n_inliers, iterations_run, best_iteration, inliers)."""
    flags = int(bool(refit)) | (2 if solver == "grunert" else 0)
    n = len(problems)
    arr = (PnpProblemC * n)()
    keep, inl = [], []
    for i, (X, uv) in enumerate(problems):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3); uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        keep += [X, uv]
        arr[i] = PnpProblemC(X.shape[0], K._dp(X), K._dp(uv), (C.c_double * 4)(*K4), max_iterations, thr, confidence, flags)
        inl.append(np.zeros(max(X.shape[0], 1), np.uint8))
    res = (PnpResultC * n)()
    ip = (K.c_uint8_p * n)(*[a.ctypes.data_as(K.c_uint8_p) for a in inl])
    L = K.lib()
    L.vdo_pnp_ransac_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(PnpProblemC), C.POINTER(PnpResultC), C.POINTER(K.c_uint8_p)]
    K.check(L.vdo_pnp_ransac_batch(ctx._h, n, arr, res, ip))
    return [dict(T=np.array(res[i].T).reshape(4, 4), n_inliers=res[i].n_inliers, iterations_run=res[i].iterations_run,
                 best_iteration=res[i].best_iteration, inliers=inl[i][:keep[2 * i].shape[0]]) for i in range(n)]
