"""Host-side mirror of the per-frame joint pose+flow optimisers (thin ctypes wrapper).

``Flow2Batch`` corresponds to one call of ``Optimizer::PoseOptimizationFlow2Cam`` (camera) or
to the per-object loop over ``Optimizer::PoseOptimizationFlow2`` in ``Tracking::Track``
(reference src/Tracking.cc:697, 785-1001): all problems of a batch run in ONE kernel launch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K


class Flow2Batch:
    def __init__(self, ctx, problems):
        self.ctx = ctx
        self.problems = list(problems)
        self._keep = []
        arr = (K.Flow2ProblemC * len(self.problems))()
        for i, p in enumerate(self.problems):
            s, keep = K.flow2_to_c(p)
            arr[i] = s
            self._keep.append(keep)
        self._arr = arr
        self._h = C.c_void_p()
        K.check(K.lib().vdo_flow2_batch_create(ctx._h, len(self.problems), arr, C.byref(self._h)))

    def run(self):
        K.check(K.lib().vdo_flow2_batch_run(self._h))

    def fetch(self):
        n = len(self.problems)
        res = (K.Flow2ResultC * n)()
        flows = [np.zeros((p.n, 2)) for p in self.problems]
        inl = [np.zeros(p.n, np.uint8) for p in self.problems]
        fp = (K.c_double_p * n)(*[K._dp(f) for f in flows])
        ip = (K.c_uint8_p * n)(*[a.ctypes.data_as(K.c_uint8_p) for a in inl])
        K.check(K.lib().vdo_flow2_batch_fetch(self._h, res, fp, ip))
        out = []
        for i in range(n):
            out.append(dict(T=np.array(res[i].T).reshape(4, 4), n_inliers=res[i].n_inliers, iterations=res[i].iterations,
                            trials=res[i].trials, stop_reason=res[i].stop_reason, initial_chi2=res[i].initial_chi2,
                            final_chi2=res[i].final_chi2, final_lambda=res[i].final_lambda, flow=flows[i], inliers=inl[i]))
        return out

    def close(self):
        if self._h:
            K.lib().vdo_flow2_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
