"""Non-joint per-frame pose refinement (Optimizer::PoseOptimizationNew / PoseOptimizationObjMot,
reference src/Optimizer.cc:2177-2331, 2544-2753): ctypes veneer over vdo_pose_* + synthetic problems."""
import ctypes as C
import dataclasses

import numpy as np

from . import _capi as K
from .synth import KITTI_H, KITTI_K, KITTI_W, _mat4, rotvec_to_R


class PoseProblemC(C.Structure):
    _fields_ = [("n", C.c_int32), ("kind", C.c_int32), ("obs", K.c_double_p), ("Xw", K.c_double_p),
                ("K", C.c_double * 4), ("P", C.c_double * 12), ("T0", C.c_double * 16),
                ("huber_delta", C.c_double), ("chi2_gate", C.c_double), ("max_iterations", C.c_int32), ("pad", C.c_int32)]


@dataclasses.dataclass
class PoseProblem:
    kind: int               # 0 camera (EdgeSE3ProjectXYZOnlyPose) ; 1 object motion (EdgeSE3ProjectXYZOnlyObjMotion)
    obs: np.ndarray         # [n,2]
    Xw: np.ndarray          # [n,3]
    K: tuple
    P: np.ndarray           # [3,4]
    T0: np.ndarray          # [4,4]
    huber_delta: float
    chi2_gate: float = float(np.float32(0.01))
    max_iterations: int = 100
    T_true: np.ndarray | None = None

    @property
    def n(self): return self.obs.shape[0]


def to_c(p: PoseProblem):
    obs = np.ascontiguousarray(p.obs, dtype=np.float64); xw = np.ascontiguousarray(p.Xw, dtype=np.float64)
    s = PoseProblemC()
    s.n = p.n; s.kind = p.kind; s.obs = K._dp(obs); s.Xw = K._dp(xw)
    s.K = (C.c_double * 4)(*p.K)
    s.P = (C.c_double * 12)(*np.asarray(p.P, dtype=np.float64).ravel())
    s.T0 = (C.c_double * 16)(*np.asarray(p.T0, dtype=np.float64).ravel())
    s.huber_delta = p.huber_delta; s.chi2_gate = p.chi2_gate; s.max_iterations = p.max_iterations
    return s, [obs, xw]


def make_pose_problem(n=1200, seed=1, kind=0, outlier_frac=0.1, pix_sigma=0.05, init_sigma_t=0.05, init_sigma_r=0.004) -> PoseProblem:
    """kind 0: world points seen by a camera at T_cw ; kind 1: object points moved by H, seen through P = K*T_cw."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = KITTI_K
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    T_cw = f32(_mat4(rotvec_to_R(rng.normal(0, 0.05, 3)), rng.normal(0, 2.0, 3)))
    if kind == 0:
        u = rng.uniform(5, KITTI_W - 5, n); v = rng.uniform(5, KITTI_H - 5, n); z = rng.uniform(4, 40, n)
    else:
        u = rng.uniform(500, 700, n); v = rng.uniform(150, 260, n); z = rng.uniform(8, 20, n)
    Xc = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1)
    Twc = np.linalg.inv(T_cw)
    Xw = f32(Xc @ Twc[:3, :3].T + Twc[:3, 3])               # UnprojectStereo* returns float
    KK = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0]], np.float64)
    if kind == 0:
        dT = _mat4(rotvec_to_R(np.array([0.0, rng.uniform(-0.01, 0.01), 0.0])), np.array([rng.normal(0, 0.02), rng.normal(0, 0.01), -0.8]))
        T_true = dT @ T_cw
        Xn = Xw @ T_true[:3, :3].T + T_true[:3, 3]
        proj = np.stack([Xn[:, 0] / Xn[:, 2] * fx + cx, Xn[:, 1] / Xn[:, 2] * fy + cy], 1)
        P = np.zeros((3, 4))
    else:
        c = Xw.mean(0)
        Hl = _mat4(rotvec_to_R(np.array([0, rng.uniform(-0.03, 0.03), 0])), np.array([rng.normal(0, 0.1), 0, rng.uniform(0.2, 0.8)]))
        T_true = _mat4(np.eye(3), c) @ Hl @ _mat4(np.eye(3), -c)        # world-frame motion about the object centre
        P = KK @ T_cw
        Xn = Xw @ T_true[:3, :3].T + T_true[:3, 3]
        m = Xn @ P[:, :3].T + P[:, 3]
        proj = m[:, :2] / m[:, 2:3]
    obs = proj + rng.normal(0, pix_sigma, (n, 2))
    outl = rng.random(n) < outlier_frac
    obs[outl] += rng.normal(0, 5.0, (int(outl.sum()), 2))
    T0 = T_true.copy()
    T0[:3, :3] = rotvec_to_R(rng.normal(0, init_sigma_r, 3)) @ T0[:3, :3]
    T0[:3, 3] += rng.normal(0, init_sigma_t, 3)
    return PoseProblem(kind=kind, obs=f32(obs), Xw=Xw, K=KITTI_K, P=P, T0=f32(T0),
                       huber_delta=float(np.sqrt(np.float32(0.01))) if kind == 0 else 0.0,
                       max_iterations=100 if kind == 0 else 200, T_true=T_true)


def _bind():
    L = K.lib()
    L.vdo_pose_batch_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(PoseProblemC), C.POINTER(C.c_void_p)]
    L.vdo_pose_batch_run.argtypes = [C.c_void_p]
    L.vdo_pose_batch_fetch.argtypes = [C.c_void_p, C.POINTER(K.Flow2ResultC), C.POINTER(K.c_uint8_p)]
    L.vdo_pose_batch_destroy.argtypes = [C.c_void_p]
    return L


class PoseBatch:
    def __init__(self, ctx, problems):
        self.ctx = ctx
        self.problems = list(problems)
        self._keep = []
        arr = (PoseProblemC * len(self.problems))()
        for i, p in enumerate(self.problems):
            arr[i], keep = to_c(p)
            self._keep.append(keep)
        self._h = C.c_void_p()
        K.check(_bind().vdo_pose_batch_create(ctx._h, len(self.problems), arr, C.byref(self._h)))

    def run(self):
        K.check(K.lib().vdo_pose_batch_run(self._h))

    def fetch(self):
        n = len(self.problems)
        res = (K.Flow2ResultC * n)()
        inl = [np.zeros(max(p.n, 1), np.uint8) for p in self.problems]
        ip = (K.c_uint8_p * n)(*[a.ctypes.data_as(K.c_uint8_p) for a in inl])
        K.check(K.lib().vdo_pose_batch_fetch(self._h, res, ip))
        return [dict(T=np.array(res[i].T).reshape(4, 4), n_inliers=res[i].n_inliers, iterations=res[i].iterations, trials=res[i].trials,
                     stop_reason=res[i].stop_reason, initial_chi2=res[i].initial_chi2, final_chi2=res[i].final_chi2,
                     final_lambda=res[i].final_lambda, inliers=inl[i][:self.problems[i].n]) for i in range(n)]

    def close(self):
        if self._h:
            K.lib().vdo_pose_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass
