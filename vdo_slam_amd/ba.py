"""Host-side handle for the batch bundle-adjustment path (thin ctypes mirror of the C-ABI).

Mirrors the role of ``Optimizer::FullBatchOptimization`` / ``PartialBatchOptimization``
(reference src/Optimizer.cc:1232-2175, 42-1230): a graph goes in, refined camera poses /
object motions / points come out.  All compute happens in libvdo_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K


class Context:
    def __init__(self, device: int = 0, stream: int | None = None, cu_mask: tuple | None = None):
        """``cu_mask=(first, count, invert)``: own stream restricted to (or excluding) a range of compute units."""
        self._h = C.c_void_p()
        if cu_mask is not None:
            L = K.lib()
            L.vdo_ctx_create_cu_mask.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
            K.check(L.vdo_ctx_create_cu_mask(device, cu_mask[0], cu_mask[1], int(cu_mask[2]), C.byref(self._h)))
        else:
            K.check(K.lib().vdo_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = device

    @property
    def stream_ptr(self) -> int:
        """The hipStream_t every call of this context is ordered on (for ExternalStream wrappers)."""
        out = C.c_void_p()
        L = K.lib()
        L.vdo_ctx_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        K.check(L.vdo_ctx_stream(self._h, C.byref(out)))
        return out.value or 0

    def synchronize(self):
        K.check(K.lib().vdo_ctx_synchronize(self._h))

    # Objects built on a context (pipelines, batches, extractors) retain it: when a garbage-collected reference cycle (e.g. the frames
    # of a failed test) finalises the context BEFORE its users - Python gives no order inside a cycle - the destroy is postponed until
    # the last user has let go, instead of pulling the stream and the arenas from under a live pipeline.
    _users = 0
    _close_pending = False

    def _retain(self):
        self._users += 1

    def _release(self):
        self._users -= 1
        if self._users <= 0 and self._close_pending:
            self._close_pending = False
            self.close()

    def close(self):
        if self._users > 0:
            self._close_pending = True
            return
        if self._h:
            K.lib().vdo_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchBA:
    def __init__(self, ctx: Context, graph):
        self.ctx = ctx
        self.graph = graph
        self._gc, self._keep = K.graph_to_c(graph)
        self._h = C.c_void_p()
        K.check(K.lib().vdo_ba_create(ctx._h, C.byref(self._gc), C.byref(self._h)))
        ctx._retain()

    def linearize(self, repeat: int = 1, timed: bool = False):
        """Run the linearisation sweep; with ``timed`` returns the mean ms of the K18 kernel."""
        ms = C.c_float(0.0)
        K.check(K.lib().vdo_ba_linearize(self._h, repeat, C.byref(ms) if timed else None))
        return ms.value if timed else None

    def profile_schur(self, repeat=100):
        """mean ms of one Schur mat-vec launch (k_schur_tile<0>) - vdo_ba_profile_schur; call after optimize()."""
        L = K.lib()
        ms = C.c_float()
        L.vdo_ba_profile_schur.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        K.check(L.vdo_ba_profile_schur(self._h, repeat, C.byref(ms)))
        return float(ms.value)

    def profile_linearize(self, repeat: int = 10):
        """(ms of the sweep kernel alone, ms of a whole linearisation, layout dims) - vdo_ba_profile_linearize."""
        ms = (C.c_float * 2)(); dims = (C.c_int64 * 8)()
        L = K.lib()
        L.vdo_ba_profile_linearize.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int64)]
        K.check(L.vdo_ba_profile_linearize(self._h, repeat, ms, dims))
        return float(ms[0]), float(ms[1]), dict(zip(("tiles", "slots", "partial_row", "max_slots", "read_bytes_eb", "read_bytes_et", "eb_entries", "hubs"), (int(v) for v in dims)))

    def dims(self) -> dict:
        """Layout facts of the tiled graph (tests): tiles, (tile, slot) pairs, ps_stride (= partial_row), max_slots, bytes read per edge."""
        d = self.profile_linearize(1)[2]
        d["ps_stride"] = d["partial_row"]
        return d

    def system(self) -> K.BASystem:
        S = K.BASystem(self.graph)
        K.check(K.lib().vdo_ba_download_system(self._h, C.byref(S.c)))
        return S

    def optimize(self, max_iterations=300, gain_threshold=1e-4, verbose=0, pcg_tolerance=0.0, pcg_max_iterations=0, solver=0):
        """solver: 0 auto, 2 Schur + chain-preconditioned PCG, 3 Schur + dense MFMA Cholesky of the reduced-camera matrix."""
        opt = K.LMOptionsC(max_iterations, gain_threshold, verbose, solver, pcg_tolerance, pcg_max_iterations)
        st = K.LMStatsC()
        K.check(K.lib().vdo_ba_optimize(self._h, C.byref(opt), C.byref(st)))
        return st

    def estimates(self):
        pose = np.zeros((self.graph.n_pose, 12)); point = np.zeros((self.graph.n_point, 3))
        K.check(K.lib().vdo_ba_get_estimates(self._h, K._dp(pose), K._dp(point)))
        return pose, point

    def set_estimates(self, pose, point):
        pose = np.ascontiguousarray(pose, dtype=np.float64); point = np.ascontiguousarray(point, dtype=np.float64)
        K.check(K.lib().vdo_ba_set_estimates(self._h, K._dp(pose), K._dp(point)))

    def close(self):
        if self._h:
            K.lib().vdo_ba_destroy(self._h)
            self._h = C.c_void_p()
            self.ctx._release()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def linearize_byte_model(graph, dims):
    """HBM bytes one linearisation has to move with this design - the floor the counters are compared with.
    sweep: the edge inputs of every ENTRY of the tiles' padded edge blocks (a few percent more than the edges) + every point once + 48 B of descriptor per
    tile + pose id and row id per (tile, pose-slot) pair (reads); we per edge, the point-point block of a ternary edge, the landmark scalar +
    right-hand side per point, one row of running sums per (tile, pose-slot) pair (writes).  finalize: the rows again (read) + the
    6x6 block and right-hand side of every pose (write)."""
    row = 8 * dims["partial_row"]
    entries = dims.get("eb_entries") or graph.n_eb
    sweep_r = dims["read_bytes_eb"] * entries + dims["read_bytes_et"] * graph.n_et + 24 * graph.n_point + 48 * dims["tiles"] + 8 * dims["slots"]
    sweep_w = 8 * (graph.n_eb + graph.n_et) + 72 * graph.n_et + 32 * graph.n_point + row * dims["slots"]
    fin = row * dims["slots"] + 336 * graph.n_pose
    return dict(sweep_read=int(sweep_r), sweep_write=int(sweep_w), sweep=int(sweep_r + sweep_w), finalize=int(fin), linearize=int(sweep_r + sweep_w + fin))


def schur_byte_model(graph, dims, n_static):
    """HBM bytes ONE Schur mat-vec launch (k_schur_tile<0>: part_q = B Hll^-1 B^T p, csrc/ba_solve.hip) has to move with the matrix-free design: per incidence of the
    tiles' padded edge blocks and per ternary incidence the slot key + the scalar `we` the sweep stored (12 B - the 6x3 block is RECOMPUTED from the point and the slot's
    pose); per point its linearisation point + landmark scalar + chain range (40 B), per point of a dynamic track its two 3x3 chain factors (144 B); per (tile, pose-slot)
    pair pose id, row id, the pose (96 B) and the two direction vectors (96 B); 48 B of descriptor per tile (reads); one row of 8 doubles per pair (write).
    `stored_hpl`: what SURVEY 8d's formula prices - a design that keeps the 6x3 pose-landmark blocks: 144 B per EdgeSE3PointXYZ, 360 B per ternary edge."""
    entries = dims.get("eb_entries") or graph.n_eb
    n_dyn = max(0, graph.n_point - int(n_static))
    rd = 12 * (entries + 2 * graph.n_et) + 40 * graph.n_point + 144 * n_dyn + 48 * dims["tiles"] + 200 * dims["slots"]
    wr = 64 * dims["slots"]
    return dict(read=int(rd), write=int(wr), matvec=int(rd + wr), stored_hpl=int(144 * graph.n_eb + 360 * graph.n_et))
