#!/usr/bin/env python
"""bench.py — headline measurement of the hot path on MI355X (driver contract in the task brief).

A *step* is one pass of the hot path over one batch of synthetic input that is already
resident in HBM when the timed region starts:
  * per-frame leg (BASELINE.json configs[1], the metric's config): one KITTI-0000-shaped frame
    (1242x375, ~2.5k ORB features, ~1.2k static + ~5 objects) through ORB + flow propagation +
    per-frame joint pose/flow LM  -> frames/sec;
  * batch leg (reported in the same JSON line): one Levenberg–Marquardt outer iteration of the
    KITTI-shaped full-batch dynamic factor graph -> ms_per_lm_iter, and the `roofline` object for
    the dominant kernel of that leg, the per-edge Jacobian sweep (SURVEY.md §8d B_sweep formula).
With N>1 ranks (torchrun) every rank processes its own shard/replica with no data-path
collective ("weak" scaling); value = units over all ranks / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 achievable)


def _dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def cpu_baseline_batch(graph, seconds_budget=20.0):
    """Oracle (CPU restatement of the reference algorithms, 1 thread) on a bounded sample of
    the same workload: LM outer iterations of the same graph until ~seconds_budget."""
    from tests import oracle_lib
    from vdo_slam_amd import _capi as K
    o = oracle_lib.load()
    gc, keep = K.graph_to_c(graph)
    # time one linearisation sweep (errors + Jacobians + accumulation) and a few LM iterations
    S = K.BASystem(graph)
    t0 = time.perf_counter()
    reps = 0
    while True:
        o.vdo_oracle_ba_linearize(C.byref(gc), C.byref(S.c))
        reps += 1
        if time.perf_counter() - t0 > 2.0 or reps >= 20:
            break
    sweep_ms = (time.perf_counter() - t0) / reps * 1e3
    its = 3
    opt = K.LMOptionsC(its, -1.0, 0, 0, 0.0, 0)
    st = K.LMStatsC()
    pose = np.zeros_like(graph.pose); point = np.zeros_like(graph.point)
    t0 = time.perf_counter()
    o.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), K._dp(pose), K._dp(point), C.byref(st))
    lm_ms = (time.perf_counter() - t0) * 1e3 / max(1, st.iterations)
    return {"sweep_ms": sweep_ms, "ms_per_lm_iter": lm_ms, "iterations": int(st.iterations)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=60, help="frames of the batch graph")
    ap.add_argument("--static", type=int, default=30000, help="static landmarks of the batch graph")
    ap.add_argument("--objects", type=int, default=5)
    ap.add_argument("--dyn-tracks", type=int, default=800, help="dynamic tracks per object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank, world, local = _dist_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in libvdo_hip)")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from vdo_slam_amd import synth
    from vdo_slam_amd.ba import BatchBA, Context

    ctx = Context(local, torch.cuda.current_stream().cuda_stream)
    g = synth.make_ba_graph(args.frames, args.static, args.objects, args.dyn_tracks, seed=1 + rank)
    ba = BatchBA(ctx, g)
    pose0, point0 = g.pose.copy(), g.point.copy()

    # ---- roofline of the dominant kernel: per-edge Jacobian sweep (binary edges)
    ba.linearize()
    sweep_ms = ba.linearize(repeat=50, timed=True)
    bytes_sweep_eb = 208 * g.n_eb          # SURVEY §8d per-unit figure x units of one launch
    achieved = bytes_sweep_eb / (sweep_ms * 1e-3) / 1e9

    # ---- LM outer iterations (steps)
    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ba.set_estimates(pose0, point0)
    ba.optimize(max_iterations=args.warmup, gain_threshold=-1.0)
    ba.set_estimates(pose0, point0)
    barrier()
    t0 = time.perf_counter()
    st = ba.optimize(max_iterations=args.steps, gain_threshold=-1.0)
    barrier()
    dt = time.perf_counter() - t0
    steps_done = int(st.iterations)
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_iter = dt * 1e3 / max(1, steps_done)

    out = {
        "metric": "LM outer iterations/sec on KITTI-shaped full-batch factor graph (frames/sec leg pending)",
        "value": world * steps_done / dt,
        "unit": "lm_iter/s",
        "n_gpus": world, "steps": steps_done, "warmup": args.warmup,
        "ms_per_step": ms_per_iter, "ms_per_lm_iter": ms_per_iter,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic KITTI-shaped full-batch dynamic BA: {g.n_cam} frames, {g.n_pose} pose/motion vertices, "
                               f"{g.n_point} points, {g.n_eb} EdgeSE3PointXYZ, {g.n_et} ternary, {g.n_ep} EdgeSE3",
                   "parallelism": f"replicas x{world}", "lm_trials": int(st.total_trials)},
        "roofline": {"bound": "hbm", "kernel": "k_sweep_eb<true>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "bytes_per_launch": bytes_sweep_eb,
                     "avg_launch_ms": sweep_ms},
    }
    if rank == 0 and not args.no_cpu_baseline:
        cb = cpu_baseline_batch(g)
        out["cpu_baseline"] = {"value": 1e3 / cb["ms_per_lm_iter"], "unit": "lm_iter/s", "cores": 1, "kind": "port",
                               "sample": f"{cb['iterations']} LM outer iterations of the same graph (oracle, 1 thread); sweep {cb['sweep_ms']:.1f} ms",
                               "ms_per_lm_iter": cb["ms_per_lm_iter"], "sweep_ms": cb["sweep_ms"]}
    if rank == 0:
        print(json.dumps(out))
    ba.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
