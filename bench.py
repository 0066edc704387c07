#!/usr/bin/env python
"""bench.py — headline measurement of the hot path on MI355X (driver contract in the task brief).

A *step* is one KITTI-0000-shaped frame (1242x375) through the per-frame hot path with every
input already resident in HBM when the timed region starts:
    K1 depth preprocess -> ORB (pyramid, FAST cells, quadtree, IC angle, blur) -> K9 static filter
    -> K10 object sampling -> per-frame joint pose+flow LM for the camera (1200 matches)
    -> the same LM for the 5 objects of the frame (one launch).
This is BASELINE.json configs[1] ("KITTI seq 0000 on 1xMI355X: ORB+flow front-end and per-frame
PoseOptimization on GPU").  Stages of TrackRGBD not yet on the GPU path (P3P RANSAC initialiser,
RenewFrameInfo, UpdateMask, tracklets — SURVEY.md §8 "next") are outside the step on BOTH sides
(GPU and CPU baseline).  The same JSON line also carries
  * ms_per_lm_iter of the full-batch dynamic BA (configs[2] shape), and
  * `roofline` for the dominant kernel of that leg, the per-edge Jacobian sweep (K18), measured
    live with HIP events on the stream the kernel is launched on, on a graph large enough to be
    HBM-bound (config[4] shape scaled to one GPU).
N>1 ranks (torchrun): every rank runs its own replica (the per-frame path does not shard,
SURVEY.md §8e) -> "weak" scaling, no data-path collective; value = frames of all ranks / max time.
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

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
N_DISTINCT_FRAMES = 4   # synthetic frames cycled through the timed loop


def _pmc_traffic_bytes(graph):
    """HBM bytes per k_sweep_tile launch from the committed PMC summary, if it was taken on this very graph (else None)."""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_sweep_pmc_hbm_traffic.txt")
    try:
        txt = open(path).read()
        m = re.search(r"n_eb (\d+) n_et (\d+) n_point (\d+)", txt)
        t = re.search(r"= ([0-9.]+) MB\s*$", txt, re.M)
        if m and t and (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (graph.n_eb, graph.n_et, graph.n_point):
            return float(t.group(1)) * 1e6
    except OSError:
        pass
    return None


def _dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def make_frame_inputs(seed0):
    from vdo_slam_amd import synth, synth_frames as SF
    frames = [SF.make_frame(seed=seed0 + k) for k in range(N_DISTINCT_FRAMES)]
    cam = [synth.make_flow2_problem(1200, seed=seed0 + 100 + k) for k in range(N_DISTINCT_FRAMES)]
    obj = [[synth.make_flow2_problem(n, seed=seed0 + 200 + 10 * k + j, is_object=True) for j, n in enumerate([800, 600, 400, 300, 200])]
           for k in range(N_DISTINCT_FRAMES)]
    return frames, cam, obj


def cpu_baseline_frames(frames, cam, obj, budget_s=14.0):
    """Oracle (1 thread) on the same frames: the same stages, chained frame to frame like FramePipeline does
    (tests/pipeline_ref.py composes the oracle's functions; the LM oracle supplies pose and inliers)."""
    from tests import oracle_lib
    from tests.pipeline_ref import OraclePipeline
    from tests.test_oracle_flow2 import run_oracle
    o = oracle_lib.load()
    pipe = OraclePipeline(o)
    n = 0
    t_lm_cam = t_lm_obj = 0.0
    t0 = time.perf_counter()
    while True:
        k = n % len(frames)
        t = time.perf_counter()
        Tc, _, inl, _, _ = run_oracle(o, cam[k])
        t_lm_cam += time.perf_counter() - t
        pipe.step(frames[k], Tc.astype(np.float32), inl)
        t = time.perf_counter()
        for p in obj[k]:
            run_oracle(o, p)
        t_lm_obj += time.perf_counter() - t
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 40:
            break
    dt = time.perf_counter() - t0
    stage = dict(pipe.stage_s, lm_cam=t_lm_cam, lm_obj=t_lm_obj)
    return n / dt, n, {k2: v / n * 1e3 for k2, v in stage.items()}


def cpu_baseline_batch(graph, its=2):
    from tests import oracle_lib
    from vdo_slam_amd import _capi as K
    o = oracle_lib.load()
    gc, keep = K.graph_to_c(graph)
    S = K.BASystem(graph)
    t0 = time.perf_counter()
    o.vdo_oracle_ba_linearize(C.byref(gc), C.byref(S.c))
    sweep_ms = (time.perf_counter() - t0) * 1e3
    opt = K.LMOptionsC(its, -1.0, 0, 0, 0.0, 0)
    st = K.LMStatsC()
    pose = np.zeros_like(graph.pose); point = np.zeros_like(graph.point)
    t0 = time.perf_counter()
    o.vdo_oracle_ba_optimize(C.byref(gc), C.byref(opt), K._dp(pose), K._dp(point), C.byref(st))
    return (time.perf_counter() - t0) * 1e3 / max(1, st.iterations), sweep_ms, int(st.iterations)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the batch-BA / roofline legs")
    ap.add_argument("--roofline-static", type=int, default=600000, help="static landmarks of the roofline graph")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank, world, local = _dist_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (libvdo_hip has no CPU fallback)")
    torch.cuda.set_device(local)
    use_dist = world > 1 or bool(os.environ.get("VDO_BENCH_FORCE_DIST"))     # FORCE: exercise the RCCL legs on a 1-GPU box
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
    stream = torch.cuda.Stream()          # non-default stream shared by torch events and libvdo_hip
    torch.cuda.set_stream(stream)

    from vdo_slam_amd import synth, synth_frames as SF
    from vdo_slam_amd.ba import BatchBA, Context
    from vdo_slam_amd.flow2 import Flow2Batch
    from vdo_slam_amd.frontend import FrameImages, ORBextractor

    n_lm_cu = int(os.environ.get("VDO_BENCH_LM_CUS", "0"))      # measured: no gain from CU partitioning (the LM kernel is not slowed by its neighbours)
    if n_lm_cu > 0:
        # the LM chain gets CUs of its own (one latency-bound workgroup per problem); everything else runs on the other CUs
        ctx = Context(local, cu_mask=(0, n_lm_cu, True))
        ctx_lm = Context(local, cu_mask=(0, n_lm_cu, False))
        ctx_ba = Context(local, stream.cuda_stream)          # batch / roofline legs: whole chip, torch's stream
    else:
        ctx = ctx_ba = Context(local, stream.cuda_stream)
        ctx_lm = Context(local)           # second HIP stream: the per-frame LM kernels overlap the ORB front-end of the same frame
    frames, cam, obj = make_frame_inputs(seed0=1000 * (rank + 1))
    W, H = synth.KITTI_W, synth.KITTI_H
    # ---- inputs resident in HBM
    dev = [dict(gray=torch.from_numpy(f["gray"]).cuda(), depth=torch.from_numpy(f["depth_raw"]).cuda(),
                flow=torch.from_numpy(f["flow"]).cuda(), mask=torch.from_numpy(f["mask"]).cuda()) for f in frames]
    cam_b = [Flow2Batch(ctx_lm, [p]) for p in cam]
    obj_b = [Flow2Batch(ctx_lm, ps) for ps in obj]
    # The per-frame sequence runs in the C++ host class FramePipeline (vdo_slam_amd/host/FramePipeline.cc: the hot
    # part of Tracking::GrabImageRGBD + Track over the C-ABI, state chained frame to frame); one ctypes call per frame.
    from vdo_slam_amd.pipeline import FramePipeline, kitti_params
    pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ))
    torch.cuda.synchronize()
    counts = pipe.counts
    n_cam = cam[0].n

    def step(i):
        # UpdateMask (K15) -> K1 -> propagation (K11) -> camera LM (K16, stream 2) || ORB (K3-K7) + K9 + K10 ->
        # scene flow (K13) + DynObjTracking -> object LMs (K17, stream 2) || RenewFrameInfo static (K14, K12) ->
        # RenewFrameInfo objects (K14, K12) -> tracklets, with the RANSAC-P3P initialisers (GetInitModelCam/Obj) in
        # front of both LM stages.  The LM problems of a frame are pre-built KITTI-shaped problems (the chained random
        # frames are not geometrically consistent: RANSAC finds no consensus and runs its full 500-hypothesis budget,
        # the LM on such data would not be representative); everything else is chained data.
        k = i % N_DISTINCT_FRAMES
        d = dev[k]
        pipe.step(d["gray"].data_ptr(), d["depth"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr(), cam_b[k], obj_b[k], n_cam, len(obj[k]))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    fps = world * args.steps / dt
    lm = cam_b[0].fetch()[0]
    n_all = args.steps + args.warmup
    sect = pipe.section_ms()

    out = {
        "metric": "frames/sec (per-frame hot path, KITTI-0000-shaped 1242x375) + ms/LM-iter (batch factor graph)",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 (LM) / u8,i32,f32 (front-end)", "data": "synthetic",
        "config": {"workload": "KITTI-0000-shaped per-frame hot path (C++ FramePipeline over the C-ABI): K15 UpdateMask, K1 depth, K11 propagation, ORB 2500 feats/8 levels "
                               "(pyramid, FAST, quadtree, angle, blur), K9 static filter, K10 object sampling, joint pose+flow LM camera (1200) + 5 objects (800..200) with "
                               "ref_quirks=1, K13 scene flow + DynObjTracking, K14/K12 RenewFrameInfo (static 1200, objects 800 each), tracklets",
                   "parallelism": f"replicas x{world}; inside a frame the LM chain (stream 2) overlaps the ORB front-end / RenewFrameInfo (stream 1)",
                   "orb_keypoints": counts.n_orb, "new_static_candidates": counts.n_static_new, "object_samples": counts.n_object_samples,
                   "static_tracked": counts.n_static_tracked, "object_points_tracked": counts.n_object_tracked, "objects": counts.n_objects,
                   "static_tracklets": counts.n_static_tracks, "dynamic_tracklets": counts.n_dynamic_tracks,
                   "camera_lm_iterations": int(lm["iterations"]),
                   "host_ms_per_section": {k_: round(v_ / n_all, 4) for k_, v_ in sect.items()}},
    }

    if not args.no_batch:
        # ---- batch leg: LM outer iterations on the KITTI-shaped full-batch graph (configs[2] shape)
        g = synth.make_ba_graph(60, 30000, 5, 800, seed=1 + rank)
        ba = BatchBA(ctx_ba, g)
        ba.optimize(max_iterations=1, gain_threshold=-1.0)
        ba.set_estimates(g.pose, g.point)
        barrier()
        t0 = time.perf_counter()
        st = ba.optimize(max_iterations=5, gain_threshold=-1.0)
        barrier()
        ms_iter = (time.perf_counter() - t0) * 1e3 / max(1, st.iterations)
        out["ms_per_lm_iter"] = ms_iter
        out["config"]["batch_graph"] = f"{g.n_cam} frames, {g.n_pose} pose/motion vertices, {g.n_point} points, {g.n_eb} EdgeSE3PointXYZ, {g.n_et} ternary"
        ba.close()
        if use_dist and not os.environ.get("VDO_BENCH_NO_SHARDED"):
            # ---- the same batch graph SHARDED over the ranks (landmark tracks; all-reduce over RCCL/xGMI)
            try:
                from vdo_slam_amd.dist import ShardedBatchBA
                gs = synth.make_ba_graph(60, 30000, 5, 800, seed=1)
                sh = ShardedBatchBA(ctx_ba, gs)
                sh.optimize(max_iterations=1, gain_threshold=-1.0)
                sh.ba.set_estimates(sh.shard.pose, sh.shard.point)
                calls0 = sh.hook.calls
                barrier()
                t0 = time.perf_counter()
                st = sh.optimize(max_iterations=5, gain_threshold=-1.0)
                barrier()
                out["ms_per_lm_iter_sharded"] = (time.perf_counter() - t0) * 1e3 / max(1, st.iterations)
                out["config"]["batch_sharding"] = (f"landmark tracks over {world} ranks: {sh.mine.size}/{gs.n_point} points on rank 0, "
                                                   f"{sh.hook.calls - calls0} all-reduces in {st.iterations} LM iterations, final chi2 {st.final_chi2:.6g}")
                sh.close()
            except Exception as e:                       # the replica numbers above stay valid
                out["batch_sharded_error"] = repr(e)[:300]
        # ---- roofline of the dominant kernel (K18 sweep) on an HBM-sized graph
        gr = synth.make_ba_graph(200, args.roofline_static, 10, 1500, seed=7 + rank)
        bar = BatchBA(ctx_ba, gr)
        bar.linearize()
        sweep_ms = bar.linearize(repeat=30, timed=True)
        bytes_launch = 208 * gr.n_eb + 452 * gr.n_et + 96 * gr.n_point     # SURVEY.md §8d B_sweep terms of this kernel
        achieved = bytes_launch / (sweep_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_sweep_tile<true>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc_traffic_bytes(gr),
                           "traffic_note": "HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes over this same "
                                           "kernel and graph (profiles/r01_sweep_pmc_hbm_traffic.txt; not re-collected inside bench.py): below the "
                                           "algorithmic bytes because the 6x3 blocks are stored factored (32 B instead of 144 B)",
                           "bytes_per_launch": int(bytes_launch), "avg_launch_ms": sweep_ms,
                           "units_per_launch": {"EdgeSE3PointXYZ": int(gr.n_eb), "LandmarkMotionTernaryEdge": int(gr.n_et), "points": int(gr.n_point)}}
        bar.close()
        if rank == 0 and not args.no_cpu_baseline:
            cb_ms, cb_sweep, cb_its = cpu_baseline_batch(g)
            out["cpu_baseline_batch"] = {"ms_per_lm_iter": cb_ms, "sweep_ms": cb_sweep, "iterations": cb_its, "cores": 1, "kind": "port"}
    if rank == 0 and not args.no_cpu_baseline:
        cfps, cn, cstage = cpu_baseline_frames(frames, cam, obj)
        out["cpu_baseline"] = {"value": cfps, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"{cn} frames of the same synthetic sequence through the same stages (oracle, 1 thread)",
                               "ms_per_stage": cstage}
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
