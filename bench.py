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


def cpu_baseline_frames(frames, cam, obj, budget_s=12.0):
    """Oracle (1 thread) on the same frames: same stages as the GPU step."""
    from tests import oracle_lib, frontend_ref as R
    from tests.test_oracle_flow2 import run_oracle
    from vdo_slam_amd import synth_frames as SF
    o = oracle_lib.load()
    n = 0
    t0 = time.perf_counter()
    stage = {"depth": 0.0, "orb": 0.0, "frame": 0.0, "lm_cam": 0.0, "lm_obj": 0.0}
    while True:
        k = n % len(frames)
        fr = frames[k]
        t = time.perf_counter()
        d = fr["depth_raw"].copy()
        o.vdo_oracle_depth_preprocess(R._fp(d), d.size, SF.BF, SF.DEPTH_MAP_FACTOR)
        stage["depth"] += time.perf_counter() - t; t = time.perf_counter()
        kp = R.extract(o, fr["gray"])
        stage["orb"] += time.perf_counter() - t; t = time.perf_counter()
        R.static_filter(o, kp["x"], kp["y"], kp["octave"], fr["mask"], d, fr["flow"], SF.TH_DEPTH_BG)
        R.object_sample(o, fr["mask"], d, fr["flow"], SF.TH_DEPTH_OBJ)
        stage["frame"] += time.perf_counter() - t; t = time.perf_counter()
        run_oracle(o, cam[k])
        stage["lm_cam"] += time.perf_counter() - t; t = time.perf_counter()
        for p in obj[k]:
            run_oracle(o, p)
        stage["lm_obj"] += time.perf_counter() - t
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 40:
            break
    dt = time.perf_counter() - t0
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

    ctx = Context(local, stream.cuda_stream)
    ctx_lm = Context(local)               # second HIP stream: the per-frame LM kernels overlap the ORB front-end of the same frame
    frames, cam, obj = make_frame_inputs(seed0=1000 * (rank + 1))
    W, H = synth.KITTI_W, synth.KITTI_H
    # ---- inputs resident in HBM
    dev = [dict(gray=torch.from_numpy(f["gray"]).cuda(), depth=torch.from_numpy(f["depth_raw"]).cuda(),
                flow=torch.from_numpy(f["flow"]).cuda(), mask=torch.from_numpy(f["mask"]).cuda()) for f in frames]
    orb = ORBextractor(ctx, W, H)
    fimg = FrameImages(ctx, W, H)
    cam_b = [Flow2Batch(ctx_lm, [p]) for p in cam]
    obj_b = [Flow2Batch(ctx_lm, ps) for ps in obj]
    torch.cuda.synchronize()

    n_kp = n_stat = n_obj = 0

    def step(i):
        # Within a frame the LM chain (camera, then objects: they consume last frame's correspondences + this frame's
        # flow) and the ORB front-end (whose keypoints are only needed by RenewFrameInfo at the END of the frame,
        # src/Tracking.cc:1168) are independent: the LM kernels go to their own stream first, the front-end (device
        # stages + host quadtree) runs meanwhile, and the frame joins both before the next one starts.
        nonlocal n_kp, n_stat, n_obj
        k = i % N_DISTINCT_FRAMES
        d = dev[k]
        fimg.upload_device(d["depth"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())   # raw inputs -> working images (D2D)
        fimg.depth_preprocess(SF.BF, SF.DEPTH_MAP_FACTOR)                                       # K1
        ctx.synchronize()                                                                        # the LM's inputs (K11 gathers) need the metric depth
        cam_b[k].run()                                                                           # K16 (stream 2)
        obj_b[k].run()                                                                           # K17 (5 objects, one launch, stream 2)
        kp = orb.extract_device(d["gray"].data_ptr(), W)                                         # K3-K7 (+ host quadtree)
        st = fimg.static_filter(kp["x"], kp["y"], SF.TH_DEPTH_BG)                                # K9
        ob = fimg.object_sample(SF.TH_DEPTH_OBJ)                                                 # K10
        ctx_lm.synchronize()                                                                     # join: RenewFrameInfo needs both
        n_kp, n_stat, n_obj = kp["x"].size, st["keep_idx"].size, ob["label"].size

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

    out = {
        "metric": "frames/sec (per-frame hot path, KITTI-0000-shaped 1242x375) + ms/LM-iter (batch factor graph)",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 (LM) / u8,i32,f32 (front-end)", "data": "synthetic",
        "config": {"workload": "KITTI-0000-shaped per-frame hot path: K1 depth, ORB 2500 feats/8 levels (pyramid, FAST, quadtree, angle, blur), "
                               "K9 static filter, K10 object sampling, joint pose+flow LM camera (1200) + 5 objects (800..200), ref_quirks=1",
                   "parallelism": f"replicas x{world}; inside a frame the LM chain (stream 2) overlaps the ORB front-end (stream 1)", "orb_keypoints": int(n_kp), "static_matches": int(n_stat), "object_points": int(n_obj),
                   "camera_lm_iterations": int(lm["iterations"])},
    }

    if not args.no_batch:
        # ---- batch leg: LM outer iterations on the KITTI-shaped full-batch graph (configs[2] shape)
        g = synth.make_ba_graph(60, 30000, 5, 800, seed=1 + rank)
        ba = BatchBA(ctx, g)
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
                sh = ShardedBatchBA(ctx, gs)
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
        bar = BatchBA(ctx, gr)
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
