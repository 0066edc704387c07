#!/usr/bin/env python
"""bench.py — headline measurement of the hot path on MI355X (driver contract in the task brief).

A *step* is one KITTI-0000-shaped frame (1242x375) through the whole per-frame hot path of TrackRGBD — the hot part
of Tracking::GrabImageRGBD + Track(), run by the C++ FramePipeline over the C-ABI — with every input (gray, raw
depth, dense flow, instance mask) already resident in HBM when the timed region starts:
    UpdateMask -> depth preprocess -> correspondence propagation -> RANSAC-P3P / motion-model initial camera model
    -> joint pose+flow LM (camera) || ORB (pyramid, FAST cells, quadtree, IC angle, blur) + static filter + object sampling
    -> scene flow + DynObjTracking -> RANSAC per object -> joint pose+flow LM of every object (one launch)
    || RenewFrameInfo (static, objects) -> tracklets.
The frames come from a geometrically consistent synthetic sequence (vdo_slam_amd/synth_seq.py), so RANSAC, the LM and
the object tracker do the work they do on KITTI (consensus found, 1200 static matches, 2-3 tracked objects).
This is BASELINE.json configs[1] ("KITTI seq 0000 on 1xMI355X: ORB+flow front-end and per-frame PoseOptimization on
GPU").  The CPU baseline runs the same full Track() composed from the oracle on the first frames of the same sequence.
The same JSON line also carries
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
N_DISTINCT_FRAMES = 4   # make_frame_inputs(): independent random frames (tools/frame_probe.py)
# SURVEY.md 8d "synthetic inputs": flow noise N(0, 0.3^2) px, 2 % invalid depth, 1 % exactly-zero flow, 5 moving objects, one
# instance mask missing for two frames (exercises UpdateMask)
FLOW_SIGMA, INVALID_DEPTH, ZERO_FLOW, N_OBJECTS, DROP_MASKS, BOX_DEPTH = 0.3, 0.02, 0.01, 5, {30: {2}, 31: {2}}, 0.9
MAX_SEQ_FRAMES = 160    # length of the consistent synthetic sequence (the objects stay in view that long)


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


def _cpu_budget():
    """CPUs this job may use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def _dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def make_frame_inputs(seed0):
    from vdo_slam_amd import synth, synth_frames as SF
    frames = [SF.make_frame(seed=seed0 + k) for k in range(N_DISTINCT_FRAMES)]
    cam = [synth.make_flow2_problem(1200, seed=seed0 + 100 + k) for k in range(N_DISTINCT_FRAMES)]
    obj = [[synth.make_flow2_problem(n, seed=seed0 + 200 + 10 * k + j, is_object=True) for j, n in enumerate([800, 600, 400, 300, 200])]
           for k in range(N_DISTINCT_FRAMES)]
    return frames, cam, obj


def cpu_baseline_frames(frames, budget_s=14.0):
    """Oracle (1 thread) on the first frames of the same sequence: the full Track() composed from the oracle's
    functions (tests/pipeline_ref.py, build_lm=True: oracle ORB, RANSAC, LM, RenewFrameInfo, UpdateMask, tracklets)."""
    from tests import oracle_lib
    from tests.pipeline_ref import OraclePipeline
    pipe = OraclePipeline(oracle_lib.load(), build_lm=True)
    n = 0
    t0 = time.perf_counter()
    while n < len(frames):
        pipe.step(frames[n])
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return n / dt, n, {k2: v / n * 1e3 for k2, v in pipe.stage_s.items()}, pipe.Tl.astype(np.float64)


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
    # stdout carries exactly ONE line (the JSON result): libraries that print banners to fd 1 (RCCL's version block on
    # communicator teardown) are sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

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

    from vdo_slam_amd import synth, synth_frames as SF, synth_seq as SQ
    from vdo_slam_amd.ba import BatchBA, Context

    n_lm_cu = int(os.environ.get("VDO_BENCH_LM_CUS", "0"))      # measured: no gain from CU partitioning (the LM kernel is not slowed by its neighbours)
    if n_lm_cu > 0:
        # the LM chain gets CUs of its own (one latency-bound workgroup per problem); everything else runs on the other CUs
        ctx = Context(local, cu_mask=(0, n_lm_cu, True))
        ctx_lm = Context(local, cu_mask=(0, n_lm_cu, False))
        ctx_ba = Context(local, stream.cuda_stream)          # batch / roofline legs: whole chip, torch's stream
    else:
        ctx = ctx_ba = Context(local, stream.cuda_stream)
        ctx_lm = Context(local)           # second HIP stream: the camera LM overlaps the ORB front-end of the same frame
    ctx_obj = Context(local)              # third HIP stream: the object LMs overlap RenewFrameInfo and the next frame's camera stage
    # ---- the sequence: geometrically consistent synthetic KITTI-shaped RGB-D + flow + masks (vdo_slam_amd/synth_seq.py),
    # one distinct frame per step, resident in HBM before the timed region
    W, H = synth.KITTI_W, synth.KITTI_H
    n_seq = min(args.steps + args.warmup, MAX_SEQ_FRAMES)
    Ts = SQ.camera_poses(n_seq)
    objs = SQ.default_objects(N_OBJECTS, box_depth=BOX_DEPTH)         # axis-aligned boxes (2 x 0.9 m deep), not planar panels
    frames = [SQ.render_frame(k, Ts, objs, flow_sigma=FLOW_SIGMA, seed=17 * rank, invalid_depth=INVALID_DEPTH, zero_flow=ZERO_FLOW, drop_masks=DROP_MASKS)
              for k in range(n_seq)]
    dev = [{q: torch.from_numpy(np.ascontiguousarray(f[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")} for f in frames]
    # The per-frame sequence runs in the C++ host class FramePipeline (vdo_slam_amd/host/FramePipeline.cc: the hot
    # part of Tracking::GrabImageRGBD + Track over the C-ABI, state chained frame to frame); one ctypes call per frame.
    from vdo_slam_amd.pipeline import FramePipeline, kitti_params
    defer = 0 if os.environ.get("VDO_BENCH_SYNC_OBJECTS") else 1
    # host threads per replica: main + 1 helper of FramePipeline (polls) + 3 quadtree helpers of ORB (sleep when idle); with fewer
    # than ~5 CPUs per rank the helpers would only steal time from each other
    cpus_per_rank = _cpu_budget() / max(1, world)
    if "VDO_ORB_THREADS" not in os.environ:             # quadtree helpers of ORB (library default 3): measured 888 / 916 / 921 frames/s with 3 / 5 / 7
        os.environ["VDO_ORB_THREADS"] = "0" if cpus_per_rank < 3 else ("5" if cpus_per_rank >= 10 else "3")
    use_worker = not os.environ.get("VDO_BENCH_NO_WORKER") and cpus_per_rank >= 5
    ctx_w = Context(local) if use_worker else None      # helper host thread of FramePipeline, own stream + arena
    # ORB on a stream of its own (device stage queued at the start of the frame, under the camera stage): supported, results identical,
    # but measured neutral (843 vs 824 frames/s, inside the run-to-run spread): the camera stage's small kernels then share the GPU
    # with ORB's - off unless asked for
    ctx_orb = Context(local) if os.environ.get("VDO_BENCH_ORB_STREAM") else None
    pipe = FramePipeline(ctx, ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, build_lm=1, defer_objects=defer), ctx_obj, ctx_w, ctx_orb)
    torch.cuda.synchronize()
    counts = pipe.counts
    agg = {"cam_lm_iterations": 0, "n_static_tracked": 0, "n_object_tracked": 0, "n_objects": 0, "n_ransac_cam": 0, "n_cam_inliers": 0, "n_ransac_obj": 0, "n_recovered_masks": 0}

    def step(i):
        # Full Track() of one frame: UpdateMask (K15) -> K1 -> propagation (K11) -> GetInitModelCam (RANSAC-P3P, motion model)
        # -> camera pose+flow LM (K16, stream 2) || ORB (K3-K7) + K9 + K10 -> scene flow (K13) + DynObjTracking ->
        # GetInitModelObj (RANSAC per object) -> object LMs (K17, one launch, stream 2) || RenewFrameInfo static (K14, K12)
        # -> RenewFrameInfo objects (K14, K12) -> tracklets.  The LM problems are built from the frame's own chained
        # correspondences (build_lm mode).  A sequence longer than MAX_SEQ_FRAMES wraps to frame 0 (a scene cut).
        d = dev[i % n_seq]
        c = pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
        for q in agg:
            agg[q] += c[q]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    pipe.flush()
    barrier()
    step_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        step(args.warmup + i)                 # the sequence continues where the warm-up left it
        step_ms.append((time.perf_counter() - ts) * 1e3)
    pipe.flush()                              # deferred mode: the object stage of the last frame ends inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    fps = world * args.steps / dt
    n_all = args.steps + args.warmup
    sect = pipe.section_ms()
    k_last = (n_all - 1) % n_seq
    Tcw = pipe.pose().astype(np.float64)
    drift = float(np.abs(Tcw[:3, 3] - frames[k_last]["Tcw"][:3, 3]).max()) if n_all <= n_seq else None
    motions = pipe.motions()

    out = {
        "metric": "frames/sec (per-frame hot path, KITTI-0000-shaped 1242x375) + ms/LM-iter (batch factor graph)",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 (LM, RANSAC) / u8,i32,f32 (front-end, tracking)", "data": "synthetic",
        "config": {"workload": "KITTI-0000-shaped TrackRGBD per frame (C++ FramePipeline over the C-ABI, full Track()): K15 UpdateMask, K1 depth, K11 propagation, "
                               "RANSAC-P3P + motion-model initialisers, ORB 2500 feats/8 levels (pyramid, FAST, quadtree, angle, blur), K9 static filter, K10 object sampling, "
                               "joint pose+flow LM for the camera (<=1200 matches) and every tracked object (ref_quirks=1) built from the frame's own correspondences, "
                               "K13 scene flow + DynObjTracking, K14/K12 RenewFrameInfo (static 1200, objects 800 each), tracklets; "
                               f"geometrically consistent synthetic sequence of {n_seq} frames: {N_OBJECTS} moving boxes, flow noise sigma {FLOW_SIGMA} px, "
                               f"{INVALID_DEPTH:.0%} invalid depth, {ZERO_FLOW:.0%} zero flow, one instance mask missing in frames {sorted(DROP_MASKS)}",
                   "parallelism": f"replicas x{world}; {4 + (ctx_orb is not None)} HIP streams per replica: camera LM (2) || ORB front-end ({5 if ctx_orb is not None else 1}); object LMs (3) || RenewFrameInfo (1) and - "
                                  f"defer_objects={defer} - the next frame's camera stage; every LM problem runs on a cluster of up to 8 workgroups; "
                                  f"{cpus_per_rank:.1f} CPUs per replica, host threads per replica: 1 + {int(ctx_w is not None)} helper (object stage of the previous frame || camera stage + ORB; K9/K10/RenewFrameInfo static || object chain) + {os.environ['VDO_ORB_THREADS']} ORB quadtree helpers",
                   "orb_keypoints": counts.n_orb, "new_static_candidates": counts.n_static_new, "object_samples": counts.n_object_samples,
                   "static_tracklets": counts.n_static_tracks, "dynamic_tracklets": counts.n_dynamic_tracks,
                   "per_frame_mean": {q: round(v / n_all, 2) for q, v in agg.items()},
                   "step_ms_p50_p90_max": [round(float(np.percentile(step_ms, 50)), 3), round(float(np.percentile(step_ms, 90)), 3), round(max(step_ms), 3)],
                   "trajectory_drift_m": drift, "object_translations_last_frame": [np.round(m["H"][:3, 3], 4).tolist() for m in motions],
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
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cb_ms, cb_sweep, cb_its = cpu_baseline_batch(g)
            out["cpu_baseline_batch"] = {"ms_per_lm_iter": cb_ms, "sweep_ms": cb_sweep, "iterations": cb_its, "cores": 1, "kind": "port"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU baseline is reported at N=1 only
        cfps, cn, cstage, _ = cpu_baseline_frames(frames)
        out["cpu_baseline"] = {"value": cfps, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"the first {cn} frames of the same sequence through the same full Track() (oracle, 1 thread)",
                               "ms_per_stage": cstage}
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
