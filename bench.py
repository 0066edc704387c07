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
`value` is measured with the reference's semantics - everything of a frame is done when its call returns; `value_deferred` is the
throughput mode (the object stage of frame k finished inside call k+1; rounds 1-2 reported that one as `value`).
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
import gc
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
# instance mask missing for two frames (exercises UpdateMask): ONE definition, shared with the parity tests (vdo_slam_amd/synth_seq.py)
from vdo_slam_amd.synth_seq import (BENCH_FLOW_SIGMA as FLOW_SIGMA, BENCH_INVALID_DEPTH as INVALID_DEPTH, BENCH_ZERO_FLOW as ZERO_FLOW, BENCH_N_OBJECTS as N_OBJECTS,  # noqa: E402
                                    BENCH_BOX_DEPTH as BOX_DEPTH, BENCH_MAX_FRAMES as MAX_SEQ_FRAMES, KITTI0000_FRAMES)
# batch graphs of the N > 1 legs (and their one-GPU numbers at N = 1): BASELINE configs[3] - Oxford-Multimotion-shaped (example/omd.yaml: 3000 features per
# frame, four swinging boxes filling a third of the view): 300 frames, 150 k static + 4 x 40 k dynamic tracks (1.9 M EdgeSE3PointXYZ, 0.8 M ternary edges,
# 1 496 pose / motion vertices) - and configs[4]: 1 M landmarks / 5 k pose vertices / 20 objects
OMD_SHAPE = (300, 150000, 4, 40000)
LARGE_SHAPE = (239, 950000, 20, 500)
SHARD_LEGS = (("control", (60, 30000, 5, 800), 5), ("omd", OMD_SHAPE, 5), ("large", LARGE_SHAPE, 3))
FULL_WARMUP = 5          # the KITTI-0000-length run: 5 warm-up + 148 timed frames = 153 (example/vdo_slam.cc:95-96)


def _pmc_traffic_bytes(graph, dims):
    """HBM bytes per k_sweep_tile launch from the committed PMC summary, if it was taken on this very graph AND this very tile layout (else None)."""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_sweep_pmc_hbm_traffic.txt")
    try:
        txt = open(path).read()
        m = re.search(r"n_eb (\d+) n_et (\d+) n_point (\d+) tiles (\d+) eb_entries (\d+)", txt)
        t = re.search(r"= ([0-9.]+) MB\s*$", txt, re.M)
        if m and t and tuple(int(v) for v in m.groups()) == (graph.n_eb, graph.n_et, graph.n_point, dims["tiles"], dims["eb_entries"]):
            return float(t.group(1)) * 1e6
    except OSError:
        pass
    return None


_PMC_EXTRA = {}        # SQ counters of the sweep kernel per launch (the third pass of _pmc_traffic_live)


def _pmc_traffic_live(n_static, graph, dims, timeout_s=150, script="sweep_only.py", kernel_like="%k_sweep_tile<true%", sq=True, extra=None):
    """HBM bytes per launch of one kernel MEASURED DURING THIS BENCH RUN: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE - each in its own run, as
    MI355X_MICROARCH.md prescribes, no trace domains beside them) over tools/<script> on the same graph, in child processes; 2*FETCH + WRITE with
    the guide's gfx950 correction.  None when rocprofv3 is not there, a pass fails or the child's graph / tile layout is not this one (the committed
    counter file, then the byte model, take over).  sq: a third pass with the SQ counters (VALU issue, LDS pipe) into `extra`."""
    import re, shutil, sqlite3, subprocess, tempfile
    if not shutil.which("rocprofv3"):
        return None
    here = os.path.dirname(os.path.abspath(__file__))
    extra = _PMC_EXTRA if extra is None else extra
    tot = {}
    tmp = tempfile.mkdtemp(prefix="vdo_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            r = subprocess.run(["rocprofv3", "--pmc", ctr, "-d", out, "--", sys.executable, os.path.join(here, "tools", script), str(n_static)],
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s, text=True)
            m = re.search(r"n_eb (\d+) n_et (\d+) n_point (\d+) tiles (\d+) eb_entries (\d+)", r.stdout or "")
            if r.returncode != 0 or not m or tuple(int(v) for v in m.groups()) != (graph.n_eb, graph.n_et, graph.n_point, dims["tiles"], dims["eb_entries"]):
                return None
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if not dbs:
                return None
            row = sqlite3.connect(dbs[0]).execute("select avg(value), count(*) from counters_collection where kernel_name like ? and counter_name = ?", (kernel_like, ctr)).fetchone()
            if not row or not row[1]:
                return None
            tot[ctr] = float(row[0]) * 1024.0                    # (the counters are in KB)
        extra.clear()
        if sq:
            try:                                                 # a third pass: what the SQ saw (VALU issue, LDS pipe) - reported beside the HBM fraction, never instead of it
                out = os.path.join(tmp, "SQ")
                ctrs = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"]
                r = subprocess.run(["rocprofv3", "--pmc"] + ctrs + ["-d", out, "--", sys.executable, os.path.join(here, "tools", script), str(n_static)],
                                   cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s, text=True)
                dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
                if r.returncode == 0 and dbs:
                    con = sqlite3.connect(dbs[0])
                    for c in ctrs:
                        row = con.execute("select avg(value), count(*) from counters_collection where kernel_name like ? and counter_name = ?", (kernel_like, c)).fetchone()
                        if row and row[1]:
                            extra[c] = float(row[0])
            except Exception:                                    # noqa: BLE001
                pass
        return 2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]       # (gfx950 tallies 128-B read requests at 64 B: FETCH_SIZE doubled)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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
    br = {k2: v / max(1, n - 1) * 1e3 for k2, v in pipe.bracket_s.items()}                    # (per tracked frame: frame 0 only initialises)
    br["object_estimate"] = pipe.bracket_s["object_estimate"] / max(1, pipe.n_object_estimates) * 1e3     # mean per object, as the reference reports it (Tracking.cc:1003-1006)
    return n / dt, n, {k2: v / n * 1e3 for k2, v in pipe.stage_s.items()}, pipe.Tl.astype(np.float64), br


def cpu_reference_source_brackets(frames, W, H, n_max=12):
    """all_timing of the reference's own src/Tracking.cc compiled verbatim (oracle/_ref/libref_track.so) over the first frames: the CPU-baseline side
    of the five-bracket comparison, in a CHILD process (the reference reads members it never initialises: it stays out of the process that carries the
    HIP runtime).  Informational: {} when the library did not travel with the snapshot."""
    try:
        import subprocess
        import tempfile
        from tests import oracle_lib as _ol
        if _ol.load_ref_track() is None:
            return {}
        from vdo_slam_amd import synth, synth_frames as SF
        from vdo_slam_amd.system import write_settings
        root = os.path.dirname(os.path.abspath(__file__))
        with tempfile.TemporaryDirectory() as td:
            cfg = write_settings(os.path.join(td, "k.yaml"), W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ)
            n_run = min(n_max, len(frames))
            d = {"n": n_run}
            for kk in range(n_run):
                for q in ("gray", "depth_raw", "flow", "mask"):
                    d[f"{q}_{kk}"] = np.ascontiguousarray(frames[kk][q])
            fin = os.path.join(td, "frames.npz")
            np.savez(fin, **d)
            code = f"import sys; sys.path.insert(0, {root!r}); from tests.ref_track import brackets_worker_main; brackets_worker_main({cfg!r}, {fin!r})"
            r = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True, cwd=root, timeout=120)
        res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"reference_source_build": [round(v, 4) for v in res["ms"]],
                "reference_source_build_note": (
                    f"all_timing of the reference's own src/Tracking.cc compiled verbatim against the mini-cv shim (oracle/ref/), {res['tracked_frames']} tracked frames, "
                    f"{res['frames_per_s']:.1f} frames/s as a whole: first-party code as the reference wrote it (one cv::Mat per 3-D point ...), but OpenCV's primitives are the oracle's scalar "
                    "restatements and the shim's containers are unoptimised - NOT a baseline for speed; the oracle pipeline is the faster CPU path and stays `cpu_baseline`")}
    except Exception as e:                                # noqa: BLE001 - informational leg
        return {"reference_source_build_error": repr(e)[:200]}


def cpu_worker(path, budget_s):
    """One of the N concurrent CPU-baseline processes: the oracle-composed Track() over the stored frames, for ~budget_s seconds."""
    from tests import oracle_lib
    from tests.pipeline_ref import OraclePipeline
    z = np.load(path)
    n_fr = int(z["n"])
    frames = [{q: z[f"{q}_{k}"] for q in ("gray", "depth_raw", "flow", "mask")} for k in range(n_fr)]
    pipe = OraclePipeline(oracle_lib.load(), build_lm=True)
    n = 0
    t0 = time.perf_counter()
    while n < n_fr:
        pipe.step(frames[n]); n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    print(json.dumps({"frames": n, "seconds": time.perf_counter() - t0}))


def cpu_baseline_frames_multi(frames, nproc, budget_s=8.0, max_frames=40):
    """the N-process throughput of the CPU path - nproc independent sequences (the reference is single-threaded; one
    process per core is how a host would be filled), same frames, run concurrently; aggregate frames/s."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "frames.npz")
        fr = frames[:max_frames]
        np.savez(path, n=len(fr), **{f"{q}_{k}": f[q] for k, f in enumerate(fr) for q in ("gray", "depth_raw", "flow", "mask")})
        env = dict(os.environ, OMP_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, "--cpu-worker-budget", str(budget_s)], stdout=subprocess.PIPE, env=env)
                 for _ in range(nproc)]
        outs = [json.loads(pr.communicate()[0].decode().strip().splitlines()[-1]) for pr in procs]
    return sum(o["frames"] / o["seconds"] for o in outs), [o["frames"] for o in outs]


def cpu_baseline_window(window_map):
    """The CPU side of value_with_windowed_ba: the oracle's Levenberg (1 thread, sparse Cholesky; at most 100 iterations, gain 1e-3 - Optimizer::PartialBatchOptimization's
    settings) on the last 20 frames of a Map the product exported, as the graph the reference's builder makes of it (tests/map_builder_ref.py).  Returns (seconds,
    iterations, (poses, points, EdgeSE3PointXYZ, EdgeSE3))."""
    from tests import map_builder_ref as SM
    from tests import oracle_lib
    from vdo_slam_amd import _capi as K
    gw, _info = SM.map_to_graph(window_map, partial_window=20)
    gcw, keepw = K.graph_to_c(gw)
    optw = K.LMOptionsC(100, 1e-3, 0, 0, 0.0, 0)
    stw = K.LMStatsC()
    pw = np.zeros_like(gw.pose); qw = np.zeros_like(gw.point)
    t0w = time.perf_counter()
    oracle_lib.load().vdo_oracle_ba_optimize(C.byref(gcw), C.byref(optw), K._dp(pw), K._dp(qw), C.byref(stw))
    return time.perf_counter() - t0w, int(stw.iterations), (gw.n_pose, gw.n_point, gw.n_eb, gw.n_ep)


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


def parity_start(seq_dir, seqs):
    """The reference's side of the parity leg: one CPU child process per sequence runs oracle/_ref/libref_full.so (the reference's own sources compiled verbatim) over the
    rendered frames; started before the HIP runtime is loaded, collected by parity_check.  seqs: (tag, spec, frames).  [] when the library did not travel with the snapshot."""
    from tests import bench_parity as BP
    cfg_par = BP.write_bench_settings(os.path.join(seq_dir, "kitti_parity.yaml"))
    jobs = []
    for tag, sp, fr_ in seqs:
        out_npz = os.path.join(seq_dir, f"ref_{tag}.npz")
        pr = BP.start_reference(cfg_par, os.path.join(seq_dir, tag), len(fr_), out_npz, BP.labels_of(sp))
        if pr is not None:
            jobs.append((tag, fr_, cfg_par, BP.labels_of(sp), pr, out_npz))
    return jobs


def parity_check(jobs):
    """The product's side and the comparison (tests/bench_parity.py; the same as tests/test_bench_sequence_gpu.py): System::TrackRGBD on the same host buffers, synchronous,
    its own RANSAC + EPnP + LM.  Raises SystemExit - nothing is timed - when the parity does not hold.  Returns {"parity": ..., "parity_full_sequence": ...}."""
    out = {}
    if not jobs:
        return out
    from tests import bench_parity as BP
    from tests.ref_track import finish_sequence
    for tag, fr_, cfg_, labels_, pr, out_npz in jobs:
        got_ = BP.product_sequence(cfg_, fr_, labels_)
        ref_ = finish_sequence(pr, out_npz, timeout_s=900)
        par = BP.compare({q: ref_[q] for q in ref_.files}, got_)
        del got_, ref_
        # a window of the driver's size must agree in EVERYTHING; over the KITTI-0000 length one weakly constrained object may leave the reference's
        # trajectory (the reference does the same to itself under one ulp of input: tests/test_bench_sequence_ref.py) - the number is in the record
        bad = BP.assert_parity(par) if par["frames"] <= 60 else BP.check_long_sequence(par)
        par["asserted"] = "everything (bit-exact parts + every object motion within 1e-4)" if par["frames"] <= 60 else \
            "camera pose, depth, static sets, max_id of every frame bit for bit; objects equal beyond frame 60; >= 90 % of the object motions within 1e-4"
        if bad:
            raise SystemExit(f"bench.py: PARITY with the reference FAILED on the {par['frames']}-frame sequence, nothing timed: " + "; ".join(bad))
        out["parity" if tag == "run" else "parity_full_sequence"] = par
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment, exactly what torchrun would set); rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=148, help="timed frames; 148 + 5 warm-up = the 153 frames of KITTI-0000 (example/vdo_slam.cc:96)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the batch-BA / roofline legs")
    ap.add_argument("--no-host-inputs", action="store_true", help="skip the System::TrackRGBD (host buffers in) leg")
    ap.add_argument("--no-windowed-ba", action="store_true", help="skip the value_with_windowed_ba leg (the KITTI-0000-length run with PartialBatchOptimization inside the timed frames)")
    ap.add_argument("--replicas-per-gpu", type=int, default=1, help="R independent sequences (FramePipelines) on every GPU; value = all of them")
    ap.add_argument("--replica-sweep", type=str, default="", help="e.g. 1,2,4,8: also report frames/s for these numbers of sequences per GPU")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes of the roofline leg (the committed counter file, then the byte model, take over)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity leg (the timed sequences through the whole reference, oracle/_ref/libref_full.so, and through System::TrackRGBD)")
    ap.add_argument("--no-full-sequence", action="store_true", help="do not add the KITTI-0000-length run (value_full_sequence) when --steps / --warmup ask for another window")
    ap.add_argument("--roofline-static", type=int, default=2200000, help="static landmarks of the roofline graph (default: 13.3 M edges, ~515 MB per sweep launch - twice the 256 MB Infinity Cache)")
    ap.add_argument("--cpu-worker", type=str, default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-budget", type=float, default=8.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker, args.cpu_worker_budget)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    # stdout carries exactly ONE line (the JSON result): libraries that print banners to fd 1 (RCCL's version block on
    # communicator teardown) are sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # ---- the sequences: geometrically consistent synthetic KITTI-shaped RGB-D + flow + masks (vdo_slam_amd/synth_seq.py), rendered by numpy-only child
    # processes BEFORE the HIP runtime is loaded: (a) the one this run times - SURVEY 8d's events (a mask missing for two frames, an object leaving, an object
    # entering; turning boxes) fall inside the timed window whatever --steps is; (b) the KITTI-0000-length one (153 frames, events at SURVEY's frames), timed
    # as `value_full_sequence` whatever --steps the caller passes.  On rank 0 of a one-GPU run both also go through THE WHOLE REFERENCE
    # (oracle/_ref/libref_full.so, CPU child processes started here, compared below before anything is timed).
    import atexit, shutil, tempfile
    from vdo_slam_amd import synth_seq as SQ
    rank, world, _ = _dist_env()
    seq_dir = tempfile.mkdtemp(prefix="vdo_bench_seq_", dir="/tmp")
    atexit.register(shutil.rmtree, seq_dir, ignore_errors=True)
    spec = SQ.bench_spec(args.warmup, args.steps, seed=17 * rank)
    spec_full = SQ.bench_spec(FULL_WARMUP, KITTI0000_FRAMES - FULL_WARMUP, seed=17 * rank)
    want_full = (not args.no_full_sequence) and spec_full != spec and world == 1
    render_workers = max(1, int(_cpu_budget() / max(1, world)))
    frames = SQ.render_bench_sequence(spec, os.path.join(seq_dir, "run"), workers=render_workers)
    frames_full = SQ.render_bench_sequence(spec_full, os.path.join(seq_dir, "full"), workers=render_workers) if want_full else None
    parity_jobs = []                # (tag, frames, settings, labels, reference child, its output)
    if rank == 0 and world == 1 and not args.no_parity:
        parity_jobs = parity_start(seq_dir, (("run", spec, frames),) + ((("full", spec_full, frames_full),) if want_full else ()))

    import torch
    import torch.distributed as dist
    # the cyclic collector stays out of the timed windows, as in timeit: with torch imported a full collection walks ~10^6 objects
    # (40-45 ms, 30 frames' worth), and where it lands depends on the allocation count of everything before it.  Reference counting
    # still frees everything the legs drop; VDO_BENCH_GC=1 leaves the collector on.
    if not os.environ.get("VDO_BENCH_GC"):
        gc.collect(); gc.freeze(); gc.disable()
    rank, world, local = _dist_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (libvdo_hip has no CPU fallback)")
    n_dev = torch.cuda.device_count()
    shared_gpu = world > n_dev            # more ranks than devices (a 1-GPU box asked for --gpus 2): ranks share devices, collectives over gloo
    local = local % n_dev
    torch.cuda.set_device(local)
    use_dist = world > 1 or bool(os.environ.get("VDO_BENCH_FORCE_DIST"))     # FORCE: exercise the RCCL legs on a 1-GPU box
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        if shared_gpu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
    stream = torch.cuda.Stream()          # non-default stream shared by torch events and libvdo_hip
    torch.cuda.set_stream(stream)

    from vdo_slam_amd import synth, synth_frames as SF
    from vdo_slam_amd.ba import BatchBA, Context

    ctx_ba = Context(local, stream.cuda_stream)            # batch / roofline legs: torch's stream
    W, H = synth.KITTI_W, synth.KITTI_H
    n_seq = spec["n_seq"]
    drop_masks, leave_at, enter_at = {k_: set(v_) for k_, v_ in spec["drop_masks"].items()}, spec["leave_at"], spec["enter_at"]
    objs = SQ.bench_objects(spec)                                                            # 5 boxes (2 x 0.9 m deep), 4 of them turning

    # ---- PARITY FIRST ("parity asserted before any timing is reported"): the timed sequences through the product's System::TrackRGBD (host buffers,
    # synchronous) against the whole reference's own run of them - pose, depth, mask, static / object sets, samples, labels, max_id, tracklets bit for bit, object
    # motions to the north star's 1e-4 (tests/bench_parity.py; the same comparison as tests/test_bench_sequence_gpu.py)
    out_parity = parity_check(parity_jobs)
    dev = [{q: torch.from_numpy(np.ascontiguousarray(f[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")} for f in frames]
    # The per-frame sequence runs in the C++ host class FramePipeline (vdo_slam_amd/host/FramePipeline.cc: the hot
    # part of Tracking::GrabImageRGBD + Track over the C-ABI, state chained frame to frame); one ctypes call per frame.
    from vdo_slam_amd.pipeline import FramePipeline, kitti_params
    import threading
    defer = 0 if os.environ.get("VDO_BENCH_SYNC_OBJECTS") else 1

    def defer_now():
        return defer
    cpus = _cpu_budget() / max(1, world)
    dump_sections = bool(os.environ.get("VDO_BENCH_DUMP_SECTIONS"))
    AGG = ("cam_lm_iterations", "n_static_tracked", "n_object_tracked", "n_objects", "n_ransac_cam", "n_cam_inliers", "n_ransac_obj", "n_recovered_masks", "n_motion_model_obj", "n_mm_inliers_obj")

    class Replica:
        """One sequence on this GPU: a FramePipeline with its own HIP streams (4 contexts), host helper thread and Map."""
        def __init__(self, cpus_here, first=False, dev_frames=None, window=(0, 0), with_map=False):
            self.dev = dev if dev_frames is None else dev_frames
            # host threads per replica: main + 1 helper of FramePipeline (polls) + the quadtree helpers of ORB (sleep when idle);
            # with fewer than ~5 CPUs per replica the helpers would only steal time from each other
            orb_threads = os.environ.get("VDO_ORB_THREADS_FORCE") or ("0" if cpus_here < 3 else ("5" if cpus_here >= 10 else "3"))
            os.environ["VDO_ORB_THREADS"] = orb_threads           # read by vdo_orb_create
            self.orb_threads = orb_threads
            self.ctx = Context(local, stream.cuda_stream) if first else Context(local)
            self.ctx_lm, self.ctx_obj = Context(local), Context(local)       # camera LM || ORB front-end; object LMs || RenewFrameInfo + next camera stage
            self.ctx_w = Context(local) if (not os.environ.get("VDO_BENCH_NO_WORKER") and cpus_here >= 5) else None
            # + ORB of the frame on a third host thread (its own context / stream) when there are CPUs for it
            self.ctx_orb = Context(local) if (self.ctx_w is not None and not os.environ.get("VDO_BENCH_NO_ORB_THREAD") and cpus_here >= 6) else None
            self.pipe = FramePipeline(self.ctx, self.ctx_lm, kitti_params(W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ,
                                                                         build_lm=1, defer_objects=defer_now(), window_size=window[0], overlap_size=window[1]), self.ctx_obj, self.ctx_w, self.ctx_orb)
            if with_map:
                self.pipe.attach_map()                            # (the untimed window-sample pass: the Map is what tests/map_builder_ref.py turns into the oracle's graph)
            elif not os.environ.get("VDO_BENCH_NO_MAP"):
                self.pipe.keep_graph()                            # "Save Graph Structure" (Tracking.cc:1031-1159): every frame is appended to the flat GraphStore the batch optimisers read
            self.agg = {q: 0 for q in AGG}
            self.step_ms = []
            self.sect_trace = []
            self.err = None

        def run(self, i0, n, timed):
            # Full Track() of one frame per step: UpdateMask (K15) -> K1 -> propagation (K11) -> GetInitModelCam (RANSAC-P3P, motion model)
            # -> camera pose+flow LM (K16) || ORB (K3-K7) + K9 + K10 -> scene flow (K13) + DynObjTracking -> GetInitModelObj (RANSAC per
            # object) -> object LMs (K17, one launch) || RenewFrameInfo static (K14, K12) -> RenewFrameInfo objects -> tracklets -> Map.
            # The LM problems are built from the frame's own chained correspondences.  A sequence longer than MAX_SEQ_FRAMES wraps (a scene cut).
            try:
                for i in range(i0, i0 + n):
                    d = self.dev[i % len(self.dev)]
                    ts = time.perf_counter()
                    c = self.pipe.step(d["gray"].data_ptr(), d["depth_raw"].data_ptr(), d["flow"].data_ptr(), d["mask"].data_ptr())
                    if timed:
                        self.step_ms.append((time.perf_counter() - ts) * 1e3)
                        if dump_sections:                     # (debug: cumulative section times after every step + the counts of the frame)
                            self.sect_trace.append((dict(self.pipe.section_ms()), dict(c)))
                    for q in AGG:
                        self.agg[q] += c[q]
                self.pipe.flush()                     # deferred mode: the object stage of the last frame ends inside the timed region
            except Exception as e:                    # noqa: BLE001 - reported by the main thread
                self.err = e

        def close(self):
            self.pipe.close()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_sequences(R, dev_frames=None, warmup=None, steps=None, window=(0, 0), with_map=False):
        """R independent sequences on this GPU (own pipelines, streams, host threads), each through warm-up + the timed steps;
        returns (seconds for the timed steps - max over ranks, replicas)."""
        warmup = args.warmup if warmup is None else warmup
        steps = args.steps if steps is None else steps
        reps = [Replica(cpus / R, first=(k == 0), dev_frames=dev_frames, window=window, with_map=with_map) for k in range(R)]
        torch.cuda.synchronize()

        def all_run(i0, n, timed):
            if R == 1:
                reps[0].run(i0, n, timed)
            else:
                th = [threading.Thread(target=r.run, args=(i0, n, timed)) for r in reps]
                for t in th: t.start()
                for t in th: t.join()
            for r in reps:
                if r.err is not None:
                    raise r.err
        all_run(0, warmup, False)
        barrier()

        def _throttled():
            try:
                return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat")) if k in ("nr_throttled", "throttled_usec", "usage_usec")}
            except (OSError, ValueError):
                return {}
        thr0 = _throttled() if os.environ.get("VDO_BENCH_DUMP_STEPS") else None
        t0 = time.perf_counter()
        all_run(warmup, steps, True)                  # the sequence continues where the warm-up left it
        barrier()
        dt = time.perf_counter() - t0
        if thr0 is not None:                          # (debug: CPU-quota throttling and CPU seconds inside the timed region)
            thr1 = _throttled()
            print("cgroup cpu.stat delta over the timed region:", {k: thr1.get(k, 0) - thr0.get(k, 0) for k in thr1}, "wall_us", int(dt * 1e6), file=sys.stderr)
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        if use_dist:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()), reps

    R = max(1, args.replicas_per_gpu)
    dt, reps = run_sequences(R)
    fps = world * R * args.steps / dt
    n_all = args.steps + args.warmup
    pipe = reps[0].pipe
    counts, agg, step_ms = pipe.counts, reps[0].agg, reps[0].step_ms
    if os.environ.get("VDO_BENCH_DUMP_STEPS"):              # (debug: where the slow steps of a run are)
        print("step_ms:", " ".join(f"{v:.2f}" for v in step_ms), file=sys.stderr)
    if dump_sections:
        prev = None
        for i, (sm, cc) in enumerate(reps[0].sect_trace):
            if prev is not None:
                print(f"step {i:3d} {step_ms[i]:.2f} ms objs {cc['n_objects']} obj_pts {cc['n_object_tracked']} mm {cc['n_motion_model_obj']} | " +
                      " ".join(f"{k_}={sm[k_] - prev[k_]:.3f}" for k_ in sm), file=sys.stderr)
            prev = sm
    sect = pipe.section_ms()
    k_last = (n_all - 1) % n_seq
    Tcw = pipe.pose().astype(np.float64)
    drift = float(np.abs(Tcw[:3, 3] - frames[k_last]["Tcw"][:3, 3]).max()) if n_all <= n_seq else None
    motions = pipe.motions()
    # accuracy of what the timed run computed, against the ground truth of the synthetic sequence - ASSERTED below (a fast wrong answer is not a result):
    # camera translation / rotation error at the last frame, and for every tracked object the error of its estimated world-frame motion k-1 -> k
    # applied to the object's centre
    rot_err = float(np.abs(Tcw[:3, :3] - frames[k_last]["Tcw"][:3, :3]).max()) if n_all <= n_seq else None
    motion_err = []
    if n_all <= n_seq and k_last >= 1:
        for m in motions:
            ob = objs[m["sem_label"] - 1]
            Hgt = SQ.object_motion(ob, k_last - 1)
            _, c_ob = SQ.object_pose(ob, k_last - 1)
            He = m["H"].astype(np.float64)
            motion_err.append(float(np.abs((He[:3, :3] @ c_ob + He[:3, 3]) - (Hgt[:3, :3] @ c_ob + Hgt[:3, 3])).max()))
    acc_bounds = {"trajectory_drift_m": 0.05 + 0.005 * n_all, "rotation_error": 0.01, "object_motion_error_m": 0.3}
    if n_all <= n_seq:
        bad = []
        if drift > acc_bounds["trajectory_drift_m"]: bad.append(f"camera drift {drift:.3f} m after {n_all} frames")
        if rot_err > acc_bounds["rotation_error"]: bad.append(f"camera rotation error {rot_err:.4f}")
        bad += [f"object motion error {e:.3f} m" for e in motion_err if e > acc_bounds["object_motion_error_m"]]
        if not motions: bad.append("no object tracked in the last frame")
        if bad:
            raise SystemExit("bench.py: the timed run computed a WRONG answer (" + "; ".join(bad) + f"; bounds {acc_bounds})")
    identical = all(np.array_equal(r.pipe.pose(), pipe.pose()) and [m["H"].tolist() for m in r.pipe.motions()] == [m["H"].tolist() for m in motions] for r in reps[1:])
    rep0 = reps[0]

    out = {
        "metric": "frames/sec (per-frame hot path, KITTI-0000-shaped 1242x375) + ms/LM-iter (batch factor graph)",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 (LM, RANSAC) / u8,i32,f32 (front-end, tracking)", "data": "synthetic",
        "config": {"workload": "KITTI-0000-shaped TrackRGBD per frame = the per-frame path of Track() (C++ FramePipeline over the C-ABI, incl. \"Save Graph Structure\": every frame appended to the GraphStore of the batch optimisers; "
                               "the windowed PartialBatchOptimization Track() fires every 16 frames, src/Tracking.cc:1168-1181, is NOT inside the timed frames of `value` / `value_full_sequence` on either side - "
                               "it is timed inside the frames under its own key, `value_with_windowed_ba` (+ cpu_baseline.with_windowed_ba), and per iteration as ms_per_lm_iter*): K15 UpdateMask, K1 depth, K11 propagation, "
                               "RANSAC (AP3P) + EPnP + motion-model initialisers, ORB 2500 feats/8 levels (pyramid, FAST, quadtree, angle, blur), K9 static filter, K10 object sampling, "
                               "joint pose+flow LM for the camera (<=1200 matches) and every tracked object (ref_quirks=1) built from the frame's own correspondences, "
                               "K13 scene flow + DynObjTracking, K14/K12 RenewFrameInfo (static 1200, objects 800 each), tracklets, graph store; "
                               f"geometrically consistent synthetic sequence of {n_seq} frames: {N_OBJECTS} moving boxes (4 turning, yaw rate <= 0.05 rad/frame), flow noise sigma {FLOW_SIGMA} px, "
                               f"{INVALID_DEPTH:.0%} invalid depth, {ZERO_FLOW:.0%} zero flow, the instance mask of object 1 missing in frames {sorted(drop_masks)}, "
                               f"object 2 leaves at frame {leave_at}, object 5 enters at frame {enter_at}",
                   "sequences_per_gpu": R, "sequences_identical": bool(identical),
                   "parallelism": f"replicas x{world}" + (f" (ranks share {n_dev} device(s), collectives over gloo)" if shared_gpu else "") + f", {R} sequence(s) per GPU; {4 + int(rep0.ctx_orb is not None)} HIP streams per sequence: camera LM (2) || ORB front-end ({5 if rep0.ctx_orb is not None else 1}) || object LMs of the last frame (3) -> RenewFrameInfo (4), then UpdateMask and the static stage (4) || the object chain (1); "
                                  f"defer_objects={defer}; every LM problem runs on a cluster of up to 8 workgroups; "
                                  f"{cpus:.1f} CPUs per rank, host threads per sequence: 1 + {int(rep0.ctx_w is not None)} helper (object stage of the previous frame || camera stage; K9/K10/RenewFrameInfo static || object chain) + {int(rep0.ctx_orb is not None)} ORB thread (K3-K7 of the frame from its start, then the tail of the last frame's object stage) + {rep0.orb_threads} ORB quadtree helpers",
                   "orb_keypoints": counts.n_orb, "new_static_candidates": counts.n_static_new, "object_samples": counts.n_object_samples,
                   "static_tracklets": counts.n_static_tracks, "dynamic_tracklets": counts.n_dynamic_tracks,
                   "per_frame_mean": {q: round(v / n_all, 2) for q, v in agg.items()},
                   "n_recovered_masks_total": int(agg["n_recovered_masks"]),
                   "n_motion_model_obj": int(agg["n_motion_model_obj"]),      # object-frames whose LM was seeded by the motion model (GetInitModelObj, Tracking.cc:1803-1825)
                   "step_ms_p50_p90_max": [round(float(np.percentile(step_ms, 50)), 3), round(float(np.percentile(step_ms, 90)), 3), round(max(step_ms), 3)],
                   "trajectory_drift_m": drift, "rotation_error_last_frame": rot_err, "object_motion_error_m_last_frame": [round(e, 4) for e in motion_err],
                   "accuracy_asserted": {"against": "ground truth of the synthetic sequence, last frame of the timed run", "bounds": acc_bounds},
                   "object_translations_last_frame": [np.round(m["H"][:3, 3], 4).tolist() for m in motions],
                   "host_ms_per_section": {k_: round(v_ / n_all, 4) for k_, v_ in sect.items()}},
    }
    if R > 1:
        out["config"]["single_sequence_latency_ms_p50"] = round(float(np.percentile(step_ms, 50)), 3)
    for r in reps:
        r.close()
    del reps, pipe, rep0
    # ---- the same device-resident sequence with the REFERENCE'S RETURN SEMANTICS: everything of a frame (object LMs, RenewFrameInfo of the
    # objects, tracklets) is done when Step returns - no work deferred into the next call.  (`value` defers the object stage of frame k
    # into Step k+1: same results, one frame of latency for the object motions.)
    if not os.environ.get("VDO_BENCH_SYNC_OBJECTS") and R == 1:
        defer_saved = defer
        defer = 0
        dts_sync, rs_sync = run_sequences(1)
        defer = defer_saved
        out["value_sync"] = world * args.steps / dts_sync
        out["config"]["value_sync"] = ("device-resident inputs, every Step complete on return (the reference's TrackRGBD semantics); camera pose identical to the "
                                       "deferred run: " + str(bool(np.array_equal(rs_sync[0].pipe.pose().astype(np.float64), Tcw))))
        # The headline is the number with the REFERENCE'S SEMANTICS: `value` = everything of a frame done when its call returns.
        # The throughput mode (rounds 1-2 reported it as `value`) stays beside it as `value_deferred`.
        out["value_deferred"], out["ms_per_step_deferred"] = out["value"], out["ms_per_step"]
        out["value"], out["ms_per_step"] = out["value_sync"], dts_sync * 1e3 / args.steps
        out["config"]["value"] = ("= value_sync: inputs resident in HBM, every frame complete when its call returns (object optimisations, RenewFrameInfo of the objects, tracklets, graph "
                                  "store) - System::TrackRGBD's semantics; value_deferred = the same sequence with the object stage of frame k finished inside call k+1 (same results; "
                                  "what rounds 1-2 reported as `value`); config.step_ms_p50_p90_max, host_ms_per_section, per_frame_mean belong to the deferred run")
        for r in rs_sync:
            r.close()
    out.update(out_parity)
    # ---- the KITTI-0000-length run (5 warm-up + 148 timed frames of the 153-frame sequence, SURVEY's event frames), whatever --steps the caller passed: the same
    # Step with the reference's return semantics (everything of a frame done when its call returns); the driver's `value` window is shorter and carries no capped object LM
    window_map = None
    def windowed_leg(dev_frames_w, warm_w, steps_w):
        nonlocal defer, window_map
        defer_saved = defer
        # ---- the same run with the windowed optimisation Track() contains INSIDE the timed frames (src/Tracking.cc:1165-1183: PartialBatchOptimization over the last
        # WINDOW_SIZE = 20 frames every 16 frames, example/kitti-0000-0013.yaml): a key of its own, never `value`
        if not args.no_windowed_ba:
            defer = 0
            dt_w, rs_w = run_sequences(1, dev_frames_w, warm_w, steps_w, window=(20, 4))
            n_pb = rs_w[0].pipe.partial_batches()
            sm_w = rs_w[0].step_ms
            out["value_with_windowed_ba"] = world * steps_w / dt_w
            out["config"]["value_with_windowed_ba"] = (f"value_full_sequence's run with window 20 / overlap 4: {n_pb} PartialBatchOptimization calls (C++ graph builder from the GraphStore, "
                                                       f"Levenberg on the GPU, at most 100 iterations, gain 1e-3) inside the {steps_w} timed frames + warm-up; step ms p50 / p90 / max "
                                                       f"{np.percentile(sm_w, 50):.3f} / {np.percentile(sm_w, 90):.3f} / {max(sm_w):.3f}")
            out["config"]["windowed_ba_calls"] = int(n_pb)
            for r in rs_w:
                r.close()
            del rs_w
            # the CPU side needs a window of THIS sequence as a graph: the SECOND window (frames 16 .. 35: like every window but the first it has no gauge prior,
            # src/Optimizer.cc:227-236, and takes ~27 Levenberg iterations where the first takes 2) through an untimed pass with a Map attached, exported and turned
            # into the graph the reference's builder makes of it (tests/map_builder_ref.py: test-side code, used by the cpu_baseline leg only)
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                try:
                    _, rs_m = run_sequences(1, dev_frames_w, 0, 36, with_map=True)
                    rs_m[0].pipe.finalize_map()
                    window_map = rs_m[0].pipe.export_map(synth.KITTI_K)
                    for r in rs_m:
                        r.close()
                    del rs_m
                except Exception as e:                            # noqa: BLE001 - the product's number above stays valid
                    window_map = None
                    out["config"]["windowed_ba_cpu_sample_error"] = repr(e)[:200]
            defer = defer_saved

    if frames_full is not None and R == 1 and not os.environ.get("VDO_BENCH_SYNC_OBJECTS"):
        dev_full = [{q: torch.from_numpy(np.ascontiguousarray(f[q])).cuda() for q in ("gray", "depth_raw", "flow", "mask")} for f in frames_full]
        defer_saved = defer
        defer = 0
        steps_full = KITTI0000_FRAMES - FULL_WARMUP
        dt_full, rs_full = run_sequences(1, dev_full, FULL_WARMUP, steps_full)
        defer = defer_saved
        out["value_full_sequence"] = world * steps_full / dt_full
        sm_full = rs_full[0].step_ms
        Tf = rs_full[0].pipe.pose().astype(np.float64)
        gt_full = frames_full[KITTI0000_FRAMES - 1]["Tcw"]
        out["config"]["value_full_sequence"] = (f"{FULL_WARMUP} warm-up + {steps_full} timed frames of the {KITTI0000_FRAMES}-frame sequence (example/vdo_slam.cc:95-96; mask of object 1 missing in frames "
                                                f"{sorted(spec_full['drop_masks'])}, object 2 leaves at {spec_full['leave_at']}, object 5 enters at {spec_full['enter_at']}), device-resident inputs, every Step complete on "
                                                f"return; step ms p50 / p90 / max {np.percentile(sm_full, 50):.3f} / {np.percentile(sm_full, 90):.3f} / {max(sm_full):.3f}; camera drift against ground truth "
                                                f"{float(np.abs(Tf[:3, 3] - gt_full[:3, 3]).max()):.3f} m; this is the sequence `parity_full_sequence` compares with the reference")
        for r in rs_full:
            r.close()
        del rs_full
        windowed_leg(dev_full, FULL_WARMUP, steps_full)
        del dev_full
    elif frames_full is None and args.steps + args.warmup == KITTI0000_FRAMES and "value_sync" in out:
        out["value_full_sequence"] = out["value"]
        out["config"]["value_full_sequence"] = "= value: this run IS the KITTI-0000-length run"
        if R == 1 and not os.environ.get("VDO_BENCH_SYNC_OBJECTS"):
            windowed_leg(None, args.warmup, args.steps)
        if "parity" in out:
            out["parity_full_sequence"] = out["parity"]
    # ---- R-sweep: aggregate frames/s for several numbers of independent sequences per GPU (the per-frame path keeps <= ~10 of the
    # 256 CUs busy: one sequence per GPU leaves the chip idle, SURVEY 8e "replicas only")
    if args.replica_sweep:
        sweep = {}
        for Rs in [int(x) for x in args.replica_sweep.split(",") if x]:
            dts, rs = run_sequences(Rs)
            same = all(np.array_equal(r.pipe.pose(), rs[0].pipe.pose()) for r in rs[1:]) and np.array_equal(rs[0].pipe.pose().astype(np.float64), Tcw)
            sweep[str(Rs)] = {"frames_per_s": world * Rs * args.steps / dts, "step_ms_p50": round(float(np.percentile(rs[0].step_ms, 50)), 3), "identical_to_single": bool(same)}
            for r in rs:
                r.close()
        out["sequences_per_gpu_sweep"] = sweep
    # ---- the same sequence through System::TrackRGBD with HOST buffers (include/System.h:45-51: the reference's real entry;
    # 7.9 MB of H2D per frame, the caller's depth converted in place, synchronous - the frame is complete when the call returns)
    if rank == 0 and world == 1 and not args.no_host_inputs:
        import tempfile
        from vdo_slam_amd.system import System, write_settings
        with tempfile.TemporaryDirectory() as td:
            host_in = [(f["gray"], f["depth_raw"], f["flow"], f["mask"]) for f in frames[:n_all]]
            for mode in ("deferred", "sync"):
                sysm = System(write_settings(os.path.join(td, "kitti.yaml"), W, H, synth.KITTI_K, SF.BF, SF.DEPTH_MAP_FACTOR, SF.TH_DEPTH_BG, SF.TH_DEPTH_OBJ, window=0, overlap=0))
                sysm.set_defer(mode == "deferred")
                # depth and mask are mutated in place by TrackRGBD; ground-truth object rows (frame, label, ...) for every object of the
                # sequence, as the reference's object_pose.txt carries them: the tracker only follows objects that have one (Tracking.cc:791-841)
                bufs = [(g_, d_.copy(), fl_, m_.copy(), np.array([[k_, lab + 1] + [0.0] * 8 for lab in range(len(objs))], np.float32))
                        for k_, (g_, d_, fl_, m_) in enumerate(host_in)]
                for k in range(args.warmup):
                    sysm.track_rgbd(*bufs[k])
                gc_n = [g_["collections"] for g_ in gc.get_stats()]
                t0 = time.perf_counter()
                Th = None
                call_ms = []
                for k in range(args.warmup, n_all):
                    tc = time.perf_counter()
                    Th = sysm.track_rgbd(*bufs[k % len(bufs)])
                    call_ms.append((time.perf_counter() - tc) * 1e3)
                sysm.flush()
                dth = time.perf_counter() - t0
                out["config"]["host_inputs_call_ms_p50_p90_max_" + mode] = [round(float(np.percentile(call_ms, 50)), 3), round(float(np.percentile(call_ms, 90)), 3), round(max(call_ms), 3)]
                if os.environ.get("VDO_BENCH_DUMP_STEPS"):
                    print(f"host-input calls ({mode}):", " ".join(f"{v:.2f}" for v in call_ms), "| gc collections in the window:",
                          [g_["collections"] - n0 for g_, n0 in zip(gc.get_stats(), gc_n)], file=sys.stderr)
                out["value_host_inputs" if mode == "deferred" else "value_host_inputs_sync"] = args.steps / dth
                same = bool(Th is not None and np.array_equal(Th.astype(np.float64), Tcw))
                sysm.close()
                if mode == "sync":
                    continue
                out["config"]["host_inputs"] = ("System::TrackRGBD on pageable host buffers (gray u8, raw depth f32 converted in place, flow 2xf32, mask i32 = 7.9 MB/frame H2D "
                                                "+ 1.9 MB D2H of the converted depth), graph store attached, no windowed optimisation; "
                                                "value_host_inputs: object stage deferred into the next call (as `value`), value_host_inputs_sync: everything done when TrackRGBD returns "
                                                "(the reference's semantics); same camera pose as the device-input run: " + str(same))

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
        # the same iterations with the reduced-camera matrix assembled densely and factorised on the fp64 MFMA units (solver 3: the
        # fallback for pose graphs that are not sets of paths; vdo_lm_options.solver)
        ba.optimize(max_iterations=1, gain_threshold=-1.0, solver=3)
        ba.set_estimates(g.pose, g.point)
        barrier()
        t0 = time.perf_counter()
        st3 = ba.optimize(max_iterations=5, gain_threshold=-1.0, solver=3)
        barrier()
        out["ms_per_lm_iter_dense_mfma"] = (time.perf_counter() - t0) * 1e3 / max(1, st3.iterations)
        out["config"]["batch_solvers"] = (f"PCG (pose-chain preconditioner): {st.iterations} iterations / {st.total_trials} trials, chi2 {st.final_chi2:.9g}; "
                                          f"dense MFMA Cholesky of the {6 * g.n_pose} x {6 * g.n_pose} reduced-camera matrix: {st3.iterations} / {st3.total_trials}, chi2 {st3.final_chi2:.9g}")
        out["config"]["batch_graph"] = f"{g.n_cam} frames, {g.n_pose} pose/motion vertices, {g.n_point} points, {g.n_eb} EdgeSE3PointXYZ, {g.n_et} ternary"
        ba.close()
        if use_dist and not os.environ.get("VDO_BENCH_NO_SHARDED"):
            # ---- batch graphs SHARDED over the ranks (landmark tracks per rank, poses replicated; all-reduces of the reduced-camera quantities issued by the
            # C-ABI over RCCL / xGMI).  north_star: "only where the graph is large enough to benefit" - so three graphs:
            #   control  the 60-frame window graph of the N = 1 line (224 k edges, ~0.45 ms per iteration on ONE GPU): too small, expected to LOSE
            #   omd      BASELINE configs[3]: Oxford-Multimotion-shaped multi-object graph (4 objects carrying a third of the points)
            #   large    BASELINE configs[4]: 1 M landmarks / 5 k pose vertices / 20 objects
            # each with the same graph on one GPU of this node beside it (every rank times its own replica: `*_1gpu` is rank 0's).
            out["sharded"] = {}
            for tag, shape, its in SHARD_LEGS:
                if tag == "large" and os.environ.get("VDO_BENCH_NO_LARGE"):
                    continue
                try:
                    from vdo_slam_amd.dist import ShardedBatchBA
                    gs = synth.make_ba_graph(*shape, seed=1 if tag == "control" else 5)       # (the same graph on every rank)
                    b1 = BatchBA(ctx_ba, gs)
                    b1.optimize(max_iterations=1, gain_threshold=-1.0)
                    b1.set_estimates(gs.pose, gs.point)
                    barrier()
                    t0 = time.perf_counter()
                    st1 = b1.optimize(max_iterations=its, gain_threshold=-1.0)
                    ms1 = (time.perf_counter() - t0) * 1e3 / max(1, st1.iterations)
                    b1.close(); del b1
                    barrier()
                    sh = ShardedBatchBA(ctx_ba, gs)
                    sh.optimize(max_iterations=1, gain_threshold=-1.0)
                    sh.ba.set_estimates(sh.shard.pose, sh.shard.point)
                    calls0, dbl0 = sh.hook.calls, sh.hook.doubles
                    barrier()
                    t0 = time.perf_counter()
                    st = sh.optimize(max_iterations=its, gain_threshold=-1.0)
                    barrier()
                    msN = (time.perf_counter() - t0) * 1e3 / max(1, st.iterations)
                    out["sharded"][tag] = {
                        "graph": f"{gs.n_cam} frames, {gs.n_pose} pose/motion vertices, {gs.n_point} points, {gs.n_eb} EdgeSE3PointXYZ, {gs.n_et} ternary, {gs.n_ep} EdgeSE3",
                        "ranks": world, "transport": sh.transport, "ms_per_lm_iter_sharded": msN, "ms_per_lm_iter_1gpu": ms1, "speedup_vs_1gpu": ms1 / msN,
                        "lm_iterations": int(st.iterations), "trials": int(st.total_trials), "same_trajectory_as_1gpu": bool(st.iterations == st1.iterations and st.total_trials == st1.total_trials),
                        "final_chi2": float(st.final_chi2), "final_chi2_1gpu": float(st1.final_chi2),
                        "allreduces_per_lm_iter": (sh.hook.calls - calls0) / max(1, st.iterations),
                        "allreduce_bytes_per_lm_iter": 8 * (sh.hook.doubles - dbl0) / max(1, st.iterations),
                        "points_on_rank0": int(sh.mine.size), "points": int(gs.n_point)}
                    out["ms_per_lm_iter_sharded" if tag == "control" else f"ms_per_lm_iter_{tag}_sharded"] = msN
                    sh.close(); del sh, gs
                except Exception as e:                       # the replica numbers above stay valid
                    out["sharded"][tag] = {"error": repr(e)[:300]}
            out["config"]["batch_sharding"] = ("landmark tracks cut into `ranks` runs of equal incidence count, all pose / motion vertices and pose-pose edges replicated; per LM iteration: "
                                               "Hpp | bp | chi2 (42 P + 2 doubles) once per linearisation, the block-Jacobi diagonal and the reduced right-hand side in ONE exchange per trial (27 P + 1; two dependent ones until round 6), 6 P doubles per CG iteration, "
                                               "3 scalars per trial (DESIGN 6); transport 'rccl' = ncclAllReduce issued by the C-ABI on its stream, 'callback' = torch.distributed through the "
                                               "host callback (ranks sharing a GPU)")
        # ---- BASELINE configs[2] (KITTI 0018-0020-shaped: ~10 k landmarks, 5 objects) and a configs[4]-shaped graph (1 M landmarks,
        # 5 k pose vertices, 20 objects) on ONE GPU: ms per LM outer iteration (PCG)
        for key, shape, its in (("ms_per_lm_iter_config3", (60, 10000, 5, 400), 5), ("ms_per_lm_iter_omd", OMD_SHAPE, 5), ("ms_per_lm_iter_large", LARGE_SHAPE, 3)):
            if key != "ms_per_lm_iter_config3" and (world > 1 or os.environ.get("VDO_BENCH_NO_LARGE")):      # (N > 1: these two are the sharded legs above, with their 1-GPU time beside them)
                continue
            gx = synth.make_ba_graph(*shape, seed=5 + rank)
            bx = BatchBA(ctx_ba, gx)
            bx.optimize(max_iterations=1, gain_threshold=-1.0)
            bx.set_estimates(gx.pose, gx.point)
            barrier()
            t0 = time.perf_counter()
            stx = bx.optimize(max_iterations=its, gain_threshold=-1.0)
            barrier()
            out[key] = (time.perf_counter() - t0) * 1e3 / max(1, stx.iterations)
            out["config"][key] = (f"{gx.n_cam} frames, {gx.n_pose} pose/motion vertices, {gx.n_point} points, {gx.n_eb} EdgeSE3PointXYZ, {gx.n_et} ternary, {gx.n_ep} EdgeSE3: "
                                  f"{stx.iterations} iterations / {stx.total_trials} trials, chi2 {stx.initial_chi2:.6g} -> {stx.final_chi2:.6g}")
            bx.close()
            del bx, gx
        # ---- roofline of the dominant kernel (K18 sweep) and of the whole linearisation on an HBM-sized graph
        gr = synth.make_ba_graph(200, args.roofline_static, 10, 1500, seed=7 + rank)
        bar = BatchBA(ctx_ba, gr)
        bar.linearize()
        bar.profile_linearize(300)                                # untimed warm-up, ~70 ms of the same launches: the device has idled while the host built the graph, and its clocks take tens of
                                                                  # milliseconds of load to come back (tools/sweep_repeat_probe.py: the first 40 launches of a process take 1.3x the steady time)
        sweep_ms, lin_ms, dims = bar.profile_linearize(100)      # hipEvents on the stream the kernels run on (vdo_ba_profile_linearize): 100 launches of the sweep, then 100 linearisations
        from vdo_slam_amd.ba import linearize_byte_model
        model = linearize_byte_model(gr, dims)
        alg_bytes = 208 * gr.n_eb + 452 * gr.n_et + 96 * gr.n_point     # SURVEY.md §8d B_sweep terms of this kernel
        traffic_live = _pmc_traffic_live(args.roofline_static, gr, dims) if (rank == 0 and world == 1 and not args.no_live_pmc) else None
        traffic = traffic_live if traffic_live is not None else _pmc_traffic_bytes(gr, dims)                               # rocprofv3 --pmc passes over this kernel and graph, if committed for this layout
        used = traffic if traffic is not None else float(model["sweep"])
        achieved = used / (sweep_ms * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": "k_sweep_tile<true, true>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "frac_basis": "counters (2*FETCH_SIZE + WRITE_SIZE)" if traffic is not None else "model_bytes (no counter file for this layout)",
            "model_bytes": model, "avg_launch_ms": sweep_ms, "linearize_ms": lin_ms,
            "achieved_model": model["sweep"] / (sweep_ms * 1e-3) / 1e9, "frac_model": model["sweep"] / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "linearize_achieved_model": model["linearize"] / (lin_ms * 1e-3) / 1e9, "linearize_frac_model": model["linearize"] / (lin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "survey_8d_bytes": int(alg_bytes), "survey_8d_rate": alg_bytes / (sweep_ms * 1e-3) / 1e9,
            "note": "`frac` = HBM bytes the sweep kernel really moves per launch (PMC counters when profiles/ holds a pass over this very graph and layout, else the "
                    "design's byte model = the floor for this layout: vdo_slam_amd/ba.py linearize_byte_model) / its mean duration (hipEvents, this run) / 8 TB/s - it "
                    "cannot exceed 1.  `survey_8d_rate` divides the bytes of SURVEY 8d's formula (208 B per EdgeSE3PointXYZ, 452 B per ternary edge, 96 B per point) "
                    "by the same time: the kernel moves a fifth of them (8 B of a 6x3 block instead of 144 B, 16 B of edge inputs instead of 64 B, one scalar for the "
                    "landmark block), so that rate is NOT a bandwidth.  `linearize_*`: sweep + expansion of the pose blocks + pose-pose edges + chi2 (one "
                    "BlockSolver::buildSystem).  `valu` / `lds`: how busy the two on-chip units next in line are (DESIGN.md 4.1).",
            "layout": dims,
            "traffic_source": ("live: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/sweep_only.py on this very graph and tile layout, run by this bench in child "
                               "processes; the duration is this process's own (hipEvents)") if traffic_live is not None else
                              (("profiles/r05_sweep_pmc_hbm_traffic.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/sweep_only.py on this very graph and tile layout "
                                "(tools/profile_round5_sweep.sh; the live passes of this run were not available); the duration is live") if traffic is not None else None),
            "graph_vs_infinity_cache": f"{model['sweep'] / 1e6:.0f} MB per sweep launch by the byte model vs 256 MB of Infinity Cache: the traffic is HBM traffic",
            "units_per_launch": {"EdgeSE3PointXYZ": int(gr.n_eb), "LandmarkMotionTernaryEdge": int(gr.n_et), "points": int(gr.n_point), "poses": int(gr.n_pose)},
            "timed_launches": 100, "warmup_launches": 300}
        if _PMC_EXTRA.get("SQ_INSTS_VALU"):
            # what else the kernel is near: VALU issue (every VALU instruction of a wave64 occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs) and the LDS pipe
            # (SQ_ACTIVE_INST_LDS counts quad-cycles per CU) over the kernel's duration at the device's clock - fractions of those two ceilings beside the HBM one
            prop = torch.cuda.get_device_properties(local)
            clk = float(getattr(prop, "clock_rate", 0) or 2.4e6) * 1e3                # Hz (torch builds without the field: the MI355X's 2.4 GHz)
            cyc = clk * sweep_ms * 1e-3
            out["roofline"]["valu"] = {"insts_per_launch": _PMC_EXTRA["SQ_INSTS_VALU"], "issue_frac": _PMC_EXTRA["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc, "clock_hz": clk,
                                       "note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x clock x kernel time)"}
            if _PMC_EXTRA.get("SQ_ACTIVE_INST_LDS"):
                out["roofline"]["lds"] = {"active_quad_cycles_per_launch": _PMC_EXTRA["SQ_ACTIVE_INST_LDS"], "bank_conflict_cycles_per_launch": _PMC_EXTRA.get("SQ_LDS_BANK_CONFLICT"),
                                          "busy_frac": _PMC_EXTRA["SQ_ACTIVE_INST_LDS"] * 4.0 / 256.0 / cyc, "note": "SQ_ACTIVE_INST_LDS x 4 / (256 CUs x clock x kernel time)"}
        # ---- the solver side of the same graph: the Schur mat-vec of a CG iteration (k_schur_tile<0>), the largest consumer of an LM iteration - the work the
        # matrix-free sweep moved out of the linearisation (it stores 8 B per edge instead of the 144-B pose-landmark block; every mat-vec recomputes the block)
        try:
            from vdo_slam_amd.ba import schur_byte_model
            st_r = bar.optimize(max_iterations=1, gain_threshold=-1.0)
            bar.profile_schur(100)                                                      # untimed warm-up
            schur_ms = bar.profile_schur(100)
            smodel = schur_byte_model(gr, dims, args.roofline_static)
            sx = {}
            s_traffic = _pmc_traffic_live(args.roofline_static, gr, dims, script="schur_only.py", kernel_like="%k_schur_tile<0>%", sq=True, extra=sx) \
                if (rank == 0 and world == 1 and not args.no_live_pmc) else None
            s_used = s_traffic if s_traffic is not None else float(smodel["matvec"])
            out["roofline_solver"] = {
                "bound": "hbm", "kernel": "k_schur_tile<0>", "achieved": s_used / (schur_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": s_used / (schur_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": s_traffic,
                "frac_basis": "counters (2*FETCH_SIZE + WRITE_SIZE, live passes over tools/schur_only.py)" if s_traffic is not None else "model_bytes (no live counter pass)",
                "model_bytes": smodel, "frac_model": smodel["matvec"] / (schur_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": schur_ms,
                "survey_8d_matvec_bytes": smodel["stored_hpl"], "survey_8d_matvec_rate": smodel["stored_hpl"] / (schur_ms * 1e-3) / 1e9,
                "launches_per_lm_iteration": "one per CG iteration (+ one each of the <1> and <2> instantiations per trial)",
                "timed_launches": 100, "lm_iteration_before": [int(st_r.iterations), int(st_r.total_trials)],
                "note": "`frac` = HBM bytes one Schur mat-vec launch moves (counters of this run when available, else the byte model of vdo_slam_amd/ba.py schur_byte_model) / its mean "
                        "duration (hipEvents on the context's stream, vdo_ba_profile_schur) / 8 TB/s.  `survey_8d_matvec_rate` divides what a design with STORED 6x3 pose-landmark blocks "
                        "would move (SURVEY 8d: 144 B per EdgeSE3PointXYZ + 360 B per ternary edge) by the same time - a comparison, not a bandwidth: this design moves 12 B per incidence "
                        "and recomputes the block (two cross products) from the point and the slot's pose in LDS"}
            if sx.get("SQ_INSTS_VALU"):
                prop = torch.cuda.get_device_properties(local)
                clk = float(getattr(prop, "clock_rate", 0) or 2.4e6) * 1e3
                cyc = clk * schur_ms * 1e-3
                out["roofline_solver"]["valu"] = {"insts_per_launch": sx["SQ_INSTS_VALU"], "issue_frac": sx["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc}
                if sx.get("SQ_ACTIVE_INST_LDS"):
                    out["roofline_solver"]["lds"] = {"active_quad_cycles_per_launch": sx["SQ_ACTIVE_INST_LDS"], "bank_conflict_cycles_per_launch": sx.get("SQ_LDS_BANK_CONFLICT"),
                                                     "busy_frac": sx["SQ_ACTIVE_INST_LDS"] * 4.0 / 256.0 / cyc}
        except Exception as e:                                   # noqa: BLE001 - the sweep's roofline above stays valid
            out["roofline_solver"] = {"error": repr(e)[:300]}
        bar.close()
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cb_ms, cb_sweep, cb_its = cpu_baseline_batch(g)
            out["cpu_baseline_batch"] = {"ms_per_lm_iter": cb_ms, "sweep_ms": cb_sweep, "iterations": cb_its, "cores": 1, "kind": "port"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU baseline is reported at N=1 only
        cfps, cn, cstage, _, cbr = cpu_baseline_frames(frames)
        out["cpu_baseline"] = {"value": cfps, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"the first {cn} frames of the same sequence through the same full Track() (oracle, 1 thread)",
                               "ms_per_stage": cstage}
        # the north star's ">= 30 x the CPU baseline", spelled out: `value` (inputs resident in HBM, the contract's headline) and the number a caller of the
        # reference's own entry point sees - System::TrackRGBD on host buffers, everything done when the call returns (PCIe inclusive, never `value`)
        out["speedup_vs_cpu_baseline"] = {"value": out["value"] / cfps,
                                           "value_host_inputs_sync": (out["value_host_inputs_sync"] / cfps) if "value_host_inputs_sync" in out else None,
                                           "target": 30.0,
                                           "note": "the product keeps ~8 host threads busy per sequence (main + helper + ORB thread + quadtree helpers) against a 1-thread baseline; "
                                                   "cpu_baseline.multi_process is the same CPU code on 8 processes"}
        if frames_full is not None and "value_full_sequence" in out:      # the same ratio over the KITTI-0000 length: both sides on the 153-frame sequence
            cfps_f, cn_f, _, _, _ = cpu_baseline_frames(frames_full, budget_s=12.0)
            out["cpu_baseline"]["full_sequence"] = {"value": cfps_f, "unit": "frames/s", "cores": 1, "kind": "port", "sample": f"the first {cn_f} frames of the {KITTI0000_FRAMES}-frame sequence (oracle, 1 thread)"}
            out["speedup_vs_cpu_baseline"]["value_full_sequence"] = out["value_full_sequence"] / cfps_f
            cfps_full = cfps_f
        elif "value_full_sequence" in out:
            out["speedup_vs_cpu_baseline"]["value_full_sequence"] = out["value_full_sequence"] / cfps
            cfps_full = cfps
        else:
            cfps_full = cfps
        if window_map is not None and "value_with_windowed_ba" in out:
            # the CPU path with the same windows: the oracle's Levenberg (1 thread, sparse Cholesky) on the first window of this sequence, once; every window has that shape
            try:
                t_win, its_w, dims_w = cpu_baseline_window(window_map)
                n_pb = out["config"]["windowed_ba_calls"]
                cfps_w = KITTI0000_FRAMES / (KITTI0000_FRAMES / cfps_full + n_pb * t_win)
                out["cpu_baseline"]["with_windowed_ba"] = {"value": cfps_w, "unit": "frames/s", "cores": 1, "kind": "port",
                                                           "sample": (f"full_sequence's frame rate + {n_pb} windows at the cost of ONE measured here: the oracle's Levenberg on the second window (frames 16 .. 35, no "
                                                                      f"windowed refinement before it) of this sequence ({dims_w[0]} poses, {dims_w[1]} points, {dims_w[2]} + {dims_w[3]} edges; {its_w} iterations, "
                                                                      f"{t_win * 1e3:.0f} ms; the first window - the only one with a gauge prior - stops after 2)")}
                out["speedup_vs_cpu_baseline"]["value_with_windowed_ba"] = out["value_with_windowed_ba"] / cfps_w
            except Exception as e:                            # noqa: BLE001
                out["cpu_baseline"]["with_windowed_ba_error"] = repr(e)[:200]
        # ---- the reference's own five clock() brackets (all_timing[0..4]: mask update, camera estimate, object tracking, object estimate per object,
        # map update = RenewFrameInfo; src/Tracking.cc:230-243, 685-703, 1366-1603, 868-1010, 1016-1020), side by side, ms per frame
        pf = out["config"]["host_ms_per_section"]; n_obj_mean = max(out["config"]["per_frame_mean"]["n_objects"], 1e-9)
        out["reference_brackets_ms"] = {
            "brackets": ["mask_update", "camera_estimate", "object_tracking", "object_estimate (mean per object)", "map_update"],
            "cpu_oracle_1_thread": [round(cbr[q], 4) for q in ("mask_update", "camera_estimate", "object_tracking", "object_estimate", "map_update")],
            "gpu_product_host_wall": [round(pf["k15_k11_objects"], 4), round(pf["k1_k11_ransac_cam"] + pf["wait_cam_lm"], 4), round(pf["k13_dynobj"], 4),
                                      round((pf["ransac_obj"] + pf["wait_obj_lm"]) / n_obj_mean, 4), round(pf["renew_static"] + pf["renew_object"], 4)],
            "note": "gpu_product_host_wall: wall time of the host sections that do the bracket's work in the throughput run (UpdateMask comes with K11 (objects) + K13 in one call; "
                    "the camera bracket = ingest + GetInitModelCam + the wait for the camera LM launched a frame ahead; the object bracket = RANSAC / EPnP / problem set-up + "
                    "the wait for the object LMs, divided by the mean number of objects); the brackets of the product OVERLAP (three host threads, five streams): they do not add up to the frame time"}
        rsb = cpu_reference_source_brackets(frames, W, H)      # the reference's OWN Tracking.cc (oracle/_ref/libref_track.so, when it travelled with the snapshot)
        out["reference_brackets_ms"].update(rsb)
        # the N-process throughput next to the 1-thread figure (N = min(8, CPUs of this job); the reference itself is single-threaded)
        try:
            nproc = int(max(1, min(8, _cpu_budget())))
            mfps, mframes = cpu_baseline_frames_multi(frames, nproc)
            out["cpu_baseline"]["multi_process"] = {"value": mfps, "unit": "frames/s", "processes": nproc, "cores_of_the_job": int(_cpu_budget()),
                                                    "sample": f"{nproc} concurrent processes, {mframes} frames each, ~8 s"}
        except Exception as e:                            # noqa: BLE001 - the 1-thread baseline above stays valid
            out["cpu_baseline"]["multi_process_error"] = repr(e)[:200]
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
