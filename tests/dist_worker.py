"""Worker of the world_size>1 tests (one process per rank, rendezvous on 127.0.0.1).

mode "oracle_sum" (CPU, gloo) - PROTOCOL ONLY: the product has no CPU path, so the shard systems are linearised
    by the oracle; what is exercised of the product is the partition, the shard construction and the AllReduceHook
    (called exactly as vdo_ba_optimize calls it) over a real 2-rank process group.  It does NOT run the sharded HIP path.
mode "gpu_lm" (GPU): ShardedBatchBA.optimize (the sharded HIP solver) vs single-GPU BatchBA.optimize on the full graph;
    backend "nccl": one GPU per rank, exchanges issued by the library itself over RCCL (transport "rccl") and - for
    comparison - through the host callback; backend "gloo": ranks share cuda:0 on a 1-GPU box (callback transport).
"""
import ctypes as C
import dataclasses
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _empty_points(g):
    z = np.zeros(0, np.int32)
    return dataclasses.replace(g, point=np.zeros((0, 3)), eb_pose=z, eb_point=z, eb_z=np.zeros((3, 0)), eb_w=np.zeros(0),
                               et_p1=z, et_p2=z, et_pose=z, et_z=np.zeros((3, 0)), et_w=np.zeros(0), point_gt=None)


def oracle_sum(rank, world, out_path):
    import torch.distributed as dist
    from tests import oracle_lib
    from vdo_slam_amd import _capi as K, dist as D, synth
    o = oracle_lib.load()
    g = synth.make_ba_graph(n_frames=14, n_static=500, n_objects=2, dyn_tracks_per_object=40, seed=3)
    owner = D.partition(g, world)
    shard, mine = D.shard_graph(g, owner, rank)

    def lin(graph):
        gc, keep = K.graph_to_c(graph)
        S = K.BASystem(graph)
        assert o.vdo_oracle_ba_linearize(C.byref(gc), C.byref(S.c)) == 0
        return S

    S_sh, S_pp = lin(shard), lin(_empty_points(g))
    # landmark-side partials = shard system minus the replicated pose-pose part
    buf = np.concatenate([(S_sh.Hpp - S_pp.Hpp).ravel(), (S_sh.bp - S_pp.bp).ravel(),
                          [S_sh.c.chi2 - S_pp.c.chi2, S_sh.c.robust_chi2 - S_pp.c.robust_chi2]])
    buf = np.ascontiguousarray(buf)
    hook = D.AllReduceHook(on_device=False)
    assert hook.cfunc(None, buf.ctypes.data, buf.size, 0) == 0
    mx = np.array([float(np.abs(S_sh.Hll.reshape(-1, 3, 3)[:, [0, 1, 2], [0, 1, 2]]).max())])
    assert hook.cfunc(None, mx.ctypes.data, 1, 1) == 0
    assert hook.calls == 2 and hook.doubles == buf.size + 1
    P = g.n_pose
    Hpp = buf[:36 * P].reshape(P, 36) + S_pp.Hpp
    bp = buf[36 * P:42 * P].reshape(P, 6) + S_pp.bp
    chi2, rchi2 = buf[42 * P] + S_pp.c.chi2, buf[42 * P + 1] + S_pp.c.robust_chi2
    S_full = lin(g)
    res = dict(rank=rank,
               hpp_err=float(np.abs(Hpp - S_full.Hpp).max() / np.abs(S_full.Hpp).max()),
               bp_err=float(np.abs(bp - S_full.bp).max() / np.abs(S_full.bp).max()),
               chi_err=float(abs(chi2 - S_full.c.chi2) / S_full.c.chi2), rchi_err=float(abs(rchi2 - S_full.c.robust_chi2) / S_full.c.robust_chi2),
               hll_equal=bool(np.array_equal(S_sh.Hll, S_full.Hll[mine]) and np.array_equal(S_sh.bl, S_full.bl[mine])),
               max_diag_err=float(abs(mx[0] - np.abs(S_full.Hll.reshape(-1, 3, 3)[:, [0, 1, 2], [0, 1, 2]]).max())),
               n_mine=int(mine.size), n_eb=int(shard.n_eb), n_et=int(shard.n_et))
    json.dump(res, open(f"{out_path}.{rank}", "w"))
    dist.barrier()


def gpu_lm(rank, world, out_path, backend):
    import torch
    import torch.distributed as dist
    from vdo_slam_amd import dist as D, synth
    from vdo_slam_amd.ba import BatchBA, Context
    dev = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    ctx = Context(dev)
    res = dict(rank=rank, cases=[])
    import hashlib
    transports = ("rccl", "callback") if backend == "nccl" else ("callback",)
    cases = (dict(n_frames=14, n_static=500, n_objects=2, dyn_tracks_per_object=40, seed=3),
             dict(n_frames=30, n_static=3000, n_objects=3, dyn_tracks_per_object=100, seed=4))
    if world > 2:
        cases = cases[1:]                                  # (many ranks on one GPU: one graph with enough tracks for every rank)
    else:
        cases = cases + (dict(n_frames=16, n_static=1200, n_objects=0, dyn_tracks_per_object=0, seed=8),)      # round 6: 96 unknowns - the reduced matrix is all-reduced and solved by ONE workgroup on every rank (k_dense_small, ba_dense.hip)
        cases = cases + (dict(n_frames=300, n_static=700, n_objects=1, dyn_tracks_per_object=30, seed=6, hubs=4),)      # round 6: 4 static points seen from all 300 cameras - hub landmarks (ba_hub.hip) on whichever rank owns them
    for transport in transports:
        for kw in cases:
            kw = dict(kw)
            n_hub = kw.pop("hubs", 0)
            g = synth.make_ba_graph(**kw)
            if n_hub:
                g = synth.with_hub_points(g, n_hub, seed=1)
            sh = D.ShardedBatchBA(ctx, g, transport=transport)
            st = sh.optimize(max_iterations=6, gain_threshold=-1.0)
            pose, point = sh.estimates()
            one = BatchBA(ctx, g)
            st1 = one.optimize(max_iterations=6, gain_threshold=-1.0)
            pose1, point1 = one.estimates()
            res["cases"].append(dict(
                transport=sh.transport, it=(st.iterations, st1.iterations), trials=(st.total_trials, st1.total_trials),
                chi=(st.final_chi2, st1.final_chi2), chi0=(st.initial_chi2, st1.initial_chi2),
                pose_err=float(np.abs(pose - pose1).max()), point_err=float(np.abs(point - point1).max()),
                hook_calls=sh.hook.calls, hook_doubles=sh.hook.doubles, n_mine=int(sh.mine.size), n_point=int(g.n_point),
                # replicated state must be the same BITS on every rank (poses, LM scalars): compared across ranks by the test
                pose_sha=hashlib.sha1(np.ascontiguousarray(pose).tobytes()).hexdigest(), lam=float(st.final_lambda),
                chi_trace_sha=hashlib.sha1(np.ascontiguousarray(np.frombuffer(bytes(st.chi2_trace), np.float64)[:st.iterations + 1]).tobytes()).hexdigest()))
            sh.close(); one.close()
    json.dump(res, open(f"{out_path}.{rank}", "w"))
    dist.barrier()


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        if mode == "oracle_sum":
            oracle_sum(rank, world, out_path)
        else:
            gpu_lm(rank, world, out_path, backend)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
