"""Multi-GPU batch BA: partition, shard construction, the all-reduce hook over a real
process group (gloo, world_size 2, 127.0.0.1), and — on a GPU box — the sharded solver end to end."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from vdo_slam_amd import dist as D
from vdo_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run_world(mode, world, tmp_path, backend="gloo", timeout=600):
    out = str(tmp_path / "res")
    port = _free_port()
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        f = open(f"{out}.log{r}", "wb")          # files, not pipes: a full pipe on one rank would stall the collective
        files.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), mode, out, backend], env=env, cwd=ROOT,
                                      stdout=f, stderr=subprocess.STDOUT))
    try:
        import time
        deadline = time.time() + timeout
        while any(p.poll() is None for p in procs):
            if time.time() > deadline or any(p.poll() not in (None, 0) for p in procs):
                break                              # a dead rank leaves its peers blocked in the collective: stop them
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        for f in files:
            f.close()
    logs = [open(f"{out}.log{r}", "rb").read().decode(errors="replace") for r in range(world)]
    if any(p.returncode != 0 for p in procs):
        raise AssertionError("\n".join(f"--- rank {r} rc={p.returncode}\n{logs[r][-2500:]}" for r, p in enumerate(procs)))
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_partition_keeps_tracks_together_and_balances(world):
    g = synth.make_ba_graph(n_frames=20, n_static=1500, n_objects=3, dyn_tracks_per_object=60, seed=2)
    owner = D.partition(g, world)
    assert owner.min() == 0 and owner.max() == world - 1
    assert np.array_equal(owner[g.et_p1], owner[g.et_p2])             # a dynamic track never straddles ranks
    inc = np.bincount(g.eb_point, minlength=g.n_point) + np.bincount(g.et_p1, minlength=g.n_point) + np.bincount(g.et_p2, minlength=g.n_point)
    load = np.bincount(owner, weights=inc, minlength=world)
    assert load.max() <= 1.15 * load.mean() + 50
    covered = np.zeros(g.n_point, bool)
    n_eb = n_et = 0
    for r in range(world):
        sh, mine = D.shard_graph(g, owner, r)
        assert not covered[mine].any()
        covered[mine] = True
        n_eb += sh.n_eb; n_et += sh.n_et
        assert sh.n_pose == g.n_pose and sh.n_ep == g.n_ep and sh.n_prior == g.n_prior     # replicated
        assert np.array_equal(sh.point, g.point[mine])
        if sh.n_eb:
            assert sh.eb_point.max() < mine.size and sh.eb_point.min() >= 0
    assert covered.all() and n_eb == g.n_eb and n_et == g.n_et


def test_partition_rejects_bad_input():
    g = synth.make_ba_graph(n_frames=6, n_static=50, n_objects=1, dyn_tracks_per_object=5, seed=1)
    from vdo_slam_amd import _capi as K
    with pytest.raises(K.VdoError):
        D.partition(g, 0)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_protocol_only_shard_systems_sum_to_the_full_system_gloo(tmp_path, oracle, world):
    """PROTOCOL ONLY (no GPU here, and the product has no CPU path): partition + shard construction + the all-reduce hook over
    a real gloo group of 2 / 4 / 8 ranks (the node sizes bench.py --gpus N is run at), with the ORACLE linearising each shard.  The
    sharded HIP solver itself is tested by the gpu tests below (2 / 4 / 8 ranks sharing one GPU over gloo; RCCL transport at world 1,
    and at world 2 where two GPUs are visible)."""
    res = _run_world("oracle_sum", world, tmp_path)
    assert sum(r["n_mine"] for r in res) > 0 and all(r["n_mine"] > 0 for r in res)
    for r in res:
        assert r["hpp_err"] < 1e-12 and r["bp_err"] < 1e-11 and r["chi_err"] < 1e-12 and r["rchi_err"] < 1e-12
        assert r["hll_equal"] and r["max_diag_err"] == 0.0


@pytest.mark.gpu
def test_sharded_lm_over_rccl_world1(tmp_path):
    """backend "nccl" (= RCCL): the exchanges issued by the library itself (vdo_rccl_comm_*, ncclAllReduce in place on the
    context's stream: SUM and MAX) and, for comparison, through the host callback.  One rank is all a 1-GPU box allows."""
    res = _run_world("gpu_lm", 1, tmp_path, backend="nccl", timeout=240)
    assert {c["transport"] for c in res[0]["cases"]} == {"rccl", "callback"}
    for c in res[0]["cases"]:
        assert c["it"][0] == c["it"][1] and c["trials"][0] == c["trials"][1], c
        assert abs(c["chi"][0] - c["chi"][1]) <= 1e-6 * c["chi"][1], c
        assert c["pose_err"] < 1e-6 and c["point_err"] < 1e-5 and c["hook_calls"] > 10
    by = {}
    for c in res[0]["cases"]:
        by.setdefault(c["n_point"], {})[c["transport"]] = c
    for pair in by.values():                              # the same exchanges whichever way they travel
        assert pair["rccl"]["hook_calls"] == pair["callback"]["hook_calls"] and pair["rccl"]["hook_doubles"] == pair["callback"]["hook_doubles"]


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (one per rank) for RCCL at world 2")
def test_sharded_lm_over_rccl_world2(tmp_path):
    """Two ranks, one GPU each, all-reduces over RCCL/xGMI issued from the C-ABI: sharded LM == single-GPU LM on every rank."""
    res = _run_world("gpu_lm", 2, tmp_path, backend="nccl", timeout=300)
    for r in res:
        for c in r["cases"]:
            assert c["it"][0] == c["it"][1] and c["trials"][0] == c["trials"][1], c
            assert abs(c["chi"][0] - c["chi"][1]) <= 1e-6 * c["chi"][1], c
            assert c["pose_err"] < 1e-6 and c["point_err"] < 1e-5 and 0 < c["n_mine"] < c["n_point"], c
    for ca, cb in zip(res[0]["cases"], res[1]["cases"]):
        assert ca["chi"][0] == cb["chi"][0] and ca["it"] == cb["it"]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_lm_matches_single_gpu(tmp_path, world):
    """2 / 4 / 8 ranks sharing cuda:0 (gloo between them; the data path is the same hook RCCL serves): the sharded LM takes the
    single-GPU LM's iterations / trials to the same chi2 and estimates, and every rank holds the same BITS of the replicated state
    (poses, chi2 trace, final lambda)."""
    res = _run_world("gpu_lm", world, tmp_path, timeout=420)
    for r in res:
        for c in r["cases"]:
            assert c["it"][0] == c["it"][1] and c["trials"][0] == c["trials"][1], c
            assert abs(c["chi0"][0] - c["chi0"][1]) <= 1e-10 * c["chi0"][1]
            assert abs(c["chi"][0] - c["chi"][1]) <= 1e-6 * c["chi"][1], c
            assert c["pose_err"] < 1e-6 and c["point_err"] < 1e-5, c
            assert 0 < c["n_mine"] < c["n_point"] and c["hook_calls"] > 10
    # every rank ends with the same answer, bit for bit
    for other in res[1:]:
        for ca, cb in zip(res[0]["cases"], other["cases"]):
            assert ca["chi"][0] == cb["chi"][0] and ca["it"] == cb["it"] and ca["trials"] == cb["trials"]
            assert ca["pose_sha"] == cb["pose_sha"] and ca["chi_trace_sha"] == cb["chi_trace_sha"] and ca["lam"] == cb["lam"]
            assert ca["hook_calls"] == cb["hook_calls"] and ca["hook_doubles"] == cb["hook_doubles"]
