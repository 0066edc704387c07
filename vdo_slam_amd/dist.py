"""Multi-GPU batch BA: landmark-track shards, one process per GPU.

Every rank holds all pose vertices and pose-pose edges (replicated) and the points it owns with
their binary/ternary edges.  The exchanges the C-ABI solver (vdo_ba_optimize) needs go one of two ways:
  * transport "rccl" (the default when the process group is "nccl" with one GPU per rank): the library owns an RCCL
    communicator (vdo_rccl_comm_*, vdo_ba_set_rccl) and issues ncclAllReduce itself, in place, on its stream - the host is
    not in the loop; torch.distributed only carries the 128-byte communicator id at start-up;
  * transport "callback": the solver calls back into :class:`AllReduceHook` (vdo_ba_set_allreduce), which runs
    ``torch.distributed.all_reduce`` - for gloo groups (CPU tests, several ranks sharing one GPU).
Nothing in this file computes: it only cuts the graph and moves bytes.
"""
import ctypes as C
import dataclasses
import os
import sys

import numpy as np

from . import _capi as K

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)


def partition(graph, world: int) -> np.ndarray:
    """Owner rank of every point (host-only C-ABI call: works without a GPU)."""
    gc, keep = K.graph_to_c(graph)
    owner = np.zeros(graph.n_point, np.int32)
    L = K.lib()
    L.vdo_ba_partition.argtypes = [C.POINTER(K.BAGraphC), C.c_int, K.c_int32_p]
    K.check(L.vdo_ba_partition(C.byref(gc), world, owner.ctypes.data_as(K.c_int32_p)))
    return owner


def shard_graph(graph, owner: np.ndarray, rank: int):
    """Shard of ``graph`` for ``rank``: (BAGraph, ids of its points in the full graph)."""
    mine = np.nonzero(owner == rank)[0].astype(np.int32)
    new_of_old = np.full(graph.n_point, -1, np.int32)
    new_of_old[mine] = np.arange(mine.size, dtype=np.int32)
    eb = np.nonzero(owner[graph.eb_point] == rank)[0] if graph.n_eb else np.zeros(0, np.int64)
    et = np.nonzero(owner[graph.et_p1] == rank)[0] if graph.n_et else np.zeros(0, np.int64)
    if graph.n_et:
        assert np.all(owner[graph.et_p2[et]] == rank), "a track was split across ranks"
    g = dataclasses.replace(
        graph,
        point=np.ascontiguousarray(graph.point[mine]),
        eb_pose=graph.eb_pose[eb], eb_point=new_of_old[graph.eb_point[eb]],
        eb_z=np.ascontiguousarray(graph.eb_z[:, eb]), eb_w=graph.eb_w[eb],
        et_p1=new_of_old[graph.et_p1[et]], et_p2=new_of_old[graph.et_p2[et]], et_pose=graph.et_pose[et],
        et_z=np.ascontiguousarray(graph.et_z[:, et]), et_w=graph.et_w[et],
        point_gt=None if graph.point_gt is None else graph.point_gt[mine])
    return g, mine


class _DevView:
    """Zero-copy view of ``count`` doubles at a raw device pointer for ``torch.as_tensor``."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None}


class AllReduceHook:
    """ctypes callback handed to ``vdo_ba_set_allreduce``: in-place sum / max over the process group."""

    def __init__(self, group=None, stream_ptr: int = 0, on_device: bool = True):
        import torch
        import torch.distributed as dist
        self.calls = 0
        self.doubles = 0
        self.error = None
        trace = bool(os.environ.get("VDO_DIST_TRACE"))

        def fn(_user, buf, count, op):
            try:
                rop = dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM
                if trace:
                    print(f"[allreduce rank {dist.get_rank(group)}] #{self.calls} count={count} op={op}", file=sys.stderr, flush=True)
                if on_device:
                    ext = torch.cuda.ExternalStream(stream_ptr) if stream_ptr else torch.cuda.current_stream()
                    with torch.cuda.stream(ext):
                        t = torch.as_tensor(_DevView(buf, count), device="cuda")
                        if dist.get_backend(group) == "gloo":       # single-box tests: stage through the host
                            h = t.cpu()
                            dist.all_reduce(h, op=rop, group=group)
                            t.copy_(h)
                        else:
                            dist.all_reduce(t, op=rop, group=group)
                else:
                    arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(int(count),))
                    dist.all_reduce(torch.from_numpy(arr), op=rop, group=group)
                self.calls += 1
                self.doubles += int(count)
                return 0
            except Exception as e:                     # never unwind through the C frames
                self.error = e
                return -1

        self.cfunc = ALLREDUCE_FN(fn)


class RcclComm:
    """RCCL communicator owned by libvdo_hip (one per context); the id travels through the torch process group."""

    def __init__(self, ctx, group=None):
        import torch.distributed as dist
        L = K.lib()
        L.vdo_rccl_unique_id.argtypes = [C.c_char_p]
        L.vdo_rccl_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.vdo_rccl_comm_destroy.argtypes = [C.c_void_p]
        L.vdo_rccl_comm_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.vdo_rccl_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = C.create_string_buffer(128)
        if rank == 0:
            K.check(L.vdo_rccl_unique_id(buf))
        box = [buf.raw]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = C.c_void_p()
        K.check(L.vdo_rccl_comm_create(ctx._h, box[0], world, rank, C.byref(self._h)))
        self.rank, self.world, self._keep = rank, world, ctx

    def stats(self):
        calls, nbytes = C.c_int64(), C.c_int64()
        K.check(K.lib().vdo_rccl_comm_stats(self._h, C.byref(calls), C.byref(nbytes)))
        return calls.value, nbytes.value

    def allreduce(self, device_ptr: int, count: int, op: int = 0):
        K.check(K.lib().vdo_rccl_allreduce(self._h, C.c_void_p(device_ptr), count, op))

    def close(self):
        if self._h:
            K.lib().vdo_rccl_comm_destroy(self._h); self._h = C.c_void_p()


class _Counter:
    def __init__(self, comm): self.comm = comm
    @property
    def calls(self): return self.comm.stats()[0]
    @property
    def doubles(self): return self.comm.stats()[1] // 8
    error = None


_probe_serial = [0]


def _one_gpu_per_rank(ctx, group):
    """RCCL needs a device of its own per rank (ncclCommInitRank fails - or hangs - when two ranks of a communicator sit on the
    same GPU).  Every rank contributes (host, device) and all take the same decision.  The exchange goes through the rendezvous
    STORE of the process group (TCP key/value), not through a collective of the group itself: in the very situation this probe is
    for - two ranks of an "nccl" group on one GPU - an all_gather over that group would be the first thing to fail."""
    import os
    import socket
    import torch.distributed as dist
    # (a launcher that narrows the visible devices per rank makes every rank's device "0": the masks are part of the identity)
    mine = repr((socket.gethostname(), int(getattr(ctx, "device", 0)), os.environ.get("HIP_VISIBLE_DEVICES", ""), os.environ.get("ROCR_VISIBLE_DEVICES", "")))
    from torch.distributed import distributed_c10d as c10d
    store = c10d._get_default_store()                            # (no store / an old torch: raises on EVERY rank alike - nothing to decide)
    ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
    _probe_serial[0] += 1                                        # (every rank constructs its sharded handles in the same order)
    tag = f"vdo_one_gpu_per_rank/{_probe_serial[0]}/{','.join(map(str, ranks))}"
    me = dist.get_rank()
    # The decision must be COLLECTIVE: ranks that pick different transports (one "rccl", one "callback") deadlock in their first all-reduce.  Two
    # phases through the store: (1) everyone publishes its identity and reads everyone's - a rank that fails here (a get that times out) says so in
    # (2), where everyone reads everyone's verdict; rccl only if every rank saw all identities and they are distinct.  A failure of phase 2 itself
    # raises (the group is broken: better an exception on this rank than a hang in a collective).
    try:
        store.set(f"{tag}/id/{me}", mine)
        everyone = [store.get(f"{tag}/id/{r}").decode() for r in ranks]     # (get blocks until the key is there)
        verdict = "1" if len(set(everyone)) == len(everyone) else "0"
    except Exception:                                            # noqa: BLE001
        verdict = "E"
    store.set(f"{tag}/ok/{me}", verdict)
    verdicts = [store.get(f"{tag}/ok/{r}").decode() for r in ranks]
    if store.add(f"{tag}/done", 1) == len(ranks):                # the last reader cleans up
        for r in ranks:
            for kind in ("id", "ok"):
                try:
                    store.delete_key(f"{tag}/{kind}/{r}")
                except Exception:                                # noqa: BLE001 - a store without delete_key: the keys stay (a few bytes)
                    pass
        try:
            store.delete_key(f"{tag}/done")
        except Exception:                                        # noqa: BLE001
            pass
    return all(v == "1" for v in verdicts)


class ShardedBatchBA:
    """Batch BA over ``dist.get_world_size()`` GPUs.  ``optimize`` returns the same LM statistics on
    every rank; ``estimates`` returns the full pose array and the full point array (shards gathered).
    transport: "rccl" (ncclAllReduce issued by the library), "callback" (torch.distributed through a host callback), or
    None = rccl for an "nccl" process group, callback otherwise."""

    def __init__(self, ctx, graph, group=None, owner=None, transport=None):
        import torch.distributed as dist
        from .ba import BatchBA
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.group = group
        self.full = graph
        self.owner = partition(graph, self.world) if owner is None else owner
        self.shard, self.mine = shard_graph(graph, self.owner, self.rank)
        self.ba = BatchBA(ctx, self.shard)
        if transport is None:
            transport = "rccl" if dist.get_backend(group) == "nccl" and _one_gpu_per_rank(ctx, group) else "callback"
        self.transport = transport
        L = K.lib()
        self.comm = None
        if transport == "rccl":
            self.comm = RcclComm(ctx, group)
            L.vdo_ba_set_rccl.argtypes = [C.c_void_p, C.c_void_p]
            K.check(L.vdo_ba_set_rccl(self.ba._h, self.comm._h))
            self.hook = _Counter(self.comm)
        else:
            self.hook = AllReduceHook(group, stream_ptr=ctx.stream_ptr)
            L.vdo_ba_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int]
            K.check(L.vdo_ba_set_allreduce(self.ba._h, self.hook.cfunc, None, self.rank))

    def optimize(self, **kw):
        try:
            st = self.ba.optimize(**kw)
        except K.VdoError as e:
            if self.hook.error is not None:
                raise RuntimeError(f"all-reduce hook failed: {self.hook.error!r}") from e
            raise
        return st

    def estimates(self):
        import torch
        import torch.distributed as dist
        pose, pt = self.ba.estimates()
        full = np.zeros((self.full.n_point, 3))
        full[self.mine] = pt
        t = torch.from_numpy(full)
        if dist.get_backend(self.group) == "nccl":
            tg = t.cuda(); dist.all_reduce(tg, group=self.group); full = tg.cpu().numpy()
        else:
            dist.all_reduce(t, group=self.group)
        return pose, full

    def close(self):
        self.ba.close()
        if self.comm is not None:
            self.comm.close()
