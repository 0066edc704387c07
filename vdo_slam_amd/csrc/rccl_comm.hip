// Direct RCCL transport for the sharded batch BA: the C-ABI owns an RCCL communicator (one process per GPU,
// ranks on the xGMI fabric of one node) and issues ncclAllReduce IN PLACE on the library's device buffers, stream-ordered on
// the context's HIP stream - no host callback, no Python, no torch dispatcher between an LM trial's ~10 small exchanges.
//   rank 0: vdo_rccl_unique_id(id)  ->  the host moves the 128 bytes to every rank (any side channel)
//   all   : vdo_rccl_comm_create(ctx, id, n_ranks, rank, &comm); vdo_ba_set_rccl(ba, comm)
// librccl is opened at run time (dlopen): a process that already carries an RCCL (e.g. torch's) shares it, and the library
// still loads on a machine without RCCL (single-GPU use never touches it).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
// The handful of RCCL types / enum values this file passes through the dlsym'ed entry points, declared here so that the library
// builds on a machine without the RCCL headers (the ABI of nccl.h: 128-byte unique id, opaque communicator, C enums).
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
}

#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/vdo_slam_hip.h"
#include "ba_host.hpp"
#include "ctx.hpp"

namespace vdo {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = "symbols missing";   // dlerror() of the failed dlopen, captured once (a second dlerror() call returns NULL)
};

static RcclApi& rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) { const char* e = dlerror(); if (e) std::snprintf(api.why, sizeof api.why, "%s", e); return; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
  });
  return api;
}

}  // namespace vdo

using namespace vdo;

struct vdo_rccl_comm {
  vdo_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int n_ranks = 0, rank = 0;
  int64_t calls = 0, bytes = 0;
};

static int rccl_err(const char* what, ncclResult_t r) {
  RcclApi& A = rccl_api();
  return set_error(VDO_ERR_NO_DEVICE, "%s: %s", what, A.GetErrorString ? A.GetErrorString(r) : "RCCL error");
}

extern "C" int vdo_rccl_unique_id(char id_out[128]) {
  RcclApi& A = rccl_api();
  if (!A.ok) return set_error(VDO_ERR_UNSUPPORTED, "librccl could not be loaded: %s", A.why);
  if (!id_out) return set_error(VDO_ERR_INVALID, "vdo_rccl_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t r = A.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_err("ncclGetUniqueId", r);
  std::memcpy(id_out, &id, 128);
  return VDO_OK;
}

extern "C" int vdo_rccl_comm_create(vdo_ctx* ctx, const char id[128], int n_ranks, int rank, vdo_rccl_comm** out) {
  RcclApi& A = rccl_api();
  if (!A.ok) return set_error(VDO_ERR_UNSUPPORTED, "librccl could not be loaded: %s", A.why);
  if (!ctx || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return set_error(VDO_ERR_INVALID, "vdo_rccl_comm_create: bad argument");
  int rc = ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  ncclUniqueId uid;
  std::memcpy(&uid, id, 128);
  vdo_rccl_comm* c = new vdo_rccl_comm();
  c->ctx = ctx; c->n_ranks = n_ranks; c->rank = rank;
  const ncclResult_t r = A.CommInitRank(&c->comm, n_ranks, uid, rank);
  if (r != ncclSuccess) { delete c; return rccl_err("ncclCommInitRank", r); }
  *out = c;
  return VDO_OK;
}

extern "C" int vdo_rccl_comm_destroy(vdo_rccl_comm* c) {
  if (!c) return VDO_OK;
  if (c->comm) { ctx_bind(c->ctx); hipStreamSynchronize(c->ctx->stream); rccl_api().CommDestroy(c->comm); }
  delete c;
  return VDO_OK;
}

extern "C" int vdo_rccl_comm_stats(const vdo_rccl_comm* c, int64_t* calls, int64_t* bytes) {
  if (!c) return set_error(VDO_ERR_INVALID, "null handle");
  if (calls) *calls = c->calls;
  if (bytes) *bytes = c->bytes;
  return VDO_OK;
}

// in-place all-reduce of `count` doubles at a device pointer, stream-ordered on the communicator's context stream
extern "C" int vdo_rccl_allreduce(vdo_rccl_comm* c, double* device_buf, int64_t count, int op) {
  if (!c || !device_buf || count < 0) return set_error(VDO_ERR_INVALID, "vdo_rccl_allreduce: bad argument");
  if (count == 0) return VDO_OK;
  const ncclResult_t r = rccl_api().AllReduce(device_buf, device_buf, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, c->ctx->stream);
  if (r != ncclSuccess) return rccl_err("ncclAllReduce", r);
  ++c->calls; c->bytes += 8 * count;
  return VDO_OK;
}

static int rccl_reduce_fn(void* user, void* buf, int64_t n, int op) { return vdo_rccl_allreduce((vdo_rccl_comm*)user, (double*)buf, n, op) == VDO_OK ? 0 : -1; }

extern "C" int vdo_ba_set_rccl(vdo_ba* ba, vdo_rccl_comm* c) {
  if (!ba) return set_error(VDO_ERR_INVALID, "vdo_ba_set_rccl: null handle");
  if (!c) return vdo_ba_set_allreduce(ba, nullptr, nullptr, 0);
  if (c->ctx != ba->ctx) return set_error(VDO_ERR_INVALID, "vdo_ba_set_rccl: the communicator and the problem must use the same context (stream)");
  return vdo_ba_set_allreduce(ba, rccl_reduce_fn, c, c->rank);
}
