// Context management and error reporting of the C-ABI (include/vdo_slam_hip.h).
#include <cstdlib>
#include <cstdarg>
#include <cstdio>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"

namespace {
thread_local char g_err[512] = "";
}

namespace vdo {
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int ctx_bind(vdo_ctx* ctx) {
  if (!ctx) return set_error(VDO_ERR_INVALID, "null context");
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e));
  return VDO_OK;
}
}  // namespace vdo

extern "C" int vdo_version(void) { return 1; }
extern "C" const char* vdo_last_error(void) { return g_err; }

extern "C" int vdo_ctx_create(int device, void* hip_stream, vdo_ctx** out) {
  if (!out) return vdo::set_error(VDO_ERR_INVALID, "vdo_ctx_create: null out");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return vdo::set_error(VDO_ERR_NO_DEVICE, "no HIP device available (%s); libvdo_hip has no CPU fallback",
                          e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return vdo::set_error(VDO_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  vdo_ctx* c = new vdo_ctx();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return vdo::set_error(VDO_ERR_NO_DEVICE, "hipSetDevice failed"); }
  if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->owns_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return vdo::set_error(VDO_ERR_NO_DEVICE, "hipStreamCreate failed"); }
    c->owns_stream = true;
  }
  *out = c;
  return VDO_OK;
}

extern "C" int vdo_ctx_destroy(vdo_ctx* ctx) {
  if (!ctx) return VDO_OK;
  hipSetDevice(ctx->device);
  if (ctx->d_arena) hipFree(ctx->d_arena);
  if (ctx->h_arena) hipHostFree(ctx->h_arena);
  if (ctx->ba_slab) hipFree(ctx->ba_slab);
  if (ctx->ba_hscal) hipHostFree(ctx->ba_hscal);
  if (ctx->ba_side) hipStreamDestroy(ctx->ba_side);
  for (hipEvent_t e : ctx->ba_ev) if (e) hipEventDestroy(e);
  if (ctx->d_stage) hipFree(ctx->d_stage);
  if (ctx->h_stage) hipHostFree(ctx->h_stage);
  if (ctx->owns_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return VDO_OK;
}

extern "C" int vdo_ctx_synchronize(vdo_ctx* ctx) {
  int rc = vdo::ctx_bind(ctx);
  if (rc != VDO_OK) return rc;
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return vdo::set_error(VDO_ERR_NO_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString(e));
  return VDO_OK;
}

extern "C" int vdo_ctx_stream(vdo_ctx* ctx, void** hip_stream_out) {
  if (!ctx || !hip_stream_out) return vdo::set_error(VDO_ERR_INVALID, "vdo_ctx_stream: null argument");
  *hip_stream_out = (void*)ctx->stream;
  return VDO_OK;
}

// A context whose (owned) stream may only use the compute units [cu_first, cu_first + cu_count) - or, with
// `invert`, every CU except those.  Used to give the latency-bound per-frame LM kernels (one workgroup per
// problem) CUs of their own while the image kernels of the same frame run on the rest of the chip.
extern "C" int vdo_ctx_create_cu_mask(int device, int cu_first, int cu_count, int invert, vdo_ctx** out) {
  if (!out || cu_first < 0 || cu_count <= 0) return vdo::set_error(VDO_ERR_INVALID, "vdo_ctx_create_cu_mask: bad argument");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return vdo::set_error(VDO_ERR_NO_DEVICE, "no HIP device available (%s); libvdo_hip has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return vdo::set_error(VDO_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  if (hipSetDevice(device) != hipSuccess) return vdo::set_error(VDO_ERR_NO_DEVICE, "hipSetDevice failed");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return vdo::set_error(VDO_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  const int ncu = prop.multiProcessorCount;
  if (cu_first + cu_count > ncu) return vdo::set_error(VDO_ERR_INVALID, "CU range [%d,%d) exceeds the %d CUs of the device", cu_first, cu_first + cu_count, ncu);
  const int words = (ncu + 31) / 32;
  uint32_t mask[32] = {0};
  for (int c = 0; c < ncu; ++c) {
    const bool in = c >= cu_first && c < cu_first + cu_count;
    if (in != (invert != 0)) mask[c / 32] |= 1u << (c % 32);
  }
  vdo_ctx* c = new vdo_ctx();
  c->device = device;
  if (hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)words, mask) != hipSuccess) { delete c; return vdo::set_error(VDO_ERR_NO_DEVICE, "hipExtStreamCreateWithCUMask failed"); }
  c->owns_stream = true;
  *out = c;
  return VDO_OK;
}
