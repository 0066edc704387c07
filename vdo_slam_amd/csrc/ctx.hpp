// Context + error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

struct vdo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // grow-only scratch of the small per-call entry points (arena.hpp): device block + pinned host block
  char* d_arena = nullptr; size_t d_cap = 0;
  char* h_arena = nullptr; size_t h_cap = 0;
  char* h_arena_dev = nullptr;                    // the pinned block as the DEVICE sees it (mapped): kernels write small outputs straight into it (Arena::out)
  // inputs of the NEXT frame's object chain, staged ahead (vdo_object_chain_prestage, tracking_logic.hip): device block + pinned mirror + what the host
  // worked out while staging (label slots); n < 0: nothing staged
  char* d_stage = nullptr; char* h_stage = nullptr; size_t stage_cap = 0;
  int stage_n = -1, stage_L = 0;
  int32_t stage_uni[64]; int32_t stage_off[65];
  // what a SMALL batch-BA handle is made of, kept between vdo_ba_create / vdo_ba_destroy on this context (capi_ba.hip): the windowed optimisation of Track() builds a
  // 20-frame graph every 16 frames, and ~45 hipMalloc / hipFree, a pinned block, a stream and four events were 2 of the 10 ms such a window took.  One handle at a
  // time owns the pool (ba_pool_busy); a second concurrent handle, and whatever does not fit the slab, is allocated as before.
  char* ba_slab = nullptr; size_t ba_slab_cap = 0, ba_slab_used = 0; bool ba_pool_busy = false;
  double* ba_hscal = nullptr; double* ba_hscal_dev = nullptr;                 // mapped pinned block of the LM scalars
  hipStream_t ba_side = nullptr; hipEvent_t ba_ev[4] = {nullptr, nullptr, nullptr, nullptr};
};

namespace vdo {
// Stores a thread-local message for vdo_last_error() and returns `code`.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// hipSetDevice(ctx->device); returns VDO_ERR_NO_DEVICE on failure.
int ctx_bind(vdo_ctx* ctx);
}  // namespace vdo
