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
};

namespace vdo {
// Stores a thread-local message for vdo_last_error() and returns `code`.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// hipSetDevice(ctx->device); returns VDO_ERR_NO_DEVICE on failure.
int ctx_bind(vdo_ctx* ctx);
}  // namespace vdo
