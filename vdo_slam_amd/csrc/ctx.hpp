// Context + error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

struct vdo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  // grow-only scratch of the small per-call entry points (arena.hpp): device block + pinned host block
  char* d_arena = nullptr; size_t d_cap = 0;
  char* h_arena = nullptr; size_t h_cap = 0;
};

namespace vdo {
// Stores a thread-local message for vdo_last_error() and returns `code`.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// hipSetDevice(ctx->device); returns VDO_ERR_NO_DEVICE on failure.
int ctx_bind(vdo_ctx* ctx);
}  // namespace vdo
