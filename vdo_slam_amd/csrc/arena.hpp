// Per-call scratch of the small Tracking-side entry points (tracking.hip, tracking_logic.hip):
// one grow-only device block and one grow-only PINNED host block owned by the context.  A call
// reserves what it needs, stages its inputs through the pinned block (H2D copies from pageable
// memory and hipMalloc/hipFree cost ~100 us each on this stack: more than the kernels they feed),
// queues its outputs into the pinned block, synchronises ONCE and hands them to the caller's arrays.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"

namespace vdo {

class Arena {
 public:
  explicit Arena(vdo_ctx* c) : c_(c), s(c->stream) {}
  hipStream_t s;

  // Make room for `bytes` of device scratch and as much pinned staging (called once, before any up/alloc:
  // growing re-allocates the blocks).
  bool reserve(size_t bytes) {
    bytes = (bytes + 4095) & ~size_t(4095);
    if (bytes > c_->d_cap) {
      hipStreamSynchronize(s);
      if (c_->d_arena) hipFree(c_->d_arena);
      c_->d_arena = nullptr; c_->d_cap = 0;
      const size_t want = bytes < (size_t(8) << 20) ? (size_t(8) << 20) : bytes * 2;
      if (hipMalloc((void**)&c_->d_arena, want) != hipSuccess) return false;
      c_->d_cap = want;
    }
    if (bytes > c_->h_cap) {
      hipStreamSynchronize(s);
      if (c_->h_arena) hipHostFree(c_->h_arena);
      c_->h_arena = nullptr; c_->h_cap = 0;
      const size_t want = bytes < (size_t(8) << 20) ? (size_t(8) << 20) : bytes * 2;
      if (hipHostMalloc((void**)&c_->h_arena, want) != hipSuccess) return false;
      c_->h_cap = want;
    }
    d_off_ = h_off_ = 0;
    return true;
  }
  static size_t bytes_for(size_t n_elems_total) { return n_elems_total * 8 + 64 * 256 + 4096; }   // 8 B per element + alignment slack

  // device buffer of n elements; with `host` != null its content is staged and copied in (stream-ordered)
  template <class T>
  T* up(const T* host, size_t n) {
    T* d = (T*)take(d_off_, c_->d_arena, c_->d_cap, n * sizeof(T));
    if (!d) return nullptr;
    if (host && n) {
      T* h = (T*)take(h_off_, c_->h_arena, c_->h_cap, n * sizeof(T));
      if (!h) return nullptr;
      std::memcpy(h, host, n * sizeof(T));
      hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, s);
    }
    return d;
  }
  // queue device -> caller copy (through the pinned block; delivered by finish())
  template <class T>
  void down(T* user, const T* dev, size_t n) {
    if (!user || !n) return;
    T* h = (T*)take(h_off_, c_->h_arena, c_->h_cap, n * sizeof(T));
    if (!h) { failed_ = true; return; }
    hipMemcpyAsync(h, dev, n * sizeof(T), hipMemcpyDeviceToHost, s);
    pend_.push_back({user, h, n * sizeof(T)});
  }
  // one synchronisation, then the queued outputs reach the caller's arrays
  int finish(const char* what) {
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return set_error(VDO_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e));
    if (failed_) return set_error(VDO_ERR_OOM, "%s: scratch arena exhausted", what);
    for (const Pend& p : pend_) std::memcpy(p.user, p.pinned, p.bytes);
    pend_.clear();
    return VDO_OK;
  }

 private:
  struct Pend { void* user; const void* pinned; size_t bytes; };
  vdo_ctx* c_;
  size_t d_off_ = 0, h_off_ = 0;
  bool failed_ = false;
  std::vector<Pend> pend_;
  static void* take(size_t& off, char* base, size_t cap, size_t bytes) {
    const size_t a = (off + 255) & ~size_t(255);
    if (!base || a + bytes > cap) return nullptr;
    off = a + bytes;
    return base + a;
  }
};

}  // namespace vdo
