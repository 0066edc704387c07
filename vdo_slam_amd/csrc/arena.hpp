// Per-call scratch of the small Tracking-side entry points (tracking.hip, tracking_logic.hip):
// one grow-only device block and one grow-only PINNED host block owned by the context.  A call
// reserves what it needs, stages its inputs through the pinned block (H2D copies from pageable
// memory and hipMalloc/hipFree cost ~100 us each on this stack: more than the kernels they feed),
// queues its outputs into the pinned block, synchronises ONCE and hands them to the caller's arrays.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/vdo_slam_hip.h"
#include "ctx.hpp"

namespace vdo {

class Arena {
 public:
  explicit Arena(vdo_ctx* c) : c_(c), s_(c->stream) {}
  // The stream to launch on.  Asking for it sends the inputs staged so far (one merged H2D copy), so a kernel launched
  // on it sees them.
  hipStream_t stream() { flush(); return s_; }

  // Make room for `bytes` of device scratch and as much pinned staging (called once, before any up/alloc:
  // growing re-allocates the blocks).
  bool reserve(size_t bytes) {
    bytes = (bytes + 4095) & ~size_t(4095);
    if (bytes > c_->d_cap) {
      hipStreamSynchronize(s_);
      if (c_->d_arena) hipFree(c_->d_arena);
      c_->d_arena = nullptr; c_->d_cap = 0;
      const size_t want = bytes < (size_t(8) << 20) ? (size_t(8) << 20) : bytes * 2;
      if (hipMalloc((void**)&c_->d_arena, want) != hipSuccess) return false;
      c_->d_cap = want;
    }
    if (bytes > c_->h_cap) {
      hipStreamSynchronize(s_);
      if (c_->h_arena) hipHostFree(c_->h_arena);
      c_->h_arena = nullptr; c_->h_cap = 0; c_->h_arena_dev = nullptr;
      const size_t want = bytes < (size_t(8) << 20) ? (size_t(8) << 20) : bytes * 2;
      if (hipHostMalloc((void**)&c_->h_arena, want, hipHostMallocMapped) != hipSuccess) return false;
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, c_->h_arena, 0) == hipSuccess) c_->h_arena_dev = (char*)dp;
      c_->h_cap = want;
    }
    off_ = 0; in_lo_ = in_hi_ = 0;
    return true;
  }
  static size_t bytes_for(size_t n_elems_total) { return n_elems_total * 8 + 64 * 256 + 4096; }   // 8 B per element + alignment slack

  // Device buffer of n elements; with `host` != null its content is staged for the next merged H2D copy.  The device block
  // and the pinned block share their layout (a buffer has the same offset in both), so adjacent staged inputs travel in ONE
  // copy and all outputs of a call come back in ONE copy (a copy costs ~5-7 us of stream time whatever its size here).
  template <class T>
  T* up(const T* host, size_t n) {
    const size_t a = (off_ + 255) & ~size_t(255), bytes = n * sizeof(T);
    if (!c_->d_arena || !c_->h_arena || a + bytes > c_->d_cap || a + bytes > c_->h_cap) return nullptr;
    off_ = a + bytes;
    if (host && n) {
      std::memcpy(c_->h_arena + a, host, bytes);
      if (in_hi_ == in_lo_) { in_lo_ = a; in_hi_ = a + bytes; }
      else if (a - in_hi_ <= 4096) in_hi_ = a + bytes;            // adjacent (up to alignment / a small output buffer nothing wrote yet)
      else { flush(); in_lo_ = a; in_hi_ = a + bytes; }
    } else if (in_hi_ != in_lo_ && bytes > 4096) {
      flush();                                                    // a large device-only buffer ends the run of adjacent inputs
    }
    return (T*)(c_->d_arena + a);
  }
  // An OUTPUT buffer the kernels write ONCE and nothing on the device reads again: its slot of the PINNED block as the device sees it (mapped host memory) -
  // the stores go over PCIe while the kernel runs and no device -> host copy operation (~10 us of stream time plus its enqueue, whatever its size) stands
  // between the last kernel and the synchronisation.  down() of such a pointer only hands the data over after finish().  Falls back to a device buffer.
  template <class T>
  T* out(size_t n) {
    static const bool off = std::getenv("VDO_ARENA_NO_MAPPED_OUT") != nullptr;
    if (off || !c_->h_arena_dev) return up<T>(nullptr, n);
    const size_t a = (off_ + 255) & ~size_t(255), bytes = n * sizeof(T);
    if (!c_->d_arena || !c_->h_arena || a + bytes > c_->d_cap || a + bytes > c_->h_cap) return nullptr;
    off_ = a + bytes;
    if (in_hi_ != in_lo_) flush();                                // (keeps a later staged input from merging its copy across this slot)
    return (T*)(c_->h_arena_dev + a);
  }
  // queue device -> caller copy (through the pinned block; delivered by finish())
  template <class T>
  void down(T* user, const T* dev, size_t n) {
    if (!user || !n) return;
    if (c_->h_arena_dev && (const char*)dev >= c_->h_arena_dev && (const char*)dev + n * sizeof(T) <= c_->h_arena_dev + c_->h_cap) {      // an out() buffer: already in the pinned block
      pend_.push_back({user, (size_t)((const char*)dev - c_->h_arena_dev), n * sizeof(T), true});
      return;
    }
    const size_t o = (size_t)((const char*)dev - c_->d_arena);
    if ((const char*)dev < c_->d_arena || o + n * sizeof(T) > c_->d_cap) { failed_ = true; return; }
    pend_.push_back({user, o, n * sizeof(T), false});
  }
  // queue a device -> pinned copy WITHOUT a hand-over to a caller array: returns where the data will be in the pinned block once
  // finish() has returned (valid until the context's arena is used again) - large outputs of which the caller reads a few rows
  template <class T>
  const T* down_view(const T* dev, size_t n) {
    const size_t o = (size_t)((const char*)dev - c_->d_arena);
    if ((const char*)dev < c_->d_arena || o + n * sizeof(T) > c_->d_cap) { failed_ = true; return nullptr; }
    if (n) pend_.push_back({nullptr, o, n * sizeof(T), false});
    return (const T*)(c_->h_arena + o);
  }
  // the outputs come back (one copy of their span, or one per buffer when the span is mostly something else), one
  // synchronisation, then they reach the caller's arrays
  // finish() in two halves for a caller with host work of its own to do while the device runs: queue_downloads() sends the staged inputs and queues the
  // device -> host copies (no wait); finish() then only waits and hands the data over.
  void queue_downloads() {
    if (queued_) return;
    queued_ = true;
    flush();
    {
      size_t lo = ~size_t(0), hi = 0, sum = 0;
      for (const Pend& p : pend_) if (!p.mapped) { lo = std::min(lo, p.off); hi = std::max(hi, p.off + p.bytes); sum += p.bytes; }
      if (sum) {
        bool overlap = false;                                     // (a merged copy must not run over a mapped slot: it would bring the device block's bytes over the kernel's)
        for (const Pend& p : pend_) if (p.mapped && p.off < hi && p.off + p.bytes > lo) overlap = true;
        if (!overlap && hi - lo <= 4 * sum + (size_t(64) << 10)) hipMemcpyAsync(c_->h_arena + lo, c_->d_arena + lo, hi - lo, hipMemcpyDeviceToHost, s_);
        else for (const Pend& p : pend_) if (!p.mapped) hipMemcpyAsync(c_->h_arena + p.off, c_->d_arena + p.off, p.bytes, hipMemcpyDeviceToHost, s_);
      }
    }
  }
  int finish(const char* what) {
    queue_downloads();
    queued_ = false;
    hipError_t e = hipStreamSynchronize(s_);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { pend_.clear(); return set_error(VDO_ERR_NO_DEVICE, "%s: %s", what, hipGetErrorString(e)); }
    if (failed_) { pend_.clear(); return set_error(VDO_ERR_OOM, "%s: scratch arena exhausted", what); }
    for (const Pend& p : pend_) if (p.user) std::memcpy(p.user, c_->h_arena + p.off, p.bytes);
    pend_.clear();
    return VDO_OK;
  }

 private:
  void flush() {
    if (in_hi_ != in_lo_) hipMemcpyAsync(c_->d_arena + in_lo_, c_->h_arena + in_lo_, in_hi_ - in_lo_, hipMemcpyHostToDevice, s_);
    in_lo_ = in_hi_ = 0;
  }
  struct Pend { void* user; size_t off, bytes; bool mapped; };
  vdo_ctx* c_;
  hipStream_t s_;
  size_t off_ = 0, in_lo_ = 0, in_hi_ = 0;
  bool failed_ = false, queued_ = false;
  std::vector<Pend> pend_;
};

}  // namespace vdo
