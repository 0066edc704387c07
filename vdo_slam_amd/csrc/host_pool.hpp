// Small persistent host thread pool shared by the host-side stages of the C-ABI (per-level quadtrees of ORB, per-problem EPnP
// refits of the RANSAC initialiser).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vdo {

// Persistent workers for the per-level quadtrees (levels are independent; level 0 holds ~40 % of the
// candidates, so 3 helpers + the calling thread bring the host stage from ~0.27 to ~0.12 ms per frame).
// Workers sleep on a condition variable between frames; the caller takes tasks too and then spins on the
// completion counter (the tail is a few microseconds).
class LevelPool {
 public:
  explicit LevelPool(int n_workers) {
    for (int i = 0; i < n_workers; ++i) th_.emplace_back([this] { worker(); });
  }
  ~LevelPool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // Runs fn(0) .. fn(n_tasks-1) on the workers and the calling thread; returns when all are done AND no worker still holds the
  // batch.  A batch is a descriptor on the caller's stack, published under mu_ with a generation count: a worker that wakes late
  // finds either the batch it was woken for, a later one, or none - never a half-reset one.  Calls from several threads (the static
  // pool of the RANSAC refits is shared by every pipeline of a process) are serialised.
  template <class F>
  void run(int n_tasks, F&& fn) {
    if (n_tasks <= 0) return;
    std::lock_guard<std::mutex> serial(run_mu_);
    Batch b;
    b.fn = [&fn](int i) { fn(i); };
    b.n = n_tasks;
    { std::lock_guard<std::mutex> g(mu_); cur_ = &b; ++gen_; }
    cv_.notify_all();
    drain(b);
    while (b.done.load(std::memory_order_acquire) < b.n) std::this_thread::yield();
    { std::lock_guard<std::mutex> g(mu_); cur_ = nullptr; }                       // from here on no worker can pick the batch up ...
    while (b.active.load(std::memory_order_acquire) != 0) std::this_thread::yield();   // ... and those that did have left it
  }

 private:
  struct Batch {
    std::function<void(int)> fn;
    int n = 0;
    std::atomic<int> next{0}, done{0};
    std::atomic<int> active{0};       // workers inside drain(); incremented under mu_ while the batch is published
  };
  static void drain(Batch& b) {
    for (;;) {
      const int i = b.next.fetch_add(1, std::memory_order_relaxed);
      if (i >= b.n) break;
      b.fn(i);
      b.done.fetch_add(1, std::memory_order_release);
    }
  }
  void worker() {
    uint64_t seen = 0;
    for (;;) {
      Batch* b;
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        b = cur_;
        if (b) b->active.fetch_add(1, std::memory_order_relaxed);
      }
      if (!b) continue;
      drain(*b);
      b->active.fetch_sub(1, std::memory_order_release);
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_;
  Batch* cur_ = nullptr;               // guarded by mu_
  uint64_t gen_ = 0;
  bool stop_ = false;
};

}  // namespace vdo
