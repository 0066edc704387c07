// How long after a kernel's end does the host know?  (a) hipStreamSynchronize, (b) hipEventRecord + hipEventSynchronize, (c) the kernel's last act is a ticket in mapped
// pinned memory the host polls.  Kernel: one workgroup spinning for ~20 us of shader clock, then writing 16 doubles (+ the ticket).  Reports launch -> host-knows wall time per variant.
//   hipcc --offload-arch=gfx950 -O3 tools/sync_latency_probe.hip -o /tmp/sync_probe && /tmp/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>

__global__ void k_ticket(volatile unsigned* ticket, unsigned value) { *ticket = value; }

__global__ void k_work(double* out, volatile unsigned* ticket, unsigned value, long long spin) {
  const long long t0 = clock64();
  while (clock64() - t0 < spin) { }
  if (threadIdx.x < 16) out[threadIdx.x] = (double)value + threadIdx.x;
  if (ticket) {
    __threadfence_system();
    if (threadIdx.x == 0) *ticket = value;
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  double* h; hipHostMalloc((void**)&h, 4096, hipHostMallocMapped);
  double* hd; hipHostGetDevicePointer((void**)&hd, h, 0);
  volatile unsigned* ticket = (volatile unsigned*)(h + 64);
  unsigned* ticket_d = (unsigned*)(hd + 64);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  for (long long spin : {2000LL, 50000LL}) {
    for (int variant = 0; variant < 5; ++variant) {
      std::vector<double> t;
      for (int it = 0; it < 300; ++it) {
        *ticket = 0;
        const double t0 = now_us();
        if (variant == 0) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, hd, (volatile unsigned*)nullptr, (unsigned)(it + 1), spin); hipStreamSynchronize(s); }
        else if (variant == 1) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, hd, (volatile unsigned*)nullptr, (unsigned)(it + 1), spin); hipEventRecord(ev, s); hipEventSynchronize(ev); }
        else if (variant == 2) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, hd, (volatile unsigned*)ticket_d, (unsigned)(it + 1), spin); while (*ticket != (unsigned)(it + 1)) { } }
        else if (variant == 3) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, hd, (volatile unsigned*)nullptr, (unsigned)(it + 1), spin); hipStreamWriteValue32(s, (void*)ticket_d, (unsigned)(it + 1), 0); while (*ticket != (unsigned)(it + 1)) { } }
        else { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, hd, (volatile unsigned*)nullptr, (unsigned)(it + 1), spin); hipLaunchKernelGGL(k_ticket, dim3(1), dim3(1), 0, s, (volatile unsigned*)ticket_d, (unsigned)(it + 1)); while (*ticket != (unsigned)(it + 1)) { } }
        const double t1 = now_us();
        if (it >= 50) t.push_back(t1 - t0);
        if (variant >= 2) hipStreamSynchronize(s);
        if (h[3] != (double)(it + 1) + 3) { printf("bad data\n"); return 1; }
      }
      std::sort(t.begin(), t.end());
      const char* nm[5] = {"hipStreamSynchronize", "hipEventRecord + hipEventSynchronize", "ticket in mapped pinned memory, host polls", "hipStreamWriteValue32 behind it, host polls", "one-thread ticket kernel behind it, host polls"};
      printf("spin %lld cycles, %-45s: launch -> host knows  p10 %.1f  p50 %.1f  p90 %.1f us\n", spin, nm[variant], t[t.size() / 10], t[t.size() / 2], t[t.size() * 9 / 10]);
    }
  }
  return 0;
}
