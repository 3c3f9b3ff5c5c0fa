// How long after a kernel's end does the host know?  hipStreamSynchronize against a host thread polling a word in pinned memory that a last tiny kernel
// writes with a system-scope store (round 6: what the 0.1 - 0.15 ms between a call's kernels and its return consist of).
//   hipcc --offload-arch=gfx950 -O2 scripts/sync_latency.hip -o /tmp/sync_latency && /tmp/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_busy(uint64_t cycles, uint32_t* sink) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1u;
}
__global__ void k_flag(volatile uint32_t* flag, uint32_t v) { __hip_atomic_store((uint32_t*)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  uint32_t* flag; (void)hipHostMalloc((void**)&flag, 64, hipHostMallocDefault); *flag = 0;
  uint32_t* sink; (void)hipMalloc((void**)&sink, 64);
  const double ms_list[] = {0.05, 0.5, 3.0, 8.0};
  for (double ms : ms_list) {
    const uint64_t cyc = (uint64_t)(ms * 1e-3 * 100e6);       // wall_clock64 ticks at 100 MHz
    double a = 0, b = 0; const int R = 20;
    for (int r = 0; r < R + 2; r++) {
      double t0 = now_us();
      hipLaunchKernelGGL(k_busy, dim3(256), dim3(64), 0, s, cyc, sink);
      (void)hipStreamSynchronize(s);
      double t1 = now_us();
      const uint32_t v = (uint32_t)(r + 1) + (uint32_t)(ms * 1000) * 100u;
      hipLaunchKernelGGL(k_busy, dim3(256), dim3(64), 0, s, cyc, sink);
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s, flag, v);
      while (*(volatile uint32_t*)flag != v) __builtin_ia32_pause();
      double t2 = now_us();
      if (r >= 2) { a += t1 - t0; b += t2 - t1; }
    }
    printf("kernel of %.2f ms: launch + hipStreamSynchronize %.1f us, launch + flag kernel + polling %.1f us (difference %.1f us)\n", ms, a / R, b / R, (a - b) / R);
  }
  return 0;
}
