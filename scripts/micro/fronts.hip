// How much HBM bandwidth does a persistent kernel get when every wave streams through ITS OWN region (6144 sequential
// fronts, the access pattern of the fused shuffle / unshuffle tasks and of the per-stream codecs) compared with teams of K
// waves sweeping one region together (6144 / K fronts) and with chunk-granular tickets (one moving front)?
// Build: hipcc --offload-arch=gfx950 -O3 fronts.hip -o fronts ; run: ./fronts
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))
enum { OP_READ = 0, OP_WRITE = 1, OP_COPY = 2 };

template <int OP, int U>
__global__ __launch_bounds__(64, 6) void k(const uint8_t* src_, uint8_t* dst_, uint32_t* ticket, uint32_t nregions, uint32_t region_bytes, uint32_t K, uint32_t* sink) {
  const GAS uint8_t* src = (const GAS uint8_t*)src_;
  GAS uint8_t* dst = (GAS uint8_t*)dst_;
  const uint32_t lane = threadIdx.x;
  const uint32_t chunk = U * 1024u, cpr = region_bytes / chunk;     // chunks per region
  uint32_t acc = 0;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    uint32_t region, c0, cstep;
    if (K == 0u) { if (t >= nregions * cpr) break; region = t / cpr; c0 = t % cpr; cstep = cpr; }   // one chunk per ticket
    else { if (t >= nregions * K) break; region = t / K; c0 = t % K; cstep = K; }
    for (uint32_t c = c0; c < cpr; c += cstep) {
      const size_t off = (size_t)region * region_bytes + (size_t)c * chunk + 16u * lane;
      v4 v[U];
      if (OP != OP_WRITE) {
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = *(const GAS v4*)(src + off + 1024u * u);
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (v4){lane, c, (uint32_t)u, region};
      }
      if (OP == OP_READ) {
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) *(GAS v4*)(dst + off + 1024u * u) = v[u];
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int OP, int U>
static float run(const uint8_t* src, uint8_t* dst, uint32_t* ticket, uint32_t nregions, uint32_t region_bytes, uint32_t K, uint32_t* sink, int waves) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int it = 0; it < 4; it++) {
    hipMemsetAsync(ticket, 0, 4, 0);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<OP, U>), dim3(waves), dim3(64), 0, 0, src, dst, ticket, nregions, region_bytes, K, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (it && ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const size_t total = (size_t)8 << 30;
  uint8_t *src, *dst; uint32_t *ticket, *sink;
  if (hipMalloc(&src, total) != hipSuccess || hipMalloc(&dst, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&ticket, 64); hipMalloc(&sink, 64);
  hipMemset(src, 1, total); hipMemset(dst, 2, total);
  const char* opn[3] = {"read", "write", "copy"};
  const int waves_list[3] = {6144, 4096, 8192};
  for (int wi = 0; wi < 3; wi++) {
    const int waves = waves_list[wi];
    for (uint32_t rb : {1u << 20, 128u << 10}) {
      const uint32_t nreg = (uint32_t)(total / rb);
      for (uint32_t K : {1u, 4u, 16u, 0u}) {
        if (K > rb / 8192u) continue;
        float r = run<OP_READ, 8>(src, dst, ticket, nreg, rb, K, sink, waves);
        float w = run<OP_WRITE, 8>(src, dst, ticket, nreg, rb, K, sink, waves);
        float c = run<OP_COPY, 8>(src, dst, ticket, nreg, rb, K, sink, waves);
        printf("waves %5d region %7u B  team K=%2u%s: read %.3f ms %.2f TB/s | write %.3f ms %.2f TB/s | copy %.3f ms %.2f TB/s (r+w)\n", waves, rb, K, K ? "" : " (chunk tickets)",
               r, total / r / 1e9, w, total / w / 1e9, c, 2.0 * total / c / 1e9);
        fflush(stdout);
      }
    }
    if (wi == 0) {   // bytes in flight per wave: 4 KiB and 16 KiB instead of 8 KiB, own regions
      const uint32_t rb = 1u << 20, nreg = (uint32_t)(total / rb);
      float r4 = run<OP_READ, 4>(src, dst, ticket, nreg, rb, 1, sink, waves), w4 = run<OP_WRITE, 4>(src, dst, ticket, nreg, rb, 1, sink, waves);
      float r16 = run<OP_READ, 16>(src, dst, ticket, nreg, rb, 1, sink, waves), w16 = run<OP_WRITE, 16>(src, dst, ticket, nreg, rb, 1, sink, waves);
      printf("own 1 MiB regions, 4 KiB in flight: read %.2f TB/s write %.2f TB/s;  16 KiB in flight: read %.2f TB/s write %.2f TB/s\n",
             total / r4 / 1e9, total / w4 / 1e9, total / r16 / 1e9, total / w16 / 1e9);
    }
  }
  (void)opn;
  return 0;
}
