// Does a bounded scratch ring keep the decoder's plane-major round trip out of HBM?  
// Traffic model of k_decode_streams on bench19: per 1 MiB destination block 512 KiB of scratch are written, read back and
// 1 MiB of destination is written.  Here every persistent wave does exactly that, with the scratch either
//   ARENA: a distinct 512 KiB per block (4 GiB total: what the engine does today), or
//   RING R: a private ring of R bytes per wave, written and read back in groups of R bytes
// so the scratch footprint is waves x R.  If the Infinity Cache (256 MiB) / L2 (8 x 4 MiB) absorb the round trip, time falls
// from the 3-pass figure towards the 1-pass one as the footprint shrinks.  Also: destination stores plain vs non-temporal.
// Build: hipcc --offload-arch=gfx950 -O3 ring.hip -o ring ; run: ./ring
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t v4 __attribute__((ext_vector_type(4)));
#define GAS __attribute__((address_space(1)))

template <int NT>
__device__ __forceinline__ void st_dst(GAS uint8_t* p, v4 v) {
  if (NT) __builtin_nontemporal_store(v, (GAS v4*)p); else *(GAS v4*)p = v;
}

// group: bytes written to the scratch before they are read back (== ring size for RING, 512 KiB for ARENA)
template <int NT>
__global__ __launch_bounds__(64, 6) void k(uint8_t* scratch_, uint8_t* dst_, uint32_t* ticket, uint32_t nblocks, uint32_t group, int arena, int do_scratch) {
  GAS uint8_t* scratch = (GAS uint8_t*)scratch_;
  GAS uint8_t* dst = (GAS uint8_t*)dst_;
  const uint32_t lane = threadIdx.x;
  const uint32_t SCR_PER_BLOCK = 512u << 10;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= nblocks) break;
    GAS uint8_t* d = dst + ((size_t)t << 20);
    GAS uint8_t* s = arena ? scratch + (size_t)t * SCR_PER_BLOCK : scratch + (size_t)blockIdx.x * group;
    for (uint32_t g0 = 0; g0 < SCR_PER_BLOCK; g0 += group) {
      GAS uint8_t* sg = arena ? s + g0 : s;
      if (do_scratch) {
        for (uint32_t o = 0; o < group; o += 4096u) {           // "decode": 4 KiB per step, 16 B per lane x 4
#pragma unroll
          for (int u = 0; u < 4; u++) *(GAS v4*)(sg + o + 1024u * u + 16u * lane) = (v4){lane, o, (uint32_t)u, t};
        }
        __builtin_amdgcn_s_waitcnt(0);
      }
      for (uint32_t o = 0; o < group; o += 4096u) {             // "unshuffle": read 4 KiB, write 8 KiB
        v4 v[4];
        if (do_scratch) {
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = *(const GAS v4*)(sg + o + 1024u * u + 16u * lane);
        } else {
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = (v4){lane, o, (uint32_t)u, t};
        }
        GAS uint8_t* dd = d + 2u * (size_t)(g0 + o);
#pragma unroll
        for (int u = 0; u < 4; u++) { st_dst<NT>(dd + 2048u * u + 32u * lane, v[u]); st_dst<NT>(dd + 2048u * u + 32u * lane + 16u, v[u]); }
      }
    }
  }
}

template <int NT>
static float run(uint8_t* scratch, uint8_t* dst, uint32_t* ticket, uint32_t nblocks, uint32_t group, int arena, int do_scratch, int waves) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int it = 0; it < 4; it++) {
    hipMemsetAsync(ticket, 0, 4, 0);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<NT>), dim3(waves), dim3(64), 0, 0, scratch, dst, ticket, nblocks, group, arena, do_scratch);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (it && ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t total = (size_t)8 << 30;
  const uint32_t nblocks = (uint32_t)(total >> 20);
  uint8_t *scratch, *dst; uint32_t* ticket;
  if (hipMalloc(&scratch, total / 2) != hipSuccess || hipMalloc(&dst, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&ticket, 64);
  hipMemset(scratch, 1, total / 2); hipMemset(dst, 2, total);
  for (int waves : {6144, 4096, 2048}) {
    float base0 = run<0>(scratch, dst, ticket, nblocks, 512u << 10, 1, 0, waves), base1 = run<1>(scratch, dst, ticket, nblocks, 512u << 10, 1, 0, waves);
    printf("waves %5d  destination only (8 GiB written): plain %.3f ms  nt %.3f ms\n", waves, base0, base1);
    float a0 = run<0>(scratch, dst, ticket, nblocks, 512u << 10, 1, 1, waves), a1 = run<1>(scratch, dst, ticket, nblocks, 512u << 10, 1, 1, waves);
    printf("waves %5d  ARENA 4 GiB scratch:                 plain %.3f ms  nt %.3f ms\n", waves, a0, a1);
    for (uint32_t R : {512u << 10, 256u << 10, 128u << 10, 64u << 10, 32u << 10, 16u << 10, 8u << 10, 4u << 10}) {
      float r0 = run<0>(scratch, dst, ticket, nblocks, R, 0, 1, waves), r1 = run<1>(scratch, dst, ticket, nblocks, R, 0, 1, waves);
      printf("waves %5d  RING %4u KiB per wave = %7.1f MiB:   plain %.3f ms  nt %.3f ms\n", waves, R >> 10, (double)waves * R / 1048576.0, r0, r1);
      fflush(stdout);
    }
  }
  return 0;
}
