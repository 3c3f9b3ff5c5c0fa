// fetch_calib.hip — what does rocprofv3's FETCH_SIZE count for the access shapes of the encode kernel?  (the x 2 correction of
// /opt/skills/guides/MI355X_MICROARCH.md is calibrated for 16-byte-per-lane streaming reads only; k_encode_streams' candidate fetches are 4 + 8 + 8 + 4
// bytes per lane at unrelated places.)  Kernels of KNOWN requested bytes over a buffer much larger than L2 + Infinity Cache:
//   k_stream16   every lane 16 contiguous bytes, wave = 1 KiB contiguous (the guide's calibration case)
//   k_cand24     every lane 24 bytes (ld4 + ld8 + ld8 + ld4, like enc_lz.h: cpre + load20) at a pseudo-random byte offset: one 24-byte
//                object per lane and step, lines not shared between lanes
//   k_rand4      every lane one dword at a pseudo-random 4-aligned offset
//   k_row256     a wave reads 256 contiguous bytes (a dword per lane) at a pseudo-random row: the fused unshuffle's plane loads, the encoder's window
// Run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace  (scripts/r05_call3.sh) and compare the counter (KiB) with `requested` printed here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;
__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_stream16(const uint8_t* __restrict__ buf, size_t bytes, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o + 16 <= bytes; o += (size_t)gridDim.x * blockDim.x * 16) {
    const uint4 v = *(const uint4*)(buf + o); acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_cand24(const uint8_t* __restrict__ buf, size_t bytes, int steps, uint32_t* sink) {
  uint32_t acc = 0; const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int s = 0; s < steps; s++) {
    const size_t o = 4 + (size_t)(((uint64_t)rnd(t * 7919u + (uint32_t)s * 104729u) * (uint64_t)(bytes - 64)) >> 32);
    acc ^= *(const u32u*)(buf + o - 4) ^ (uint32_t)*(const u64u*)(buf + o) ^ (uint32_t)*(const u64u*)(buf + o + 8) ^ *(const u32u*)(buf + o + 16);
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_rand4(const uint8_t* __restrict__ buf, size_t bytes, int steps, uint32_t* sink) {
  uint32_t acc = 0; const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int s = 0; s < steps; s++) {
    const size_t o = ((size_t)(((uint64_t)rnd(t * 7919u + (uint32_t)s * 104729u) * (uint64_t)(bytes - 64)) >> 32)) & ~(size_t)3;
    acc ^= *(const uint32_t*)(buf + o);
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_row256(const uint8_t* __restrict__ buf, size_t bytes, int steps, uint32_t* sink) {
  uint32_t acc = 0; const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  for (int s = 0; s < steps; s++) {
    const size_t o = ((size_t)(((uint64_t)rnd(w * 7919u + (uint32_t)s * 104729u) * (uint64_t)(bytes - 512)) >> 32)) & ~(size_t)255;
    acc ^= *(const uint32_t*)(buf + o + 4 * lane);
  }
  if (acc == 0x12345678u) *sink = acc;
}
int main() {
  const size_t bytes = (size_t)4 << 30;      // 4 GiB: far beyond 32 MiB of L2 and 256 MiB of Infinity Cache
  uint8_t* buf; uint32_t* sink;
  CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&sink, 4)); CK(hipMemset(buf, 1, bytes));
  const int grid = 256 * 16, block = 256, steps = 64;
  const double lanes = (double)grid * block;
  hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(block), 0, 0, buf, bytes, sink);
  hipLaunchKernelGGL(k_cand24, dim3(grid), dim3(block), 0, 0, buf, bytes, steps, sink);
  hipLaunchKernelGGL(k_rand4, dim3(grid), dim3(block), 0, 0, buf, bytes, steps, sink);
  hipLaunchKernelGGL(k_row256, dim3(grid), dim3(block), 0, 0, buf, bytes, steps, sink);
  CK(hipDeviceSynchronize());
  printf("requested KiB: k_stream16 %.0f  k_cand24 %.0f (objects %.0f)  k_rand4 %.0f (dwords %.0f)  k_row256 %.0f (rows %.0f)\n",
         bytes / 1024.0, lanes * steps * 24 / 1024.0, lanes * steps, lanes * steps * 4 / 1024.0, lanes * steps, lanes * steps * 4 / 1024.0, lanes * steps / 64);
  return 0;
}
