// lat_probe.hip — is the decode kernel's slow / fast placement state visible to SIMPLE kernels run over the same arena?
// A shared library for scripts/placement_latprobe.py:  lat_probe(region, bytes, mode, iters, &ms)  times one launch of
//   mode 0  chase   4096 one-wave workgroups; every wave makes `iters` DEPENDENT reads of 256 contiguous bytes at pseudo-random rows of the region
//   mode 1  rowwr   every wave writes 1 KiB rows one after the other into 128 KiB planes of its own (wave w: planes w, w + 4096, ...) and waits for
//                   each row before the next (the decoder's row flush with its latency exposed)
//   mode 2  rowrd   the same with reads
//   mode 3  copy    grid-stride 16 bytes per lane: first half of the region -> second half (bandwidth)
// Build:  hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o scripts/micro/liblat_probe.so scripts/micro/lat_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(64) void k_chase(const uint8_t* __restrict__ buf, uint32_t rows, int iters, uint32_t* sink) {
  uint32_t r = rnd(blockIdx.x * 2654435761u + 17u), acc = 0;
  for (int i = 0; i < iters; i++) {
    const uint32_t row = (uint32_t)(((uint64_t)r * rows) >> 32);
    const uint32_t v = *(const uint32_t*)(buf + (size_t)row * 256 + threadIdx.x * 4);
    const uint32_t v0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    acc ^= v; r = rnd(r + (v0 & 0xffu) + (uint32_t)i);
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(64) void k_rowwr(uint8_t* __restrict__ buf, uint32_t planes, int nplanes) {
  const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3u, 4u);
  for (int k = 0; k < nplanes; k++) {
    const uint32_t p = (blockIdx.x + (uint32_t)k * gridDim.x) % planes;
    uint8_t* q = buf + (size_t)p * (128u << 10) + threadIdx.x * 16;
    for (int row = 0; row < 128; row++) { *(uint4*)(q + (size_t)row * 1024) = v; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
}
__global__ __launch_bounds__(64) void k_rowrd(const uint8_t* __restrict__ buf, uint32_t planes, int nplanes, uint32_t* sink) {
  uint32_t acc = 0;
  for (int k = 0; k < nplanes; k++) {
    const uint32_t p = (blockIdx.x + (uint32_t)k * gridDim.x) % planes;
    const uint8_t* q = buf + (size_t)p * (128u << 10) + threadIdx.x * 16;
    for (int row = 0; row < 128; row++) { const uint4 v = *(const uint4*)(q + (size_t)row * 1024); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc ^= v.x ^ v.w; }
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

extern "C" int lat_probe(void* region, size_t bytes, int mode, int iters, float* ms) {
  static uint32_t* sink = nullptr;
  if (!sink && hipMalloc((void**)&sink, 256) != hipSuccess) return -1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint8_t* buf = (uint8_t*)region;
  hipEventRecord(e0, 0);
  if (mode == 0) k_chase<<<4096, 64>>>(buf, (uint32_t)(bytes / 256), iters, sink);
  else if (mode == 1) k_rowwr<<<4096, 64>>>(buf, (uint32_t)(bytes >> 17), iters);
  else if (mode == 2) k_rowrd<<<4096, 64>>>(buf, (uint32_t)(bytes >> 17), iters, sink);
  else k_copy<<<2048, 256>>>((const uint4*)buf, (uint4*)(buf + bytes / 2), bytes / 32);
  hipEventRecord(e1, 0);
  const hipError_t e = hipEventSynchronize(e1);
  if (e != hipSuccess) { fprintf(stderr, "lat_probe: %s\n", hipGetErrorString(e)); return -2; }
  hipEventElapsedTime(ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}
