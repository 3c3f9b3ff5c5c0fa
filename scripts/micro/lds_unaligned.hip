// Does gfx950 serve byte-unaligned ds_read/ds_write b32/b64/b128 correctly (hipcc emits them for align-1 types)?
// Also measures the rough cost of unaligned vs aligned LDS vector access.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef v4u32 __attribute__((aligned(1))) v4u32_una;
typedef uint64_t __attribute__((aligned(1))) u64una;
typedef uint32_t __attribute__((aligned(1))) u32una;
#define LDSP(T, p) ((__attribute__((address_space(3))) T*)(p))
__global__ void k(const uint32_t* offs, uint8_t* out, int reps, uint64_t* cycles) {
  __shared__ uint8_t ring[16384];
  const int l = threadIdx.x;
  __attribute__((address_space(3))) uint8_t* r = (__attribute__((address_space(3))) uint8_t*)ring;
  for (int i = l; i < 16384; i += 64) r[i] = (uint8_t)(i * 7 + (i >> 8));
  __syncthreads();
  const uint32_t o = offs[l];
  // read 16/8/4 bytes at arbitrary byte offsets, write them at other arbitrary offsets, read back bytewise
  v4u32 a = *LDSP(v4u32_una, r + o);
  uint64_t b = *LDSP(u64una, r + o + 3000);
  uint32_t c = *LDSP(u32una, r + o + 5000);
  __syncthreads();
  *LDSP(v4u32_una, r + 8192 + 40 * l + (o & 7)) = a;
  *LDSP(u64una, r + 8192 + 40 * l + 17 + (o & 3)) = b;
  *LDSP(u32una, r + 8192 + 40 * l + 29 + (o & 1)) = c;
  __syncthreads();
  for (int i = 0; i < 40; i++) out[l * 40 + i] = r[8192 + 40 * l + i];
  for (int i = 0; i < 28; i++) out[2560 + l * 28 + i] = 0;
  // timing: dependent chain of unaligned vs aligned b128 reads
  uint32_t p = o & 8191u;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; i++) { v4u32 x = *LDSP(v4u32_una, r + p); p = (p + x.x * 0 + 33u) & 8191u; a += x; }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  p &= ~15u;
  for (int i = 0; i < reps; i++) { v4u32 x = *LDSP(v4u32, r + p); p = (p + x.x * 0 + 32u) & 8176u; a += x; }
  uint64_t t2 = __builtin_amdgcn_s_memtime();
  if (l == 0) { cycles[0] = t1 - t0; cycles[1] = t2 - t1; }
  out[4352 + l] = (uint8_t)(a.x + a.y + a.z + a.w);
}
int main() {
  std::vector<uint32_t> offs(64);
  for (int l = 0; l < 64; l++) offs[l] = (uint32_t)((l * 37 + (l % 16)) % 2000);
  uint32_t* d_offs; uint8_t* d_out; uint64_t* d_cyc;
  hipMalloc((void**)&d_offs, 256); hipMalloc((void**)&d_out, 8192); hipMalloc((void**)&d_cyc, 16);
  hipMemcpy(d_offs, offs.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_offs, d_out, 1000, d_cyc);
  std::vector<uint8_t> out(8192); uint64_t cyc[2];
  hipMemcpy(out.data(), d_out, 8192, hipMemcpyDeviceToHost); hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost);
  auto ring = [](uint32_t i) { return (uint8_t)(i * 7 + (i >> 8)); };
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    uint8_t exp[40]; for (int i = 0; i < 40; i++) exp[i] = ring(8192 + 40 * l + i);
    const uint32_t o = offs[l];
    for (int i = 0; i < 16; i++) exp[(o & 7) + i] = ring(o + i);
    for (int i = 0; i < 8; i++) exp[17 + (o & 3) + i] = ring(o + 3000 + i);
    for (int i = 0; i < 4; i++) exp[29 + (o & 1) + i] = ring(o + 5000 + i);
    for (int i = 0; i < 40; i++) if (out[l * 40 + i] != exp[i]) { if (bad < 5) printf("lane %d byte %d: got %u want %u\n", l, i, out[l * 40 + i], exp[i]); bad++; }
  }
  printf("lds_unaligned: %s (%d mismatches); 1000 dependent ds_read_b128: unaligned %llu cycles, aligned %llu cycles\n",
         bad ? "WRONG" : "OK", bad, (unsigned long long)cyc[0], (unsigned long long)cyc[1]);
  return bad != 0;
}
