#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int N> __device__ uint32_t row_shr(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true); }
__global__ void k(uint32_t* out) {
  int lane = threadIdx.x;
  uint32_t v = lane + 1;
  uint32_t incl = v;
  incl += row_shr<1>(incl); incl += row_shr<2>(incl); incl += row_shr<4>(incl); incl += row_shr<8>(incl);
  out[lane] = incl;
  out[64 + lane] = row_shr<1>(v);
}
int main() {
  uint32_t* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  uint32_t h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("incl:"); for (int i = 0; i < 20; i++) printf(" %u", h[i]); printf("\nshr1:"); for (int i = 0; i < 20; i++) printf(" %u", h[64 + i]); printf("\n");
}
