// hip_emu_runtime.h — the few HIP runtime calls the host engine makes (engine.hip, blosc_api.hip), on top of the wavefront emulator:
// "device" memory is host memory, copies are memcpy, a kernel launch runs the kernel's workgroups one after the other, each as a group of
// fibers (wave_emu::run_group).  Sequential workgroups are enough for this code base: its persistent kernels only ever wait for a task
// that sits EARLIER in the same per-XCD queue, and the first workgroup of an XCD drains that queue alone.  TEST INFRASTRUCTURE.
#pragma once
#include <functional>
#include <map>
#include <string.h>

enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
typedef struct hipStreamEmu* hipStream_t;
typedef struct hipEventEmu* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; char gcnArchName[256]; size_t totalGlobalMem; };
#define hipHostMallocDefault 0u

namespace wave_emu {
inline std::map<uintptr_t, size_t>& device_ranges() { static std::map<uintptr_t, size_t> m; return m; }
inline void launch(dim3 grid, dim3 block, const std::function<void()>& kernel) {
  const dim3 save_idx = g_block_idx, save_dim = g_block_dim, save_grid = g_grid_dim;
  g_block_dim = block; g_grid_dim = grid;
  struct Ctx { const std::function<void()>* k; } ctx = {&kernel};
  for (unsigned z = 0; z < grid.z; z++) for (unsigned y = 0; y < grid.y; y++) for (unsigned x = 0; x < grid.x; x++) {
    g_block_idx = dim3(x, y, z);
    run_group((int)(block.x * block.y * block.z), [](int, void* a) { (*((Ctx*)a)->k)(); }, &ctx);
  }
  g_block_idx = save_idx; g_block_dim = save_dim; g_grid_dim = save_grid;
}
}  // namespace wave_emu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) wave_emu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })

inline hipError_t hipMalloc(void** p, size_t n) {
  void* q = aligned_alloc(256, (n + 511) / 256 * 256);
  if (!q) return hipErrorOutOfMemory;
  memset(q, 0xD7, n);                                   // device memory starts out as garbage
  wave_emu::device_ranges()[(uintptr_t)q] = n; *p = q;
  return hipSuccess;
}
inline hipError_t hipFree(void* p) { if (p) { wave_emu::device_ranges().erase((uintptr_t)p); free(p); } return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(256, (n + 511) / 256 * 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "wavefront emulator"); strcpy(p->gcnArchName, "gfx950"); p->multiProcessorCount = 1; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  memset(a, 0, sizeof *a);
  auto& m = wave_emu::device_ranges();
  auto it = m.upper_bound((uintptr_t)p);
  a->type = hipMemoryTypeUnregistered;
  if (it != m.begin()) { --it; if ((uintptr_t)p < it->first + it->second) a->type = hipMemoryTypeDevice; }
  return a->type == hipMemoryTypeDevice ? hipSuccess : hipErrorInvalidValue;
}
