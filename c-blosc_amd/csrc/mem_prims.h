// mem_prims.h — address-space-explicit memory helpers for the gfx950 kernels.
//
// Pointers that reach a kernel through a descriptor table are "generic" to the compiler, which
// then emits flat_load/flat_store (slower issue, and they tie vmcnt to lgkmcnt).  Every kernel
// therefore casts them once to address space 1 (global) and uses these helpers, which lower to
// global_load/global_store with arbitrary byte alignment (gfx950 global memory accepts
// unaligned dword..dwordx4 accesses; amdhsa enables unaligned-access-mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bamd {

#define BAMD_GAS __attribute__((address_space(1)))
typedef BAMD_GAS uint8_t gu8;

typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef v4u32 __attribute__((aligned(1))) v4u32_una;
typedef uint32_t __attribute__((aligned(1))) u32una;
typedef uint64_t __attribute__((aligned(1))) u64una;
typedef uint16_t __attribute__((aligned(1))) u16una;

__device__ __forceinline__ const gu8* as_global(const uint8_t* p) { return (const gu8*)p; }
__device__ __forceinline__ gu8* as_global(uint8_t* p) { return (gu8*)p; }

__device__ __forceinline__ uint4 g_ld16(const gu8* p) {
  v4u32 t = *(const BAMD_GAS v4u32_una*)p;
  return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void g_st16(gu8* p, uint4 v) {
  v4u32 t = {v.x, v.y, v.z, v.w};
  *(BAMD_GAS v4u32_una*)p = t;
}
// non-temporal variants (compiler builtins: the nt bit; streaming data that nobody on the chip reads again soon is the first to
// leave the caches).  Hand-written `asm volatile("global_store_dwordx4 ... nt")` stores are NOT an option: the compiler does
// not count them in vmcnt and reuses their data registers while they are in flight (measured: wrong output).
__device__ __forceinline__ void g_st16_nt(gu8* p, uint4 v) {
  v4u32 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, (BAMD_GAS v4u32_una*)p);
}
__device__ __forceinline__ uint4 g_ld16_nt(const gu8* p) {
  v4u32 t = __builtin_nontemporal_load((const BAMD_GAS v4u32_una*)p);
  return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint32_t g_ld4_nt(const gu8* p) { return __builtin_nontemporal_load((const BAMD_GAS u32una*)p); }
__device__ __forceinline__ uint32_t g_ld4(const gu8* p) { return *(const BAMD_GAS u32una*)p; }
__device__ __forceinline__ uint64_t g_ld8(const gu8* p) { return *(const BAMD_GAS u64una*)p; }
__device__ __forceinline__ uint32_t g_ld2(const gu8* p) { return (uint32_t)*(const BAMD_GAS u16una*)p; }
__device__ __forceinline__ void g_st2(gu8* p, uint32_t v) { *(BAMD_GAS u16una*)p = (uint16_t)v; }
__device__ __forceinline__ void g_st4(gu8* p, uint32_t v) { *(BAMD_GAS u32una*)p = v; }
__device__ __forceinline__ void g_st8(gu8* p, uint64_t v) { *(BAMD_GAS u64una*)p = v; }
__device__ __forceinline__ void g_st8_nt(gu8* p, uint64_t v) { __builtin_nontemporal_store(v, (BAMD_GAS u64una*)p); }

// ---- LDS (address space 3): byte-unaligned dword .. dwordx4 accesses are fine on gfx950 (scripts/micro/lds_unaligned.hip) ----
#define BAMD_LAS __attribute__((address_space(3)))
typedef BAMD_LAS uint8_t lu8;
// compiler-level ordering between LDS phases in which lanes read what OTHER lanes wrote (the hardware runs one
// wave's DS operations in order; this keeps the compiler from reordering them on per-thread alias reasoning)
#define LDS_ORDER() asm volatile("" ::: "memory")
__device__ __forceinline__ uint4 l_ld16(const lu8* p) { v4u32 t = *(const BAMD_LAS v4u32_una*)p; return make_uint4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void l_st16(lu8* p, uint4 v) { v4u32 t = {v.x, v.y, v.z, v.w}; *(BAMD_LAS v4u32_una*)p = t; }
__device__ __forceinline__ uint64_t l_ld8(const lu8* p) { return *(const BAMD_LAS u64una*)p; }
__device__ __forceinline__ void l_st8(lu8* p, uint64_t v) { *(BAMD_LAS u64una*)p = v; }
__device__ __forceinline__ uint32_t l_ld4(const lu8* p) { return *(const BAMD_LAS u32una*)p; }
__device__ __forceinline__ void l_st4(lu8* p, uint32_t v) { *(BAMD_LAS u32una*)p = v; }

// every vector memory operation of this wave has completed (stores have reached L2)
#ifndef BAMD_WAIT_STORES        // (the wavefront emulator of tests/tools/wave_emu defines it away)
#define BAMD_WAIT_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

__device__ __forceinline__ int32_t g_ld_i32le(const gu8* p) { return (int32_t)g_ld4(p); }   // device is little endian
__device__ __forceinline__ void g_st_i32le(gu8* p, int32_t v) { g_st4(p, (uint32_t)v); }

}  // namespace bamd
