// wave_emu.h — runs wave64 device code of c-blosc_amd/csrc on the CPU.  TEST INFRASTRUCTURE: the product never sees it.
//
// The kernels of this repo are written per wavefront: 64 lanes in lock step that talk to each other through cross-lane
// instructions (ds_bpermute, v_readlane, DPP, ballots).  Here each lane is a coroutine (ucontext); every cross-lane
// intrinsic is a rendezvous: a lane deposits its operand, yields, and picks up what it needs once the scheduler has gone
// round all lanes.  A lane that reaches a DIFFERENT rendezvous than the others (a cross-lane instruction inside divergent
// control flow, which on hardware would read inactive lanes) is reported and aborts the run.
//
// What this models: control flow, arithmetic, LDS / global memory CONTENT, cross-lane data movement.
// What it does not: lock step between rendezvous points.  Code that lets one lane read memory another lane wrote "in the
// same instruction slot" without a cross-lane instruction in between (the LZ decoders' overlapping match copies, the
// LDS strips of the Zstd / Deflate writers) needs an explicit wave_emu::sync() in the source to run here; the LZ4 /
// BloscLZ / LZ4HC encoders read only their input and treat their hash tables as hints, so they run unmodified
// (what they find in a contended bucket may differ from the hardware's choice, never the validity of the output).
//
// Compile device sources with the ROCm clang++ for the host (ext_vector_type, address_space attributes):
//   /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -I tests/tools/wave_emu -x c++ ...
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <signal.h>
#include <unistd.h>

#define BAMD_WAVE_EMU 1
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r = {x, y, z, w}; return r; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r = {x, y}; return r; }
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

namespace wave_emu {

constexpr int LANES = 64;
constexpr int MAXT = 1024;               // fibers of one workgroup (16 wavefronts)
constexpr size_t STACK_BYTES = 256 * 1024;

// One workgroup: `nthreads` fibers, wavefront w = fibers 64 w .. 64 w + 63.  Cross-lane operations meet inside a wavefront,
// __syncthreads() across the workgroup.  (A single wavefront is the workgroup of 64.)
struct Wave {
  ucontext_t sched;
  ucontext_t ctx[MAXT];
  char* stack[MAXT];
  bool done[MAXT];
  int cur;                                // fiber running now
  int nthreads;
  // rendezvous state, double buffered by the parity of the fiber's operation counter
  uint64_t val[2][MAXT];
  uint64_t stamp[2][MAXT];
  const char* site[2][MAXT];
  uint64_t opno[MAXT];
  void (*body)(int lane, void* arg);
  void* arg;
  uint64_t rendezvous;     // statistics
  // __syncthreads
  uint64_t bar_gen; int bar_arrived;
  uint64_t spins;          // consecutive s_sleep yields without progress (a wait nobody will ever satisfy)
};
extern thread_local Wave* g_wave;
extern thread_local dim3 g_block_idx, g_thread_base, g_block_dim, g_grid_dim;

inline int lane() { return g_wave->cur & (LANES - 1); }
inline int wave_base() { return g_wave->cur & ~(LANES - 1); }
inline int fiber() { return g_wave->cur; }

[[noreturn]] inline void fail(const char* what, const char* where) {
  fprintf(stderr, "wave_emu: %s at %s (lane %d)\n", what, where ? where : "?", g_wave ? g_wave->cur : -1);
  abort();
}

inline void yield_lane() {
  Wave* w = g_wave;
  swapcontext(&w->ctx[w->cur], &w->sched);
}

// deposit v, wait for the round, return the operation number (all lanes of the round must show the same one)
inline uint64_t deposit(uint64_t v, const char* where) {
  Wave* w = g_wave;
  const int l = w->cur;
  w->spins = 0;
  const uint64_t k = ++w->opno[l];
  const int b = (int)(k & 1u);
  w->val[b][l] = v; w->stamp[b][l] = k; w->site[b][l] = where;
  yield_lane();
  w->rendezvous++;
  return k;
}
inline uint64_t fetch(uint64_t k, int from, const char* where) {
  Wave* w = g_wave;
  const int b = (int)(k & 1u);
  from = wave_base() + (from & (LANES - 1));
  if (w->stamp[b][from] == k && w->site[b][from] == where) return w->val[b][from];
  if (w->done[from]) return 0;                        // a lane that left before this operation: hardware returns the register's last content; 0 here
  if (w->stamp[b][from] != k || w->site[b][from] != where) fail("cross-lane operation in divergent control flow (source lane is elsewhere)", where);
  return w->val[b][from];
}
inline void check_all(uint64_t k, const char* where) {
  Wave* w = g_wave;
  const int b = (int)(k & 1u);
  for (int l = wave_base(); l < wave_base() + LANES; l++) if ((w->stamp[b][l] != k || w->site[b][l] != where) && !w->done[l]) fail("wave-wide operation in divergent control flow", where);
}

inline uint32_t xlane(uint32_t v, int from, const char* where) { const uint64_t k = deposit(v, where); return (uint32_t)fetch(k, from, where); }
inline uint32_t readfirst(uint32_t v, const char* where) {
  const uint64_t k = deposit(v, where);
  Wave* w = g_wave;
  check_all(k, where);
  for (int l = wave_base(); l < wave_base() + LANES; l++) if (w->stamp[k & 1u][l] == k) return (uint32_t)w->val[k & 1u][l];
  return v;
}
inline uint64_t ballot(bool p, const char* where) {
  const uint64_t k = deposit(p ? 1u : 0u, where);
  Wave* w = g_wave;
  check_all(k, where);
  uint64_t m = 0;
  for (int l = 0; l < LANES; l++) if (w->stamp[k & 1u][wave_base() + l] == k && w->val[k & 1u][wave_base() + l]) m |= 1ull << l;
  return m;
}
inline uint32_t dpp(uint32_t old, uint32_t src, int ctrl, bool bound_ctrl, const char* where) {
  const uint64_t k = deposit(src, where);
  const int l = lane();
  if (ctrl >= 0x111 && ctrl <= 0x11f) {               // row_shr:N
    const int n = ctrl - 0x110;
    if ((l & 15) >= n) return (uint32_t)fetch(k, l - n, where);
    return bound_ctrl ? 0u : old;
  }
  if (ctrl >= 0x101 && ctrl <= 0x10f) {               // row_shl:N
    const int n = ctrl - 0x100;
    if ((l & 15) + n <= 15) return (uint32_t)fetch(k, l + n, where);
    return bound_ctrl ? 0u : old;
  }
  if (ctrl == 0x138) {                                // wave_shr:1 (GFX9): lane l <- lane l-1 across the whole wave
    if (l >= 1) return (uint32_t)fetch(k, l - 1, where);
    return bound_ctrl ? 0u : old;
  }
  if (ctrl >= 0 && ctrl <= 0xff) return (uint32_t)fetch(k, (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3), where);      // quad_perm:[a,b,c,d]
  fail("DPP control not modelled", where);
}
// explicit lock-step point for code that communicates through memory between cross-lane instructions
inline void sync(const char* where = "sync") { const uint64_t k = deposit(0, where); check_all(k, where); }
// __syncthreads(): every fiber of the workgroup that has not returned
inline void syncthreads(const char* where) {
  Wave* w = g_wave;
  w->spins = 0;
  int live = 0;
  for (int t = 0; t < w->nthreads; t++) live += w->done[t] ? 0 : 1;
  const uint64_t gen = w->bar_gen;
  if (++w->bar_arrived >= live) { w->bar_arrived = 0; w->bar_gen++; return; }
  while (w->bar_gen == gen) yield_lane();
  (void)where;
}
// s_sleep inside a wait loop: let the others run; a wait that nothing in this (sequentially emulated) launch can satisfy is reported
inline void sleep_yield(const char* where) {
  Wave* w = g_wave;
  if (++w->spins > 2000000ull) fail("waiting for something no running workgroup will ever produce", where);
  yield_lane();
}

inline uint32_t perm(uint32_t a, uint32_t b, uint32_t sel) {     // v_perm_b32 D = perm({a, b}, sel): bytes 0-3 from b, 4-7 from a
  const uint64_t ab = ((uint64_t)a << 32) | b;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xffu;
    uint32_t byte;
    if (s <= 7u) byte = (uint32_t)(ab >> (8 * s)) & 0xffu;
    else if (s == 0x0cu) byte = 0u;
    else if (s >= 0x0du) byte = 0xffu;
    else { const uint32_t src = (uint32_t)(ab >> (16 * (s - 8u) + 8)) & 0x80u; byte = src ? 0xffu : 0u; }   // 8..11: sign of byte 1,3,5,7
    r |= byte << (8 * i);
  }
  return r;
}

void run(void (*body)(int lane, void* arg), void* arg);      // one wavefront: 64 lanes through `body`
void run_group(int nthreads, void (*body)(int thread, void* arg), void* arg);   // one workgroup of `nthreads` (a multiple of 64 or less) fibers
uint64_t last_rendezvous();

}  // namespace wave_emu

#define WAVE_EMU_STR2(x) #x
#define WAVE_EMU_STR(x) WAVE_EMU_STR2(x)
#define WAVE_EMU_HERE __FILE__ ":" WAVE_EMU_STR(__LINE__)

// ---- the AMDGPU builtins the device code uses ----
#define __builtin_amdgcn_readlane(v, l) ((int)wave_emu::xlane((uint32_t)(v), (int)(l), WAVE_EMU_HERE))
#define __builtin_amdgcn_readfirstlane(v) (wave_emu::readfirst((uint32_t)(v), WAVE_EMU_HERE))
#define __builtin_amdgcn_ds_bpermute(idx, v) ((int)wave_emu::xlane((uint32_t)(v), ((int)(idx) >> 2), WAVE_EMU_HERE))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ((int)wave_emu::dpp((uint32_t)(old), (uint32_t)(src), (ctrl), (bc), WAVE_EMU_HERE))
#define __ballot(p) (wave_emu::ballot((p) != 0, WAVE_EMU_HERE))
#define __shfl_xor(v, m, w) ((int)wave_emu::xlane((uint32_t)(v), wave_emu::lane() ^ (int)(m), WAVE_EMU_HERE))
#define __shfl_up(v, d, w) ((int)wave_emu::xlane((uint32_t)(v), wave_emu::lane() >= (int)(d) ? wave_emu::lane() - (int)(d) : wave_emu::lane(), WAVE_EMU_HERE))
#define __builtin_amdgcn_alignbyte(hi, lo, sh) ((uint32_t)(((((uint64_t)(uint32_t)(hi)) << 32) | (uint32_t)(lo)) >> (8u * ((sh) & 3u))))
#define __builtin_amdgcn_perm(a, b, sel) (wave_emu::perm((a), (b), (sel)))
#define __builtin_amdgcn_mbcnt_lo(mask, add) ((uint32_t)(add) + (uint32_t)__builtin_popcount((uint32_t)(mask) & (wave_emu::lane() >= 32 ? 0xffffffffu : ((1u << wave_emu::lane()) - 1u))))
#define __builtin_amdgcn_mbcnt_hi(mask, add) ((uint32_t)(add) + (uint32_t)__builtin_popcount((uint32_t)(mask) & (wave_emu::lane() < 32 ? 0u : ((1u << (wave_emu::lane() - 32)) - 1u))))
#define __builtin_amdgcn_s_memtime() ((uint64_t)0)
#define __builtin_amdgcn_s_sleep(n) (wave_emu::sleep_yield(WAVE_EMU_HERE))
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
#define __builtin_amdgcn_s_getreg(r) (wave_emu::g_block_idx.x & 7u)     /* HW_REG_XCC_ID: workgroups are dealt round-robin to 8 XCDs */
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define BAMD_WAIT_STORES() ((void)0)
#define BAMD_LDS_SYNC() (wave_emu::sync(WAVE_EMU_HERE))
#define BAMD_MEM_SYNC() (wave_emu::sync(WAVE_EMU_HERE))
#define __syncthreads() (wave_emu::syncthreads(WAVE_EMU_HERE))
#ifndef __HIP_MEMORY_SCOPE_WAVEFRONT
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
// (__hip_atomic_* are clang builtins on every target)
template <typename T, typename V> static inline T atomicAdd(T* p, V v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <typename T> static inline T min(T a, T b) { return b < a ? b : a; }
template <typename T> static inline T max(T a, T b) { return a < b ? b : a; }
template <typename T, typename V> static inline T atomicMin(T* p, V v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }

#define threadIdx (wave_emu::thread_idx())
#define blockIdx (wave_emu::g_block_idx)
namespace wave_emu { inline dim3 thread_idx() { return dim3((unsigned)fiber() + g_thread_base.x, 0, 0); } }
#define blockDim (wave_emu::g_block_dim)
#define gridDim (wave_emu::g_grid_dim)

#ifdef WAVE_EMU_IMPLEMENTATION
namespace wave_emu {
thread_local Wave* g_wave = nullptr;
thread_local dim3 g_block_idx(0, 0, 0), g_thread_base(0, 0, 0), g_block_dim(64, 1, 1), g_grid_dim(1, 1, 1);
static thread_local uint64_t g_last_rendezvous = 0;
uint64_t last_rendezvous() { return g_last_rendezvous; }

// a fault inside emulated device code: say which lane it was and where that lane last met the others
static void on_fault(int sig) {
  Wave* w = g_wave;
  char buf[512];
  const char* where = "?";
  if (w) { const int l = w->cur; const uint64_t k = w->opno[l]; where = w->site[k & 1u][l] ? w->site[k & 1u][l] : "?"; }
  const int n = snprintf(buf, sizeof buf, "wave_emu: signal %d in fiber %d, last rendezvous at %s\n", sig, w ? w->cur : -1, where);
  if (n > 0) { ssize_t r = write(2, buf, (size_t)n); (void)r; }
  _exit(134);
}
static void trampoline(unsigned lo, unsigned hi) {
  Wave* w = (Wave*)(((uint64_t)hi << 32) | lo);
  const int l = w->cur;
  w->body(l, w->arg);
  w->done[l] = true;
  // a fiber that leaves while others wait at a barrier may be the one they were waiting for
  int live = 0;
  for (int t = 0; t < w->nthreads; t++) live += w->done[t] ? 0 : 1;
  if (live > 0 && w->bar_arrived >= live) { w->bar_arrived = 0; w->bar_gen++; }
  swapcontext(&w->ctx[l], &w->sched);
}

static thread_local char* g_stack_pool[MAXT];

void run_group(int nthreads, void (*body)(int thread, void* arg), void* arg) {
  static bool handlers = false;
  if (!handlers) { handlers = true; signal(SIGFPE, on_fault); signal(SIGSEGV, on_fault); signal(SIGBUS, on_fault); }
  if (nthreads < 1 || nthreads > MAXT) { fprintf(stderr, "wave_emu: workgroup of %d threads\n", nthreads); abort(); }
  Wave* w = (Wave*)calloc(1, sizeof(Wave));
  Wave* outer = g_wave;
  g_wave = w;
  w->body = body; w->arg = arg; w->nthreads = nthreads;
  const int padded = (nthreads + LANES - 1) / LANES * LANES;
  for (int l = nthreads; l < padded; l++) w->done[l] = true;        // the idle lanes of a partly filled last wavefront
  for (int l = 0; l < nthreads; l++) {
    if (!g_stack_pool[l]) g_stack_pool[l] = (char*)malloc(STACK_BYTES);
    w->stack[l] = g_stack_pool[l];
    getcontext(&w->ctx[l]);
    w->ctx[l].uc_stack.ss_sp = w->stack[l];
    w->ctx[l].uc_stack.ss_size = STACK_BYTES;
    w->ctx[l].uc_link = nullptr;
    makecontext(&w->ctx[l], (void (*)())trampoline, 2, (unsigned)(uint64_t)w, (unsigned)((uint64_t)w >> 32));
  }
  for (;;) {
    bool any = false;
    for (int l = 0; l < nthreads; l++) {
      if (w->done[l]) continue;
      any = true;
      w->cur = l;
      swapcontext(&w->sched, &w->ctx[l]);
    }
    if (!any) break;
  }
  g_last_rendezvous = w->rendezvous;
  free(w);
  g_wave = outer;
}
void run(void (*body)(int lane, void* arg), void* arg) { run_group(LANES, body, arg); }
}  // namespace wave_emu
#endif
