// engine.hip — host engine implementation (see engine.h).  Compiled together with the kernels.
#include "engine.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>

#include <chrono>
#include "blosc_format.h"
#include "dev_types.h"
#include "queue_order.h"

#include "k_filters.hip"
#include "k_decode.hip"
#include "k_encode.hip"
#include "k_zstd.hip"
#include "k_zstd2.hip"
#include "k_zlib.hip"

namespace bamd {

// ---------------------------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------------------------
static std::atomic<bool> g_warned{false};
#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      fprintf(stderr, "blosc_amd: HIP error '%s' at %s:%d (%s)\n", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
      return -1;                                                                                  \
    }                                                                                             \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Experiment switches of the placement / scheduling studies (scripts/placement_*.py): read ONCE per process - a getenv on a production path races with a
// concurrent setenv of the caller - and clamped: the skew shifts the arena's base inside its allocation, 0 .. 64 MiB.
static size_t arena_skew_bytes() {
#ifdef BAMD_ENV_EVERY_CALL      // (the placement scripts move the skew between re-allocations inside ONE process: make tune NAME=env DEFS=-DBAMD_ENV_EVERY_CALL)
  const
#else
  static const
#endif
  size_t v = [] {
    const char* sk = getenv("BLOSC_AMD_ARENA_SKEW_KIB");
    long k = sk ? atol(sk) : 0;
    if (k < 0) k = 0;
    if (k > 65536) k = 65536;
    return (size_t)k << 10;
  }();
  return v;
}
static bool debug_cost_enabled() {
  static const bool v = getenv("BLOSC_AMD_DEBUG_COST") != nullptr;
  return v;
}
struct DeviceArena {   // one grow-only device allocation carved up per call
  uint8_t* base = nullptr;
  uint8_t* raw = nullptr;   // what hipMalloc returned (base = raw + skew; BLOSC_AMD_ARENA_SKEW_KIB, a placement experiment: scripts/placement_probe.py)
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (raw) { (void)hipFree(raw); raw = base = nullptr; cap = 0; }
    size_t want = align_up(bytes + bytes / 8, 1 << 20);
    const size_t skew = arena_skew_bytes();
    HIP_TRY(hipMalloc((void**)&raw, want + skew));
    base = raw + skew;
    cap = want;
    if (getenv("BLOSC_AMD_DEBUG")) fprintf(stderr, "[blosc_amd] device arena %p, %zu MiB\n", (void*)base, want >> 20);
    return 0;
  }
  void release() { if (raw) (void)hipFree(raw); raw = base = nullptr; cap = 0; }
};
struct PinnedArena {
  uint8_t* base = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (base) { (void)hipHostFree(base); base = nullptr; cap = 0; }
    size_t want = align_up(bytes + bytes / 4, 1 << 16);
    HIP_TRY(hipHostMalloc((void**)&base, want, hipHostMallocDefault));
    cap = want;
    return 0;
  }
  void release() { if (base) (void)hipHostFree(base); base = nullptr; cap = 0; }
};

struct ProfEntry { double ms = 0; int launches = 0; };

// per-launch feedback words of the persistent kernels: [0,256) cycles per plane index, [256] tasks taken by the
// stream kernel, [257] streams taken by the Zstd kernel, [259] streams taken by the Zlib kernel - the host compares them with
// what it queued
constexpr size_t kCostWords = 264;
static int check_done(const uint32_t* fb, size_t expect, size_t expect_zstd, const char* what, size_t expect_zlib = 0) {
  if (fb[256] == expect && fb[257] == expect_zstd && fb[259] == expect_zlib) return 0;
  fprintf(stderr, "blosc_amd: %s: the device took %u of %zu queued tasks (Zstd: %u of %zu, Zlib: %u of %zu) - results discarded\n",
          what, fb[256], expect, fb[257], expect_zstd, fb[259], expect_zlib);
  return -1;
}

struct EngineState {
  std::mutex mu;
  bool device_ok = false;
  int device = -1;
  std::atomic<int> device_hint{-1};   // = device once the context is set up; read without the lock by callers choosing where to queue
  DeviceArena dev;      // descriptors + scratch
  DeviceArena io;       // staging for host-pointer calls
  PinnedArena pin;
  // profiling (switched on for all contexts at once: g_prof)
  std::map<std::string, ProfEntry> prof_acc;
  struct Pending { std::string name; hipEvent_t a, b; };
  std::vector<Pending> prof_pending;
  std::vector<hipEvent_t> ev_pool;
  // Scheduling feedback: cycles the previous batch spent per plane index (stream index inside its block),
  // summed by the kernels.  Streams of one batch differ by 100x in cost and the expensive ones are, call
  // after call, the same byte planes; the queue builders use this to keep them out of the kernels' tails.
  uint32_t enc_cost[256] = {0}, dec_cost[256] = {0};
  bool enc_cost_valid = false, dec_cost_valid = false;
  // false: the device deals workgroups round-robin to 8 XCDs whose ids the kernels can read (SPX-mode MI355X),
  // so per-XCD queues and in-kernel hand-offs through one XCD's L2 are valid.  true: anything else (probe below,
  // or BLOSC_AMD_SINGLE_QUEUE=1) - one queue, shuffle / unshuffle in kernels of their own.
  bool single_queue = false;
  int cus = 0;          // compute units of the selected device (persistent grids)
  // Tables that are a function of a batch's GEOMETRY alone - the block table and the task queues (0.5 MB for 128 chunks of 64 MiB) - stay on the
  // device from one call to the next (round 6).  A call that finds its own block table, fusion flags and SET of expensive planes (order_signature) equal to the ones the tables
  // were made from neither builds the queues nor uploads anything but its chunk descriptors: callers send equal-shaped chunks call after call
  // (bench/bench.c:383 does), and the host side of a call is time the device stands idle.  Buffers of their own: the call arena is overwritten by whatever call comes next.
  struct TableCache {
    std::vector<BlockDesc> blocks; std::vector<uint32_t> modes; std::vector<int> order; int nq = 0;
    DeviceArena tabs; size_t o_queues = 0, o_zqueues = 0, sh_at = 0; int32_t ntasks = 0; bool valid = false;
    bool same(const std::vector<BlockDesc>& b, const std::vector<uint32_t>& m, const std::vector<int>& o, int q) const {
      return valid && nq == q && blocks.size() == b.size() && modes == m && order == o && (b.empty() || memcmp(blocks.data(), b.data(), b.size() * sizeof(BlockDesc)) == 0);
    }
    void drop() { valid = false; tabs.release(); }
  } enc_tabs, dec_tabs;
  hipStream_t own = nullptr;   // host-buffer calls that name no stream run here (non-blocking: see the contexts below)
};

// Contexts (round 3).  blosc_compress_ctx / blosc_decompress_ctx / blosc_getitem are re-entrant in the reference: every call builds
// a context of its own and callers on different threads run side by side (blosc/blosc.c:1288-1305, :1560-1572, :1618-1690).
// A call here needs a workspace - arenas, pinned tables, an event pool, the cost feedback of its last batch - and there are
// ctx_count() of them (BLOSC_AMD_CONTEXTS, default and at most 8 - one per GPU of a node for the multi-GPU calls).  A caller takes the first one that is free, so a single-threaded
// program only ever touches context 0 and never pays for the others; with every context busy a caller queues on one of them in
// turn.  A host-buffer call that names no stream runs on its context's own non-blocking stream: the staging copies of one caller
// overlap the kernels of another instead of lining up on the null stream (the PCIe-bound stock ABI is where that pays).
// Device-pointer calls keep the caller's stream and its ordering.  The selected device is process-wide (g_device); a context
// notices a change the next time it is used.
constexpr int kMaxCtx = 8;
static EngineState g_ctx[kMaxCtx];
static std::atomic<int> g_device{-1};      // -1: whatever device is current when the library is first used
// The device of the CALLING THREAD's calls when >= 0 (engine_thread_device): the multi-GPU entry points run one host thread per
// GPU inside one process, each bound to its device, without touching the process-wide choice other threads rely on.
static thread_local int tl_device = -1;
static bool g_forked = false;
// The device a call of THIS thread runs on: the thread's own binding (multi-GPU entry points), else the process-wide one.  While
// nobody has chosen one, "the current HIP device of the first caller" is resolved HERE and pinned - never "whatever device the context
// that happens to be free lives on": after a _multi call the contexts live on devices 0 .. N-1, and a plain call with device-0
// pointers must not be handed to (and must not silently re-select) another GPU.  -1 only when HIP has no device at all.
static int wanted_device() {
  if (tl_device >= 0) return tl_device;
  int d = g_device.load();
  if (d >= 0 || g_forked) return d;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur < 0) { (void)hipGetLastError(); return -1; }
  int none = -1;
  (void)g_device.compare_exchange_strong(none, cur);
  return g_device.load();
}
static std::atomic<bool> g_prof{false};
static std::atomic<unsigned> g_ctx_turn{0};
static int ctx_count() {
#ifdef BAMD_WAVE_EMU
  return 1;                                // the wavefront emulator (tests/tools) runs one launch at a time
#else
  static const int n = [] { const char* e = getenv("BLOSC_AMD_CONTEXTS"); int v = e ? atoi(e) : kMaxCtx; return v < 1 ? 1 : (v > kMaxCtx ? kMaxCtx : v); }();
  return n;
#endif
}
static std::mutex g_pick_mu;                // context selection is serialised: a probing thread never makes another one miss "its" context
struct CtxGuard {                          // owns one context for the duration of a call
  EngineState* st = nullptr;
  CtxGuard() {
    const int n = ctx_count();
    // a free context that already lives on the device this thread wants (its arenas stay), else a free one that has no device
    // yet, else any free one (it moves: ensure_device), else wait - for one that lives on the wanted device when there is one
    const int want = wanted_device();
    int wait_on = -1;
    {
      std::lock_guard<std::mutex> pick(g_pick_mu);
      for (int pass = 0; pass < 3 && !st; pass++)
        for (int i = 0; i < n && !st; i++) {
          if (!g_ctx[i].mu.try_lock()) continue;
          const bool ok = pass == 2 || (pass == 0 ? (g_ctx[i].device_ok && g_ctx[i].device == want) : !g_ctx[i].device_ok);
          if (ok) st = &g_ctx[i]; else g_ctx[i].mu.unlock();
        }
      if (!st) {
        // every context is busy.  `device` of a busy context is only read as a hint here (its owner may be moving it): a wrong
        // guess costs a migration in ensure_device, never correctness
        const unsigned turn = g_ctx_turn.fetch_add(1u);
        for (int k = 0; k < n && wait_on < 0; k++) { const int i = (int)((turn + (unsigned)k) % (unsigned)n); if (g_ctx[i].device_hint.load(std::memory_order_relaxed) == want) wait_on = i; }
        if (wait_on < 0) wait_on = (int)(turn % (unsigned)n);
      }
    }
    if (!st) { st = &g_ctx[wait_on]; st->mu.lock(); }
  }
  ~CtxGuard() { st->mu.unlock(); }
  CtxGuard(const CtxGuard&) = delete;
  CtxGuard& operator=(const CtxGuard&) = delete;
};

// Where do 64 consecutive workgroups land?  Expected on an SPX-mode MI355X: XCC ids 0..7, eight workgroups each.
__global__ void k_probe_xcc(uint32_t* hist) {
  if (threadIdx.x == 0) atomicAdd(&hist[__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u], 1u);
}
static void probe_topology(EngineState& st) {
  st.single_queue = getenv("BLOSC_AMD_SINGLE_QUEUE") && atoi(getenv("BLOSC_AMD_SINGLE_QUEUE")) != 0;
  if (st.single_queue) return;
  uint32_t* d = nullptr; uint32_t h[16] = {0};
  bool ok = hipMalloc((void**)&d, sizeof h) == hipSuccess && hipMemset(d, 0, sizeof h) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_probe_xcc, dim3(64), dim3(64), 0, 0, d);
    ok = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
  }
  if (d) (void)hipFree(d);
  for (int x = 0; ok && x < 16; x++) ok = h[x] == (x < 8 ? 8u : 0u);
  if (!ok) {
    st.single_queue = true;
    if (getenv("BLOSC_AMD_DEBUG")) fprintf(stderr, "blosc_amd: workgroups are not dealt round-robin to 8 XCDs here; using one task queue and unfused filters\n");
  }
}

// fork(): the reference re-creates its thread pool in the child (blosc/blosc.c:2210-2221 blosc_atfork_child).  A HIP
// context does not survive fork(), so there is nothing to re-create here: the child is marked and every compute call
// in it fails loudly (-1) instead of touching the parent's device state.  prepare/parent keep the context mutexes
// consistent across the fork (a forking thread never inherits one locked by somebody else).
// (g_pick_mu first, as CtxGuard takes it: a child forked while another thread was choosing a context would inherit it locked and hang in its first
//  call instead of failing with the message below)
static void atfork_prepare() { g_pick_mu.lock(); for (int i = 0; i < kMaxCtx; i++) g_ctx[i].mu.lock(); }
static void atfork_parent() { for (int i = kMaxCtx - 1; i >= 0; i--) g_ctx[i].mu.unlock(); g_pick_mu.unlock(); }
static void atfork_child() { for (int i = kMaxCtx - 1; i >= 0; i--) g_ctx[i].mu.unlock(); g_pick_mu.unlock(); g_forked = true; }

static int ensure_device(EngineState& st) {
  if (g_forked) {
    fprintf(stderr, "blosc_amd: this process was forked after the library had initialised its HIP device; a device context does not survive fork() - call exec() or use the library only in the parent\n");
    return -1;
  }
  static std::once_flag atfork_once;
  std::call_once(atfork_once, [] { (void)pthread_atfork(atfork_prepare, atfork_parent, atfork_child); });
  // the HIP current device is per host thread: every entry point (they all come through here, holding a
  // context) re-selects the engine's device for the calling thread
  const int want_dev = wanted_device();
  if (st.device_ok && want_dev == st.device) { HIP_TRY(hipSetDevice(st.device)); return 0; }
  if (st.device_ok) {                      // the process moved to another device (engine_set_device through another context)
    (void)hipSetDevice(st.device);
    st.dev.release(); st.io.release(); st.enc_tabs.drop(); st.dec_tabs.drop();     // arenas, stream and events belong to the device they were created on
    if (st.own) { (void)hipStreamDestroy(st.own); st.own = nullptr; }
    for (auto& p : st.prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    st.prof_pending.clear();
    for (hipEvent_t e : st.ev_pool) (void)hipEventDestroy(e);
    st.ev_pool.clear();
    st.device_ok = false;
  }
  st.device = want_dev;
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0) {
    if (!g_warned.exchange(true))
      fprintf(stderr, "blosc_amd: no usable HIP device (%s); this library has no CPU path\n",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return -1;
  }
  if (st.device >= 0) HIP_TRY(hipSetDevice(st.device));
  else { HIP_TRY(hipGetDevice(&st.device)); int none = -1; (void)g_device.compare_exchange_strong(none, st.device); }
  probe_topology(st);
  hipDeviceProp_t pr;
  const bool have_props = hipGetDeviceProperties(&pr, st.device) == hipSuccess;
  st.cus = (have_props && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  // The in-kernel hand-offs (k_decode.hip: decode_one_stream, k_encode.hip) are "drain stores, relaxed atomic at the XCD's L2" - an argument
  // about gfx942 / gfx950's write-through L1 and one L2 per XCC, outside the HIP memory model.  Any other part gets the single-queue path,
  // which has no in-kernel hand-off at all, whatever the probe above saw.
  if (!st.single_queue && !(have_props && (strncmp(pr.gcnArchName, "gfx950", 6) == 0 || strncmp(pr.gcnArchName, "gfx942", 6) == 0))) {
    st.single_queue = true;
    if (getenv("BLOSC_AMD_DEBUG")) fprintf(stderr, "blosc_amd: %s is not on the allow-list of the relaxed in-kernel hand-off; using one task queue and unfused filters\n", have_props ? pr.gcnArchName : "(unknown device)");
  }
  st.enc_cost_valid = st.dec_cost_valid = false;
  st.device_ok = true;
  st.device_hint.store(st.device, std::memory_order_relaxed);
  return 0;
}

// ---- profiling helpers ------------------------------------------------------------------------
static hipEvent_t prof_event(EngineState& st) {
  if (!st.ev_pool.empty()) { hipEvent_t e = st.ev_pool.back(); st.ev_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
// BLOSC_AMD_HOSTTIME=1: wall time of the host side of the batched calls per phase, printed when the library is released
// (where do the ~0.4 ms per call outside the kernels go?)
struct HostTime { double t[2][6] = {}; long calls[2] = {}; };
static HostTime g_ht;
static bool hosttime_on() { static const bool on = getenv("BLOSC_AMD_HOSTTIME") && atoi(getenv("BLOSC_AMD_HOSTTIME")) != 0; return on; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::mutex g_ht_mu;     // several contexts run at once: the accumulators are shared
#define HT_MARK(dir, i) do { if (hosttime_on()) { const double t_ = now_ms(); { std::lock_guard<std::mutex> l_(g_ht_mu); g_ht.t[dir][i] += t_ - ht_last; } ht_last = t_; } } while (0)
struct ProfScope {
  EngineState& st; hipStream_t s; const char* name; hipEvent_t a{}, b{}; bool on;
  ProfScope(EngineState& st_, hipStream_t s_, const char* n) : st(st_), s(s_), name(n), on(g_prof.load(std::memory_order_relaxed)) {
    if (on) { a = prof_event(st); b = prof_event(st); (void)hipEventRecord(a, s); }
  }
  ~ProfScope() { if (on) { (void)hipEventRecord(b, s); st.prof_pending.push_back({name, a, b}); } }
};
static void prof_collect(EngineState& st) {   // call after the stream has been synchronised
  for (auto& p : st.prof_pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { auto& e = st.prof_acc[p.name]; e.ms += ms; e.launches++; }
    st.ev_pool.push_back(p.a); st.ev_pool.push_back(p.b);
  }
  st.prof_pending.clear();
}

// ---- gather of the 16-byte headers of device-resident chunks ------------------------------------
__global__ void k_gather_headers(const uint8_t* const* __restrict__ srcs, uint8_t* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 16) return;
  out[i] = srcs[i >> 4][i & 15];
}

// ---------------------------------------------------------------------------------------------
// layout helper: sub-allocations inside one arena
// ---------------------------------------------------------------------------------------------
struct Carver {
  size_t off = 0;
  size_t take(size_t bytes, size_t align = 256) { off = align_up(off, align); size_t r = off; off += bytes; return r; }
};

// grid of a persistent one-wave-per-workgroup kernel: as many waves as the device keeps resident
static unsigned persistent_grid(const EngineState& st, size_t nitems, int waves_per_cu) {
  const int cus = st.cus > 0 ? st.cus : 256;
  size_t g = (size_t)cus * (size_t)waves_per_cu;
  if (nitems < g) g = nitems;
  // workgroups are dealt round-robin to the 8 XCDs and every XCD serves only its own queue: never fewer than 8
  return (unsigned)(g < 8 ? 8 : g);
}

static dim3 grid1(size_t n, int per) { return dim3((unsigned)((n + per - 1) / per)); }

// ---------------------------------------------------------------------------------------------
// compress
// ---------------------------------------------------------------------------------------------
// BLOSC_AMD_FUSE=0 keeps the byte (un)shuffle in kernels of its own (k_shuffle / k_unshuffle) instead of
// running it as work of the encode / decode kernels
// BLOSC_AMD_SPANS=0: decoded periodic planes go through the scratch like every other plane
static bool span_enabled() { static const bool on = !(getenv("BLOSC_AMD_SPANS") && atoi(getenv("BLOSC_AMD_SPANS")) == 0); return on; }
// BLOSC_AMD_PERIODIC=0: every plane goes through the match finder (A/B switch for the periodic-plane shortcut of the fused shuffle)
#ifndef BAMD_LZ4HC_DEFAULT
#define BAMD_LZ4HC_DEFAULT 1   // "lz4hc" without BLOSC_AMD_LZ4HC in the environment: 1 = LZ4HC-grade search, 0 = plain LZ4 match finder
#endif
static bool lz4hc_search_enabled() { const char* e = getenv("BLOSC_AMD_LZ4HC"); return e ? atoi(e) != 0 : (BAMD_LZ4HC_DEFAULT != 0); }
#ifndef BAMD_ZSTD_TABLES_DEFAULT
#define BAMD_ZSTD_TABLES_DEFAULT 1   // measured on MI355X (profiles/r03/r03a_encopts_bench_cfg4t.json): bench19 ratio 18.3 -> 23.8 for +5.6 % encode time
#endif
static bool zstd_tables_enabled() { const char* e = getenv("BLOSC_AMD_ZSTD_TABLES"); return e ? atoi(e) != 0 : (BAMD_ZSTD_TABLES_DEFAULT != 0); }
static bool env_flag(const char* name) { const char* e = getenv(name); return e && atoi(e) != 0; }
static bool env_flag_or(const char* name, bool dflt) { const char* e = getenv(name); return e ? atoi(e) != 0 : dflt; }
static bool periodic_enabled() { static const bool on = !(getenv("BLOSC_AMD_PERIODIC") && atoi(getenv("BLOSC_AMD_PERIODIC")) == 0); return on; }
// BLOSC_AMD_ZSTD2: 2 (default) = two-phase path, 16 frames per wave, tables in a global scratch (k_zstd2.hip);
// 1 = the same with the tables in LDS (one wave per CU); 0 = one wave per frame for everything (k_zstd_streams).
// 8 GiB of reference-written frames, mode 0 / 2: bench19 107 / 50 ms, linspace 17.8 / 15.9, random walk 23.7 / 16.5
// (profiles/r02/r02f_zstd_decode_modes.txt)
static int zstd2_mode() { static const int m = getenv("BLOSC_AMD_ZSTD2") ? atoi(getenv("BLOSC_AMD_ZSTD2")) : 2; return m; }
// typesizes whose byte (un)shuffle runs inside the codec kernels (enc_shuffle.h, k_decode.hip: unshuffle_block_wave); the others, and everything
// under BLOSC_AMD_FUSE=0 / BLOSC_AMD_SINGLE_QUEUE=1, go through the stand-alone filter kernels
static bool fused_typesize(int T) { return T >= 2 && T <= 32; }          // 2 / 4 / 8 / 16: register transposes; the others up to 32 (round 4): an LDS tile of the wave
static bool fused_fast_typesize(int T) { return T == 8 || T == 4 || T == 2 || T == 16; }     // what the Zstd / zlib kernels' own-block unshuffle handles
// bitshuffle chunks of these typesizes are (un)shuffled inside the codec kernels as well (round 4: bitshuffle_block_wave_T / bitunshuffle_block_wave)
static bool bitunshuffle_fused_host(int T) { return T == 1 || T == 2 || T == 4 || T == 8; }      // 8: round 5 (float64 + bitshuffle is a mainstream caller setting)
static bool fuse_enabled() { static const bool on = !(getenv("BLOSC_AMD_FUSE") && atoi(getenv("BLOSC_AMD_FUSE")) == 0); return on; }

// the stream a call runs on: the caller's, or - host buffers and no stream named - the context's own
// What the queue builders make of the cost feedback for the stream counts of this batch's blocks: the part of a queue's identity that is not in the block table
static bool table_cache_enabled() { static const bool on = !(getenv("BLOSC_AMD_TABLE_CACHE") && atoi(getenv("BLOSC_AMD_TABLE_CACHE")) == 0); return on; }
static void order_signature(const std::vector<BlockDesc>& blocks, const uint32_t* cost, bool valid, std::vector<int>& sig) {
  bool seen[257] = {false};
  for (const BlockDesc& b : blocks) seen[b.nstreams < 0 ? 0 : (b.nstreams > 256 ? 256 : b.nstreams)] = true;
  sig.clear();
  std::vector<int> order; int nheavy = 0;
  for (int T = 2; T <= 256; T++) {
    if (!seen[T]) continue;
    plane_order(cost, valid && sched_enabled(), T, order, &nheavy);
    // WHICH planes are the expensive ones, not their order: the cheap planes of the benchmark data cost within a few percent of each other and
    // change places from call to call (with the exact order as signature no second call ever found its tables: profiles/r06zo_*); any order of
    // a queue is a correct one, and the one thing the order is there for - the expensive planes first - is what the set says
    std::sort(order.begin(), order.begin() + (nheavy < T ? nheavy : T));
    sig.push_back(T); sig.push_back(nheavy); sig.insert(sig.end(), order.begin(), order.begin() + (nheavy < T ? nheavy : 0));
  }
}

static int call_stream(EngineState& st, bool host_buffers, hipStream_t* stream) {
  if (!host_buffers || *stream != (hipStream_t)0 || ctx_count() == 1) return 0;
  if (!st.own) HIP_TRY(hipStreamCreateWithFlags(&st.own, hipStreamNonBlocking));
  *stream = st.own;
  return 0;
}

int engine_compress_batch(const CompressParams& p, int n, const Job* jobs, int* results, bool device_ptrs,
                          hipStream_t stream) {
  if (n <= 0) return 0;
  CtxGuard ctx;
  EngineState& st = *ctx.st;
  if (ensure_device(st) || call_stream(st, !device_ptrs, &stream)) return -1;

  double ht_last = hosttime_on() ? now_ms() : 0.0; if (hosttime_on()) { std::lock_guard<std::mutex> l_(g_ht_mu); g_ht.calls[0]++; }
  std::vector<ChunkDesc> chunks((size_t)n);
  std::vector<BlockDesc> blocks;
  size_t nstr = 0;                                   // the stream table itself is made on the device (k_encode_plan)
  std::vector<uint8_t> live((size_t)n, 0);
  size_t filt_bytes = 0, stage_bytes = 0, io_src = 0, io_dst = 0;
  int tiles_shuf = 0, tiles_bit = 0;
  bool any_shuf = false, any_bit = false;

  // ---- per-chunk parameter checks and geometry (blosc.c:1062-1145, :1148-1247) ----
  for (int i = 0; i < n; i++) {
    ChunkDesc& c = chunks[(size_t)i];
    memset(&c, 0, sizeof c);
    c.mode = CH_SKIP;
    size_t nbytes = jobs[i].srcsize, destsize = jobs[i].dstsize, typesize = p.typesize;
    if (nbytes > (size_t)kMaxBufferSize) { results[i] = 0; continue; }
    if (destsize < (size_t)kMaxOverhead) { results[i] = 0; continue; }
    if (destsize - kMaxOverhead > nbytes) destsize = nbytes + kMaxOverhead;
    if (p.clevel < 0 || p.clevel > 9) { results[i] = -10; continue; }
    if (p.doshuffle != 0 && p.doshuffle != 1 && p.doshuffle != 2) { results[i] = -10; continue; }
    if (typesize == 0) { results[i] = -10; continue; }
    if (typesize > (size_t)kMaxTypeSize) typesize = 1;
    const int codec = p.codec;
    if (codec != kBloscLZ && codec != kLZ4 && codec != kLZ4HC && codec != kZlib && codec != kZstd) { results[i] = -5; continue; }  // blosc.c:1197-1207 (Snappy: not built)
    const int32_t T = (int32_t)typesize, nb = (int32_t)nbytes;
    const int32_t bs = compute_blocksize(p.clevel, T, nb, p.forced_blocksize, codec, p.splitmode);
    int32_t nblocks = nb / bs;
    const int32_t leftover = nb % bs;
    if (leftover > 0) nblocks++;
    int flags = 0;
    bool memcpyed = (p.clevel == 0) || (nb < kMinBufferSize);
    if (memcpyed) flags |= kFlagMemcpyed;
    if (p.doshuffle == 1) flags |= kFlagShuffle;
    if (p.doshuffle == 2) flags |= kFlagBitShuffle;
    const int split = split_block(codec, T, bs, p.splitmode);
    flags |= (!split) << 4;
    flags |= codec_to_format(codec) << 5;
    if (memcpyed && (size_t)nb + kMaxOverhead > destsize) { results[i] = 0; continue; }  // blosc.c:1254-1257

    c.src = (const uint8_t*)jobs[i].src; c.dst = (uint8_t*)jobs[i].dst;
    c.nbytes = nb; c.cbytes = (int32_t)destsize; c.blocksize = bs; c.typesize = T;
    c.nblocks = nblocks; c.leftover = leftover; c.nsplits = split ? T : 1;
    c.fmt = codec_to_format(codec); c.clevel = (codec == kLZ4HC) ? 9 : p.clevel; c.hdr_flags = flags;
    c.mode = 0;
    c.first_block = (int32_t)blocks.size(); c.first_stream = (int32_t)nstr;
    if (memcpyed) c.mode |= CH_MEMCPYED;
    else if (p.doshuffle == 1 && T > 1) { c.mode |= CH_SHUFFLE; if (fused_typesize(T) && fuse_enabled() && !st.single_queue) c.mode |= CH_FUSED_SHUF; }
    else if (p.doshuffle == 2) { c.mode |= CH_BITSHUFFLE; if (bitunshuffle_fused_host(T) && fuse_enabled() && !st.single_queue) c.mode |= CH_FUSED_SHUF; }
    live[(size_t)i] = 1;
    results[i] = 0;
    if (!device_ptrs) { io_src = align_up(io_src, 256) + (size_t)nb; io_dst = align_up(io_dst, 256) + destsize; }
    const bool filtered = (c.mode & (CH_SHUFFLE | CH_BITSHUFFLE)) != 0;
    if (filtered) {
      const int32_t N = bs / T;
      if (c.mode & CH_FUSED_SHUF) { /* shuffled by tasks of the encode kernel */ }
      else if (c.mode & CH_SHUFFLE) { any_shuf = true; int t = (N + shuffle_tile_elems(T) - 1) / shuffle_tile_elems(T); if (t < 1) t = 1; if (t > tiles_shuf) tiles_shuf = t; }
      else { any_bit = true; int t = (N + bitshuffle_tile_elems(T) - 1) / bitshuffle_tile_elems(T); if (t < 1) t = 1; if (t > tiles_bit) tiles_bit = t; }
    }
    // blocks; their streams (in = the block's bytes in the filtered image or the source, out = its staging slot) are k_encode_plan's
    if (blocks.capacity() < blocks.size() + (size_t)nblocks) blocks.reserve(std::max(blocks.size() + (size_t)nblocks, (size_t)(n - i) * (size_t)nblocks + blocks.size()));   // (equal chunks: one allocation)
    for (int32_t j = 0; j < nblocks; j++) {
      BlockDesc b;
      b.chunk = i; b.blk = j; b.first_stream = (int32_t)nstr;
      const bool last = (j == nblocks - 1) && leftover > 0;
      b.nstreams = memcpyed ? 0 : ((split && !last) ? T : 1);
      b.bsize = last ? leftover : bs; b.flags = 0;
      nstr += (size_t)b.nstreams;
      blocks.push_back(b);
    }
    if (!memcpyed) {
      if (filtered) filt_bytes = align_up(filt_bytes, 256) + (size_t)nb;
      stage_bytes = align_up(stage_bytes, 256) + (size_t)nb;
    }
  }

  const size_t nblk = blocks.size();
  HT_MARK(0, 0);     // per-chunk geometry + block / stream tables
  // ---- device workspace ----
  Carver cv;
  const size_t o_chunks = cv.take(sizeof(ChunkDesc) * (size_t)n);
  const size_t o_streams = cv.take(sizeof(StreamDesc) * (nstr ? nstr : 1));
  const size_t o_blkoff = cv.take(sizeof(int32_t) * (nblk ? nblk : 1));
  const size_t o_results = cv.take(sizeof(int32_t) * (size_t)n + 96);   // + the 8 ticket counters of the encode queues + the 8 of the shuffle lists
  // the block table and the queues: still on the device from the last call of this geometry, or built and uploaded now (EngineState::TableCache)
  EngineState::TableCache& tc = st.enc_tabs;
  const int nq = st.single_queue ? 1 : 8;
  std::vector<uint32_t> tmodes((size_t)n);
  for (int i = 0; i < n; i++) tmodes[(size_t)i] = chunks[(size_t)i].mode & CH_FUSED_SHUF;
  std::vector<int> tsig;
  order_signature(blocks, st.enc_cost, st.enc_cost_valid, tsig);
  const bool tabs_hit = table_cache_enabled() && tc.same(blocks, tmodes, tsig, nq);
  if (debug_cost_enabled()) fprintf(stderr, "[blosc_amd] compress: block table and queues %s\n", tabs_hit ? "still on the device" : "built and uploaded");
  std::vector<int32_t> queues; size_t sh_at = tc.sh_at;
  Carver tcv;
  const size_t t_blocks = tcv.take(sizeof(BlockDesc) * (nblk ? nblk : 1));
  if (!tabs_hit) {
    tc.valid = false;
    build_encode_queues(blocks, chunks, st.enc_cost, st.enc_cost_valid, queues, nq, &sh_at);
    tc.o_queues = tcv.take(sizeof(int32_t) * queues.size());
    if (tc.tabs.ensure(tcv.off)) return -1;
    tc.sh_at = sh_at; tc.ntasks = queues[8];
  }
  const size_t o_ready = cv.take(sizeof(uint32_t) * (nblk ? nblk : 1));
  const size_t o_cost = cv.take(sizeof(uint32_t) * kCostWords);
  const size_t o_filt = cv.take(filt_bytes + 256);
  const size_t o_stage = cv.take(stage_bytes + 256);
  // Zstd: the predefined FSE tables and one sequence scratch per persistent wave
  const bool zstd = p.codec == kZstd, zlibc = p.codec == kZlib;
  // "lz4hc": the LZ4HC-grade search of k_encode.hip (lz4hc_encode_wave); BLOSC_AMD_LZ4HC=0 serves the name with the plain LZ4
  // match finder at its highest effort instead (read per call, so that a test can compare the two in one process)
  const bool hc = p.codec == kLZ4HC && lz4hc_search_enabled();
  // Zstd: sequence tables made per block (k_encode.hip: zt_make_tables) instead of the predefined ones; opt-in (BLOSC_AMD_ZSTD_TABLES=1)
  // until it has been timed on the device, read per call
  const bool ztab = zstd && zstd_tables_enabled();
  // the LZ4HC-grade search in front of the Zstd writer (with per-block tables) / the zlib writer: BLOSC_AMD_ZSTD_SEARCH=1, BLOSC_AMD_ZLIB_SEARCH=1
  // Defaults after the device timings of round 3 (profiles/r03/r03a_encopts_bench_*.json, 8 GiB bench19): Zstd - the search costs 2.6 x the
  // encode time (35.8 -> 94.9 ms) for ratio 23.8 -> 35.1, so it serves the upper clevels (the reference maps clevel >= 6 to its
  // lazy / optimal strategies, blosc.c:502-504 + clevels.h) and stays off at the default clevel; zlib - whoever names zlib wants its
  // ratio: search + dynamic codes give 73.4 (reference 47.4, fixed codes without search 40.5) at 57 ms per 8 GiB, still 150 GB/s.
  const bool zsearch = (zstd && env_flag_or("BLOSC_AMD_ZSTD_SEARCH", p.clevel >= 6)) || (zlibc && env_flag_or("BLOSC_AMD_ZLIB_SEARCH", true));
  // Huffman-coded literals (with the per-block tables, or tables + search): BLOSC_AMD_ZSTD_HUFFMAN=1 on top of either switch
  const bool zhuf = zstd && (ztab || zsearch) && env_flag("BLOSC_AMD_ZSTD_HUFFMAN");
  const int enc_wpc_lz = (zstd || zlibc) ? ENC_WAVES_PER_CU : ENC_LZ_WAVES_PER_CU;
  const int enc_wpc = zsearch ? (160 * 1024) / (HC_TAB_BYTES + ZS_LDS_BYTES) : (hc ? HC_WAVES_PER_CU : enc_wpc_lz);   // what fits into a CU's LDS
  const bool zdyn = zlibc && env_flag_or("BLOSC_AMD_ZLIB_DYNAMIC", true);      // zlib with dynamic Huffman codes: two passes, the tokens in the sequence scratch
  const size_t zwaves = (zstd || zdyn) ? (size_t)(st.cus > 0 ? st.cus : 256) * (size_t)enc_wpc : 0;
  const size_t o_ctabs = cv.take(sizeof(zenc::CTabs) + 64);
  const size_t o_seqbufs = cv.take(zwaves * (zdyn ? (size_t)ZD_SCRATCH_U64 : (size_t)ZS_SEQCAP) * sizeof(uint64_t) + 64);
  if (st.dev.ensure(cv.off)) return -1;
  uint8_t* D = st.dev.base;
  uint8_t *io_s = nullptr, *io_d = nullptr;
  if (!device_ptrs) {
    if (st.io.ensure(align_up(io_src + 256, 256) + io_dst + 512)) return -1;
    io_s = st.io.base; io_d = st.io.base + align_up(io_src + 256, 256);
  }
  if (zstd) {
    static zenc::CTabs host_tabs;
    static bool built = false;
    if (!built) { zenc::build_predefined(host_tabs); built = true; }
    HIP_TRY(hipMemcpyAsync(D + o_ctabs, &host_tabs, sizeof host_tabs, hipMemcpyHostToDevice, stream));
  }
  // ---- patch pointers ----
  {
    size_t fo = 0, so = 0, is = 0, id = 0;
    for (int i = 0; i < n; i++) {
      if (!live[(size_t)i]) continue;
      ChunkDesc& c = chunks[(size_t)i];
      if (!device_ptrs) {
        is = align_up(is, 256); id = align_up(id, 256);
        HIP_TRY(hipMemcpyAsync(io_s + is, jobs[i].src, (size_t)c.nbytes, hipMemcpyHostToDevice, stream));
        c.src = io_s + is; c.dst = io_d + id;
        is += (size_t)c.nbytes; id += (size_t)c.cbytes;
      }
      if (c.mode & CH_MEMCPYED) continue;
      const bool filtered = (c.mode & (CH_SHUFFLE | CH_BITSHUFFLE)) != 0;
      if (filtered) { fo = align_up(fo, 256); c.filt = D + o_filt + fo; fo += (size_t)c.nbytes; }
      so = align_up(so, 256); c.stage = D + o_stage + so; so += (size_t)c.nbytes;
    }
  }
  HT_MARK(0, 1);     // queues, workspace, pointer patching
  // ---- upload tables ----
  Carver pc;
  const size_t p_chunks = pc.take(sizeof(ChunkDesc) * (size_t)n);
  const size_t p_blocks = pc.take(tabs_hit ? 0 : sizeof(BlockDesc) * (nblk ? nblk : 1));
  const size_t p_results = pc.take(sizeof(int32_t) * (size_t)n);
  const size_t p_queues = pc.take(sizeof(int32_t) * queues.size());
  const size_t p_cost = pc.take(sizeof(uint32_t) * kCostWords);
  if (st.pin.ensure(pc.off)) return -1;
  uint8_t* P = st.pin.base;
  uint8_t* TB = tc.tabs.base;
  memcpy(P + p_chunks, chunks.data(), sizeof(ChunkDesc) * (size_t)n);
  if (!tabs_hit) {
    if (nblk) memcpy(P + p_blocks, blocks.data(), sizeof(BlockDesc) * nblk);
    memcpy(P + p_queues, queues.data(), sizeof(int32_t) * queues.size());
    HIP_TRY(hipMemcpyAsync(TB + tc.o_queues, P + p_queues, sizeof(int32_t) * queues.size(), hipMemcpyHostToDevice, stream));
    if (nblk) HIP_TRY(hipMemcpyAsync(TB + t_blocks, P + p_blocks, sizeof(BlockDesc) * nblk, hipMemcpyHostToDevice, stream));
  }
  // results + tickets | block-ready flags | cost words lie back to back in the workspace (taken in that order above): one fill for the three of them
  HIP_TRY(hipMemsetAsync(D + o_results, 0, (o_cost + sizeof(uint32_t) * kCostWords) - o_results, stream));
  HIP_TRY(hipMemcpyAsync(D + o_chunks, P + p_chunks, sizeof(ChunkDesc) * (size_t)n, hipMemcpyHostToDevice, stream));
  uint32_t* d_ticket = (uint32_t*)(D + o_results + sizeof(int32_t) * (size_t)n + 32);

  ChunkDesc* d_chunks = (ChunkDesc*)(D + o_chunks);
  BlockDesc* d_blocks = (BlockDesc*)(TB + t_blocks);
  StreamDesc* d_streams = (StreamDesc*)(D + o_streams);
  int32_t* d_blkoff = (int32_t*)(D + o_blkoff);
  int32_t* d_results = (int32_t*)(D + o_results);

  HT_MARK(0, 2);     // table copies to pinned memory + upload enqueues
  // ---- pipeline ----
  if (nstr) hipLaunchKernelGGL(k_encode_plan, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, stream, d_chunks, d_blocks, d_streams, (int)nblk);
  if (any_shuf && nblk) {
    ProfScope ps(st, stream, "k_shuffle");
    hipLaunchKernelGGL(k_shuffle, dim3((unsigned)nblk, (unsigned)tiles_shuf), dim3(FT_THREADS), 0, stream, d_chunks, d_blocks);
  }
  if (any_bit && nblk) {
    const bool bitfast = true;       // full tiles of typesize 1 / 2 / 4 / 8 through k_bitfilter_fast, the rest through the generic kernel
    ProfScope ps(st, stream, "k_bitshuffle");
    if (bitfast) hipLaunchKernelGGL(k_bitfilter_fast<0>, dim3((unsigned)nblk, (unsigned)tiles_bit), dim3(FT_THREADS), 0, stream, d_chunks, d_blocks);
    hipLaunchKernelGGL(k_bitshuffle, dim3((unsigned)nblk, (unsigned)tiles_bit), dim3(FT_THREADS), 0, stream, d_chunks, d_blocks, bitfast ? 1 : 0);
  }
  if (nstr) {
    ProfScope ps(st, stream, zstd ? "k_zstd_encode" : (zlibc ? "k_zlib_encode" : (hc ? "k_lz4hc_encode" : "k_encode_streams")));
    const int32_t* d_qoff = (const int32_t*)(TB + tc.o_queues); const int32_t* d_qlist = d_qoff + 9;
    uint32_t* d_ready = (uint32_t*)(D + o_ready);
    const size_t ntasks = (size_t)tc.ntasks;
    const int32_t* d_shoff = d_qoff + sh_at;
    uint64_t* d_seqbufs = (zstd || zdyn) ? (uint64_t*)(D + o_seqbufs) : nullptr;
    const zenc::CTabs* d_ctabs = zstd ? (const zenc::CTabs*)(D + o_ctabs) : nullptr;
    const int detect = (!zstd && !zlibc && periodic_enabled()) ? 1 : 0;
    const dim3 grid(persistent_grid(st, ntasks, enc_wpc)), block(64 * ENC_WAVES);
#ifdef BAMD_PROFILE_DECODE
    uint32_t* d_prof = nullptr;
    if (getenv("BLOSC_AMD_ENC_PROFILE")) { (void)hipMalloc((void**)&d_prof, nstr * 64); (void)hipMemsetAsync(d_prof, 0, nstr * 64, stream); }
#define BAMD_ENC_LAUNCH(MODE) hipLaunchKernelGGL(k_encode_streams_t<MODE>, grid, block, 0, stream, d_streams, d_ticket, d_qlist, d_qoff, d_shoff, d_chunks, d_blocks, d_ready, (uint32_t*)(D + o_cost), st.single_queue ? 1 : 0, d_seqbufs, d_ctabs, detect, d_prof)
#else
#define BAMD_ENC_LAUNCH(MODE) hipLaunchKernelGGL(k_encode_streams_t<MODE>, grid, block, 0, stream, d_streams, d_ticket, d_qlist, d_qoff, d_shoff, d_chunks, d_blocks, d_ready, (uint32_t*)(D + o_cost), st.single_queue ? 1 : 0, d_seqbufs, d_ctabs, detect)
#endif
    if (zstd && zsearch) { if (zhuf) BAMD_ENC_LAUNCH(ENC_ZSTD_HCH); else BAMD_ENC_LAUNCH(ENC_ZSTD_HC); }
    else if (zstd && ztab) { if (zhuf) BAMD_ENC_LAUNCH(ENC_ZSTD_TH); else BAMD_ENC_LAUNCH(ENC_ZSTD_T); }
    else if (zstd) BAMD_ENC_LAUNCH(ENC_ZSTD);
    else if (zlibc && zdyn) { if (zsearch) BAMD_ENC_LAUNCH(ENC_ZLIB_DYN_HC); else BAMD_ENC_LAUNCH(ENC_ZLIB_DYN); }
    else if (zlibc && zsearch) BAMD_ENC_LAUNCH(ENC_ZLIB_HC);
    else if (zlibc) BAMD_ENC_LAUNCH(ENC_ZLIB);
    else if (hc) BAMD_ENC_LAUNCH(ENC_HC);
    else BAMD_ENC_LAUNCH(ENC_LZ);
#undef BAMD_ENC_LAUNCH
#ifdef BAMD_PROFILE_DECODE
    if (d_prof) {
      std::vector<uint32_t> h(nstr * 16);
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpy(h.data(), d_prof, nstr * 64, hipMemcpyDeviceToHost);
      FILE* f = fopen(getenv("BLOSC_AMD_ENC_PROFILE"), "wb");
      if (f) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
      (void)hipFree(d_prof);
    }
#endif
  }
  {
    ProfScope ps(st, stream, "k_chunk_scan");
    hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)n), dim3(SCAN_THREADS), 0, stream, d_chunks, d_blocks, d_streams, d_blkoff, d_results);
  }
  if (nblk) {
    ProfScope ps(st, stream, "k_chunk_compact");
    hipLaunchKernelGGL(k_chunk_compact, dim3((unsigned)nblk), dim3(COMPACT_THREADS), 0, stream, d_chunks, d_blocks, d_streams, d_blkoff);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(P + p_results, d_results, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(P + p_cost, D + o_cost, sizeof(uint32_t) * kCostWords, hipMemcpyDeviceToHost, stream));
  HT_MARK(0, 3);     // kernel launches
  HIP_TRY(hipStreamSynchronize(stream));
  HT_MARK(0, 4);     // waiting for the device
  prof_collect(st);
  if (nstr && check_done((const uint32_t*)(P + p_cost), (size_t)tc.ntasks, 0, "compress")) return -1;
  if (!tabs_hit) { tc.blocks.swap(blocks); tc.modes.swap(tmodes); tc.order.swap(tsig); tc.nq = nq; tc.valid = true; }      // (only now: the uploads are known to have arrived)
  if (nstr >= 4096) { memcpy(st.enc_cost, P + p_cost, sizeof st.enc_cost); st.enc_cost_valid = true; }   // small calls say little
  const int32_t* r = (const int32_t*)(P + p_results);
  for (int i = 0; i < n; i++) if (live[(size_t)i]) results[i] = r[i];
  if (!device_ptrs) {
    for (int i = 0; i < n; i++) {
      if (!live[(size_t)i] || results[i] <= 0) continue;
      HIP_TRY(hipMemcpyAsync(jobs[i].dst, chunks[(size_t)i].dst, (size_t)results[i], hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// decompress
// ---------------------------------------------------------------------------------------------
// header validation shared by decompress and getitem; returns 1 = go on, else *res holds the result
static int classify_for_decompress(const Header& h, size_t srcsize, size_t destsize, int* res, int* fmt) {
  if (h.nbytes == 0) { *res = 0; return 0; }                                   // blosc.c:1463-1466
  if (h.blocksize <= 0 || (size_t)h.blocksize > destsize || h.blocksize > kMaxBlockSize || h.typesize <= 0) { *res = -1; return 0; }
  if (h.version != kVersionFormat) { *res = -1; return 0; }                    // blosc.c:1474-1477
  if (h.flags & kFlagReserved) { *res = -1; return 0; }                        // blosc.c:1478-1481
  // A NEGATIVE nbytes (bit 31 set: damaged input only) is not "> destsize" for the reference's signed comparison (blosc.c:1490); it then counts
  // nbytes / blocksize <= 0 blocks (C's truncating division, the leftover is <= 0: blosc.c:1485-1487), passes or fails the remaining header
  // checks with that count, runs no block and returns 0 with nothing written (serial_blosc's loop, blosc.c:814).  Same here (round 6; rounds 1 - 5
  // answered -1).
  const bool neg = h.nbytes < 0;
  if (!neg && (size_t)h.nbytes > destsize) { *res = -1; return 0; }            // blosc.c:1490-1492
  if (srcsize && (h.cbytes < 0 || (size_t)h.cbytes > srcsize)) { *res = -1; return 0; }  // extension: caller told us the buffer size
  if (h.flags & kFlagMemcpyed) {
    if ((int32_t)((uint32_t)h.nbytes + (uint32_t)kMaxOverhead) != h.cbytes) { *res = -1; return 0; }   // blosc.c:1494-1499
    *fmt = 0;
    if (neg) { *res = 0; return 0; }
    return 1;
  }
  const int f = (h.flags & 0xe0) >> 5;                                         // blosc.c:525-574
  if (f != FMT_BLOSCLZ && f != FMT_LZ4 && f != FMT_ZLIB && f != FMT_ZSTD) { *res = -5; return 0; }   // Snappy: not built, like a stock build without it
  if (h.versionlz != 1) { *res = -9; return 0; }
  *fmt = f;
  int32_t nblocks = h.nbytes / h.blocksize + ((h.nbytes % h.blocksize) > 0 ? 1 : 0);
  if (nblocks > (h.cbytes - 16) / 4) { *res = -1; return 0; }                  // blosc.c:1504-1507
  if (neg) { *res = 0; return 0; }                                             // nblocks <= 0: nothing runs
  return 1;
}

static int fetch_headers(EngineState& st, int n, const Job* jobs, bool device_ptrs, hipStream_t stream,
                         std::vector<Header>& hdrs) {
  hdrs.resize((size_t)n);
  if (!device_ptrs) {
    for (int i = 0; i < n; i++) hdrs[(size_t)i] = parse_header((const uint8_t*)jobs[i].src);
    return 0;
  }
  Carver cv;
  const size_t o_ptrs = cv.take(sizeof(void*) * (size_t)n);
  const size_t o_hdr = cv.take(16 * (size_t)n);
  if (st.dev.ensure(cv.off)) return -1;
  if (st.pin.ensure(cv.off)) return -1;
  const void** pp = (const void**)(st.pin.base + o_ptrs);
  // an entry whose caller-stated size cannot hold a header is never dereferenced: it reads the (zeroed) slot 0 of
  // the header area instead and is rejected by classify_for_decompress (cbytes 0 > ... version 0)
  bool any_short = false;
  for (int i = 0; i < n; i++) { const bool sh = jobs[i].srcsize && jobs[i].srcsize < (size_t)kMaxOverhead; any_short |= sh; pp[i] = sh ? (const void*)(st.dev.base + o_hdr) : jobs[i].src; }
  if (any_short) HIP_TRY(hipMemsetAsync(st.dev.base + o_hdr, 0, 16, stream));
  // the kernel reads the pointer table from, and writes the headers to, the pinned (device-mapped) table memory itself: one device operation in
  // front of the synchronisation instead of three (late in round 6; the call's host side is time the device stands idle)
  hipLaunchKernelGGL(k_gather_headers, grid1((size_t)n * 16, 256), dim3(256), 0, stream,
                     (const uint8_t* const*)(st.pin.base + o_ptrs), st.pin.base + o_hdr, n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(stream));
  for (int i = 0; i < n; i++) hdrs[(size_t)i] = parse_header(st.pin.base + o_hdr + 16 * (size_t)i);
  for (int i = 0; i < n; i++) if (jobs[i].srcsize && jobs[i].srcsize < (size_t)kMaxOverhead) { hdrs[(size_t)i] = Header{}; hdrs[(size_t)i].nbytes = -1; hdrs[(size_t)i].version = -1; }
  return 0;
}

// Builds the tables for decoding blocks [j0, j1) of one validated chunk.
static void add_decode_chunk(const Header& h, int fmt, int chunk_index, int32_t j0, int32_t j1,
                             ChunkDesc& c, std::vector<BlockDesc>& blocks, size_t& nstreams) {
  memset(&c, 0, sizeof c);
  const int32_t T = h.typesize, bs = h.blocksize;
  c.nbytes = h.nbytes; c.cbytes = h.cbytes; c.blocksize = bs; c.typesize = T;
  c.nblocks = h.nbytes / bs + ((h.nbytes % bs) ? 1 : 0);
  c.leftover = h.nbytes % bs;
  c.fmt = fmt;
  const bool dont_split = (h.flags & kFlagDontSplit) != 0;
  const bool split = !dont_split && T <= kMaxSplits && bs / T >= kMinBufferSize;   // blosc.c:749-757
  c.nsplits = split ? T : 1;
  c.mode = 0;
  if (h.flags & kFlagMemcpyed) c.mode |= CH_MEMCPYED;
  else if ((h.flags & kFlagShuffle) && T > 1) c.mode |= CH_SHUFFLE;            // blosc.c:739-741
  else if (h.flags & kFlagBitShuffle) c.mode |= CH_BITSHUFFLE;
  c.first_block = (int32_t)blocks.size();
  c.first_stream = (int32_t)nstreams;
  if (c.mode & CH_MEMCPYED) return;
  for (int32_t j = j0; j < j1; j++) {
    BlockDesc b;
    b.chunk = chunk_index; b.blk = j; b.first_stream = (int32_t)nstreams;
    const bool last = (j == c.nblocks - 1) && c.leftover > 0;
    b.nstreams = (split && !last) ? T : 1;
    b.bsize = last ? c.leftover : bs; b.flags = fmt == FMT_ZSTD ? BLK_Z : (fmt == FMT_ZLIB ? (BLK_Z | BLK_ZLIB) : 0);   // BLK_Z: not in k_decode_streams' queues
    nstreams += (size_t)b.nstreams;
    blocks.push_back(b);
  }
}

// Deals whole blocks round-robin to 8 per-XCD queues; queue x lists the stream ids of its blocks.
// Layout: qoff[9] (int32) followed by qlist[nstr].
struct DecodeLaunch {
  ChunkDesc* d_chunks; BlockDesc* d_blocks; StreamDesc* d_streams; int32_t* d_status; uint32_t* d_ticket; uint32_t* d_blkdone;
  uint32_t* d_spans; uint8_t* d_pat;             // periodic spans of the fused unshuffle (k_decode.hip: SpanCtx)
  uint32_t* d_cost;                              // [256] cycles per plane index (scheduling feedback)
  uint32_t* d_zticket; bool any_zstd;            // Zstd frames: k_zstd_entropy + k_zstd_exec (two-phase), the rest through k_zstd_streams
  bool any_zlib;                                 // zlib streams: k_zlib_streams with per-XCD queues of its own (ticket words d_zticket[8..15])
  const int32_t* d_zqlist; const int32_t* d_zqoff; size_t nstr_zlib;
  ZMeta* d_zmeta; ptrdiff_t zseq_delta;          // nullptr: everything through k_zstd_streams
  ZgLds* d_zgscr;                                // table scratch of the global two-phase variant (nullptr: not allocated)
  ZcTab* d_zctab;                                // its 16-bit sequence tables, one dense record per frame (k_zstd_seq; nullptr: the 32-bit ones inside d_zgscr)
  const int32_t* d_qlist; const int32_t* d_qoff;   // per-XCD stream queues
  size_t nblk, nstr; int nchunks;
  bool any_shuf, any_bit, any_copy; int tiles_shuf, tiles_bit;
  size_t nstr_queued;                              // streams left to k_decode_streams
};

static int launch_decode(EngineState& st, const DecodeLaunch& L, hipStream_t stream) {
  if (L.nblk) {
    {
      ProfScope ps(st, stream, "k_decode_plan");
      hipLaunchKernelGGL(k_decode_plan, grid1(L.nblk, 256), dim3(256), 0, stream, L.d_chunks, L.d_blocks, L.d_streams, L.d_status, (int)L.nblk);
    }
    if (L.nstr_queued) {
      ProfScope ps(st, stream, "k_decode_streams");
      const dim3 dgrid(persistent_grid(st, L.nstr_queued ? L.nstr_queued : 1, DEC_WAVES_PER_CU));
#ifdef BAMD_PROFILE_DECODE
      uint32_t* d_prof = nullptr;
      if (getenv("BLOSC_AMD_DEC_PROFILE")) { (void)hipMalloc((void**)&d_prof, L.nstr * 64); (void)hipMemsetAsync(d_prof, 0, L.nstr * 64, stream); }
      hipLaunchKernelGGL(k_decode_streams, dgrid, dim3(64 * DEC_WAVES), 0, stream, L.d_streams, L.d_status, L.d_ticket, L.d_qlist, L.d_qoff, L.d_chunks, L.d_blocks, L.d_blkdone, L.d_spans, L.d_pat, L.d_cost, st.single_queue ? 1 : 0, d_prof);
      if (d_prof) {
        std::vector<uint32_t> h(L.nstr * 16);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), d_prof, L.nstr * 64, hipMemcpyDeviceToHost);
        FILE* f = fopen(getenv("BLOSC_AMD_DEC_PROFILE"), "wb");
        if (f) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
        (void)hipFree(d_prof);
      }
#else
      hipLaunchKernelGGL(k_decode_streams, dgrid, dim3(64 * DEC_WAVES), 0, stream, L.d_streams, L.d_status, L.d_ticket, L.d_qlist, L.d_qoff, L.d_chunks, L.d_blocks, L.d_blkdone, L.d_spans, L.d_pat, L.d_cost, st.single_queue ? 1 : 0);
#endif
    }
    // single-block frames through k_zstd_entropy (16 frames per wave) + k_zstd_exec (zstd2_mode above); every other frame
    // shape is left to k_zstd_streams.
    // (Round 4 sent the batch through in 4 / 8 slices, every slice's entropy -> seq -> exec chain on a stream of its own with the entropy
    //  kernels in slice order, so that one slice's sequence chains run underneath the other slices' phases: 30.2 -> 28.5 ms on the reference
    //  frames of config 4, 11.0 -> 13.4 ms on linspace.  The timeline (profiles/r04/r04zp_*): the phases do overlap, but every kernel is slower
    //  in company - k_zstd_seq is bound by its scattered table reads, not by an idle chip - and 8 streams share 4 hardware queues.  Not kept.)
    const int zstd2 = zstd2_mode();
    const uint32_t* d_taken = nullptr;
    if (L.any_zstd && L.d_zmeta && zstd2) {
      {
        ProfScope ps(st, stream, "k_zstd_entropy");
        if (zstd2 == 2 && L.d_zgscr) hipLaunchKernelGGL(k_zstd_entropy_t<true>, grid1(L.nstr, ZG_FRAMES), dim3(64), 0, stream, L.d_streams, (int)L.nstr, L.d_chunks, L.d_blocks, L.d_zmeta, L.zseq_delta, L.d_zgscr, L.d_zctab);
        else hipLaunchKernelGGL(k_zstd_entropy_t<false>, grid1(L.nstr, ZG_FRAMES), dim3(64), 0, stream, L.d_streams, (int)L.nstr, L.d_chunks, L.d_blocks, L.d_zmeta, L.zseq_delta, (ZgLds*)nullptr, (ZcTab*)nullptr);
      }
      if (zstd2 == 2 && L.d_zgscr && BAMD_ZSTD_SEQ_KERNEL && !BAMD_ZSTD_LDS_FSE) {
        ProfScope ps(st, stream, "k_zstd_seq");      // the sequence streams of the frames phase A took, one lane per frame
        hipLaunchKernelGGL(k_zstd_seq, grid1(L.nstr, ZSEQ_FRAMES), dim3(64), 0, stream, L.d_streams, (int)L.nstr, L.d_chunks, L.d_blocks, L.d_zmeta, L.zseq_delta, L.d_zgscr, L.d_zctab);
      }
      {
        ProfScope ps(st, stream, "k_zstd_exec");
        hipLaunchKernelGGL(k_zstd_exec, dim3(persistent_grid(st, L.nstr, ZEXEC_WAVES_PER_CU)), dim3(64), 0, stream, L.d_streams, (int)L.nstr, L.d_status, L.d_zticket + 1,
                           L.d_chunks, L.d_blocks, L.d_zmeta, L.zseq_delta);
      }
      d_taken = (const uint32_t*)L.d_zmeta;
    }
    if (L.any_zstd) {
      ProfScope ps(st, stream, "k_zstd_streams");
#ifdef BAMD_PROFILE_DECODE
      uint32_t* d_zprof = nullptr;
      if (getenv("BLOSC_AMD_ZSTD_PROFILE")) { (void)hipMalloc((void**)&d_zprof, L.nstr * 64); (void)hipMemsetAsync(d_zprof, 0, L.nstr * 64, stream); }
      hipLaunchKernelGGL(k_zstd_streams, dim3(persistent_grid(st, L.nstr, ZSTD_WAVES_PER_CU)), dim3(64), 0, stream, L.d_streams, (int)L.nstr, L.d_status,
                         L.d_zticket, L.d_chunks, L.d_blocks, L.d_cost + 257, d_taken, d_zprof);
      if (d_zprof) {
        std::vector<uint32_t> h(L.nstr * 16);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), d_zprof, L.nstr * 64, hipMemcpyDeviceToHost);
        FILE* f = fopen(getenv("BLOSC_AMD_ZSTD_PROFILE"), "wb");
        if (f) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
        (void)hipFree(d_zprof);
      }
#else
      hipLaunchKernelGGL(k_zstd_streams, dim3(persistent_grid(st, L.nstr, ZSTD_WAVES_PER_CU)), dim3(64), 0, stream, L.d_streams, (int)L.nstr, L.d_status,
                         L.d_zticket, L.d_chunks, L.d_blocks, L.d_cost + 257, d_taken);
#endif
    }
    if (L.any_zlib) {
      ProfScope ps(st, stream, "k_zlib_streams");
      hipLaunchKernelGGL(k_zlib_streams, dim3(persistent_grid(st, L.nstr_zlib ? L.nstr_zlib : 1, ZLIB_WAVES_PER_CU)), dim3(64), 0, stream, L.d_streams, L.d_status,
                         L.d_zticket + 8, L.d_zqlist, L.d_zqoff, L.d_cost + 259, L.d_chunks, L.d_blocks, L.d_blkdone, st.single_queue ? 1 : 0);
    }
    if (L.any_shuf) {
      ProfScope ps(st, stream, "k_unshuffle");
      hipLaunchKernelGGL(k_unshuffle, dim3((unsigned)L.nblk, (unsigned)L.tiles_shuf), dim3(FT_THREADS), 0, stream, L.d_chunks, L.d_blocks);
    }
    if (L.any_bit) {
      const bool bitfast = true;       // full tiles of typesize 1 / 2 / 4 / 8 through k_bitfilter_fast, the rest through the generic kernel
      ProfScope ps(st, stream, "k_bitunshuffle");
      if (bitfast) hipLaunchKernelGGL(k_bitfilter_fast<1>, dim3((unsigned)L.nblk, (unsigned)L.tiles_bit), dim3(FT_THREADS), 0, stream, L.d_chunks, L.d_blocks);
      hipLaunchKernelGGL(k_bitunshuffle, dim3((unsigned)L.nblk, (unsigned)L.tiles_bit), dim3(FT_THREADS), 0, stream, L.d_chunks, L.d_blocks, bitfast ? 1 : 0);
    }
  }
  if (L.any_copy) {
    ProfScope ps(st, stream, "k_copy_chunks");
    hipLaunchKernelGGL(k_copy_chunks, dim3(64, (unsigned)L.nchunks), dim3(COMPACT_THREADS), 0, stream, L.d_chunks, kMaxOverhead);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}


static void filter_tiles(ChunkDesc& c, bool& any_shuf, bool& any_bit, int& tiles_shuf, int& tiles_bit, bool may_fuse) {
  const int32_t T = c.typesize, N = c.blocksize / T;
  // (Zstd chunks: only unsplit ones - the wave that decodes a block's one stream unshuffles it, k_decode.hip: fused_unshuffle_own_block;
  //  zlib chunks: split ones too, k_zlib_streams has per-XCD queues and the hand-off of the LZ4 kernel)
  const bool zfmt = c.fmt == FMT_ZSTD || c.fmt == FMT_ZLIB;
  if ((c.mode & CH_SHUFFLE) && fuse_enabled() && may_fuse && (zfmt ? (fused_fast_typesize(T) && (c.fmt == FMT_ZLIB || c.nsplits == 1)) : fused_typesize(T))) { c.mode |= CH_FUSED_UNSHUF; return; }
  if (c.mode & CH_SHUFFLE) {
    any_shuf = true;
    int t = (N + shuffle_tile_elems(T) - 1) / shuffle_tile_elems(T); if (t < 1) t = 1;
    if (t > tiles_shuf) tiles_shuf = t;
  } else if ((c.mode & CH_BITSHUFFLE) && bitunshuffle_fused_host(T) && fuse_enabled() && may_fuse && !zfmt) {
    c.mode |= CH_FUSED_BITUNSH;      // round 4: the decode kernel bit-unshuffles every block when its last stream is done (k_decode.hip: bitunshuffle_block_wave)
  } else if (c.mode & CH_BITSHUFFLE) {
    any_bit = true;
    int t = (N + bitshuffle_tile_elems(T) - 1) / bitshuffle_tile_elems(T); if (t < 1) t = 1;
    if (t > tiles_bit) tiles_bit = t;
  }
}

int engine_decompress_batch(int n, const Job* jobs, int* results, bool device_ptrs, hipStream_t stream) {
  if (n <= 0) return 0;
  CtxGuard ctx;
  EngineState& st = *ctx.st;
  if (ensure_device(st) || call_stream(st, !device_ptrs, &stream)) return -1;

  double ht_last = hosttime_on() ? now_ms() : 0.0; if (hosttime_on()) { std::lock_guard<std::mutex> l_(g_ht_mu); g_ht.calls[1]++; }
  std::vector<Header> hdrs;
  if (fetch_headers(st, n, jobs, device_ptrs, stream, hdrs)) return -1;
  HT_MARK(1, 0);     // header gather (kernel + copy + sync)

  std::vector<ChunkDesc> chunks((size_t)n);
  std::vector<BlockDesc> blocks;
  std::vector<uint8_t> live((size_t)n, 0);
  size_t nstr = 0, nstr_z = 0, nstr_zlib = 0, filt_bytes = 0, zlit_bytes = 0, io_src = 0, io_dst = 0;      // nstr_z: streams of Zstd / zlib chunks (their own kernels')
  DecodeLaunch L{};
  for (int i = 0; i < n; i++) {
    ChunkDesc& c = chunks[(size_t)i];
    memset(&c, 0, sizeof c);
    c.mode = CH_SKIP;
    int res = -1, fmt = 0;
    if (!classify_for_decompress(hdrs[(size_t)i], jobs[i].srcsize, jobs[i].dstsize, &res, &fmt)) { results[i] = res; continue; }
    add_decode_chunk(hdrs[(size_t)i], fmt, i, 0, hdrs[(size_t)i].nbytes / hdrs[(size_t)i].blocksize + ((hdrs[(size_t)i].nbytes % hdrs[(size_t)i].blocksize) ? 1 : 0),
                     c, blocks, nstr);
    c.src = (const uint8_t*)jobs[i].src; c.dst = (uint8_t*)jobs[i].dst;
    live[(size_t)i] = 1;
    results[i] = c.nbytes;
    if (c.mode & CH_MEMCPYED) L.any_copy = true;
    filter_tiles(c, L.any_shuf, L.any_bit, L.tiles_shuf, L.tiles_bit, !st.single_queue);
    if (c.mode & (CH_SHUFFLE | CH_BITSHUFFLE)) filt_bytes = align_up(filt_bytes, 256) + (size_t)c.nblocks * filt_block_stride(c);
    if (c.fmt == FMT_ZSTD && !(c.mode & CH_MEMCPYED)) { L.any_zstd = true; zlit_bytes = align_up(zlit_bytes, 256) + (size_t)c.nbytes; }
    if (c.fmt == FMT_ZLIB && !(c.mode & CH_MEMCPYED)) L.any_zlib = true;
    if (c.fmt == FMT_ZSTD || c.fmt == FMT_ZLIB) nstr_z += nstr - (size_t)c.first_stream;     // (memcpyed chunks have no streams)
    if (c.fmt == FMT_ZLIB) nstr_zlib += nstr - (size_t)c.first_stream;
    if (!device_ptrs) { io_src = align_up(io_src, 256) + (size_t)c.cbytes; io_dst = align_up(io_dst, 256) + (size_t)c.nbytes; }
  }
  const size_t nblk = blocks.size();
  Carver cv;
  const size_t o_chunks = cv.take(sizeof(ChunkDesc) * (size_t)n);
  const size_t o_streams = cv.take(sizeof(StreamDesc) * (nstr ? nstr : 1));
  const size_t o_status = cv.take(sizeof(int32_t) * (size_t)n + 64 + sizeof(uint32_t) * (nblk ? nblk : 1));   // + 8 tickets + per-block arrival counters
  const size_t o_cost = cv.take(sizeof(uint32_t) * kCostWords);
  const size_t o_spans = cv.take(8 * (nstr ? nstr : 1));
  const size_t o_pat = cv.take(span_enabled() ? (size_t)2048 * (nstr ? nstr : 1) : 256);
  const size_t o_filt = cv.take(filt_bytes + 256);
  const size_t o_zlit = cv.take(zlit_bytes + 256);      // literal scratch of the Zstd chunks
  const size_t o_zseq = cv.take(zlit_bytes + 512);      // sequence triples of the two-phase Zstd path, same layout as the literal scratch
  const size_t o_zmeta = cv.take(L.any_zstd ? sizeof(ZMeta) * (nstr ? nstr : 1) : 64);
  const size_t o_zticket = cv.take(64);
  const size_t o_zgscr = cv.take((L.any_zstd && zstd2_mode() == 2) ? sizeof(ZgLds) * (nstr ? nstr : 1) : 64);
  const bool use_zctab = L.any_zstd && zstd2_mode() == 2 && BAMD_ZSTD_SEQ_KERNEL && !BAMD_ZSTD_LDS_FSE;
  const size_t o_zctab = cv.take(use_zctab ? sizeof(ZcTab) * (nstr ? nstr : 1) : 64);
  if (st.dev.ensure(cv.off)) return -1;
  uint8_t* D = st.dev.base;
  uint8_t *io_s = nullptr, *io_d = nullptr;
  if (!device_ptrs) {
    if (st.io.ensure(align_up(io_src + 256, 256) + io_dst + 512)) return -1;
    io_s = st.io.base; io_d = st.io.base + align_up(io_src + 256, 256);
  }
  {
    size_t fo = 0, zo = 0, is = 0, id = 0;
    for (int i = 0; i < n; i++) {
      if (!live[(size_t)i]) continue;
      ChunkDesc& c = chunks[(size_t)i];
      if (!device_ptrs) {
        is = align_up(is, 256); id = align_up(id, 256);
        HIP_TRY(hipMemcpyAsync(io_s + is, jobs[i].src, (size_t)c.cbytes, hipMemcpyHostToDevice, stream));
        c.src = io_s + is; c.dst = io_d + id;
        is += (size_t)c.cbytes; id += (size_t)c.nbytes;
      }
      if (c.mode & (CH_SHUFFLE | CH_BITSHUFFLE)) { fo = align_up(fo, 256); c.filt = D + o_filt + fo; fo += (size_t)c.nblocks * filt_block_stride(c); }
      if (c.fmt == FMT_ZSTD && !(c.mode & CH_MEMCPYED)) { zo = align_up(zo, 256); c.stage = D + o_zlit + zo; zo += (size_t)c.nbytes; }
    }
  }
  // the block table and the queues: still on the device from the last call of this geometry, or built and uploaded now (EngineState::TableCache)
  EngineState::TableCache& tc = st.dec_tabs;
  const int nq = st.single_queue ? 1 : 8;
  std::vector<uint32_t> tmodes;                      // (the decode queues are made of the block table and the plane order alone)
  std::vector<int> tsig;
  order_signature(blocks, st.dec_cost, st.dec_cost_valid, tsig);
  const bool tabs_hit = table_cache_enabled() && tc.same(blocks, tmodes, tsig, nq);
  if (debug_cost_enabled()) fprintf(stderr, "[blosc_amd] decompress: block table and queues %s\n", tabs_hit ? "still on the device" : "built and uploaded");
  std::vector<int32_t> queues, zqueues;
  Carver tcv;
  const size_t t_blocks = tcv.take(sizeof(BlockDesc) * (nblk ? nblk : 1));
  if (!tabs_hit) {
    tc.valid = false;
    build_xcd_queues(blocks, nstr, st.dec_cost, st.dec_cost_valid, queues, nq);
    if (L.any_zlib) build_xcd_queues(blocks, nstr, nullptr, false, zqueues, nq, BLK_ZLIB);
    tc.o_queues = tcv.take(sizeof(int32_t) * (9 + (nstr ? nstr : 1)));
    tc.o_zqueues = tcv.take(sizeof(int32_t) * (9 + (nstr ? nstr : 1)));      // k_zlib_streams' queues
    if (tc.tabs.ensure(tcv.off)) return -1;
  }
  uint8_t* TB = tc.tabs.base;
  Carver pc;
  const size_t p_chunks = pc.take(sizeof(ChunkDesc) * (size_t)n);
  const size_t p_blocks = pc.take(tabs_hit ? 0 : sizeof(BlockDesc) * (nblk ? nblk : 1));
  const size_t p_status = pc.take(sizeof(int32_t) * (size_t)n);
  const size_t p_queues = pc.take(sizeof(int32_t) * queues.size());
  const size_t p_zqueues = pc.take(sizeof(int32_t) * (zqueues.size() + 1));
  const size_t p_cost = pc.take(sizeof(uint32_t) * kCostWords);
  if (st.pin.ensure(pc.off)) return -1;
  uint8_t* P = st.pin.base;
  memcpy(P + p_chunks, chunks.data(), sizeof(ChunkDesc) * (size_t)n);
  HIP_TRY(hipMemcpyAsync(D + o_chunks, P + p_chunks, sizeof(ChunkDesc) * (size_t)n, hipMemcpyHostToDevice, stream));
  if (!tabs_hit) {
    if (nblk) memcpy(P + p_blocks, blocks.data(), sizeof(BlockDesc) * nblk);
    memcpy(P + p_queues, queues.data(), sizeof(int32_t) * queues.size());
    if (nblk) HIP_TRY(hipMemcpyAsync(TB + t_blocks, P + p_blocks, sizeof(BlockDesc) * nblk, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(TB + tc.o_queues, P + p_queues, sizeof(int32_t) * queues.size(), hipMemcpyHostToDevice, stream));
    if (!zqueues.empty()) {
      memcpy(P + p_zqueues, zqueues.data(), sizeof(int32_t) * zqueues.size());
      HIP_TRY(hipMemcpyAsync(TB + tc.o_zqueues, P + p_zqueues, sizeof(int32_t) * zqueues.size(), hipMemcpyHostToDevice, stream));
    }
  }
  // status words + tickets + arrival counters | cost words lie back to back in the workspace (taken in that order above): one fill
  HIP_TRY(hipMemsetAsync(D + o_status, 0, (o_cost + sizeof(uint32_t) * kCostWords) - o_status, stream));

  L.d_chunks = (ChunkDesc*)(D + o_chunks); L.d_blocks = (BlockDesc*)(TB + t_blocks);
  L.d_streams = (StreamDesc*)(D + o_streams); L.d_status = (int32_t*)(D + o_status);
  L.d_ticket = (uint32_t*)(D + o_status + sizeof(int32_t) * (size_t)n + 32);
  L.d_blkdone = (uint32_t*)(D + o_status + sizeof(int32_t) * (size_t)n + 64);
  L.d_qoff = (const int32_t*)(TB + tc.o_queues); L.d_qlist = L.d_qoff + 9;
  L.d_zqoff = (const int32_t*)(TB + tc.o_zqueues); L.d_zqlist = L.d_zqoff + 9; L.nstr_zlib = nstr_zlib;
  L.d_spans = span_enabled() ? (uint32_t*)(D + o_spans) : nullptr; L.d_pat = D + o_pat;
  L.d_cost = (uint32_t*)(D + o_cost);
  L.d_zticket = (uint32_t*)(D + o_zticket);
  if (L.any_zstd || L.any_zlib) HIP_TRY(hipMemsetAsync(D + o_zticket, 0, 64, stream));
  L.d_zmeta = L.any_zstd ? (ZMeta*)(D + o_zmeta) : nullptr; L.zseq_delta = (ptrdiff_t)o_zseq - (ptrdiff_t)o_zlit + 8;
  L.d_zgscr = (L.any_zstd && zstd2_mode() == 2) ? (ZgLds*)(D + o_zgscr) : nullptr;
  L.d_zctab = use_zctab ? (ZcTab*)(D + o_zctab) : nullptr;
  L.nblk = nblk; L.nstr = nstr; L.nchunks = n;
  L.nstr_queued = nstr - nstr_z;
  HT_MARK(1, 2);     // tables, queues, uploads
  if (launch_decode(st, L, stream)) return -1;
  HIP_TRY(hipMemcpyAsync(P + p_status, D + o_status, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(P + p_cost, D + o_cost, sizeof(uint32_t) * kCostWords, hipMemcpyDeviceToHost, stream));
  HT_MARK(1, 3);     // kernel launches
  HIP_TRY(hipStreamSynchronize(stream));
  HT_MARK(1, 4);     // waiting for the device
  prof_collect(st);
  if (nblk && check_done((const uint32_t*)(P + p_cost), nstr - nstr_z, L.any_zstd ? nstr : 0, "decompress", nstr_zlib)) return -1;
  if (!tabs_hit) { tc.blocks.swap(blocks); tc.modes.swap(tmodes); tc.order.swap(tsig); tc.nq = nq; tc.valid = true; }      // (only now: the uploads are known to have arrived)
  if (nstr >= 4096) { memcpy(st.dec_cost, P + p_cost, sizeof st.dec_cost); st.dec_cost_valid = true; }
  if (debug_cost_enabled()) {
    fprintf(stderr, "[blosc_amd] decode plane costs:");
    for (int k = 0; k < 16; k++) fprintf(stderr, " %u", st.dec_cost[k]);
    fprintf(stderr, "\n");
  }
  const int32_t* stt = (const int32_t*)(P + p_status);
  for (int i = 0; i < n; i++) {
    if (!live[(size_t)i]) continue;
    if (stt[i] < 0) results[i] = -1;       // blosc.c:1511-1514: every block-level error surfaces as -1
  }
  if (!device_ptrs) {
    for (int i = 0; i < n; i++) {
      if (!live[(size_t)i] || results[i] <= 0) continue;
      HIP_TRY(hipMemcpyAsync(jobs[i].dst, chunks[(size_t)i].dst, (size_t)results[i], hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// getitem (blosc/blosc.c:1574-1703): decode only the blocks overlapping [start, start+nitems)
// ---------------------------------------------------------------------------------------------
int engine_getitem(const void* src, int start, int nitems, void* dest, bool src_dev, bool dst_dev, hipStream_t stream) {
  CtxGuard ctx;
  EngineState& st = *ctx.st;
  if (ensure_device(st) || call_stream(st, !src_dev && !dst_dev, &stream)) return -1;
  Job job{src, nullptr, 0, 0};
  std::vector<Header> hdrs;
  if (fetch_headers(st, 1, &job, src_dev, stream, hdrs)) return -1;
  const Header h = hdrs[0];
  const int stop = start + nitems;
  if (h.version != kVersionFormat) return -9;                                    // blosc.c:1603-1604
  if (h.blocksize <= 0 || h.blocksize > h.nbytes || h.blocksize > kMaxBlockSize || h.typesize <= 0) return -1;
  const int32_t T = h.typesize, bs = h.blocksize;
  const int32_t nblocks = h.nbytes / bs + ((h.nbytes % bs) ? 1 : 0);
  int fmt = 0;
  if (h.flags & kFlagMemcpyed) {
    if (h.nbytes + kMaxOverhead != h.cbytes) return -1;
  } else {
    const int f = (h.flags & 0xe0) >> 5;
    if (f != FMT_BLOSCLZ && f != FMT_LZ4 && f != FMT_ZLIB && f != FMT_ZSTD) return -5;
    if (h.versionlz != 1) return -9;
    fmt = f;
    if (nblocks >= (h.cbytes - 16) / 4) return -1;                               // blosc.c:1630-1632 (sic: >=)
  }
  if (start < 0 || (int64_t)start * T > h.nbytes) { fprintf(stderr, "`start` out of bounds"); return -1; }
  if (stop < 0 || (int64_t)stop * T > h.nbytes) { fprintf(stderr, "`start`+`nitems` out of bounds"); return -1; }
  const int64_t lo = (int64_t)start * T, hi = (int64_t)stop * T;
  if (hi <= lo) return 0;
  const size_t want = (size_t)(hi - lo);
  const hipMemcpyKind out_kind = dst_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;

  const uint8_t* dsrc = (const uint8_t*)src;
  if (!src_dev) {   // bring the chunk to the device
    if (h.cbytes < kMaxOverhead) return -1;
    if (st.io.ensure((size_t)h.cbytes + 256)) return -1;
    HIP_TRY(hipMemcpyAsync(st.io.base, src, (size_t)h.cbytes, hipMemcpyHostToDevice, stream));
    dsrc = st.io.base;
  }
  if (h.flags & kFlagMemcpyed) {
    HIP_TRY(hipMemcpyAsync(dest, dsrc + kMaxOverhead + lo, want, dst_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return (int)want;
  }
  const int32_t j0 = (int32_t)(lo / bs), j1 = (int32_t)((hi + bs - 1) / bs);
  ChunkDesc c;
  std::vector<BlockDesc> blocks;
  size_t nstr = 0;
  add_decode_chunk(h, fmt, 0, j0, j1, c, blocks, nstr);
  const size_t nblk = blocks.size();
  const size_t span = (size_t)(j1 - j0) * (size_t)bs;
  Carver cv;
  const size_t o_chunks = cv.take(sizeof(ChunkDesc));
  const size_t o_blocks = cv.take(sizeof(BlockDesc) * nblk);
  const size_t o_streams = cv.take(sizeof(StreamDesc) * nstr);
  const size_t o_status = cv.take(sizeof(int32_t) + 64 + sizeof(uint32_t) * nblk);
  const size_t o_queues = cv.take(sizeof(int32_t) * (9 + nstr));
  const size_t o_cost = cv.take(sizeof(uint32_t) * kCostWords);
  const size_t o_spans = cv.take(8 * nstr);
  const size_t o_pat = cv.take(span_enabled() ? (size_t)2048 * nstr : 256);
  const size_t o_out = cv.take(span + 256);
  const size_t o_filt = cv.take(span + 256);   // room for the padded plane layout of a fused chunk
  const size_t o_zlit = cv.take((fmt == FMT_ZSTD ? span : 0) + 256);
  const size_t o_zticket = cv.take(64);
  if (st.dev.ensure(cv.off)) return -1;
  uint8_t* D = st.dev.base;
  c.src = dsrc;
  // kernels address block j at base + j*blocksize: bias the bases so that block j0 lands at offset 0
  c.dst = D + o_out - (size_t)j0 * bs;
  c.stage = (fmt == FMT_ZSTD) ? D + o_zlit - (size_t)j0 * bs : nullptr;
  DecodeLaunch L{};
  filter_tiles(c, L.any_shuf, L.any_bit, L.tiles_shuf, L.tiles_bit, !st.single_queue);   // may set CH_FUSED_UNSHUF: before the upload
  c.filt = (c.mode & (CH_SHUFFLE | CH_BITSHUFFLE)) ? D + o_filt - (size_t)j0 * filt_block_stride(c) : nullptr;   // (the block stride depends on the mode just chosen)
  Carver pc;
  const size_t p_chunks = pc.take(sizeof(ChunkDesc));
  const size_t p_blocks = pc.take(sizeof(BlockDesc) * nblk);
  const size_t p_status = pc.take(sizeof(int32_t));
  const size_t p_cost = pc.take(sizeof(uint32_t) * kCostWords);
  std::vector<int32_t> queues;     // (one chunk, one format: the zlib kernel's queues when it is a zlib chunk, k_decode_streams' otherwise)
  build_xcd_queues(blocks, nstr, st.dec_cost, st.dec_cost_valid, queues, st.single_queue ? 1 : 8, fmt == FMT_ZLIB ? (uint32_t)BLK_ZLIB : 0u);
  const size_t p_queues = pc.take(sizeof(int32_t) * queues.size());
  if (st.pin.ensure(pc.off)) return -1;
  uint8_t* P = st.pin.base;
  memcpy(P + p_queues, queues.data(), sizeof(int32_t) * queues.size());
  memcpy(P + p_chunks, &c, sizeof c);
  memcpy(P + p_blocks, blocks.data(), sizeof(BlockDesc) * nblk);
  HIP_TRY(hipMemcpyAsync(D + o_chunks, P + p_chunks, sizeof c, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(D + o_blocks, P + p_blocks, sizeof(BlockDesc) * nblk, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(D + o_queues, P + p_queues, sizeof(int32_t) * queues.size(), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemsetAsync(D + o_status, 0, sizeof(int32_t) + 64 + sizeof(uint32_t) * nblk, stream));
  L.d_chunks = (ChunkDesc*)(D + o_chunks); L.d_blocks = (BlockDesc*)(D + o_blocks);
  L.d_streams = (StreamDesc*)(D + o_streams); L.d_status = (int32_t*)(D + o_status);
  L.d_ticket = (uint32_t*)(D + o_status + 32);
  L.d_blkdone = (uint32_t*)(D + o_status + 68);
  L.d_qoff = (const int32_t*)(D + o_queues); L.d_qlist = L.d_qoff + 9;
  L.d_spans = span_enabled() ? (uint32_t*)(D + o_spans) : nullptr; L.d_pat = D + o_pat;
  L.d_cost = (uint32_t*)(D + o_cost);   // a handful of blocks: the plane costs are not fed back, only the task count is checked
  HIP_TRY(hipMemsetAsync(D + o_cost, 0, sizeof(uint32_t) * kCostWords, stream));
  L.any_zstd = fmt == FMT_ZSTD; L.any_zlib = fmt == FMT_ZLIB; L.d_zticket = (uint32_t*)(D + o_zticket);
  L.d_zqoff = L.d_qoff; L.d_zqlist = L.d_qlist; L.nstr_zlib = L.any_zlib ? nstr : 0;
  if (L.any_zstd || L.any_zlib) HIP_TRY(hipMemsetAsync(D + o_zticket, 0, 64, stream));
  L.nblk = nblk; L.nstr = nstr; L.nchunks = 1; L.nstr_queued = (L.any_zstd || L.any_zlib) ? 0 : nstr;    // a handful of blocks: always through k_decode_streams (Zstd / zlib: their own kernels)
  if (launch_decode(st, L, stream)) return -1;
  HIP_TRY(hipMemcpyAsync(P + p_status, D + o_status, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(P + p_cost, D + o_cost, sizeof(uint32_t) * kCostWords, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  prof_collect(st);
  if (check_done((const uint32_t*)(P + p_cost), L.nstr_queued, L.any_zstd ? nstr : 0, "getitem", L.any_zlib ? nstr : 0)) return -1;
  const int32_t stt = *(const int32_t*)(P + p_status);
  if (stt < 0) return stt;                                                        // blosc.c:1689-1692: blosc_d's code is returned as is
  HIP_TRY(hipMemcpyAsync(dest, D + o_out + (size_t)(lo - (int64_t)j0 * bs), want, out_kind, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return (int)want;
}

// ---------------------------------------------------------------------------------------------
// stand-alone filter calls on host buffers (the reference exports blosc_internal_* for its own
// tests under BLOSC_TESTING, blosc/shuffle.h:34-61 + blosc/blosc-export.h:38-43)
// kind: 0 shuffle, 1 unshuffle, 2 bitshuffle, 3 bitunshuffle
// ---------------------------------------------------------------------------------------------
int engine_filter(int kind, size_t typesize, size_t blocksize, const void* src, void* dst) {
  CtxGuard ctx;
  EngineState& st = *ctx.st;
  if (ensure_device(st)) return -1;
  if (blocksize == 0) return 0;
  if (typesize == 0 || typesize > 255 || blocksize > (size_t)kMaxBlockSize) return -1;
  hipStream_t stream = 0;
  const int32_t T = (int32_t)typesize, bs = (int32_t)blocksize;
  Carver cv;
  const size_t o_chunk = cv.take(sizeof(ChunkDesc));
  const size_t o_block = cv.take(sizeof(BlockDesc));
  const size_t o_in = cv.take(blocksize + 256);
  const size_t o_out = cv.take(blocksize + 256);
  if (st.dev.ensure(cv.off)) return -1;
  uint8_t* D = st.dev.base;
  ChunkDesc c;
  memset(&c, 0, sizeof c);
  c.nbytes = bs; c.blocksize = bs; c.typesize = T; c.nblocks = 1; c.leftover = 0; c.nsplits = 1;
  const bool fwd = (kind == 0 || kind == 2);
  c.mode = (kind < 2) ? CH_SHUFFLE : CH_BITSHUFFLE;
  // forward kernels read c.src and write c.filt; inverse kernels read c.filt and write c.dst
  if (fwd) { c.src = D + o_in; c.filt = D + o_out; c.dst = nullptr; }
  else { c.filt = D + o_in; c.dst = D + o_out; c.src = nullptr; }
  BlockDesc b{0, 0, 0, 1, bs, 0};
  HIP_TRY(hipMemcpyAsync(D + o_chunk, &c, sizeof c, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(D + o_block, &b, sizeof b, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemcpyAsync(D + o_in, src, blocksize, hipMemcpyHostToDevice, stream));
  const int32_t N = bs / T;
  int tiles;
  if (kind < 2) tiles = (N + shuffle_tile_elems(T) - 1) / shuffle_tile_elems(T);
  else tiles = (N + bitshuffle_tile_elems(T) - 1) / bitshuffle_tile_elems(T);
  if (tiles < 1) tiles = 1;
  const ChunkDesc* dc = (const ChunkDesc*)(D + o_chunk);
  const BlockDesc* db = (const BlockDesc*)(D + o_block);
  switch (kind) {
    case 0: hipLaunchKernelGGL(k_shuffle, dim3(1, (unsigned)tiles), dim3(FT_THREADS), 0, stream, dc, db); break;
    case 1: hipLaunchKernelGGL(k_unshuffle, dim3(1, (unsigned)tiles), dim3(FT_THREADS), 0, stream, dc, db); break;
    case 2: hipLaunchKernelGGL(k_bitfilter_fast<0>, dim3(1, (unsigned)tiles), dim3(FT_THREADS), 0, stream, dc, db);
            hipLaunchKernelGGL(k_bitshuffle, dim3(1, (unsigned)tiles), dim3(FT_THREADS), 0, stream, dc, db, 1); break;
    default: hipLaunchKernelGGL(k_bitfilter_fast<1>, dim3(1, (unsigned)tiles), dim3(FT_THREADS), 0, stream, dc, db);
             hipLaunchKernelGGL(k_bitunshuffle, dim3(1, (unsigned)tiles), dim3(FT_THREADS), 0, stream, dc, db, 1); break;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(dst, D + o_out, blocksize, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------------------------
int engine_set_device(int dev) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || dev < 0 || dev >= cnt) return -1;
  g_device.store(dev);
  // ensure_device() probes the topology of THIS device (per-XCD queues are only valid where the probe says so) and resets
  // everything a context cached about the previous one; the other contexts do the same the next time they are used
  CtxGuard ctx;
  return ensure_device(*ctx.st);
}

// Binds the calling thread's calls to `dev` (-1: back to the process-wide device).  Used by the multi-GPU entry points
// (blosc_api.hip: one host thread per GPU); a context that lives on `dev` is preferred, so every GPU keeps its own arenas.
int engine_thread_device(int dev) {
  if (dev >= 0) { int cnt = 0; if (hipGetDeviceCount(&cnt) != hipSuccess || dev >= cnt) return -1; }
  tl_device = dev;
  return 0;
}
int engine_device_count() { int cnt = 0; if (hipGetDeviceCount(&cnt) != hipSuccess) { (void)hipGetLastError(); return 0; } return cnt; }

void engine_release() {
  if (hosttime_on()) {
    const char* nm[2] = {"compress", "decompress"};
    for (int d = 0; d < 2; d++) if (g_ht.calls[d])
      fprintf(stderr, "blosc_amd host time per %s call (ms, %ld calls): phase0 %.3f  phase1 %.3f  phase2 %.3f  launches %.3f  wait %.3f\n", nm[d], g_ht.calls[d],
              g_ht.t[d][0] / g_ht.calls[d], g_ht.t[d][1] / g_ht.calls[d], g_ht.t[d][2] / g_ht.calls[d], g_ht.t[d][3] / g_ht.calls[d], g_ht.t[d][4] / g_ht.calls[d]);
  }
  if (g_forked) return;
  for (int i = 0; i < kMaxCtx; i++) {
    EngineState& st = g_ctx[i];
    std::lock_guard<std::mutex> lock(st.mu);
    if (!st.device_ok) continue;
    st.dev.release(); st.io.release(); st.pin.release(); st.enc_tabs.drop(); st.dec_tabs.drop();
    if (st.own) { (void)hipStreamDestroy(st.own); st.own = nullptr; }
  }
}

bool engine_is_device_pointer(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

void engine_prof_enable(int on) { g_prof.store((on & 1) != 0); sched_override_off().store((on & 2) != 0); }      // bit 1: plain queue order for the calls that follow (queue_order.h)
void engine_prof_reset() {
  for (int i = 0; i < kMaxCtx; i++) { std::lock_guard<std::mutex> lock(g_ctx[i].mu); g_ctx[i].prof_acc.clear(); }
}
int engine_prof_get(const char* kernel, double* total_ms, int* launches) {      // summed over the contexts
  double ms = 0; int n = 0; bool found = false;
  for (int i = 0; i < kMaxCtx; i++) {
    std::lock_guard<std::mutex> lock(g_ctx[i].mu);
    auto it = g_ctx[i].prof_acc.find(kernel);
    if (it == g_ctx[i].prof_acc.end()) continue;
    ms += it->second.ms; n += it->second.launches; found = true;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  return found ? 0 : -1;
}

}  // namespace bamd
