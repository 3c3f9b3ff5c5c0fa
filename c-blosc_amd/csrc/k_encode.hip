// k_encode.hip — per-stream LZ4 / BloscLZ encoders and the chunk assembly kernels
// (rows K4, K6, C1, S1, T1 of SURVEY §8a).
//
// Replaces blosc_c's split loop (blosc/blosc.c:635-719) with its codec calls
//   LZ4_compress_fast   (lz4.c:1453 -> LZ4_compress_generic_validated :930-1338)
//   blosclz_compress    (blosc/blosclz.c:421-613)
// and serial_blosc/t_blosc's output placement (blosc/blosc.c:814-860, :1843-1860).
//
// The compressed BYTES are not the reference's (no reference test pins them, and the
// reference's own multi-threaded output order is nondeterministic); the contract is the
// FORMAT: every stream written here decodes with stock LZ4_decompress_safe /
// blosclz_decompress, every chunk with stock blosc_decompress (tests/ check exactly that).
//
// One kernel per batch (k_encode_streams): persistent wavefronts draw tasks from per-XCD queues - "shuffle
// block b" (typesize 4 / 8; the transposes run underneath the match finding of other waves) or "encode
// stream s".  Match finder, one wavefront per stream: the 64 lanes look at 64 consecutive positions at once
// (own bytes out of a 768-byte register window), probe a 2048-entry tagged hash table in LDS (LZ4's 4-byte
// multiplicative hash) plus distance 1, rank the candidates by exact match length (up to 20 bytes), extend
// the winner backwards and forwards with ballots, emit the sequence with the whole wave, and pick again
// among the lanes behind the match while it ends inside the step.  DESIGN.md 3.3 has the measurements.
//
// Layout: this file holds the per-stream dispatch (encode_one_stream), the persistent kernel in its modes and the chunk
// assembly kernels; the wave-level code it dispatches to sits in enc_lz.h (match finders: plain and LZ4HC-grade),
// enc_zstd.h (Zstd frames, per-block tables), enc_zlib.h (zlib streams) and enc_shuffle.h (the fused byte shuffle and the
// periodic-plane shortcut) - textual parts of this translation unit, included below inside namespace bamd.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"
#include "wave_prims.h"
#include "zstd_enc.h"
#include "deflate_enc.h"

namespace bamd {

// Optional phase profiling of the encoder (prof build, scripts/enc_phase.py); same slot layout as k_decode.hip's.
// slots: 0 steps, 1 steps without a match, 2 forward extensions, 3 literal runs copied from memory, 6 sequences,
//        5 backward extensions tried, 7 of those > 0 bytes, 4 of those > 4 bytes
//        (Zstd: 4 cycles tail literals + offset values, 5 cycles sequences section)
//        13 cycles: waiting for the block's shuffle task
//        8 cycles: window+probe, 9 candidates+select, 10 extension, 11 emit, 12 tail
#ifdef BAMD_PROFILE_DECODE
#define EPROF_ARG , DecProf& prof_
#define EPROF_PASS , prof_
#else
#define EPROF_ARG
#define EPROF_PASS
#endif

#include "enc_lz.h"
#include "enc_lz4p.h"
#include "enc_zstd.h"
#include "enc_zlib.h"
#include "enc_shuffle.h"

// one stream, not inlined into the queue loop (see decode_one_stream in k_decode.hip for why)
// MODE: 0 = LZ4 / BloscLZ, 1 = Zstd, 2 = Zlib, 3 = LZ4 with the LZ4HC-grade search, 4 = Zstd with per-block sequence tables -
// a batch has ONE codec, so every kernel carries only its own code path
// 5 = Zstd with per-block tables behind the LZ4HC-grade search, 6 = zlib behind the LZ4HC-grade search
// 7 / 8 = 4 / 5 with Huffman-coded literals
// 9 / 10 = zlib with dynamic Huffman codes (two passes), plain match finder / LZ4HC-grade search
enum { ENC_LZ = 0, ENC_ZSTD = 1, ENC_ZLIB = 2, ENC_HC = 3, ENC_ZSTD_T = 4, ENC_ZSTD_HC = 5, ENC_ZLIB_HC = 6, ENC_ZSTD_TH = 7, ENC_ZSTD_HCH = 8,
       ENC_ZLIB_DYN = 9, ENC_ZLIB_DYN_HC = 10 };
constexpr bool enc_mode_hc(int mode) { return mode == ENC_HC || mode == ENC_ZSTD_HC || mode == ENC_ZLIB_HC || mode == ENC_ZSTD_HCH || mode == ENC_ZLIB_DYN_HC; }
constexpr bool enc_mode_zlib(int mode) { return mode == ENC_ZLIB || mode == ENC_ZLIB_HC || mode == ENC_ZLIB_DYN || mode == ENC_ZLIB_DYN_HC; }
constexpr bool enc_mode_zstd(int mode) { return mode == ENC_ZSTD || mode == ENC_ZSTD_T || mode == ENC_ZSTD_HC || mode == ENC_ZSTD_TH || mode == ENC_ZSTD_HCH; }
#ifndef BAMD_ENC_HELP
#define BAMD_ENC_HELP 256     // a wave whose stream's block is not shuffled yet takes shuffle tasks instead of sleeping, up to this many blocks ahead (0: never)
#endif
template <int MODE>
__device__ __attribute__((noinline)) void encode_one_stream(StreamDesc* sd_, enc_entry_t* tab_, const ChunkDesc* chunks_, uint32_t* blk_ready_, int lane,
                                                            const BlockDesc* blocks_, uint32_t sid_, uint32_t* plane_cost_, uint64_t* seqbuf_
#ifdef BAMD_PROFILE_DECODE
                                                            , uint32_t* profslot_
#endif
                                                            ) {
  StreamDesc* sd; enc_entry_t* tab; const ChunkDesc* chunks; uint32_t* blk_ready; const BlockDesc* blocks; uint32_t sid; uint32_t* plane_cost; uint64_t* seqbuf;
#ifdef BAMD_PROFILE_DECODE
  uint32_t* profslot;
#endif
  PROF_DECL
#ifdef BAMD_PROFILE_DECODE
  prof_.c[4] = 0; prof_.c[5] = 0;
#endif
  // The arguments of a real call arrive in vector registers; every one of them is wave-uniform and most are needed again behind the encoder's loop.
  // Left where they are they cost the loop 16 of its 80 registers - or are spilled and RELOADED inside it, and a scratch reload waits for vmcnt(0),
  // i.e. for the step's stores (round 6: the parallel LZ4 step at 9.2 ms instead of 8.2 until these moved to scalar registers).
  sd = uni_gp(sd_); tab = uni_gp(tab_); chunks = uni_gp(chunks_); blk_ready = uni_gp(blk_ready_); blocks = uni_gp(blocks_);
  plane_cost = uni_gp(plane_cost_); seqbuf = uni_gp(seqbuf_); sid = uni(sid_);
#ifdef BAMD_PROFILE_DECODE
  profslot = uni_gp(profslot_);
#endif
  const uint32_t n = uni((uint32_t)sd->in_size), cap = uni((uint32_t)sd->out_size);
  const uint32_t aux = uni((uint32_t)sd->aux);
  const int clevel = (int)(aux & 15u);
  if (uni(chunks[uni((uint32_t)sd->chunk)].mode) & CH_FUSED_SHUF) {
    // the block's shuffle task sits earlier in this XCD's queue, so a running wave already owns it
    const uint32_t gb = aux >> 4;
    while (__hip_atomic_load(&blk_ready[gb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(16);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  PROF_LAP(13);
  const uint64_t cost_t0 = __builtin_amdgcn_s_memtime();
  uint32_t r;
  const int32_t hint = (int32_t)uni((uint32_t)sd->result);      // < 0: the shuffle task found this plane periodic (period -hint)
  if (MODE == ENC_ZSTD) r = seqbuf ? zstd_encode_wave(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZSTD_HCH) r = seqbuf ? zstd_encode_wave<true, true, true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZSTD_TH) r = seqbuf ? zstd_encode_wave<true, false, true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZSTD_HC) r = seqbuf ? zstd_encode_wave<true, true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZLIB_DYN) r = seqbuf ? zlib_dyn_encode_wave<false>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZLIB_DYN_HC) r = seqbuf ? zlib_dyn_encode_wave<true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZLIB_HC) r = zlib_encode_wave<true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, lane EPROF_PASS);
  else if (MODE == ENC_ZSTD_T) r = seqbuf ? zstd_encode_wave<true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZLIB) r = zlib_encode_wave(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, lane EPROF_PASS);
  else if (hint < 0) r = emit_periodic_stream(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, (uint32_t)-hint, uni((uint32_t)sd->fmt) == (uint32_t)FMT_LZ4, lane);
  else if (MODE == ENC_HC) r = lz4hc_encode_wave(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, tab, lane);
  else if (sd->fmt == FMT_LZ4) r = BAMD_ENC_PAR ? lz4_encode_wave_auto(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, lane EPROF_PASS)
                                                : lz_encode_wave<EF_LZ4>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, lane EPROF_PASS);
  else r = lz_encode_wave<EF_BLOSCLZ>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, lane EPROF_PASS);
  if (lane == 0) sd->result = (int32_t)r;
  // cost feedback for the host's queue order (queue_order.h: build_encode_queues): cycles per plane index
  if (plane_cost && lane == 0) {
    const uint32_t j = sid - (uint32_t)blocks[aux >> 4].first_stream;
    atomicAdd(plane_cost + (j & 255u), (uint32_t)((__builtin_amdgcn_s_memtime() - cost_t0) >> 10));
  }
#ifdef BAMD_PROFILE_DECODE
  prof_.c[15] = (uint32_t)(prof_.t0 >> 6);
  if (lane == 0 && profslot) for (int i_ = 0; i_ < 16; i_++) profslot[i_] = prof_.c[i_];
#endif
}

// Persistent waves + per-XCD ticket queues, like k_decode_streams (stream costs differ by orders of
// magnitude).  A queue entry >= 0 is a stream to encode; an entry < 0 is "shuffle block -(entry+1)".  The
// host puts every block's shuffle task a few dozen entries ahead of its streams (queue_order.h:
// build_encode_queues), so the bandwidth-bound transposes run underneath the latency/issue-bound match
// finding of other waves instead of in a kernel of their own.
template <int MODE>
// (waves per SIMD the register allocator plans for: the 24 KiB table of the HC modes leaves room for 1.5, the Zstd modes' LDS for 5)
__global__ __launch_bounds__(64 * ENC_WAVES, enc_mode_hc(MODE) ? 2 : ((MODE == ENC_ZSTD_T || MODE == ENC_ZSTD_TH) ? 5 : (MODE == ENC_LZ ? BAMD_ENC_LZ_MINWAVES : BAMD_ENC_MINWAVES))) void k_encode_streams_t(
    StreamDesc* __restrict__ streams, uint32_t* __restrict__ tickets /*[8]*/, const int32_t* __restrict__ qlist,
    const int32_t* __restrict__ qoff /*[9]*/, const int32_t* __restrict__ shoff /*[9] | shuffle list*/, const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks,
    uint32_t* __restrict__ blk_ready, uint32_t* __restrict__ plane_cost, int single_queue,
    uint64_t* __restrict__ seqbufs, const zenc::CTabs* __restrict__ ctabs, int detect_periodic
#ifdef BAMD_PROFILE_DECODE
    , uint32_t* __restrict__ profbuf
#endif
    ) {
  constexpr bool ZSTD = enc_mode_zstd(MODE);
  constexpr int TABBYTES = enc_mode_hc(MODE) ? HC_TAB_BYTES : ENC_TAB_BYTES;      // the match finder's table; the writers' LDS sits behind it
  __shared__ __attribute__((aligned(16))) enc_entry_t tabs[ENC_WAVES][(TABBYTES + (ZSTD ? ZS_LDS_BYTES : (enc_mode_zlib(MODE) ? DFL_LDS_BYTES : (MODE == ENC_LZ ? ENC_SCR_BYTES : 0)))) / 4];      // ENC_LZ: the 64 scratch dwords of the parallel LZ4 emitter behind the table (enc_lz4p.h)
  static_assert(ENC_WAVES == 1, "one stream per wave, one wave per workgroup");
  const int lane = threadIdx.x & 63;
  uint64_t* seqbuf = nullptr;
  if (ZSTD) {       // the predefined FSE tables of the sequence coder, once per persistent wave
    const uint32_t* g = (const uint32_t*)ctabs;
    for (uint32_t k = (uint32_t)lane; k < sizeof(zenc::CTabs) / 4u; k += 64u) tabs[0][TABBYTES / 4 + k] = g[k];
    seqbuf = seqbufs + (size_t)blockIdx.x * ZS_SEQCAP;
  }
  if (MODE == ENC_ZLIB_DYN || MODE == ENC_ZLIB_DYN_HC) seqbuf = seqbufs + (size_t)blockIdx.x * ZD_SCRATCH_U64;   // the tokens of the first pass
  // HW_REG_XCC_ID[3:0]; queue 0 for everybody in the single-queue fallback (no in-kernel hand-offs there)
  const uint32_t xcc = single_queue ? 0u : (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
  const uint32_t qbase = (uint32_t)qoff[xcc], qlen = (uint32_t)qoff[xcc + 1] - qbase;
  // Shuffle tasks are claimed in list order through a counter of their own (queue_order.h): a negative queue entry means "one
  // shuffle task", and a wave whose stream's block is not ready yet claims shuffle tasks too instead of sleeping.  No deadlock: a
  // shuffle task never waits, and a stream waits only for a block whose task either is claimed by a running wave or is still on
  // the list - where the waiting wave itself reaches it (claims are in list order, the list is finite).
  const uint32_t shbase = (uint32_t)shoff[xcc], shlen = (uint32_t)shoff[xcc + 1] - shbase;
  const int32_t* shlist = shoff + 9;
  auto shuffle_one = [&]() -> bool {
    const uint32_t s = take_ticket(tickets + 8 + xcc, lane);
    if (s >= shlen) return false;
    shuffle_block_task(chunks, blocks, uni((uint32_t)shlist[shbase + s]), blk_ready, streams, detect_periodic, lane, (volatile uint32_t*)tabs[0]);
    return true;
  };
  uint32_t t = take_ticket(tickets + xcc, lane);
  uint32_t ndone = 0;      // tasks this wave took: summed into plane_cost[256], the host checks the total
  while (t < qlen) {
    const int32_t task = (int32_t)uni((uint32_t)qlist[qbase + t]);
    if (task < 0) {
      shuffle_one();
    } else {
      const StreamDesc* sd = streams + task;
      if (uni(chunks[uni((uint32_t)sd->chunk)].mode) & CH_FUSED_SHUF) {
        const uint32_t gb = uni((uint32_t)sd->aux) >> 4;
        // only for the LDS-tile shuffle of the "other" typesizes (a task of theirs takes a wave several times as long as the register forms
        // of 2 / 4 / 8 / 16, so the few waves that drew the negative entries cannot keep incompressible data coming: random bytes, typesize 6:
        // 13.3 -> 9.6 ms).  With the fast forms helping costs 4 % on compressible data and gains nothing on random bytes
        // (profiles/r04/r04z2_enc_ab_help_bound_variants.txt).
        const ChunkDesc* cd = chunks + uni((uint32_t)sd->chunk);
        bool more = !(uni(cd->mode) & CH_BITSHUFFLE) && shuffle_generic_T(uni((uint32_t)cd->typesize));
        while (__hip_atomic_load(&blk_ready[gb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          // ... but not further ahead of its own block than BAMD_ENC_HELP blocks of the list: planes shuffled much earlier than they are
          // encoded have left the L2 / MALL by then (unbounded: config 2 + 7 %, profiles/r04/r04z_enc_ab_help_unbounded_vs_off.txt)
          if (BAMD_ENC_HELP && more && __hip_atomic_load(tickets + 8 + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (single_queue ? gb : gb / 8u) + (uint32_t)BAMD_ENC_HELP) more = shuffle_one();
          else __builtin_amdgcn_s_sleep(16);
        }
      }
#ifdef BAMD_PROFILE_DECODE
      encode_one_stream<MODE>(streams + task, tabs[0], chunks, blk_ready, lane, blocks, (uint32_t)task, plane_cost, seqbuf, profbuf ? profbuf + (size_t)task * 16 : nullptr);
#else
      encode_one_stream<MODE>(streams + task, tabs[0], chunks, blk_ready, lane, blocks, (uint32_t)task, plane_cost, seqbuf);
#endif
    }
    ndone++;
    t = take_ticket(tickets + xcc, lane);
  }
  if (lane == 0 && ndone) atomicAdd(plane_cost + 256, ndone);
}

// The stream table of a compress call, one thread per block: stream s of block j reads the block's split s out of the filtered image
// (or the source when the chunk has no filter) and writes into its staging slot of the same size (blosc/blosc.c:608-672: the per-split loop
// of blosc_c).  Made here instead of on the host: for 8 GiB that table is 2.6 MB - built, copied to pinned memory and uploaded per call
// it cost 0.2 ms of host time in front of every launch (round 4, BLOSC_AMD_HOSTTIME).
__global__ __launch_bounds__(256) void k_encode_plan(const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks,
                                                     StreamDesc* __restrict__ streams, int nblocks_total) {
  const int g = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (g >= nblocks_total) return;
  const BlockDesc b = blocks[g];
  if (b.nstreams <= 0) return;
  const ChunkDesc& c = chunks[b.chunk];
  const uint8_t* inbase = (c.mode & (CH_SHUFFLE | CH_BITSHUFFLE)) ? c.filt : c.src;
  const int32_t neblock = b.bsize / b.nstreams;
  const size_t at = (size_t)b.blk * (size_t)c.blocksize;
  for (int32_t k = 0; k < b.nstreams; k++) {
    StreamDesc sd;
    sd.in = inbase + at + (size_t)k * (size_t)neblock;
    sd.out = c.stage + at + (size_t)k * (size_t)neblock;
    sd.in_size = neblock; sd.out_size = neblock; sd.chunk = b.chunk; sd.fmt = c.fmt; sd.aux = c.clevel | (int32_t)((uint32_t)g << 4); sd.result = 0;
    streams[(size_t)b.first_stream + k] = sd;
  }
}

// ---------------------------------------------------------------------------------------------
// chunk assembly
// ---------------------------------------------------------------------------------------------
// k_chunk_scan: one workgroup per chunk.  Sums 4 + csize over each block's streams, prefix-sums
// the blocks IN BLOCK ORDER (deterministic layout; the reference's threaded path is completion
// order, blosc.c:1845-1860), writes the 16-byte header (blosc.c:1154-1244, :1275) and bstarts
// (absolute offsets, blosc.c:816), and decides the whole-chunk fallbacks (blosc.c:1264-1272):
//   total <= maxbytes            -> regular chunk,          result = total
//   else nbytes + 16 <= maxbytes -> MEMCPYED chunk,         result = nbytes + 16
//   else                         -> does not fit,           result = 0
constexpr int SCAN_THREADS = 256;

__device__ __forceinline__ void st_i32(gu8* p, int32_t v) { g_st_i32le(p, v); }

__global__ __launch_bounds__(SCAN_THREADS) void k_chunk_scan(ChunkDesc* __restrict__ chunks,
                                                            const BlockDesc* __restrict__ blocks,
                                                            const StreamDesc* __restrict__ streams,
                                                            int32_t* __restrict__ blk_off,   // [nblocks_total] out
                                                            int32_t* __restrict__ results) { // [nchunks] out
  __shared__ int32_t part[SCAN_THREADS];
  __shared__ int32_t carry_s;
  const int cid = blockIdx.x, tid = threadIdx.x;
  ChunkDesc& c = chunks[cid];
  if (c.mode & CH_SKIP) return;
  gu8* d = as_global(c.dst);
  if (c.mode & CH_MEMCPYED) {  // decided on the host (clevel 0 / nbytes < 128, blosc.c:1219-1229)
    if (tid == 0) {
      d[0] = 2; d[1] = 1; d[2] = (uint8_t)c.hdr_flags; d[3] = (uint8_t)c.typesize;  // versionlz byte: see below
      st_i32(d + 4, c.nbytes); st_i32(d + 8, c.blocksize); st_i32(d + 12, c.nbytes + 16);
      results[cid] = c.nbytes + 16;
    }
    return;
  }
  if (tid == 0) carry_s = 16 + 4 * c.nblocks;
  __syncthreads();
  for (int base = 0; base < c.nblocks; base += SCAN_THREADS) {
    const int j = base + tid;
    int32_t mine = 0;
    if (j < c.nblocks) {
      const BlockDesc b = blocks[c.first_block + j];
      for (int s = 0; s < b.nstreams; s++) {
        const StreamDesc& sd = streams[b.first_stream + s];
        mine += 4 + (sd.result > 0 ? sd.result : sd.in_size);
      }
    }
    part[tid] = mine;
    __syncthreads();
    // inclusive Hillis-Steele scan over the 256 partials
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {
      int32_t v = (tid >= o) ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    const int32_t carry = carry_s;
    // saturate instead of wrapping: a chunk cannot exceed INT_MAX anyway
    if (j < c.nblocks) {
      int64_t start = (int64_t)carry + part[tid] - mine;
      blk_off[c.first_block + j] = start > 0x7fffffff ? 0x7fffffff : (int32_t)start;
    }
    __syncthreads();
    if (tid == SCAN_THREADS - 1) {
      int64_t nc = (int64_t)carry + part[tid];
      carry_s = nc > 0x7fffffff ? 0x7fffffff : (int32_t)nc;
    }
    __syncthreads();
  }
  const int32_t total = carry_s;
  const int32_t maxbytes = c.cbytes;
  int32_t res;
  uint32_t flags = (uint32_t)c.hdr_flags;
  if (total <= maxbytes) {
    res = total;
    for (int j = tid; j < c.nblocks; j += SCAN_THREADS) st_i32(d + 16 + 4 * (size_t)j, blk_off[c.first_block + j]);
  } else if ((int64_t)c.nbytes + 16 <= (int64_t)maxbytes) {
    res = c.nbytes + 16;
    flags |= 0x2u;
    if (tid == 0) c.mode |= CH_MEMCPYED;   // compact kernel copies the raw input instead
  } else {
    res = 0;
    if (tid == 0) c.mode |= CH_SKIP;
  }
  if (tid == 0) {
    d[0] = 2;                      // BLOSC_VERSION_FORMAT (blosc.h:29)
    d[1] = 1;                      // codec format version, 1 for every codec (blosc.h:104-109)
    d[2] = (uint8_t)flags; d[3] = (uint8_t)c.typesize;
    st_i32(d + 4, c.nbytes); st_i32(d + 8, c.blocksize); st_i32(d + 12, res);
    results[cid] = res;
  }
}

// k_chunk_compact: one workgroup per block; moves the block's streams to their final place.
constexpr int COMPACT_THREADS = 256;

__device__ __forceinline__ void wg_copy(gu8* dst, const gu8* src, uint32_t n) {
  const uint32_t tid = threadIdx.x;
  uint32_t full = n & ~15u;
  for (uint32_t k = tid * 16u; k < full; k += COMPACT_THREADS * 16u) st16u(dst + k, ld16u(src + k));
  for (uint32_t k = full + tid; k < n; k += COMPACT_THREADS) dst[k] = src[k];
}

__global__ __launch_bounds__(COMPACT_THREADS) void k_chunk_compact(const ChunkDesc* __restrict__ chunks,
                                                                  const BlockDesc* __restrict__ blocks,
                                                                  const StreamDesc* __restrict__ streams,
                                                                  const int32_t* __restrict__ blk_off) {
  const BlockDesc b = blocks[blockIdx.x];
  const ChunkDesc& c = chunks[b.chunk];
  if (c.mode & CH_SKIP) return;
  const uint32_t bsize = (uint32_t)b.bsize;
  if (c.mode & CH_MEMCPYED) {   // payload = raw input right after the header (blosc.c:825-830)
    wg_copy(as_global(c.dst) + 16 + (size_t)b.blk * c.blocksize, as_global(c.src) + (size_t)b.blk * c.blocksize, bsize);
    return;
  }
  uint32_t pos = (uint32_t)blk_off[blockIdx.x];
  for (int s = 0; s < b.nstreams; s++) {
    const StreamDesc& sd = streams[b.first_stream + s];
    const bool raw = sd.result <= 0;
    const uint32_t sz = raw ? (uint32_t)sd.in_size : (uint32_t)sd.result;
    if (threadIdx.x == 0) st_i32(as_global(c.dst) + pos, (int32_t)sz);
    wg_copy(as_global(c.dst) + pos + 4, as_global(raw ? sd.in : (const uint8_t*)sd.out), sz);
    pos += 4u + sz;
  }
}

// plain byte copy of whole chunks (MEMCPYED chunks on the decompress side, blosc.c:843-848)
__global__ __launch_bounds__(COMPACT_THREADS) void k_copy_chunks(const ChunkDesc* __restrict__ chunks, int src_skip) {
  const ChunkDesc& c = chunks[blockIdx.y];
  if (!(c.mode & CH_MEMCPYED) || (c.mode & CH_SKIP)) return;
  const uint64_t per = (uint64_t)COMPACT_THREADS * 16u * 8u;  // 32 KiB per workgroup step
  for (uint64_t lo = (uint64_t)blockIdx.x * per; lo < (uint64_t)c.nbytes; lo += (uint64_t)gridDim.x * per) {
    uint64_t hi = lo + per < (uint64_t)c.nbytes ? lo + per : (uint64_t)c.nbytes;
    wg_copy(as_global(c.dst) + lo, as_global(c.src) + src_skip + lo, (uint32_t)(hi - lo));
  }
}

}  // namespace bamd
