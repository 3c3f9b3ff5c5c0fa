// k_decode.hip — per-stream LZ4 / BloscLZ decoders + the block "plan" kernel (rows K5, K7, D1 of
// SURVEY §8a).
//
// Replaces  blosc_d's split loop (blosc/blosc.c:760-787) and the codecs it calls:
//   LZ4_decompress_safe   (internal-complibs/lz4-1.10.0/lz4.c:2451 -> LZ4_decompress_generic :2023-2445)
//   blosclz_decompress    (blosc/blosclz.c:679-789)
// One wavefront decodes one stream (= one split of one block).  Both grammars (dec_ring.h): the wave's recent output lives in an LDS ring,
// sequences are parsed and executed 16 at a time out of LDS, complete 1 KiB rows leave for global memory (the plane-major scratch, or the
// destination when the chunk has no filter) as coalesced stores nobody waits for.  The wave that completes a block's last stream
// (bit-)unshuffles the block into the destination (below).
//
// Algorithmic HBM bytes per stream: csize read + neblock written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "dev_types.h"
#include "wave_prims.h"

namespace bamd {

constexpr int DEC_WAVES = 1;  // one stream per workgroup: streams differ 1000x in sequence count, a
                              // multi-wave workgroup would hold its slots until the slowest is done

// ---------------------------------------------------------------------------------------------
// plan: walk every block's csize chain once (blosc/blosc.c:760-771), emit StreamDesc entries
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t ld_i32(const uint8_t* p) { return g_ld_i32le(as_global(p)); }

__global__ void k_decode_plan(const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks,
                              StreamDesc* __restrict__ streams, int32_t* __restrict__ status, int nblocks_total) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nblocks_total) return;
  const BlockDesc b = blocks[g];
  const ChunkDesc& c = chunks[b.chunk];
  const int32_t bsize = b.bsize;
  const int32_t neblock = bsize / b.nstreams;
  const bool to_filt = (c.mode & (CH_SHUFFLE | CH_BITSHUFFLE)) != 0;
  // (The scratch of block g at slot g % S of ONE arena, without any synchronisation - the most a ring of block slots could ever give: round 3,
  //  768 ... 1536 slots: no change; round 4 with the LDS ring decoder and the short lead, 256 ... 2048 slots: - 5 ... - 12 % below 640 slots, where
  //  several live blocks share a slot, - 0 ... - 7 % and not monotonic from 768 up (profiles/r04/r04zd_*, r04ze_*).  About 1 600 blocks are live at a
  //  time - a block lives as long as its slowest stream - so a real ring needs that many slots and gains a few per cent at best: not built.)
  uint8_t* out = to_filt ? c.filt + (size_t)b.blk * filt_block_stride(c) : c.dst + (size_t)b.blk * c.blocksize;
  const uint32_t pstride = (to_filt && b.nstreams > 1) ? filt_plane_stride(c, (uint32_t)bsize, b.nstreams) : (uint32_t)neblock;   // split blocks: one stream per plane
  int32_t off = ld_i32(c.src + 16 + 4 * (size_t)b.blk);
  bool bad = false;
  for (int j = 0; j < b.nstreams; j++) {
    StreamDesc s;
    s.chunk = b.chunk; s.fmt = c.fmt; s.aux = g; s.result = 0;
    s.out = out + (size_t)j * pstride; s.out_size = neblock;
    s.in = nullptr; s.in_size = -1;  // -1: nothing to decode (chain broken)
    if (!bad) {
      if (off < 0 || off > c.cbytes - 4) bad = true;
      else {
        int32_t cs = ld_i32(c.src + off);
        off += 4;
        if (cs < 0 || cs > c.cbytes - off) bad = true;
        else { s.in = c.src + off; s.in_size = cs; off += cs; }
      }
    }
    streams[b.first_stream + j] = s;
  }
  if (bad) atomicMin(&status[b.chunk], (int32_t)ST_BADCHAIN);
}

// most sequences one batched step takes (dec_ring.h: dr_step, both grammars)
constexpr uint32_t BATCH_MAXSEQ = 16;      // (32 - a fifth doubling round, the second DPP row, two passes of 4-lane pieces: half of bench19's steps hold more than 16 sequences, yet + 1 %: profiles/r04/r04zw_*)

// Optional phase profiling (build with -DBAMD_PROFILE_DECODE -> libblosc_amd_prof.so, scripts/dec_phase.py):
// 16 wave-uniform counters per stream, cycles from s_memtime.
#ifdef BAMD_PROFILE_DECODE
struct DecProf { uint64_t t0; uint32_t c[16]; };
#define PROF_DECL DecProf prof_; for (int i_ = 0; i_ < 16; i_++) prof_.c[i_] = 0; prof_.t0 = __builtin_amdgcn_s_memtime(); \
  prof_.c[14] = (uint32_t)(prof_.t0 >> 6); prof_.c[4] = __builtin_amdgcn_s_getreg(0xF804); prof_.c[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
#define PROF_ARG , DecProf& prof_
#define PROF_PASS , prof_
#define PROF_ADD(i, v) prof_.c[i] += (uint32_t)(v)
#define PROF_LAP(i) do { uint64_t t_ = __builtin_amdgcn_s_memtime(); prof_.c[i] += (uint32_t)(t_ - prof_.t0); prof_.t0 = t_; } while (0)
#else
#define PROF_DECL
#define PROF_ARG
#define PROF_PASS
#define PROF_ADD(i, v)
#define PROF_LAP(i)
#endif
// counter slots: 0 batches, 1 sequences done in batches, 2 of those "others" (sequential), 3 scalar-path sequences,
// 8 cycles: seek+gather+parse, 9 chain walk, 10 literals+pieces issue, 11 others, 12 scalar path, 13 raw/other

__device__ __forceinline__ uint32_t bperm(uint32_t src_lane, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
// value of lane (l - n) within the same row of 16 lanes, 0 when that lane is outside the row (DPP row_shr:n)
template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true);
}
// one hop through a next-position table held one entry per lane; position 64 ("stop") is absorbing
__device__ __forceinline__ uint32_t hop(uint32_t table, uint32_t x) {
  const uint32_t t = bperm(x & 63u, table);
  return x >= 64u ? 64u : t;
}

// ---------------------------------------------------------------------------------------------
// Periodic spans.  After a byte shuffle the planes of the high-order bytes are typically constant or
// repeat with a short power-of-two period; their whole stream is "a few literals + one match that runs
// to the end".  Writing such a plane to the scratch only to read it back in the unshuffle is two wasted
// passes over HBM.  When a match is long (>= 16 KiB) and its distance a power of two <= 2048, the decoder
// therefore writes only the part up to the next 1 KiB boundary and the part behind the last one, stores
// the 2 KiB pattern table pat[i] = plane[q], q = i (mod 2048), and records the skipped range [lo, hi).
// The fused unshuffle (unshuffle_block_wave) reads those positions from the table (L1/L2-resident)
// instead of the scratch.  A later match that reaches back into the skipped range makes the decoder
// fill it in after all ("materialise"); only split blocks of fused chunks use spans.
// ---------------------------------------------------------------------------------------------
struct SpanCtx { uint32_t enabled, lo, hi, off; gu8* pat;
#ifdef BAMD_LOO_PLANES      // timing-only builds (wrong output): the planes of this mask never reach the scratch and are never read back by the unshuffle
  uint32_t loo;
#endif
};
constexpr uint32_t SPAN_PAT = 2048u;

// (real calls, made a few times per stream at most, with plain arguments: the helpers' registers must not count against the hot
//  loops of the caller, and a SpanCtx handed over by reference would live in memory - its `hi` is tested in every step)
__device__ __attribute__((noinline)) void span_fill_call(gu8* out_, uint32_t lo_, uint32_t off_, uint32_t n_, int lane) {
  wave_match_copy(uni_ptr(out_), uni(lo_), uni(off_), uni(n_), lane);      // out[lo - off, lo) was written by the head copy
}
// Periods ABOVE the pattern table's 2 KiB (any power of two up to 64 KiB; round 3).  Byte planes of slowly varying data repeat with the
// period of the generator (bench19's second byte: 32 KiB; a third of the plane is content, the rest ONE match).  Such a span needs no
// table: the plane itself holds the period, in the `off` bytes in front of the skipped range.  With ob = the 4-aligned position at or behind
// mpos - off,   plane[q] = plane[ob + ((q - ob) & (off - 1))]   for q in [lo, hi)   (every step back by `off` stays inside the match), and a
// dword at a 4-aligned q never straddles the wrap.  The unshuffle computes that address per lane (SPAN_SELF); ob and off travel in the first
// 8 bytes of the stream's table slot.  Both decoders take their spans through dec_ring.h: dr_span_long_match / dr_span_call.

}  // namespace bamd
#include "dec_ring.h"
namespace bamd {

// (Both stream decoders - lz4_decode_wave and blosclz_decode_wave - live in dec_ring.h, on the LDS ring.)

// ---------------------------------------------------------------------------------------------
// decode kernel: persistent waves + ticket queue, block = 64 * DEC_WAVES
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Fused byte-unshuffle of one block by ONE wavefront (typesize 2, 4, 8 or 16), used by the decode kernel when the
// last stream of a block has been decoded.  Plane-major scratch -> element-major destination
// (blosc/shuffle-generic.h:61-81).  Per step each lane loads 16 bytes of every plane (coalesced 1 KiB
// rows), transposes bytes in registers (v_perm_b32) and stores 16-byte pieces of its own contiguous
// T*16-byte output; the 128 steps of a 1 MiB block keep two steps of loads in flight.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void transpose4x4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3,
                                             uint32_t& t0, uint32_t& t1, uint32_t& t2, uint32_t& t3) {
  // r_j = bytes (e0,e1,e2,e3) of plane j  ->  t_e = bytes (plane0,plane1,plane2,plane3) of element e
  const uint32_t a01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), b01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
  const uint32_t a23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), b23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
  t0 = __builtin_amdgcn_perm(a23, a01, 0x05040100u); t1 = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
  t2 = __builtin_amdgcn_perm(b23, b01, 0x05040100u); t3 = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
}

// 4 x 4 byte transpose ACROSS the four lanes of a quad: lane i ends up with (byte i of lane 0's dword, byte i of lane 1's, of lane 2's,
// of lane 3's).  Four DPP quad broadcasts and three v_perm.  It is its own inverse, and it is what makes the 16-byte accesses of
// typesize 16 coalesce: a lane that holds a plane dword "elements 4l .. 4l+3" afterwards holds "elements 16Q + i, + 4, + 8, + 12"
// (Q = l >> 2, i = l & 3), so the four lanes of a quad store (or load) 64 CONTIGUOUS bytes per instruction instead of 16-byte
// pieces 64 bytes apart (measured before: the fused typesize-16 unshuffle cost more than the stand-alone kernel it replaced).
__device__ __forceinline__ uint32_t quad_byte_transpose(uint32_t d, int lane) {
  const uint32_t i = (uint32_t)lane & 3u, sel = (i | ((4u + i) << 8)) * 0x00010001u;
  const uint32_t b0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x00, 0xf, 0xf, true), b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x55, 0xf, 0xf, true);
  const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0xAA, 0xf, 0xf, true), b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0xFF, 0xf, 0xf, true);
  const uint32_t lo = __builtin_amdgcn_perm(b1, b0, sel), hi = __builtin_amdgcn_perm(b3, b2, sel);      // low two bytes: (b0[i], b1[i]) / (b2[i], b3[i])
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// One step: lane l owns elements e + 4l .. e + 4l + 3: a 4-byte load from every plane (each wave load
// instruction reads 256 contiguous bytes of one plane), a byte transpose in registers, and T*4 contiguous
// output bytes per lane (consecutive lanes are consecutive in memory: fully coalesced 16-byte stores).
template <int T>
struct Rows { uint32_t r[T]; };

template <int T>
struct PlanePtrs { const gu8* p[T]; };

// the final output of the fused unshuffles leaves with non-temporal stores (round 2: bench19 -3.8 %, linspace -1.5 %, profiles/r02/r02f_decode_nt_variants.txt;
// round 4, plain against non-temporal with the ring decoder: no difference, profiles/r04/r04i_dec_ab_dst_stores_plane_loads_nt.txt); the planes are read with
// plain loads (the rows the ring decoder flushed are in L2: non-temporal loads cost 3-6 %, same file)
__device__ __forceinline__ void st16_dst(gu8* p, uint4 v) { g_st16_nt(p, v); }
__device__ __forceinline__ uint32_t ld4_plane(const gu8* p) { return g_ld4(p); }
// Which 4 elements of a 256-element step a lane's plane dwords cover (byte offset into the plane).  Typesize 2 / 4: elements
// 4 l .. 4 l + 3, i.e. T * 4 contiguous output bytes per lane and contiguous lanes - every store instruction is one contiguous run.
// Typesize 8 (round 4): a lane's 32 output bytes are two 16-byte stores, and dealt that way every store instruction wrote only half of
// each 32-byte sector - the fused unshuffle is bound by exactly those stores (profiles/r04/r04h_*: leaving the LOADS out changed nothing,
// leaving the stores out gave the kernel 1.5 of its 1.8 ms back; profiles/r04/r04j_*: the same bytes as contiguous 1 KiB stores: -12 ... 16 %).
// So lanes work in pairs: lane 2 i loads the dword of elements 4 i .. 4 i + 3, lane 2 i + 1 that of elements 128 + 4 i .. (each wave load still
// covers two full 128-byte lines), the pair swaps halves (one DPP quad_perm per plane) and lane l then holds elements 2 l, 2 l + 1 of the
// step's first 128 elements AND of its second 128: two stores of 1 KiB, each fully contiguous.
template <int T>
__device__ __forceinline__ uint32_t unsh_l4(int lane) {
  if (T == 8) return 4u * (((uint32_t)lane >> 1) + 32u * ((uint32_t)lane & 1u));
  return 4u * (uint32_t)lane;
}
template <int T>
__device__ __forceinline__ void unshuffle_store(gu8* dst, uint32_t e, int lane, const Rows<T>& x) {
  if constexpr (T == 8) {
    const bool odd = (lane & 1) != 0;
    // x.r[j] = plane j's bytes of elements 4 i .. 4 i + 3 (even lane) / 128 + 4 i .. (odd lane), y = the pair partner's.  First store: elements
    // 2 l, 2 l + 1 = the even lane's low half (even l) or high half (odd l); second store: the odd lane's low / high half
    const uint32_t sel0 = odd ? 0x0c0c0706u : 0x0c0c0100u, sel1 = odd ? 0x0c0c0302u : 0x0c0c0504u;
    uint32_t h0[8], h1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x.r[j], 0xB1, 0xf, 0xf, true);      // quad_perm [1, 0, 3, 2]: the pair partner's dword
      h0[j] = __builtin_amdgcn_perm(y, x.r[j], sel0); h1[j] = __builtin_amdgcn_perm(y, x.r[j], sel1);       // (byte 0: element 2 l, byte 1: element 2 l + 1)
    }
    gu8* o = dst + (size_t)(e + 2u * (uint32_t)lane) * 8u;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const uint32_t* h = s ? h1 : h0;
      const uint32_t x01 = __builtin_amdgcn_perm(h[1], h[0], 0x05010400u), x23 = __builtin_amdgcn_perm(h[3], h[2], 0x05010400u);
      const uint32_t x45 = __builtin_amdgcn_perm(h[5], h[4], 0x05010400u), x67 = __builtin_amdgcn_perm(h[7], h[6], 0x05010400u);
      st16_dst(o + 1024 * s, make_uint4(__builtin_amdgcn_perm(x23, x01, 0x05040100u), __builtin_amdgcn_perm(x67, x45, 0x05040100u),
                                        __builtin_amdgcn_perm(x23, x01, 0x07060302u), __builtin_amdgcn_perm(x67, x45, 0x07060302u)));
    }
    return;
  }
  gu8* o = dst + (size_t)(e + 4u * (uint32_t)lane) * T;
  if constexpr (T == 2) {                            // elements 0..3 (2 bytes each): one 8-byte store per lane
    const uint32_t lo = __builtin_amdgcn_perm(x.r[1], x.r[0], 0x05010400u), hi = __builtin_amdgcn_perm(x.r[1], x.r[0], 0x07030602u);
    const uint64_t v = (uint64_t)lo | ((uint64_t)hi << 32);
    g_st8_nt(o, v);
  } else {                                           // typesize 4: elements 0..3 (4 bytes each), one 16-byte store per lane
    uint32_t t0, t1, t2, t3;
    transpose4x4(x.r[0], x.r[1], x.r[2], x.r[3], t0, t1, t2, t3);
    st16_dst(o, make_uint4(t0, t1, t2, t3));
  }
}

// spans: {lo | flags, hi} per plane (nullptr: none), pat: SPAN_PAT bytes per plane - see SpanCtx above.
// Flag bit 0, `small`: the span's period divides 256.  A lane's dword of such a plane is then the same in every
// 256-element step (element e + 4 lane, e a multiple of 256), so it is loaded ONCE per block and kept in a register.
// That matters more than it looks: the pattern tables of the blocks in flight (16 KiB per block, 768 blocks per XCD) do
// not fit the 4 MiB L2, so table reads come from HBM / Infinity Cache like the scratch itself (all-zero input: 8 GiB
// written AND 8 GiB "read" per launch, 3.1 ms; scripts/micro/fronts.hip: a plain 8 GiB fill takes 1.6 ms).
// Flag bit 1, `raw`: the split was stored raw (blosc/blosc.c:773-776) and has NOT been copied to the scratch: the plane is
// read where it lies in the chunk (raw[j], any byte alignment).  Noisy float64 data stores 3 of 4 planes raw: the copy
// was a read and a write of the plane for nothing.
// Flag bit 2, `self`: a period above the table's 2 KiB (span_long_match): the skipped positions are read from the plane's own bytes
// in front of the span, at ob + ((q - ob) & (off - 1)) per lane; ob and off are the first two words of the stream's table slot.
constexpr uint32_t SPAN_SMALL = 1u, SPAN_RAW = 2u, SPAN_SELF = 4u;
#ifndef BAMD_UNSH_BYTES
#define BAMD_UNSH_BYTES 0         // plane bytes a wave keeps in flight per group of the fused unshuffle; 0 = one quarter-group (1024 elements) whatever the typesize
#endif
template <int T>
__device__ void unshuffle_block_wave_T(const gu8* src, gu8* dst, uint32_t bsize, int lane, const uint32_t* spans, const gu8* pat, const StreamDesc* sds, uint32_t pstride) {
  const uint32_t N = bsize / T;
  uint32_t lo[T], hi[T], pr[T], ob[T];                   // ob: a self span's base | log2(period) << 24
  const gu8* pl[T];                                      // where plane j lies: the scratch, or the chunk itself (raw)
  uint32_t small = 0, self = 0;                          // wave-uniform plane masks
#pragma unroll
  for (int j = 0; j < T; j++) {
    const uint32_t w = spans ? uni(spans[2 * j]) : 0u;
    lo[j] = w & ~1023u; hi[j] = spans ? uni(spans[2 * j + 1]) : 0u;
    pr[j] = 0u; ob[j] = 0u;
    pl[j] = src + (size_t)j * pstride;
#ifdef BAMD_LOO_PLANES
    if (spans && (((uint32_t)(BAMD_LOO_PLANES) >> j) & 1u)) { small |= 1u << j; lo[j] = 0u; hi[j] = N; pr[j] = g_ld4(pat + (size_t)j * SPAN_PAT + unsh_l4<T>(lane)); continue; }
#endif
    if (w & SPAN_RAW) { pl[j] = uni_ptr(as_global(sds[j].in)); hi[j] = 0u; }
    else if ((w & SPAN_SMALL) && hi[j] > lo[j]) { small |= 1u << j; pr[j] = g_ld4(pat + (size_t)j * SPAN_PAT + unsh_l4<T>(lane)); }
    else if ((w & SPAN_SELF) && hi[j] > lo[j]) {
      self |= 1u << j;
      ob[j] = uni(g_ld4(pat + (size_t)j * SPAN_PAT)) | ((31u - (uint32_t)__builtin_clz(uni(g_ld4(pat + (size_t)j * SPAN_PAT + 4)))) << 24);
    }
  }
  // the register rows are in before the loop: otherwise the compiler, which cannot tell whether they are still in flight,
  // waits for vmcnt(0) at the top of EVERY iteration - i.e. for the previous iteration's stores
  __builtin_amdgcn_s_waitcnt(0);
  uint32_t e = 0;
  // A group = G quarter-groups of 4 steps (1024 elements each): all its loads are issued before its first store.  Span bounds are multiples of
  // 1024, so one decision per plane and quarter-group picks the plane, the pattern table or the register.
  const uint32_t l4 = unsh_l4<T>(lane);
  // (Two things round 4 tried on this loop and dropped.  A software pipeline - two register sets, the loads of group g + 1 issued before the
  //  stores of group g -: 32 % SLOWER, 5.55 against 4.20 ms; the span branches around the loads leave the compiler no exact vmcnt and the second
  //  register set spills - round 3 had seen -10 % at 96 registers (profiles/r04/r04g_dec_ab_bisect_walk_pipe_rowfill.txt, r03i_*).  And spreading the
  //  block's stores thin behind the steps of the streams the wave decodes next: no gain - what the unshuffle costs is its STORES, however they
  //  are issued (profiles/r04/r04h_dec_ab_unshuffle_loads_vs_stores.txt, r04l_dec_ab_background_stores_experiment.txt).)
  auto group = [&](auto gtag) {
    constexpr int G = decltype(gtag)::value;
    for (; e + 1024u * G <= N; e += 1024u * G) {
      Rows<T> x[4 * G];
#pragma unroll
      for (int h = 0; h < G; h++) {
        const uint32_t eh = e + 1024u * (uint32_t)h;
#pragma unroll
        for (int j = 0; j < T; j++) {
          const bool in_span = eh >= lo[j] && eh < hi[j];          // wave-uniform
          const bool reg = in_span && ((small >> j) & 1u);
          if (reg) { x[4 * h].r[j] = x[4 * h + 1].r[j] = x[4 * h + 2].r[j] = x[4 * h + 3].r[j] = pr[j]; }   // scalar branch: no load at all
          else if (in_span && ((self >> j) & 1u)) {                 // the plane's own period in front of the span, address per lane
            const uint32_t o = ob[j] & 0xffffffu, m = (1u << (ob[j] >> 24)) - 1u;
            const uint32_t q = eh + l4 - o;
            x[4 * h].r[j] = ld4_plane(pl[j] + o + (q & m)); x[4 * h + 1].r[j] = ld4_plane(pl[j] + o + ((q + 256u) & m));
            x[4 * h + 2].r[j] = ld4_plane(pl[j] + o + ((q + 512u) & m)); x[4 * h + 3].r[j] = ld4_plane(pl[j] + o + ((q + 768u) & m));
          } else {
            const gu8* p = in_span ? pat + (size_t)j * SPAN_PAT + (eh & (SPAN_PAT - 1u)) : pl[j] + eh;
            x[4 * h].r[j] = ld4_plane(p + l4); x[4 * h + 1].r[j] = ld4_plane(p + l4 + 256u); x[4 * h + 2].r[j] = ld4_plane(p + l4 + 512u); x[4 * h + 3].r[j] = ld4_plane(p + l4 + 768u);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4 * G; k++) unshuffle_store<T>(dst, e + 256u * (uint32_t)k, lane, x[k]);
    }
  };
  // bytes in flight per wave and group: G * 1024 * T (typesize 8: 8 KiB per quarter-group)
  // Round 5 (profiles/r05a_dec_ab.txt, reference-written bench19 chunks): 16 KiB in flight are worth - 4.5 % at typesize 2 (8 quarter-groups) and
  // - 1.8 % at typesize 4 (4), + 4 % at typesize 8 (2: the second register set pushes the callee's saves up); 32 KiB are slower everywhere.
  constexpr int GDEF = T == 2 ? 8 : (T == 4 ? 4 : 1);
  constexpr int GMAIN = BAMD_UNSH_BYTES ? (BAMD_UNSH_BYTES / (1024 * T) > 1 ? BAMD_UNSH_BYTES / (1024 * T) : 1) : GDEF;
  if constexpr (GMAIN > 1) group(std::integral_constant<int, GMAIN>{});
  group(std::integral_constant<int, 1>{});
  // behind the last multiple of 1024 nothing is skipped
  for (; e + 256u <= N; e += 256u) {
    Rows<T> x;
#pragma unroll
    for (int j = 0; j < T; j++) x.r[j] = g_ld4(pl[j] + e + l4);
    unshuffle_store<T>(dst, e, lane, x);
  }
  // tail: fewer than 256 elements, then the bytes that do not form a whole element (never a split block: src is the scratch)
#pragma unroll
  for (int j = 0; j < T; j++)
    for (uint32_t el = e + (uint32_t)lane; el < N; el += 64u) dst[(size_t)el * T + j] = pl[j][el];
  for (uint32_t k = N * T + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}

// Typesize 16 (round 3).  The same steps - lane l owns elements e + 4l .. e + 4l + 3, 4 bytes of every plane in, 64 contiguous
// bytes out - but sixteen planes' worth of span bounds, flags and plane pointers do not fit the scalar registers next to the
// loop: they live in a LANE TABLE (lane j holds plane j's words) and a step reads what it needs with v_readlane.  Two steps
// (8 KiB of loads) in flight per wave, as many bytes as typesize 8 keeps in flight with four.  Planes whose period divides 256
// read their dword from the first row of the pattern table in every iteration (the same 256 bytes: cache-resident) instead
// of holding it in a register per plane.
__device__ __forceinline__ void unshuffle_store16(gu8* dst, uint32_t e, int lane, const uint32_t (&x)[16]) {
  // x[j]: plane j's bytes of elements e + 4l .. e + 4l + 3.  After the quad transpose the lane holds elements e + 16Q + i + 4k (k = 0..3):
  // store k of a quad covers 64 contiguous bytes
  gu8* o = dst + (size_t)(e + 16u * ((uint32_t)lane >> 2) + ((uint32_t)lane & 3u)) * 16u;
  uint32_t t[4][4];
#pragma unroll
  for (int q = 0; q < 4; q++)
    transpose4x4(quad_byte_transpose(x[4 * q], lane), quad_byte_transpose(x[4 * q + 1], lane), quad_byte_transpose(x[4 * q + 2], lane), quad_byte_transpose(x[4 * q + 3], lane),
                 t[q][0], t[q][1], t[q][2], t[q][3]);
#pragma unroll
  for (int k = 0; k < 4; k++) st16_dst(o + 64 * k, make_uint4(t[0][k], t[1][k], t[2][k], t[3][k]));      // element e + 16Q + i + 4k
}
__device__ __attribute__((noinline)) void unshuffle_block_wave_16(const gu8* src_, gu8* dst_, uint32_t bsize_, int lane, const uint32_t* spans_,
                                                                  const gu8* pat_, const StreamDesc* sds_, uint32_t pstride_) {
  const gu8* src = uni_ptr(src_); gu8* dst = uni_ptr(dst_); const gu8* pat = uni_ptr(pat_);
  const uint32_t bsize = uni(bsize_), pstride = uni(pstride_);
  const uint64_t sv = (uint64_t)spans_, dv = (uint64_t)sds_;
  const uint32_t* spans = (const uint32_t*)(((uint64_t)uni((uint32_t)(sv >> 32)) << 32) | uni((uint32_t)sv));
  const StreamDesc* sds = (const StreamDesc*)(((uint64_t)uni((uint32_t)(dv >> 32)) << 32) | uni((uint32_t)dv));
  const uint32_t N = bsize / 16u;
  // the lane table: plane j = lane & 15
  uint32_t t_lo = 0u, t_hi = 0u, t_kind = 0u, t_ob = 0u, t_om = 0u;       // kind 1: period divides 256, 2: self span, 0: pattern table (or no span)
  uint64_t t_pl;
  {
    const uint32_t j = (uint32_t)lane & 15u;
    t_pl = (uint64_t)(src + (size_t)j * pstride);
    if (spans) {
      const uint32_t w = spans[2u * j];
      t_lo = w & ~1023u; t_hi = spans[2u * j + 1u];
      if (w & SPAN_RAW) { t_pl = (uint64_t)as_global(sds[j].in); t_hi = 0u; }
      else if (t_hi > t_lo) {
        if (w & SPAN_SMALL) t_kind = 1u;
        else if (w & SPAN_SELF) { t_kind = 2u; t_ob = g_ld4(pat + (size_t)j * SPAN_PAT); t_om = g_ld4(pat + (size_t)j * SPAN_PAT + 4) - 1u; }
      } else t_hi = 0u;
    }
  }
  const uint32_t t_pl_lo = (uint32_t)t_pl, t_pl_hi = (uint32_t)(t_pl >> 32);
  const uint32_t l4 = 4u * (uint32_t)lane;
  auto plane = [&](int j) -> const gu8* {
    return (const gu8*)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)t_pl_hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)t_pl_lo, j));
  };
  uint32_t e = 0;
  for (; e + 512u <= N; e += 512u) {
    uint32_t a[16], b[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)t_lo, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)t_hi, j);
      const gu8* pl = plane(j);
      if (!(e >= lo && e < hi)) { a[j] = ld4_plane(pl + e + l4); b[j] = ld4_plane(pl + e + l4 + 256u); continue; }      // wave-uniform
      const uint32_t kind = (uint32_t)__builtin_amdgcn_readlane((int)t_kind, j);
      const gu8* pj = pat + (size_t)j * SPAN_PAT;
      if (kind == 1u) { a[j] = b[j] = g_ld4(pj + l4); }
      else if (kind == 2u) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)t_ob, j), m = (uint32_t)__builtin_amdgcn_readlane((int)t_om, j);
        const uint32_t q = e + l4 - o;
        a[j] = ld4_plane(pl + o + (q & m)); b[j] = ld4_plane(pl + o + ((q + 256u) & m));
      } else {
        const gu8* p = pj + (e & (SPAN_PAT - 1u));
        a[j] = ld4_plane(p + l4); b[j] = ld4_plane(p + l4 + 256u);
      }
    }
    unshuffle_store16(dst, e, lane, a); unshuffle_store16(dst, e + 256u, lane, b);
  }
  // behind the last multiple of 1024 (span bounds are multiples of 1024) nothing is skipped
  for (; e + 256u <= N; e += 256u) {
    uint32_t a[16];
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = g_ld4(plane(j) + e + l4);
    unshuffle_store16(dst, e, lane, a);
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const gu8* pl = plane(j);
    for (uint32_t el = e + (uint32_t)lane; el < N; el += 64u) dst[(size_t)el * 16u + j] = pl[el];
  }
  for (uint32_t k = N * 16u + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}

__device__ __attribute__((noinline)) void unshuffle_block_wave(const uint8_t* src, uint8_t* dst, uint32_t bsize_, int typesize_, int lane,
                                                               const uint32_t* spans_, const uint8_t* pat, const StreamDesc* sds_, uint32_t pstride_ = 0) {
  // arguments of a real (non-inlined) call count as divergent for the compiler: without the readfirstlanes below the loop
  // bounds, the span decisions and the plane pointers all lived in VGPRs (64-bit pointer pairs spilled inside the loop)
  const uint32_t bsize = uni(bsize_); const int typesize = (int)uni((uint32_t)typesize_);
  const uint64_t sv = (uint64_t)spans_, dv = (uint64_t)sds_;
  const uint32_t* spans = (const uint32_t*)(((uint64_t)uni((uint32_t)(sv >> 32)) << 32) | uni((uint32_t)sv));
  const StreamDesc* sds = (const StreamDesc*)(((uint64_t)uni((uint32_t)(dv >> 32)) << 32) | uni((uint32_t)dv));
  const uint32_t pstride = uni(pstride_) ? uni(pstride_) : bsize / (uint32_t)typesize;      // 0: the plain plane-major image
  if (typesize == 8) unshuffle_block_wave_T<8>(uni_ptr(as_global(src)), uni_ptr(as_global(dst)), bsize, lane, spans, uni_ptr(as_global(pat)), sds, pstride);
  else if (typesize == 4) unshuffle_block_wave_T<4>(uni_ptr(as_global(src)), uni_ptr(as_global(dst)), bsize, lane, spans, uni_ptr(as_global(pat)), sds, pstride);
  else if (typesize == 2) unshuffle_block_wave_T<2>(uni_ptr(as_global(src)), uni_ptr(as_global(dst)), bsize, lane, spans, uni_ptr(as_global(pat)), sds, pstride);
  else unshuffle_block_wave_16(as_global(src), as_global(dst), bsize, lane, spans, as_global(pat), sds, pstride);      // typesize 16, out of line: its registers must not count against the loops above
}

// ---------------------------------------------------------------------------------------------
// The fused unshuffle of the entropy-coded formats (round 3).  A Zstd or zlib chunk is normally not split (blosc/blosc.c:929-959:
// only BloscLZ / LZ4 split in the default mode): a block is ONE stream that decodes into the plane-major scratch.  The wave that
// has decoded it (k_zstd_exec, k_zstd_streams, k_zlib_streams) transposes it into the destination right away - its own stores,
// no hand-off between waves, the bandwidth-bound transposes run underneath the latency-bound decoding of the other waves instead
// of in a kernel of their own behind them (k_unshuffle: 3.4 ms per 8 GiB).  No spans, and fewer steps in flight than
// unshuffle_block_wave: k_zstd_exec runs 8 waves per SIMD on 64 registers.
// ---------------------------------------------------------------------------------------------
#ifdef BAMD_WAVE_EMU
inline unsigned long long g_emu_own_block_unshuffles = 0;      // emulator only: blocks unshuffled by the wave that decoded them (fused_unshuffle_own_block)
#endif
template <int T, int STEPS>
__device__ __forceinline__ void unshuffle_plain_T(const gu8* src, gu8* dst, uint32_t bsize, int lane) {
  const uint32_t N = bsize / T;
  uint32_t e = 0;
  for (; e + 256u * STEPS <= N; e += 256u * STEPS) {
    Rows<T> x[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++)
#pragma unroll
      for (int j = 0; j < T; j++) x[s].r[j] = ld4_plane(src + (size_t)j * N + e + 256u * s + unsh_l4<T>(lane));
#pragma unroll
    for (int s = 0; s < STEPS; s++) unshuffle_store<T>(dst, e + 256u * s, lane, x[s]);
  }
  for (; e + 256u <= N; e += 256u) {
    Rows<T> x;
#pragma unroll
    for (int j = 0; j < T; j++) x.r[j] = g_ld4(src + (size_t)j * N + e + unsh_l4<T>(lane));
    unshuffle_store<T>(dst, e, lane, x);
  }
#pragma unroll
  for (int j = 0; j < T; j++)
    for (uint32_t el = e + (uint32_t)lane; el < N; el += 64u) dst[(size_t)el * T + j] = src[(size_t)j * N + el];
  for (uint32_t k = N * T + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}
__device__ __forceinline__ void unshuffle_plain_16(const gu8* src, gu8* dst, uint32_t bsize, int lane) {
  const uint32_t N = bsize / 16u;
  uint32_t e = 0;
  for (; e + 256u <= N; e += 256u) {
    uint32_t a[16];
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = ld4_plane(src + (size_t)j * N + e + 4u * (uint32_t)lane);
    unshuffle_store16(dst, e, lane, a);
  }
#pragma unroll
  for (int j = 0; j < 16; j++)
    for (uint32_t el = e + (uint32_t)lane; el < N; el += 64u) dst[(size_t)el * 16u + j] = src[(size_t)j * N + el];
  for (uint32_t k = N * 16u + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}
__device__ __attribute__((noinline)) void unshuffle_block_plain(const uint8_t* src_, uint8_t* dst_, uint32_t bsize_, int typesize_, int lane) {
  const gu8* src = uni_ptr(as_global(src_)); gu8* dst = uni_ptr(as_global(dst_));
  const uint32_t bsize = uni(bsize_); const int typesize = (int)uni((uint32_t)typesize_);
  if (typesize == 8) unshuffle_plain_T<8, 2>(src, dst, bsize, lane);
  else if (typesize == 4) unshuffle_plain_T<4, 4>(src, dst, bsize, lane);
  else if (typesize == 2) unshuffle_plain_T<2, 4>(src, dst, bsize, lane);
  else unshuffle_plain_16(src, dst, bsize, lane);
}
// Called by the wave that has just written the whole of stream `sd` (the only stream of its block) into the scratch.
__device__ __forceinline__ void fused_unshuffle_own_block(const ChunkDesc* c, const BlockDesc* b, int lane) {
  if (!(uni(c->mode) & CH_FUSED_UNSHUF) || uni((uint32_t)b->nstreams) != 1u) return;
#ifdef BAMD_WAVE_EMU
  if (lane == 0) g_emu_own_block_unshuffles++;
#endif
  BAMD_WAIT_STORES();     // this wave's own stores are what it reads back: no other wave, no other cache involved
  BAMD_MEM_SYNC();
  const uint32_t blk = uni((uint32_t)b->blk);
  unshuffle_block_plain(c->filt + (size_t)blk * filt_block_stride(*c), c->dst + (size_t)blk * (size_t)uni((uint32_t)c->blocksize),
                        uni((uint32_t)b->bsize), (int)uni((uint32_t)c->typesize), lane);
}

// ---------------------------------------------------------------------------------------------
// Fused byte-unshuffle of one block by ONE wavefront for every OTHER typesize up to 32 (3, 5, 6, 7, 9 ... 15, 17 ... 32; round 4: the
// north star says typesize 1 - 32, these took three passes).  blosc/shuffle-generic.h:61-81 for any typesize; the tiled form of
// shuffle-sse2.c:217-270 for the wide ones.  No register transposes here - a typesize that is not a power of two has no fixed byte pattern
// per lane - but the wave's own LDS (9 KiB, idle once its stream is done): per pass of 256 elements a dword of every plane per lane (a wave
// load reads 256 contiguous bytes of a plane), its four bytes into an element-major tile in LDS, and the tile out again as 16-byte pieces:
// every store instruction writes 1 KiB of contiguous destination.  Blocks of typesize > 16 are never split (blosc/blosc.c:929-959):
// their one stream's wave unshuffles its own block.  No periodic spans on this path (sp.enabled stays 0 in decode_one_stream).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool unshuffle_generic_T(uint32_t T) { return T >= 3u && T <= 32u && T != 4u && T != 8u && T != 16u; }
__device__ __attribute__((noinline)) void unshuffle_block_generic(volatile uint32_t* lds_, const uint8_t* src_, uint8_t* dst_, uint32_t bsize_, uint32_t T_, uint32_t pstride_, int lane) {
  const uint64_t lv = (uint64_t)lds_;
  lu8* S = (lu8*)(BAMD_LAS uint32_t*)(volatile uint32_t*)(((uint64_t)uni((uint32_t)(lv >> 32)) << 32) | uni((uint32_t)lv));
  const gu8* src = uni_ptr(as_global(src_)); gu8* dst = uni_ptr(as_global(dst_));
  const uint32_t bsize = uni(bsize_), T = uni(T_), N = bsize / T;
  const uint32_t pstride = uni(pstride_) ? uni(pstride_) : N;
  static_assert(256u * 32u <= DR_LDS_BYTES, "the element-major tile of a pass must fit the LDS a wave owns");
  const uint32_t l4 = 4u * (uint32_t)lane;
  // as many 256-element tiles per pass as the LDS holds (narrow elements: a pass of ONE tile of typesize 3 would move 768 bytes)
  const uint32_t M = (DR_LDS_BYTES / 256u) / T;
  uint32_t e = 0;
  while (e + 256u <= N) {
    const uint32_t m_here = (N - e) / 256u < M ? (N - e) / 256u : M;
    for (uint32_t m = 0; m < m_here; m++) {
      lu8* Sm = S + m * 256u * T;
      if ((T & 3u) == 0u) {                              // whole dwords per element: four planes -> 4 x 4 byte transpose -> one dword of each of the lane's four elements
        for (uint32_t j = 0; j < T; j += 4u) {
          const gu8* p = src + (size_t)j * pstride + e + 256u * m + l4;
          const uint32_t r0 = g_ld4(p), r1 = g_ld4(p + pstride), r2 = g_ld4(p + 2 * (size_t)pstride), r3 = g_ld4(p + 3 * (size_t)pstride);
          uint32_t t0, t1, t2, t3;
          transpose4x4(r0, r1, r2, r3, t0, t1, t2, t3);
          BAMD_LAS uint32_t* d = (BAMD_LAS uint32_t*)(Sm + l4 * T + j);
          d[0] = t0; d[T / 4u] = t1; d[2u * (T / 4u)] = t2; d[3u * (T / 4u)] = t3;
        }
        continue;
      }
      for (uint32_t j = 0; j < T; j += 4u) {             // four planes per round: their loads travel together
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) v[k] = (j + k < T) ? g_ld4(src + (size_t)(j + k) * pstride + e + 256u * m + l4) : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++)
          if (j + k < T) {
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++) Sm[(l4 + i) * T + j + k] = (uint8_t)(v[k] >> (8u * i));
          }
      }
    }
    LDS_ORDER(); BAMD_LDS_SYNC();
    gu8* out = dst + (size_t)e * T;
    for (uint32_t q = (uint32_t)lane; q < 16u * T * m_here; q += 64u) st16_dst(out + 16u * q, l_ld16(S + 16u * q));
    LDS_ORDER(); BAMD_LDS_SYNC();
    e += 256u * m_here;
  }
  // fewer than 256 elements, then the bytes that do not form a whole element
  for (uint32_t k = (uint32_t)lane; k < (N - e) * T; k += 64u) { const uint32_t el = k / T, j = k - el * T; dst[(size_t)e * T + k] = src[(size_t)j * pstride + e + el]; }
  for (uint32_t k = N * T + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}

// ---------------------------------------------------------------------------------------------
// Fused bit-unshuffle of one block by ONE wavefront (typesize 1, 2, 4; 8 since round 5), round 4: the wave that completes a block's last stream runs it
// out of its XCD's L2, exactly like the byte unshuffle above - there is no k_bitunshuffle pass over the batch any more (3.9 ms per 8 GiB on
// config #3, 2 x nbytes of traffic that SURVEY 8d says must not be credited).  Inverse of blosc_internal_bitshuffle
// (blosc/shuffle.c:393-443, bitshuffle-generic.c:125-139, :208-220): the filtered block is 8 T bit rows of N / 8 bytes (row 8 j + b = bit b
// of byte j of every element).  Per pass of 2048 elements every lane owns 32 of them: one dword of every bit row (a wave load reads 256
// contiguous bytes of a row), 8 x 8 bit-matrix transposes in registers (k_filters.hip: bit_transpose8), its 32 T element bytes into the
// wave's LDS (stream decoding is over: the rings are free) in chunks 16 bytes apart (bank-conflict-free 128-bit accesses), and out again as
// 16-byte pieces in element order: every store instruction writes 1 KiB of contiguous destination.  The reference's corner rules are kept:
// filter not applied when bsize < T (blosc.c:608-609), whole block copied when the element count is not a multiple of 8
// (shuffle.c:412-414), trailing bsize mod T bytes copied.
// ---------------------------------------------------------------------------------------------
// EPL = elements a lane owns per pass: 32 (typesize 1 / 2 / 4: a dword of every bit row per lane) or 16 (typesize 8, round 5: two bytes of each of
// its 64 rows per lane - a wave load still reads 128 contiguous bytes of a row - so that the pass's 8 KiB of element bytes fit the wave's LDS
// and 32 registers)
template <int T, int EPL>
__device__ __forceinline__ void bitunshuffle_pass(lu8* S, const gu8* src, gu8* dst, uint32_t rowlen, uint32_t e0, uint32_t nchunks, int lane) {
  constexpr uint32_t CB = (uint32_t)EPL * T, CS = CB + 16u, NDW = CB / 4u;
  constexpr int G = EPL / 8;
  static_assert(EPL == 32 || EPL == 16, "a dword or two bytes of every bit row per lane");
  static_assert(64u * CS <= DR_LDS_BYTES, "the staging tile of a pass must fit the LDS a wave owns (a 4 KiB history ring - profiles/r04/r04u_* - would need 32-chunk passes)");
  const uint32_t m0 = e0 >> 3, t = (uint32_t)lane;
  if (t < nchunks) {
    uint32_t w[NDW];
#pragma unroll
    for (uint32_t k = 0; k < NDW; k++) w[k] = 0u;
#pragma unroll
    for (int j = 0; j < T; j++) {
      uint32_t rw[8];
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const gu8* rp = src + (size_t)(8 * j + b) * rowlen + m0 + (uint32_t)G * t;
        rw[b] = G == 4 ? g_ld4(rp) : g_ld2(rp);
      }
#pragma unroll
      for (int g = 0; g < G; g++) {
        uint64_t v = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) v |= (uint64_t)((rw[b] >> (8 * g)) & 0xffu) << (8 * b);
        v = bit_transpose8(v);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int idx = (8 * g + k) * T + j;
          w[idx >> 2] |= (uint32_t)((v >> (8 * k)) & 0xff) << (8 * (idx & 3));
        }
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < NDW / 4u; k++) l_st16(S + t * CS + 16u * k, make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]));
  }
  LDS_ORDER(); BAMD_LDS_SYNC();
  gu8* out = dst + (size_t)e0 * T;
  const uint32_t npieces = nchunks * CB / 16u;
#pragma unroll
  for (uint32_t i = 0; i < CB / 16u; i++) {
    const uint32_t q = (uint32_t)lane + 64u * i, off = 16u * q, c = off / CB;
    if (q < npieces) st16_dst(out + off, l_ld16(S + c * CS + (off - c * CB)));
  }
  LDS_ORDER(); BAMD_LDS_SYNC();
}
template <int T>
__device__ void bitunshuffle_block_wave_T(lu8* S, const gu8* src, gu8* dst, uint32_t bsize, int lane) {
  constexpr int EPL = T == 8 ? 16 : 32;
  const uint32_t N = bsize / T;
  if (bsize < T || (N & 7u)) {                         // not filtered at all / copied verbatim by the filter
    for (uint32_t k = 16u * (uint32_t)lane; k < bsize; k += 1024u) {
      if (k + 16u <= bsize) g_st16(dst + k, g_ld16(src + k));
      else for (uint32_t t = k; t < bsize; t++) dst[t] = src[t];
    }
    return;
  }
  const uint32_t rowlen = N >> 3;
  uint32_t e0 = 0;
  for (; e0 + 64u * EPL <= N; e0 += 64u * EPL) bitunshuffle_pass<T, EPL>(S, src, dst, rowlen, e0, 64u, lane);
  if (N - e0 >= (uint32_t)EPL) { const uint32_t nch = (N - e0) / (uint32_t)EPL; bitunshuffle_pass<T, EPL>(S, src, dst, rowlen, e0, nch, lane); e0 += (uint32_t)EPL * nch; }
  // fewer than EPL elements left (a multiple of 8): one byte of every bit row per lane, eight elements each
  if ((uint32_t)lane < ((N - e0) >> 3)) {
    const uint32_t m = (e0 >> 3) + (uint32_t)lane;
#pragma unroll
    for (int j = 0; j < T; j++) {
      uint64_t v = 0;
#pragma unroll
      for (int b = 0; b < 8; b++) v |= (uint64_t)src[(size_t)(8 * j + b) * rowlen + m] << (8 * b);
      v = bit_transpose8(v);
#pragma unroll
      for (int k = 0; k < 8; k++) dst[(size_t)(8u * m + (uint32_t)k) * T + (uint32_t)j] = (uint8_t)(v >> (8 * k));
    }
  }
  for (uint32_t k = N * T + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}
__device__ __forceinline__ bool bitunshuffle_fused_T(int T) { return T == 1 || T == 2 || T == 4 || T == 8; }
__device__ __attribute__((noinline)) void bitunshuffle_block_wave(volatile uint32_t* lds_, const uint8_t* src_, uint8_t* dst_, uint32_t bsize_, int typesize_, int lane) {
  const uint64_t lv = (uint64_t)lds_;
  lu8* S = (lu8*)(BAMD_LAS uint32_t*)(volatile uint32_t*)(((uint64_t)uni((uint32_t)(lv >> 32)) << 32) | uni((uint32_t)lv));
  const gu8* src = uni_ptr(as_global(src_)); gu8* dst = uni_ptr(as_global(dst_));
  const uint32_t bsize = uni(bsize_); const int T = (int)uni((uint32_t)typesize_);
  if (T == 8) bitunshuffle_block_wave_T<8>(S, src, dst, bsize, lane);
  else if (T == 4) bitunshuffle_block_wave_T<4>(S, src, dst, bsize, lane);
  else if (T == 2) bitunshuffle_block_wave_T<2>(S, src, dst, bsize, lane);
  else bitunshuffle_block_wave_T<1>(S, src, dst, bsize, lane);
}

// One stream, start to finish.  Deliberately NOT inlined into the queue loop below: with the decoders
// inlined, the compiler restructured the loop with partial exec masks and re-read the ticket with lane 0
// masked off (an endless loop on stream 0).  A real call keeps the loop's control flow trivial.
__device__ __attribute__((noinline)) void decode_one_stream(StreamDesc* sd, int32_t* status, volatile uint32_t* scr,
                                                            const ChunkDesc* chunks, const BlockDesc* blocks, uint32_t* blk_done, int lane,
                                                            uint32_t sid, uint32_t* spans, uint8_t* pat, uint32_t* plane_cost
#ifdef BAMD_PROFILE_DECODE
                                                            , uint32_t* profslot
#endif
                                                            ) {
  PROF_DECL
  const gu8* in = as_global(sd->in);
  const int32_t csize = (int32_t)uni((uint32_t)sd->in_size);
  const int32_t want = (int32_t)uni((uint32_t)sd->out_size);
  gu8* out = as_global(sd->out);
  if (csize < 0) return;  // chain error already recorded by the plan kernel
  const ChunkDesc* c = chunks + uni((uint32_t)sd->chunk);
  const uint32_t mode = uni(c->mode);
  const uint32_t gb = uni((uint32_t)sd->aux);
  const BlockDesc* b = blocks + gb;
  const uint32_t nstreams = uni((uint32_t)b->nstreams);
  SpanCtx sp;
  sp.enabled = ((mode & CH_FUSED_UNSHUF) && nstreams == uni((uint32_t)c->typesize) && spans && !unshuffle_generic_T(uni((uint32_t)c->typesize))) ? 1u : 0u;
  sp.lo = 0; sp.hi = 0; sp.off = 0; sp.pat = uni_ptr(as_global(pat)) + (size_t)sid * SPAN_PAT;
#ifdef BAMD_LOO_PLANES
  sp.loo = (sp.enabled && (((uint32_t)(BAMD_LOO_PLANES) >> ((sid - uni((uint32_t)b->first_stream)) & 31u)) & 1u)) ? 1u : 0u;
#endif
  const uint64_t cost_t0 = __builtin_amdgcn_s_memtime();
  int got;
  bool raw_in_place = false;
  if (sd->fmt == FMT_ZSTD || sd->fmt == FMT_ZLIB) {
    return;               // k_zstd_* / k_zlib_streams own every stream of those chunks, the ones stored raw included (round 3)
  } else if (csize == want) {    // split stored raw (blosc/blosc.c:773-776)
    // fused split blocks: the unshuffle reads the plane where it lies in the chunk (SPAN_RAW), no copy to the scratch
    raw_in_place = sp.enabled != 0u;
    if (!raw_in_place) wave_copy_disjoint(out, in, (uint32_t)want, lane);
    got = want;
  } else if (sd->fmt == FMT_LZ4) {
    got = lz4_decode_wave(in, csize, out, want, scr, lane, sp PROF_PASS);
  } else {
    got = blosclz_decode_wave(in, csize, out, want, scr, lane, sp);
  }
#ifdef BAMD_PROFILE_DECODE
  PROF_LAP(13);
  prof_.c[15] = (uint32_t)(prof_.t0 >> 6);
  if (lane == 0 && profslot) for (int i_ = 0; i_ < 16; i_++) profslot[i_] = prof_.c[i_];
#endif
  if (lane == 0) {
    sd->result = got;
    if (got != want) atomicMin(&status[sd->chunk], (int32_t)ST_BADCODEC);  // blosc.c:780-782
    // cost feedback for the host's queue order (queue_order.h: build_xcd_queues): cycles per plane index
    if (plane_cost) atomicAdd(plane_cost + ((sid - (uint32_t)b->first_stream) & 255u), (uint32_t)((__builtin_amdgcn_s_memtime() - cost_t0) >> 10));
  }
  // ---- fused unshuffle: the wave that completes a block's LAST stream transposes the block ----
  if (!(mode & (CH_FUSED_UNSHUF | CH_FUSED_BITUNSH)) || got != want) return;
  if (spans && lane == 0) {
    spans[2 * (size_t)sid] = raw_in_place ? SPAN_RAW : (sp.lo | ((sp.hi && sp.off <= 256u) ? SPAN_SMALL : 0u) | ((sp.hi && sp.off > SPAN_PAT) ? SPAN_SELF : 0u));
    spans[2 * (size_t)sid + 1] = sp.hi;
  }
  // The hand-off, and why it holds on gfx950 without a release fence:
  //  * all streams of one block are handed out from the SAME per-XCD queue (k_decode_streams; engine.hip: probe_topology checks that
  //    workgroups really are dealt round-robin to 8 XCDs whose id the waves read) - producers and consumer share ONE L2;
  //  * the vector L1 (TCP) is write-through: a global store leaves the CU for the L2 of its XCD, and vmcnt counts a store down only
  //    when that L2 has acknowledged it (CDNA3/4 ISA, s_waitcnt: "memory writes: decremented when the data has been written to the L2
  //    cache") - after `s_waitcnt vmcnt(0)` below every row this wave flushed IS in the XCD's L2;
  //  * the counter is an atomic executed AT that L2 (agent scope: `global_atomic_add ... sc1`), issued after the wait: whoever sees
  //    the incremented value reads the same L2 the rows are in;
  //  * the consumer's `buffer_inv sc1` (the acquire fence behind the test) drops its CU's L1 lines, so its loads go to that L2.
  // What an agent-scope RELEASE would add is `buffer_wbl2`: a write-back of the XCD's whole L2 to memory, needed only when the reader
  // sits behind a DIFFERENT L2.  Timed in round 4 (profiles/r04/r04s_dec_ab_handoff_release_vs_relaxed.txt): 7.63 against 4.09 ms per
  // 8 GiB with one release per stream - kept relaxed.  tests/test_gpu_handoff_stress.py runs two contexts' persistent kernels on the
  // same XCDs for 2 000 launches and compares every byte; the single-queue fallback (no in-kernel hand-off at all) is what a device
  // without this topology gets.
  BAMD_WAIT_STORES();
  uint32_t old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(&blk_done[gb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (uint32_t)__builtin_amdgcn_readlane((int)old, 0);
  if (old + 1u != nstreams) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // buffer_inv sc1: this CU's L1 forgets the block's scratch lines
  const size_t boff = (size_t)uni((uint32_t)b->blk) * (size_t)uni((uint32_t)c->blocksize);
  if (mode & CH_FUSED_BITUNSH) {
    bitunshuffle_block_wave(scr, c->filt + (size_t)uni((uint32_t)b->blk) * filt_block_stride(*c), c->dst + boff, uni((uint32_t)b->bsize), (int)uni((uint32_t)c->typesize), lane);
    return;
  }
  const bool split = spans && nstreams == uni((uint32_t)c->typesize);
  const uint32_t fs = uni((uint32_t)b->first_stream);

#ifdef BAMD_PROFILE_DECODE
  const uint64_t ut0 = __builtin_amdgcn_s_memtime();
#endif
  if (unshuffle_generic_T(uni((uint32_t)c->typesize))) {
    unshuffle_block_generic(scr, c->filt + (size_t)uni((uint32_t)b->blk) * filt_block_stride(*c), c->dst + boff, uni((uint32_t)b->bsize), uni((uint32_t)c->typesize),
                            filt_plane_stride(*c, uni((uint32_t)b->bsize), (int)nstreams), lane);
    return;
  }
  unshuffle_block_wave(c->filt + (size_t)uni((uint32_t)b->blk) * filt_block_stride(*c), c->dst + boff, uni((uint32_t)b->bsize), (int)uni((uint32_t)c->typesize), lane,
                       split ? spans + 2 * (size_t)fs : nullptr, pat + (size_t)fs * SPAN_PAT, sd - (sid - fs),
                       filt_plane_stride(*c, uni((uint32_t)b->bsize), (int)nstreams));
#ifdef BAMD_PROFILE_DECODE
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0 && profslot) { profslot[6] = (uint32_t)(__builtin_amdgcn_s_memtime() - ut0); profslot[7] = 1u; }      // cycles of the block's fused unshuffle, charged to the stream that arrived last
#endif
}

#ifndef BAMD_DEC_MINWAVES
#define BAMD_DEC_MINWAVES 4   // waves per SIMD.  Round 4: the 9.25 KiB of LDS a wave owns (dec_ring.h) allow 16 - 17 waves per CU anyway, and at 96 registers the ring decoder spills inside its step (6.7 against 4.5 ms, profiles/r04/r04b_dec_ab_ring_variants.txt).  Before:  Round 2: 6 = 5 < 7 < 8 (spills cost more than occupancy gives).  Round 3, with the pipelined loops of dec_bulk.h and the fused unshuffle (both keep two register sets in flight): 5 beats 6 by 4-6 % on every data set (profiles/r03/r03h_dec_ab_rowfill_off.txt)
#endif                       // 6 = 5 (5.90 / 5.95 ms) < 7 (6.33) < 8 (7.0): spills cost more than occupancy gives
constexpr int DEC_WAVES_PER_CU = 4 * BAMD_DEC_MINWAVES;
// Persistent launch: the grid is sized to what the chip can hold (engine.hip) and every wave pulls stream
// indices from a ticket queue until it is empty.  Streams of one batch differ by 1000x in cost (a byte
// plane of zeros is two sequences, a noisy plane thousands); with one workgroup per stream the dispatcher
// kept only 4-9 waves per CU busy, with queues every resident wave stays busy.
// There is one queue per XCD (the host deals whole BLOCKS round-robin to the 8 queues, qlist/qoff): a wave
// reads its XCC id and serves only that queue, so every stream of a block - and the block's fused
// unshuffle - runs on one XCD and shares one L2.  Placement is used for speed and for the cheap hand-off
// above; a wave that finds its own queue empty simply exits (queues are equal shares of the same work).
__global__ __launch_bounds__(64 * DEC_WAVES, BAMD_DEC_MINWAVES) void k_decode_streams(
    StreamDesc* __restrict__ streams, int32_t* __restrict__ status, uint32_t* __restrict__ tickets /*[8]*/,
    const int32_t* __restrict__ qlist, const int32_t* __restrict__ qoff /*[9]*/,
    const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks, uint32_t* __restrict__ blk_done,
    uint32_t* __restrict__ spans, uint8_t* __restrict__ pat, uint32_t* __restrict__ plane_cost, int single_queue
#ifdef BAMD_PROFILE_DECODE
    , uint32_t* __restrict__ profbuf
#endif
    ) {
  __shared__ __attribute__((aligned(16))) uint32_t scr[DEC_WAVES][DR_LDS_BYTES / 4];   // per wave: 64 scratch dwords of the batched steps | input ring | history ring (dec_ring.h)
  static_assert(DEC_WAVES == 1, "one stream per wave, one wave per workgroup");
  const int lane = threadIdx.x & 63;
  // HW_REG_XCC_ID[3:0]; queue 0 for everybody in the single-queue fallback (no in-kernel hand-offs there)
  const uint32_t xcc = single_queue ? 0u : (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
  const uint32_t qbase = (uint32_t)qoff[xcc], qlen = (uint32_t)qoff[xcc + 1] - qbase;
  uint32_t t = take_ticket(tickets + xcc, lane);
  uint32_t ndone = 0;      // streams this wave took: summed into plane_cost[256], the host checks the total
  while (t < qlen) {
    const uint32_t sid = (uint32_t)qlist[qbase + t];
#ifdef BAMD_PROFILE_DECODE
    decode_one_stream(streams + sid, status, scr[0], chunks, blocks, blk_done, lane, sid, spans, pat, plane_cost, profbuf ? profbuf + (size_t)sid * 16 : nullptr);
#else
    decode_one_stream(streams + sid, status, scr[0], chunks, blocks, blk_done, lane, sid, spans, pat, plane_cost);
#endif
    ndone++;
    t = take_ticket(tickets + xcc, lane);
  }
  if (lane == 0 && ndone) atomicAdd(plane_cost + 256, ndone);
}

}  // namespace bamd
