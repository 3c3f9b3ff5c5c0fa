// k_zstd.hip — Zstandard frame decode on the GPU (codec row K8 of SURVEY §8a, decode direction):
// zstd_wrap_decompress (blosc/blosc.c:515-522) -> ZSTD_decompress for every stream of a Zstd chunk.  blosc
// writes one frame with one <= 128 KiB block per blosc block; the decoder is nevertheless the
// general single-frame one (several blocks; raw / RLE / compressed; treeless literals; repeat tables).
//
// First version, correctness before speed.  One wavefront per frame, persistent waves + ticket queue:
//   * the inherently serial parts - headers, FSE / Huffman table builds, the Huffman literal streams (four
//     lanes, one per stream), the FSE sequence stream - run the plain-C++ primitives of zstd_serial.h on single
//     lanes with their tables in LDS (12 KiB per wave); tests/test_zstd_serial_cpu.py checks exactly that
//     code on the CPU;
//   * lane 0 decodes the sequences 64 at a time into LDS; the whole wave then EXECUTES them with the same
//     wave-cooperative copies as the LZ4 decoder (wave_copy_disjoint for literals, wave_match_copy for
//     matches: byte-exact forward semantics for overlapping distances).
// The literals of a block are regenerated into a per-stream scratch (ChunkDesc::stage) first.
// Chunks of this codec are never "fused": k_unshuffle / k_bitunshuffle run afterwards as kernels of their own.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"
#include "wave_prims.h"
#include "zstd_serial.h"

namespace bamd {

// phase profiling (prof build, scripts/zstd_phase.py): cycles per frame in
// 0 literal header + Huffman table, 1 Huffman streams, 2 sequence tables + stream init, 3 sequence decode (lane 0),
// 4 sequence execution, 5 rest (block copies, tails); 8 sequences, 9 literals, 10 blocks
#ifdef BAMD_PROFILE_DECODE
struct ZProf { uint64_t t; uint32_t c[16]; };
#define ZP_ARG , ZProf& zp
#define ZP_PASS , zp
#define ZP_LAP(i) do { __builtin_amdgcn_s_waitcnt(0); const uint64_t t_ = __builtin_amdgcn_s_memtime(); zp.c[i] += (uint32_t)(t_ - zp.t); zp.t = t_; } while (0)
#define ZP_ADD(i, v) zp.c[i] += (uint32_t)(v)
#else
#define ZP_ARG
#define ZP_PASS
#define ZP_LAP(i)
#define ZP_ADD(i, v)
#endif

struct ZstdLds {
  uint16_t huf[2048];
  uint32_t ll[512], of[512], ml[512];
  uint32_t ftab[64];
  uint16_t next[256];
  int16_t norm[64];
  uint8_t w[256];
  uint32_t seq[3 * 64];
};

__device__ __forceinline__ uint32_t lane0_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 0); }

// fill n bytes with v
__device__ __forceinline__ void wave_fill(gu8* dst, uint32_t v, uint32_t n, int lane) {
  for (uint32_t k = (uint32_t)lane; k < n; k += 64u) dst[k] = (uint8_t)v;
}

// Executes the sequences seq[base .. base + m), m <= 16; lane r < m owns sequence r (the scheme of the LZ4
// decoder's batched step, k_decode.hip): a DPP row scan gives every sequence its literal source and its output
// position; every lane copies its own literals; matches whose source ends before this group's output are
// independent of the group and are copied by their lanes at once; the others - long, or reading what this
// group produces - follow in stream order through wave_match_copy.
// ---------------------------------------------------------------------------------------------
// Groups with several DEPENDENT matches, assembled in LDS (round 3; the Zstd twin of lz4_step_lds in k_decode.hip).  A frame the
// reference writes for a shuffled 128 KiB block is ~4000 sequences with matches of a few dozen bytes whose sources lie in the
// output of the sequences just before them: zstd_exec16 below ran those one after the other, each waiting for its own load to come
// back from L2 before it could store (k_zstd_exec: 16 ms per 8 GiB of reference-written bench19 frames against 6.3 ms for frames
// written here, whose matches are long).  Here the last ZXB_HIST bytes of output come in with ONE load, literals and independent
// matches go to memory AND to the buffer, and the dependent matches become LDS copies in stream order (a match whose distance
// is shorter than its length is the periodic extension of the bytes before it: every lane reads only bytes that are final)
// whose values the lanes also store to memory.
// ---------------------------------------------------------------------------------------------
#ifdef BAMD_WAVE_EMU
// emulator only: how often the round-3 paths ran (the CPU tests assert that reference-written frames really take them):
// [0] groups assembled in LDS, [1] of them with the history kept from the group before, [2] with literals out of the window, [3] frames whose
// sequences k_zstd_seq decoded, [4] blocks unshuffled by the wave that decoded them
inline unsigned long long g_emu_zstd_paths[5] = {0, 0, 0, 0, 0};
#endif
constexpr uint32_t ZXB_HIST = 1024u, ZXB_STEP = 2048u, ZXB_MAXM = 512u;
constexpr uint32_t ZXB_LW = 1024u, ZXB_LW_OFF = ZXB_HIST + ZXB_STEP + 128u;      // the literal window: ZXB_LW bytes of the frame's literal buffer, behind the group buffer
constexpr uint32_t ZXB_WORDS = (ZXB_LW_OFF + ZXB_LW + 16u) / 4u;
// what a run of LDS-assembled groups keeps in the buffer: the history (the last ZXB_HIST bytes of output) and a window of the literals
struct ZxState { uint32_t hist_valid, lw_valid, lw_base; };
#ifndef BAMD_ZXB_GROUPSTORE
#define BAMD_ZXB_GROUPSTORE 1    // an LDS-assembled group goes to memory in one pass of coalesced stores (0: every piece is stored where it is made)
#endif
#ifndef BAMD_ZXB_MIN_REST
#define BAMD_ZXB_MIN_REST 2      // dependent matches a group needs before the LDS form is used (0: never)
#endif
typedef volatile __attribute__((address_space(3))) uint8_t zlds_u8;
__device__ __forceinline__ void zlds_st16(zlds_u8* l, const uint4& v) { v4u32 t = {v.x, v.y, v.z, v.w}; *(volatile __attribute__((address_space(3))) v4u32_una*)l = t; }
// lane_copy_disjoint (wave_prims.h) with a second destination in LDS
template <bool G = true>      // G: to memory as well as to LDS
__device__ __forceinline__ void lane_copy_dual(gu8* d, zlds_u8* l, const gu8* s, uint32_t n) {
  if (n >= 16u) {
    uint32_t k = 0;
    for (; k + 64u <= n; k += 64u) {
      const uint4 a = ld16u(s + k), b = ld16u(s + k + 16), c = ld16u(s + k + 32), e = ld16u(s + k + 48);
      if (G) { st16u(d + k, a); st16u(d + k + 16, b); st16u(d + k + 32, c); st16u(d + k + 48, e); }
      zlds_st16(l + k, a); zlds_st16(l + k + 16, b); zlds_st16(l + k + 32, c); zlds_st16(l + k + 48, e);
    }
    const uint32_t r = n - k;
    uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a;
    if (r >= 16u) a = ld16u(s + k);
    if (r >= 32u) b = ld16u(s + k + 16);
    if (r >= 48u) c = ld16u(s + k + 32);
    const uint4 z = ld16u(s + n - 16u);
    if (r >= 16u) { if (G) st16u(d + k, a); zlds_st16(l + k, a); }
    if (r >= 32u) { if (G) st16u(d + k + 16, b); zlds_st16(l + k + 16, b); }
    if (r >= 48u) { if (G) st16u(d + k + 32, c); zlds_st16(l + k + 32, c); }
    if (G) st16u(d + n - 16u, z);
    zlds_st16(l + n - 16u, z);
  } else if (n >= 8u) {
    const uint64_t a = g_ld8(s), b = g_ld8(s + n - 8u);
    if (G) { *(BAMD_GAS u64una*)d = a; *(BAMD_GAS u64una*)(d + n - 8u) = b; }
    *(volatile __attribute__((address_space(3))) u64una*)l = a; *(volatile __attribute__((address_space(3))) u64una*)(l + n - 8u) = b;
  } else if (n >= 4u) {
    const uint32_t a = g_ld4(s), b = g_ld4(s + n - 4u);
    if (G) { g_st4(d, a); g_st4(d + n - 4u, b); }
    *(volatile __attribute__((address_space(3))) u32una*)l = a; *(volatile __attribute__((address_space(3))) u32una*)(l + n - 4u) = b;
  } else if (n) {
    const uint8_t a = s[0], b = s[n >> 1], c = s[n - 1u];
    if (G) { d[0] = a; d[n >> 1] = b; d[n - 1u] = c; }
    l[0] = a; l[n >> 1] = b; l[n - 1u] = c;
  }
}
// the same with the source in LDS (the literal window, the history): no memory load at all
__device__ __forceinline__ uint4 zlds_ld16(const zlds_u8* l) { const v4u32 t = *(const volatile __attribute__((address_space(3))) v4u32_una*)l; return make_uint4(t.x, t.y, t.z, t.w); }
template <bool G = true>
__device__ __forceinline__ void lane_copy_dual_lds(gu8* d, zlds_u8* l, const zlds_u8* s, uint32_t n) {
  if (n >= 16u) {
    uint32_t k = 0;
    for (; k + 32u <= n; k += 32u) {
      const uint4 a = zlds_ld16(s + k), b = zlds_ld16(s + k + 16);
      if (G) { st16u(d + k, a); st16u(d + k + 16, b); }
      zlds_st16(l + k, a); zlds_st16(l + k + 16, b);
    }
    const uint32_t r = n - k;                     // 0..31 bytes left: one whole piece and one that ends exactly at n
    uint4 a = make_uint4(0, 0, 0, 0);
    if (r >= 16u) a = zlds_ld16(s + k);
    const uint4 z = zlds_ld16(s + n - 16u);
    if (r >= 16u) { if (G) st16u(d + k, a); zlds_st16(l + k, a); }
    if (G) st16u(d + n - 16u, z);
    zlds_st16(l + n - 16u, z);
  } else if (n >= 8u) {
    const uint64_t a = *(const volatile __attribute__((address_space(3))) u64una*)s, b = *(const volatile __attribute__((address_space(3))) u64una*)(s + n - 8u);
    if (G) { *(BAMD_GAS u64una*)d = a; *(BAMD_GAS u64una*)(d + n - 8u) = b; }
    *(volatile __attribute__((address_space(3))) u64una*)l = a; *(volatile __attribute__((address_space(3))) u64una*)(l + n - 8u) = b;
  } else if (n >= 4u) {
    const uint32_t a = *(const volatile __attribute__((address_space(3))) u32una*)s, b = *(const volatile __attribute__((address_space(3))) u32una*)(s + n - 4u);
    if (G) { g_st4(d, a); g_st4(d + n - 4u, b); }
    *(volatile __attribute__((address_space(3))) u32una*)l = a; *(volatile __attribute__((address_space(3))) u32una*)(l + n - 4u) = b;
  } else if (n) {
    const uint8_t a = s[0], b = s[n >> 1], c = s[n - 1u];
    if (G) { d[0] = a; d[n >> 1] = b; d[n - 1u] = c; }
    l[0] = a; l[n >> 1] = b; l[n - 1u] = c;
  }
}
// (a real call: the common group must not pay for its registers).  indep / dep: this lane's match is copied at once / in stream order.
#ifndef BAMD_ZXB_INLINE
#define BAMD_ZXB_INLINE 1        // zstd_exec16_lds inlined into its callers: bench19 10.8 -> 10.5 ms, linspace 6.8 -> 5.8 (profiles/r03/r03zm_zent_split_rcp_modulo_inline.txt); 0: a real call
#endif
#if BAMD_ZXB_INLINE
#define ZXB_FN __device__ __forceinline__
#else
#define ZXB_FN __device__ __attribute__((noinline))
#endif
ZXB_FN void zstd_exec16_lds(gu8* out_, const gu8* lit_, volatile uint32_t* xbuf_generic, uint32_t ll, uint32_t ml, uint32_t off,
                                                          uint32_t excl, uint32_t lexcl, bool indep, bool dep, uint32_t op_, uint32_t lp_, int lane,
                                                          uint32_t hist_valid_, uint32_t total_out_, uint32_t lw_, uint32_t regen_) {
  // lw_: 0 = literals from memory; else bit 0 set and (lw_ >> 1) = the window's base when it already covers this group's literals,
  // 0xffffffff = use the window but (re)load it from lp first
  gu8* out = uni_ptr(out_); const gu8* lit = uni_ptr(lit_);
  const uint32_t op = uni(op_), lp = uni(lp_), total_out = uni(total_out_), lw = uni(lw_), regen = uni(regen_);
  const bool hist_valid = uni(hist_valid_) != 0u;           // the buffer's first ZXB_HIST bytes already are out[op - H, op): the group before left them there
  zlds_u8* lb = (zlds_u8*)(volatile __attribute__((address_space(3))) uint32_t*)xbuf_generic;
  zlds_u8* lwin = lb + ZXB_LW_OFF;
  constexpr uint32_t H = ZXB_HIST;
#ifdef BAMD_WAVE_EMU
  if (lane == 0) { g_emu_zstd_paths[0]++; g_emu_zstd_paths[1] += hist_valid ? 1 : 0; g_emu_zstd_paths[2] += lw ? 1 : 0; }
#endif
  uint32_t lbase = lw >> 1;
  if (lw == 0xffffffffu) {                                   // the literal window from lp on (never a byte behind the literal buffer)
    lbase = lp;
    const uint32_t q = lp + 16u * (uint32_t)lane;
    if (q + 16u <= regen) zlds_st16(lwin + 16u * (uint32_t)lane, g_ld16(lit + q));
    else for (uint32_t t = q; t < regen && t < q + 16u; t++) lwin[t - lp] = lit[t];
  }
  if (!hist_valid) zlds_st16(lb + 16u * (uint32_t)lane, g_ld16(out + op - H + 16u * (uint32_t)lane));      // the history: one load
  BAMD_LDS_SYNC();
  // BAMD_ZXB_GROUPSTORE: everything is assembled in LDS only and the finished group leaves in ONE pass of coalesced 16-byte stores (below),
  // instead of a dozen small stores - literal runs, match pieces, 64 single bytes per dependent match - per group
  constexpr bool GS = BAMD_ZXB_GROUPSTORE != 0;
  if (ll) {                                                  // literals (ll <= 256 here)
    if (lw) lane_copy_dual_lds<!GS>(out + op + excl, lb + H + excl, lwin + (lp + lexcl - lbase), ll);
    else lane_copy_dual<!GS>(out + op + excl, lb + H + excl, lit + lp + lexcl, ll);
  }
  if (indep) {
    gu8* d = out + op + excl + ll;
    if (lw && off <= excl + ll + H) lane_copy_dual_lds<!GS>(d, lb + H + excl + ll, lb + H + excl + ll - off, ml);      // the source lies in the buffer's history
    else lane_copy_dual<!GS>(d, lb + H + excl + ll, d - off, ml);
  }
  BAMD_LDS_SYNC();
  uint32_t rest = (uint32_t)__ballot(dep) & 0xffffu;
  while (rest) {
    const int sl = __builtin_ctz(rest);
    rest &= rest - 1u;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)ml, sl), o = (uint32_t)__builtin_amdgcn_readlane((int)off, sl);
    const uint32_t mr = (uint32_t)__builtin_amdgcn_readlane((int)excl, sl) + (uint32_t)__builtin_amdgcn_readlane((int)ll, sl);
    // periodic extension of the o bytes before the match (o >= m: a plain copy; m <= ZXB_MAXM)
    // k mod o per lane with a float reciprocal (k, o < 512: the quotient is exact or one too small, which the compare repairs).  The
    // integer division this replaces was 23 scalar instructions per match on a scalar unit that SQ counters show 83 % busy in this kernel
    // (profiles/r03/r03zl_zstd_sq_counters.txt: 393 scalar + 307 vector instructions per group of 16 sequences)
    const float ro = o < m ? __builtin_amdgcn_rcpf((float)o) : 0.0f;
    for (uint32_t k = (uint32_t)lane; k < m; k += 64u) {
      uint32_t kk = k;
      if (o < m) { kk = k - (uint32_t)((float)k * ro) * o; kk = kk >= o ? kk - o : kk; }
      const uint8_t v = lb[H + mr - o + kk];
      lb[H + mr + k] = v;
      if (BAMD_ZXB_GROUPSTORE == 0) out[op + mr + k] = v;
    }
    BAMD_LDS_SYNC();
  }
  if (BAMD_ZXB_GROUPSTORE) {                                 // the group, whole: 1 KiB per store instruction, never a byte behind it
    for (uint32_t q = 16u * (uint32_t)lane; q < total_out; q += 1024u) {
      if (q + 16u <= total_out) st16u(out + op + q, zlds_ld16(lb + H + q));
      else for (uint32_t t = q; t < total_out; t++) out[op + t] = lb[H + t];
    }
  }
  // slide: the last H bytes of history + group become the next group's history, so that a run of such groups loads its history once
  // (every group's load of the bytes the group before had just stored was a trip to L2 that the group's first LDS copy waited for)
  {
    const v4u32 t = *(volatile __attribute__((address_space(3))) v4u32_una*)(lb + total_out + 16u * (uint32_t)lane);
    BAMD_LDS_SYNC();
    *(volatile __attribute__((address_space(3))) v4u32_una*)(lb + 16u * (uint32_t)lane) = t;
    BAMD_LDS_SYNC();
  }
}

// (ll_b, ml_b, off_b: lane i of the batch holds sequence i; this group is sequences base .. base + m)
// xbuf: ZXB_WORDS words of LDS of this wave for the LDS-assembled form above, or nullptr
__device__ __forceinline__ bool zstd_exec16(uint32_t ll_b, uint32_t ml_b, uint32_t off_b, int base, int m, uint8_t* out_, uint32_t cap, uint32_t& op,
                                            const uint8_t* lit_, uint32_t& lp, uint32_t regen, int lane, volatile uint32_t* xbuf = nullptr, ZxState* zx = nullptr) {
  gu8* out = as_global(out_); const gu8* lit = as_global(lit_);
  const bool mine = lane < m;
  const uint32_t sel = (uint32_t)(base + lane) & 63u;
  const uint32_t ll_g = bperm(sel, ll_b), ml_g = bperm(sel, ml_b), off_g = bperm(sel, off_b);
  const uint32_t ll = mine ? ll_g : 0u, ml = mine ? ml_g : 0u;
  const uint32_t off = mine ? off_g : 1u;
  const uint32_t tot = ll + ml;
  uint32_t incl = tot, lincl = ll;                       // inclusive prefix sums over the 16 rank lanes (one DPP row)
  incl += row_shr<1>(incl); incl += row_shr<2>(incl); incl += row_shr<4>(incl); incl += row_shr<8>(incl);
  lincl += row_shr<1>(lincl); lincl += row_shr<2>(lincl); lincl += row_shr<4>(lincl); lincl += row_shr<8>(lincl);
  const uint32_t excl = incl - tot, lexcl = lincl - ll;
  const uint32_t total_out = (uint32_t)__builtin_amdgcn_readlane((int)incl, m - 1);
  const uint32_t total_lit = (uint32_t)__builtin_amdgcn_readlane((int)lincl, m - 1);
  if ((uint64_t)lp + total_lit > (uint64_t)regen || (uint64_t)op + total_out > (uint64_t)cap) return false;
  if (__ballot(mine && (off == 0u || off > op + excl + ll))) return false;     // a source before the start of the output
  uint32_t biglit = (uint32_t)__ballot(mine && ll > 256u) & 0xffffu;      // long runs: the whole wave copies them
  if (BAMD_ZXB_MIN_REST && xbuf && !biglit && op >= ZXB_HIST && total_out <= ZXB_STEP) {
    // several matches that read this group's own output (or the bytes just before it): assemble the group in LDS
    const bool indep0 = mine && ml <= 128u && off >= excl + tot;
    const bool dep0 = mine && ml && !indep0;
    const uint32_t ndep = (uint32_t)__builtin_popcountll(__ballot(dep0));
    const bool fits = __ballot(dep0 && (ml > ZXB_MAXM || off > excl + ll + ZXB_HIST)) == 0ull;      // short, and the source inside the buffer
    if (ndep >= (uint32_t)BAMD_ZXB_MIN_REST && fits) {
      uint32_t lw = 0u;                        // literals out of the window?  (see zstd_exec16_lds)
      if (zx && total_lit <= ZXB_LW) {
        const bool covered = zx->lw_valid && lp >= zx->lw_base && lp + total_lit <= zx->lw_base + ZXB_LW;
        lw = covered ? ((zx->lw_base << 1) | 1u) : 0xffffffffu;
        if (!covered) { zx->lw_valid = 1u; zx->lw_base = lp; }
      }
      zstd_exec16_lds(out, lit, xbuf, ll, ml, off, excl, lexcl, indep0, dep0, op, lp, lane, zx ? zx->hist_valid : 0u, total_out, lw, regen);
      if (zx) zx->hist_valid = 1u;
      op += total_out; lp += total_lit;
      return true;
    }
  }
  if (zx) zx->hist_valid = 0u;               // this group goes to memory only
  while (biglit) {
    const int sl = __builtin_ctz(biglit);
    biglit &= biglit - 1u;
    wave_copy_disjoint(out + op + (uint32_t)__builtin_amdgcn_readlane((int)excl, sl), lit + lp + (uint32_t)__builtin_amdgcn_readlane((int)lexcl, sl),
                       (uint32_t)__builtin_amdgcn_readlane((int)ll, sl), lane);
  }
  if (mine && ll && ll <= 256u) lane_copy_disjoint(out + op + excl, lit + lp + lexcl, ll);   // literals: disjoint buffers, every lane its own run
  const bool indep = mine && ml <= 128u && off >= excl + tot;      // source ends at or before op: nothing of this group in it
  if (indep) { gu8* d = out + op + excl + ll; lane_copy_disjoint(d, d - off, ml); }   // source ends at or before op: disjoint from everything this group writes
  uint32_t rest = (uint32_t)__ballot(mine && !indep) & 0xffffu;
  while (rest) {
    const int sl = __builtin_ctz(rest);
    rest &= rest - 1u;
    const uint32_t mm = (uint32_t)__builtin_amdgcn_readlane((int)ml, sl), oo = (uint32_t)__builtin_amdgcn_readlane((int)off, sl);
    const uint32_t pos = op + (uint32_t)__builtin_amdgcn_readlane((int)excl, sl) + (uint32_t)__builtin_amdgcn_readlane((int)ll, sl);
    wave_match_copy(out, pos, oo, mm, lane);
  }
  op += total_out; lp += total_lit;
  return true;
}

// ---------------------------------------------------------------------------------------------
// Wave-uniform sequence decoding.  The FSE sequence stream is serial, but nothing says it has to run on ONE LANE
// with its bytes fetched from memory one load at a time (the first version: 2200 cycles per sequence,
// profiles/r02/r02_e_zstd_decode_phases.txt).  Here every lane runs the same scalar program: the stream lives in a
// 256-byte register window (one dword per lane, refilled with one coalesced load as the reader moves towards the
// start), bits come out of a 64-bit accumulator refilled with v_readlane, table entries are uniform LDS reads, and
// lane i keeps the fields of the batch's i-th sequence in its own registers for the execution step.
// Same rules as zd::seq_begin / zd::seq_next (zstd_serial.h), which stay the CPU-checked statement of the format.
// ---------------------------------------------------------------------------------------------
struct WBack {
  const gu8* p; int len;
  uint32_t win; int wbase;           // lane l holds stream bytes [wbase + 4 l, + 4); wbase is a multiple of 4, may be negative
  uint64_t acc; int nacc; int off; int bytepos;
};
__device__ __forceinline__ void wb_fetch(WBack& b, int lane) {
  const int q = b.wbase + 4 * lane;
  uint32_t v = 0;
  if (q >= 0 && q + 4 <= b.len) v = g_ld4(b.p + q);
  else for (int k = 0; k < 4; k++) if (q + k >= 0 && q + k < b.len) v |= (uint32_t)b.p[q + k] << (8 * k);
  b.win = v;
}
// 4 stream bytes at pos .. pos + 3 (little endian), 0 <= pos, pos + 4 <= len
__device__ __forceinline__ uint32_t wb_get4(WBack& b, int pos, int lane) {
  if (pos < b.wbase) { b.wbase = (pos - 192) & ~3; wb_fetch(b, lane); }
  const int r = pos - b.wbase, i = r >> 2, sh = (r & 3) * 8;
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)b.win, i);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)b.win, (i + 1) & 63);
  return (uint32_t)((((uint64_t)c << 32) | a) >> sh);
}
__device__ __forceinline__ uint32_t wb_get1(WBack& b, int pos, int lane) {
  if (pos < b.wbase) { b.wbase = (pos - 192) & ~3; wb_fetch(b, lane); }
  const int r = pos - b.wbase;
  return ((uint32_t)__builtin_amdgcn_readlane((int)b.win, r >> 2) >> ((r & 3) * 8)) & 0xffu;
}
// The accumulator keeps its `nacc` valid bits at the TOP of the 64-bit word: a read is two shifts, bits past the
// start of the stream read as zero by themselves (`off` going negative is what the callers check).
__device__ __forceinline__ void wb_refill(WBack& b, int lane) {      // afterwards nacc > 32 unless the stream start is near
  if (b.nacc > 32) return;
  if (b.bytepos >= 4) {
    const uint32_t v = wb_get4(b, b.bytepos - 4, lane);
    b.bytepos -= 4; b.acc |= (uint64_t)v << (32 - b.nacc); b.nacc += 32;
    return;
  }
  while (b.nacc <= 56 && b.bytepos > 0) { b.bytepos--; b.acc |= (uint64_t)wb_get1(b, b.bytepos, lane) << (56 - b.nacc); b.nacc += 8; }
}
__device__ __forceinline__ bool wb_init(WBack& b, const gu8* p_, int len_, int lane) {
  // arguments of a real (non-inlined) device function count as divergent: without these two readfirstlanes the whole
  // decoder below is compiled as exec-masked VALU code instead of a scalar program
  const gu8* p = uni_ptr(p_); const int len = (int)uni((uint32_t)len_);
  if (len <= 0) return false;
  b.p = p; b.len = len; b.wbase = (len - 252) & ~3;      // the last byte sits at window offset <= 254
  wb_fetch(b, lane);
  const uint32_t lastb = wb_get1(b, len - 1, lane);
  if (lastb == 0u) return false;
  const int top = 31 - __builtin_clz(lastb);
  b.bytepos = len - 1;
  b.nacc = top;
  b.acc = top ? (uint64_t)(lastb & ((1u << top) - 1u)) << (64 - top) : 0ull;
  b.off = (len - 1) * 8 + top;
  return true;
}
// n <= 32 bits, caller has refilled (nacc > 32 or the stream start reached)
__device__ __forceinline__ uint32_t wb_take(WBack& b, int n) {
  const uint32_t v = n ? (uint32_t)(b.acc >> (64 - n)) : 0u;
  b.acc = n ? b.acc << n : b.acc;
  b.nacc = b.nacc > n ? b.nacc - n : 0;
  b.off -= n;
  return v;
}
__device__ __forceinline__ uint32_t wb_read(WBack& b, int n, int lane) { wb_refill(b, lane); return wb_take(b, n); }
struct WSeqState { WBack b; uint32_t sl, so, sm; uint32_t rep[3]; };
__device__ __forceinline__ bool wseq_begin(WSeqState& st, int al_ll, int al_of, int al_ml, const gu8* src, int len, int lane) {
  if (!wb_init(st.b, src, len, lane)) return false;
  st.sl = wb_read(st.b, (int)uni((uint32_t)al_ll), lane); st.so = wb_read(st.b, (int)uni((uint32_t)al_of), lane); st.sm = wb_read(st.b, (int)uni((uint32_t)al_ml), lane);
  return true;
}
// the tables are in LDS and their addresses in registers: through zd::SeqTabs (a struct the caller holds by reference,
// i.e. in scratch memory) every sequence paid three scratch loads and three flat loads
typedef __attribute__((address_space(3))) const uint32_t* lds_u32p;
__device__ __forceinline__ bool wseq_next(WSeqState& st, lds_u32p lle, lds_u32p ofe, lds_u32p mle, bool last, zd::Seq& q, int lane) {
  const uint32_t el = uni(lle[st.sl]), eo = uni(ofe[st.so]), em = uni(mle[st.sm]);
  const int lc = zd::fse_sym(el), oc = zd::fse_sym(eo), mc = zd::fse_sym(em);
  if (oc > 31 || mc > 52 || lc > 35) return false;
  const uint32_t ov = (1u << oc) + wb_read(st.b, oc, lane);          // <= 31 bits
  wb_refill(st.b, lane);                                              // <= 16 + 16 bits
  q.ml = zd::ml_base(mc) + wb_take(st.b, zd::ml_bits(mc));
  q.ll = zd::ll_base(lc) + wb_take(st.b, zd::ll_bits(lc));
  if (!last) {
    wb_refill(st.b, lane);                                            // <= 9 + 9 + 8 bits
    st.sl = zd::fse_base(el) + wb_take(st.b, zd::fse_nb(el));
    st.sm = zd::fse_base(em) + wb_take(st.b, zd::fse_nb(em));
    st.so = zd::fse_base(eo) + wb_take(st.b, zd::fse_nb(eo));
  }
  if (st.b.off < 0) return false;
  if (ov > 3) { q.off = ov - 3u; st.rep[2] = st.rep[1]; st.rep[1] = st.rep[0]; st.rep[0] = q.off; }
  else {
    uint32_t idx = ov - 1u;
    if (q.ll == 0) idx++;
    if (idx == 0) q.off = st.rep[0];
    else {
      q.off = idx < 3 ? (idx == 1 ? st.rep[1] : st.rep[2]) : st.rep[0] - 1u;
      if (q.off == 0) return false;
      if (idx > 1) st.rep[2] = st.rep[1];
      st.rep[1] = st.rep[0]; st.rep[0] = q.off;
    }
  }
  return true;
}

// one compressed block.  Returns true and advances op on success.  huf_valid / tabs state persist over the frame.
__device__ bool zstd_block_wave(const uint8_t* b, int size, uint8_t* out, uint32_t cap, uint32_t& op, uint8_t* lit, ZstdLds* L,
                                zd::Huf& huf, bool& huf_valid, zd::SeqTabs& tb, uint32_t* rep, int lane ZP_ARG) {
  ZP_LAP(5);
  // ---- literals section header (lane 0, broadcast) ----
  zd::LitHdr lh = {0, 0, 0, 1, 0};
  uint32_t ok = 0;
  if (lane == 0) ok = zd::lit_header(b, size, lh) ? 1u : 0u;
  if (!lane0_u32(ok)) return false;
  const int ltype = (int)lane0_u32((uint32_t)lh.type), regen = (int)lane0_u32((uint32_t)lh.regen), csize = (int)lane0_u32((uint32_t)lh.csize);
  const int nstreams = (int)lane0_u32((uint32_t)lh.nstreams);
  int p = (int)lane0_u32((uint32_t)lh.hdr);
  // every literal ends up in the output: more literals than output room left is corruption - and the literal
  // scratch of this stream is only as large as its output
  if ((uint32_t)regen > cap - op) return false;
  if (ltype == 0) {
    if (p + regen > size) return false;
    wave_copy_disjoint(as_global(lit), as_global(b + p), (uint32_t)regen, lane);
    p += regen;
  } else if (ltype == 1) {
    if (p + 1 > size) return false;
    wave_fill(as_global(lit), b[p], (uint32_t)regen, lane);
    p += 1;
  } else {
    if (p + csize > size) return false;
    const uint8_t* hs = b + p; int hlen = csize;
    if (ltype == 2) {
      int used = -1;
      if (lane == 0) { used = zd::huf_read_table(huf, hs, hlen, L->w, L->ftab, L->next, L->norm); }
      used = (int)lane0_u32((uint32_t)used);
      if (used < 0) return false;
      huf.maxbits = (int)lane0_u32((uint32_t)huf.maxbits);
      huf_valid = true;
      hs += used; hlen -= used;
    } else if (!huf_valid) return false;
    ZP_LAP(0);
    uint32_t good = 1;
    if (nstreams == 1) {
      if (lane == 0) good = zd::huf_decode_stream(huf, hs, hlen, lit, regen) ? 1u : 0u;
    } else {
      if (hlen < 6) return false;
      const int s1 = hs[0] | (hs[1] << 8), s2 = hs[2] | (hs[3] << 8), s3 = hs[4] | (hs[5] << 8), s4 = hlen - 6 - s1 - s2 - s3;
      const int q = (regen + 3) / 4;
      if (s4 < 1 || 3 * q > regen) return false;
      if (lane < 4) {                                          // four independent backward bit streams: one lane each
        const int so = lane == 0 ? 0 : (lane == 1 ? s1 : (lane == 2 ? s1 + s2 : s1 + s2 + s3));
        const int sl = lane == 0 ? s1 : (lane == 1 ? s2 : (lane == 2 ? s3 : s4));
        const int n = lane < 3 ? q : regen - 3 * q;
        good = zd::huf_decode_stream(huf, hs + 6 + so, sl, lit + lane * q, n) ? 1u : 0u;
      }
    }
    if (__ballot(good == 0u)) return false;
    p += csize;
    ZP_LAP(1);
  }
  ZP_ADD(9, regen); ZP_ADD(10, 1);
  // ---- sequences section ----
  int nseq = 0, u0 = -1;
  if (lane == 0) u0 = zd::seq_count(b + p, size - p, &nseq);
  u0 = (int)lane0_u32((uint32_t)u0); nseq = (int)lane0_u32((uint32_t)nseq);
  if (u0 < 0) return false;
  p += u0;
  uint32_t lp = 0;
  if (nseq == 0 && p != size) return false;       // nothing may follow the count of an empty sequences section
  if (nseq > 0) {
    if (p >= size) return false;
    const int modes = b[p++];
    if (modes & 3) return false;
    int used = -1;
    if (lane == 0) {
      int q = p, u;
      bool fine = true;
      if (fine && (u = zd::seq_table(tb.ll, tb.have_ll, 0, modes >> 6, b + q, size - q, L->norm, L->next)) >= 0) q += u; else fine = false;
      if (fine && (u = zd::seq_table(tb.of, tb.have_of, 1, (modes >> 4) & 3, b + q, size - q, L->norm, L->next)) >= 0) q += u; else fine = false;
      if (fine && (u = zd::seq_table(tb.ml, tb.have_ml, 2, (modes >> 2) & 3, b + q, size - q, L->norm, L->next)) >= 0) q += u; else fine = false;
      if (fine && size - q >= 1) used = q;
    }
    used = (int)lane0_u32((uint32_t)used);
    if (used < 0) return false;
    // the accuracy logs were set by lane 0: every lane needs them for the uniform decoder
    tb.ll.al = (int)lane0_u32((uint32_t)tb.ll.al); tb.of.al = (int)lane0_u32((uint32_t)tb.of.al); tb.ml.al = (int)lane0_u32((uint32_t)tb.ml.al);
    WSeqState st;
    st.rep[0] = uni(rep[0]); st.rep[1] = uni(rep[1]); st.rep[2] = uni(rep[2]);
    const lds_u32p lle = (lds_u32p)L->ll, ofe = (lds_u32p)L->of, mle = (lds_u32p)L->ml;
    if (!wseq_begin(st, tb.ll.al, tb.of.al, tb.ml.al, as_global(b + used), size - used, lane)) return false;
    ZP_LAP(2); ZP_ADD(8, nseq);
    for (int done = 0; done < nseq; done += 64) {
      const int m = nseq - done < 64 ? nseq - done : 64;
      uint32_t ll_b = 0, ml_b = 0, off_b = 1;
      bool fine = true;
      for (int i = 0; i < m; i++) {                            // serial in the stream, uniform across the wave
        zd::Seq q;
        if (!wseq_next(st, lle, ofe, mle, done + i + 1 == nseq, q, lane)) { fine = false; break; }
        if (lane == i) { ll_b = q.ll; ml_b = q.ml; off_b = q.off; }
      }
      if (fine && done + m == nseq && st.b.off != 0) fine = false;   // the bit stream must be consumed exactly
      if (!fine) return false;
      ZP_LAP(3);
      for (int g = 0; g < m; g += 16)                          // wave-parallel execution, 16 sequences per group
        if (!zstd_exec16(ll_b, ml_b, off_b, g, m - g < 16 ? m - g : 16, out, cap, op, lit, lp, (uint32_t)regen, lane)) return false;
      ZP_LAP(4);
    }
    rep[0] = st.rep[0]; rep[1] = st.rep[1]; rep[2] = st.rep[2];

  }
  const uint32_t rest = (uint32_t)regen - lp;
  if ((uint64_t)op + rest > (uint64_t)cap) return false;
  if (rest) wave_copy_disjoint(as_global(out) + op, as_global(lit) + lp, rest, lane);
  op += rest;
  return true;
}

// one frame -> out[0..cap); returns bytes produced, 0 on any error (zstd_wrap_decompress's contract)
__device__ __attribute__((noinline)) int zstd_decode_wave(const uint8_t* in, int n, uint8_t* out, int cap, uint8_t* lit, ZstdLds* L, int lane ZP_ARG) {
  long long fcs = -1; bool checksum = false;
  int ip = -1;
  if (lane == 0) ip = zd::frame_header(in, n, &fcs, &checksum);
  ip = (int)lane0_u32((uint32_t)ip);
  if (ip < 0) return 0;
  const uint32_t fcs_lo = lane0_u32((uint32_t)fcs), fcs_hi = lane0_u32((uint32_t)((unsigned long long)fcs >> 32));
  const long long fcs_u = (long long)(((unsigned long long)fcs_hi << 32) | fcs_lo);
  const bool has_checksum = lane0_u32(checksum ? 1u : 0u) != 0u;
  if (fcs_u >= 0 && fcs_u > cap) return 0;
  zd::Huf huf = {L->huf, 0}; bool huf_valid = false;
  zd::SeqTabs tb = {{L->ll, 0}, {L->of, 0}, {L->ml, 0}, false, false, false};
  uint32_t rep[3] = {1u, 4u, 8u};
  uint32_t op = 0;
  bool ok = false;
  for (;;) {
    if (ip + 3 > n) break;
    const uint32_t bh = (uint32_t)in[ip] | ((uint32_t)in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16);   // same address in every lane
    ip += 3;
    const int last = bh & 1u, type = (bh >> 1) & 3u, bsize = (int)(bh >> 3);
    if (type == 0) {
      if (ip + bsize > n || op + (uint32_t)bsize > (uint32_t)cap) break;
      wave_copy_disjoint(as_global(out) + op, as_global(in + ip), (uint32_t)bsize, lane);
      op += (uint32_t)bsize; ip += bsize;
    } else if (type == 1) {
      if (ip + 1 > n || op + (uint32_t)bsize > (uint32_t)cap) break;
      wave_fill(as_global(out) + op, in[ip], (uint32_t)bsize, lane);
      op += (uint32_t)bsize; ip += 1;
    } else if (type == 2) {
      if (bsize > (1 << 17) || ip + bsize > n) break;
      // the table state lives in lane 0's registers + LDS; al / have flags are broadcast after each block
      if (!zstd_block_wave(in + ip, bsize, out, (uint32_t)cap, op, lit, L, huf, huf_valid, tb, rep, lane ZP_PASS)) break;
      tb.ll.al = (int)lane0_u32((uint32_t)tb.ll.al); tb.of.al = (int)lane0_u32((uint32_t)tb.of.al); tb.ml.al = (int)lane0_u32((uint32_t)tb.ml.al);
      tb.have_ll = lane0_u32(tb.have_ll) != 0u; tb.have_of = lane0_u32(tb.have_of) != 0u; tb.have_ml = lane0_u32(tb.have_ml) != 0u;
      ip += bsize;
    } else break;
    if (last) { ok = true; break; }
  }
  if (!ok) return 0;
  if (has_checksum) {              // low 32 bits of XXH64 of the content (lane 0, serial: blosc itself never writes such frames)
    if (ip + 4 > n) return 0;
    uint32_t good = 0;
    BAMD_MEM_SYNC();                  // lane 0 reads what all lanes have stored
    if (lane == 0) {
      const uint32_t want = (uint32_t)in[ip] | ((uint32_t)in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16) | ((uint32_t)in[ip + 3] << 24);
      __builtin_amdgcn_s_waitcnt(0);
      good = (uint32_t)zd::xxh64(out, op) == want ? 1u : 0u;
    }
    if (!lane0_u32(good)) return 0;
    ip += 4;
  }
  if (ip != n) return 0;
  if (fcs_u >= 0 && fcs_u != (long long)op) return 0;
  return (int)op;
}

// Persistent waves over ALL streams of the launch; the streams of Zstd chunks are taken here, splits stored raw included
// (k_decode_streams leaves them alone); an unsplit block is unshuffled by the wave that decoded it (fused_unshuffle_own_block).
constexpr int ZSTD_WAVES_PER_CU = 12;
__global__ __launch_bounds__(64, 3) void k_zstd_streams(StreamDesc* __restrict__ streams, int nstreams, int32_t* __restrict__ status,
                                                     uint32_t* __restrict__ ticket, const ChunkDesc* __restrict__ chunks,
                                                     const BlockDesc* __restrict__ blocks, uint32_t* __restrict__ done,
                                                     const uint32_t* __restrict__ taken /* ZMeta words (k_zstd2.hip), 8 per stream; nullptr: take everything */
#ifdef BAMD_PROFILE_DECODE
                                                     , uint32_t* __restrict__ profbuf
#endif
                                                     ) {
  __shared__ ZstdLds lds;
  const int lane = threadIdx.x & 63;
  // Tickets.  Without the two-phase path every Zstd stream is this kernel's: one stream per ticket.  With it (taken != nullptr)
  // this kernel only sees what phase A left alone - frames of several blocks, raw / RLE blocks, splits stored raw: usually nothing -
  // so a ticket is a run of 64 consecutive streams, every lane looks at one of them and the wave walks the few that are its own.
  // (One ticket per stream made the empty case 65 536 atomics on one word: 0.8 ms per call for nothing)
  const uint32_t run = taken ? 64u : 1u;
  uint32_t ndone = 0;
  for (;;) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(ticket, run);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
    if (base >= (uint32_t)nstreams) break;
    const uint32_t cnt = (uint32_t)nstreams - base < run ? (uint32_t)nstreams - base : run;
    ndone += cnt;
    uint64_t todo = 1ull;
    if (taken) {
      const uint32_t s_l = base + (uint32_t)lane;
      const bool m = (uint32_t)lane < cnt && taken[8 * (size_t)s_l] == 0u && streams[s_l].fmt == FMT_ZSTD && streams[s_l].in_size >= 0;   // 0 = ZM_FALLBACK
      todo = __ballot(m);
    }
    while (todo) {
      const uint32_t sid = base + (uint32_t)__builtin_ctzll(todo);
      todo &= todo - 1ull;
      StreamDesc* sd = streams + sid;
      const int32_t csize = (int32_t)uni((uint32_t)sd->in_size), want = (int32_t)uni((uint32_t)sd->out_size);
      if (uni((uint32_t)sd->fmt) != (uint32_t)FMT_ZSTD || csize < 0) continue;
      const ChunkDesc* c = chunks + uni((uint32_t)sd->chunk);
      const BlockDesc* b = blocks + uni((uint32_t)sd->aux);
      int got;
      if (csize == want) {    // split stored raw (blosc/blosc.c:773-776)
        wave_copy_disjoint(uni_ptr(as_global(sd->out)), uni_ptr(as_global(sd->in)), (uint32_t)want, lane);
        got = want;
      } else {
        // literal scratch: this stream's slice of the chunk's `stage` area (same offset as its output)
        const size_t boff = (size_t)uni((uint32_t)b->blk) * (size_t)uni((uint32_t)c->blocksize) +
                            (size_t)(sid - uni((uint32_t)b->first_stream)) * (size_t)want;
#ifdef BAMD_PROFILE_DECODE
        ZProf zp; for (int i_ = 0; i_ < 16; i_++) zp.c[i_] = 0; zp.t = __builtin_amdgcn_s_memtime();
#endif
        got = zstd_decode_wave(sd->in, csize, sd->out, want, c->stage + boff, &lds, lane ZP_PASS);
#ifdef BAMD_PROFILE_DECODE
        ZP_LAP(5);
        if (profbuf && lane == 0) for (int i_ = 0; i_ < 16; i_++) profbuf[(size_t)sid * 16 + i_] = zp.c[i_];
#endif
      }
      if (lane == 0) {
        sd->result = got;
        if (got != want) atomicMin(&status[sd->chunk], (int32_t)ST_BADCODEC);   // blosc.c:780-782
      }
      if (got == want) fused_unshuffle_own_block(c, b, lane);
    }
  }
  if (lane == 0 && ndone) atomicAdd(done, ndone);
}

}  // namespace bamd
