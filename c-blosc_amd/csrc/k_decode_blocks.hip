// k_decode_blocks.hip — LDS-resident decode of byte-shuffled, split blocks: LZ decode + unshuffle of one block by
// ONE workgroup, no plane-major image of the block in HBM (rows D1 + K2 + K5/K7 of SURVEY §8a fused).
//
// Replaces, for blocks written with the byte shuffle and split into `typesize` streams (the BASELINE geometry),
//   blosc_d's split loop + blosc_internal_unshuffle    blosc/blosc.c:725-800, blosc/shuffle-generic.h:61-81
//   LZ4_decompress_safe                                  lz4.c:2023-2451
// and k_decode_streams' scratch round trip (k_decode.hip), which moved 26 GB per 8 GiB (profiles/r01_final_traffic.json).
//
// One workgroup = one block, wave j = byte plane j (stream j of the block).  Every wave keeps the most recent
// R = 8 KiB of its plane in an LDS ring; all waves advance in lock-step SLICES of S = 4 KiB of plane positions:
//   decode plane j up to the slice end (matches read their source from the ring: ~64-cycle LDS round trips instead of
//   L2 / HBM ones)  ->  s_barrier  ->  the workgroup transposes the slice out of the T rings straight into the
//   destination (coalesced 16-byte stores)  ->  next slice into the other half of the ring (R = 2 S: one barrier
//   per slice is enough, see decode_block).
// A match whose source lies further back than the ring reads it from `far`, a plane-major copy of the slices
// already written out that every wave keeps of ITS OWN plane (same wave stores and loads: no cross-wave ordering
// needed); a plane whose stream ends in one long periodic match (constant / short-period planes: half of bench19)
// writes no far copy at all and, once the ring holds a whole number of periods, costs nothing per slice.
//
// Algorithmic HBM bytes per block: csize read + bsize written.  Real traffic adds the far copy of the planes that
// need one (write once, read only by far matches).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"
#include "wave_prims.h"

namespace bamd {

// Rings come in two sizes: the planes with the most compressed bytes - the ones with real LZ work - get BD_RBIG, the
// others BD_RSMALL (decode_block); a slice is half the small ring, so that every ring holds two slices.
constexpr uint32_t BD_RBIG = 8192u, BD_RSMALL = 4096u;
constexpr uint32_t BD_NBIG = 2u;          // big rings per workgroup
constexpr uint32_t BD_S = BD_RSMALL / 2u; // slice: plane positions between two barriers
// History within BD_NEAR bytes of the write frontier is read from the ring, anything older from the far copy.  The gap
// to R keeps a history chunk (<= 1024 bytes + one 16-byte piece + the slack of a step's literal scatter) from
// overwriting ring slots that lanes of the SAME chunk still have to read: lanes of one chunk take different code paths
// (16-byte pieces, byte loops) and their loads and stores are not ordered against each other.
constexpr uint32_t BD_NEAR_GAP = 1088u;

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

// Optional phase profiling (-DBAMD_PROFILE_DECODE -> libblosc_amd_prof.so, scripts/bd_phase.py): wave-uniform counters,
// cycles from s_memtime.  slots: 0 batch steps, 1 sequences in them, 2 of those done one by one ("rest"), 3 tokens parsed by
// the scalar path, 4 cycles write-out + far copy, 5 history chunks with a far piece, 6 cycles window seek (fetch waited for),
// 7 cycles parse + chain of a step, 8 cycles decoding, 9 waiting at the slice barrier, 10 cycles of a step up to its rest loop,
// 11 rest loops, 12 far chunks, 13 scalar match phase
#ifdef BAMD_PROFILE_DECODE
struct BdProf { uint32_t c[16]; };
#define BDP_ARG , BdProf& bp
#define BDP_PASS , bp
#define BDP_ADD(i, v) bp.c[i] += (uint32_t)(v)
#define BDP_T0(t) const uint64_t t = __builtin_amdgcn_s_memtime()
#define BDP_LAP(i, t) bp.c[i] += (uint32_t)(__builtin_amdgcn_s_memtime() - t)
#else
#define BDP_ARG
#define BDP_PASS
#define BDP_ADD(i, v)
#define BDP_T0(t)
#define BDP_LAP(i, t)
#endif

// One plane's output: the LDS ring (positions [W - R, W) of the plane, W = write frontier) and the far copy
// (positions [0, flushed), plane-major in global memory).
struct PlaneOut {
  lu8* ring;            // `mask` + 1 bytes
  uint32_t mask;        // ring size - 1
  uint32_t near;        // ring size - BD_NEAR_GAP
  gu8* far;             // this plane's slice of the workgroup's far area
  uint32_t flushed;     // slice start: every position below it is in `far`
};

// n <= 1024 bytes from the compressed stream (global) to plane positions [pos, pos + n): the destination lies in
// one slice, so its ring image is contiguous
__device__ __forceinline__ void ring_put_global(const PlaneOut& o, uint32_t pos, const gu8* src, uint32_t n, int lane) {
  lu8* d = o.ring + (pos & o.mask);
  const uint32_t n16 = n >> 4;
  if ((uint32_t)lane < n16) l_st16(d + 16u * (uint32_t)lane, g_ld16(src + 16u * (uint32_t)lane));
  const uint32_t done = n16 << 4;
  if ((uint32_t)lane < n - done) d[done + lane] = src[done + lane];
}
__device__ __forceinline__ void ring_put_global_long(const PlaneOut& o, uint32_t pos, const gu8* src, uint32_t n, int lane) {
  uint32_t done = 0;
  while (n - done > 1024u) { ring_put_global(o, pos + done, src + done, 1024u, lane); done += 1024u; }
  ring_put_global(o, pos + done, src + done, n - done, lane);
}

// History chunk: plane positions [src, src + n) -> [dst, dst + n), n <= 1024, n <= dst - src (no overlap inside the
// chunk), W = write frontier of the ring (>= dst; everything at or above W - BD_NEAR is read from the ring, everything
// below from `far`: W - BD_NEAR <= flushed - 960).  16 bytes per lane; a piece that wraps around the ring end or
// straddles the ring / far border goes byte by byte (rare).
__device__ __forceinline__ void hist_copy(const PlaneOut& o, uint32_t dst, uint32_t src, uint32_t n, uint32_t W, int lane BDP_ARG) {
  const uint32_t off16 = 16u * (uint32_t)lane;
#ifdef BAMD_PROFILE_DECODE
  const bool anyfar = (int32_t)src < (int32_t)W - (int32_t)o.near;
  BDP_ADD(5, anyfar ? 1 : 0);
  BDP_T0(tfar);
#endif
  if (off16 < n) {
  const uint32_t q = src + off16, cnt = n - off16 < 16u ? n - off16 : 16u;
  const int32_t lo = (int32_t)W - (int32_t)o.near;        // first position read from the ring (may be negative); older ones come from `far`
  lu8* d = o.ring + ((dst + off16) & o.mask);
  const uint32_t qi = q & o.mask;
  if (cnt == 16u && (int32_t)q >= lo && qi <= o.mask - 15u) l_st16(d, l_ld16(o.ring + qi));
  else if (cnt == 16u && (int32_t)(q + 16u) <= lo) l_st16(d, g_ld16(o.far + q));
  else for (uint32_t b = 0; b < cnt; b++) {
    const uint32_t qq = q + b;
    d[b] = ((int32_t)qq >= lo) ? o.ring[qq & o.mask] : (uint8_t)o.far[qq];
  }
  }
#ifdef BAMD_PROFILE_DECODE
  if (anyfar) { __builtin_amdgcn_s_waitcnt(0); BDP_LAP(12, tfar); }
#endif
}

// LZ match out[pos + k] = out[pos - off + k], k < len, byte-wise forward semantics (lz4.c:2387-2434,
// blosc/fastcopy.c:530-639), destination inside one slice.  All arguments wave-uniform; off >= 1, off <= pos.
__device__ __forceinline__ void ring_match(const PlaneOut& o, uint32_t pos, uint32_t off, uint32_t len, uint32_t W, int lane BDP_ARG) {
  uint32_t done = 0, off_e = off;
  if (off < 64u && off < len) {
    // short period: fetch the pattern once (always in the ring: off < 64), lane i holds pattern byte i mod off, then
    // store G = off * floor(64 / off) bytes per step without further loads
    const uint32_t pat = ((uint32_t)lane < off) ? (uint32_t)o.ring[(pos - off + (uint32_t)lane) & o.mask] : 0u;
    const uint32_t M = 65536u / off + 1u;
    const uint32_t reps = (64u * M) >> 16, G = reps * off;
    const uint32_t i_mod = (uint32_t)lane - (((uint32_t)lane * M) >> 16) * off;
    const uint32_t val = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i_mod << 2), (int)pat);
    lu8* d = o.ring + (pos & o.mask);
    const uint32_t head = len < 256u ? len : 128u;         // long runs switch to 16-byte copies below
    while (done < head) {
      const uint32_t chunk = head - done < G ? head - done : G;
      if ((uint32_t)lane < chunk) d[done + lane] = (uint8_t)val;
      done += chunk;
    }
    if (done >= len) return;
    LDS_ORDER();
    off_e = G;                                             // the prefix written so far is periodic with period off | G
  }
  while (done < len) {
    const uint32_t rem = len - done;
    while (off_e < 1024u && 2u * off_e <= off + done) off_e *= 2u;   // history grew: lengthen the stride
    uint32_t chunk = rem < 1024u ? rem : 1024u;
    if (chunk > off_e) chunk = off_e;
    hist_copy(o, pos + done, pos + done - off_e, chunk, W > pos + done ? W : pos + done, lane BDP_PASS);
    LDS_ORDER();
    done += chunk;
  }
}

// ---------------------------------------------------------------------------------------------
// resumable LZ4 block decoder of one plane (rules of lz4.c:2215-2445 as in lz4_decode_wave, k_decode.hip)
// ---------------------------------------------------------------------------------------------
enum : uint32_t { PH_TOKEN = 0, PH_LIT = 1, PH_FINAL_LIT = 2, PH_MATCHHDR = 3, PH_MATCH = 4, PH_DONE = 5, PH_ERROR = 6, PH_RAW = 7 };

struct PlaneDec {
  Window w;
  const gu8* in;
  uint32_t n, cap;        // compressed size, plane size (neblock)
  uint32_t ip, op;
  uint32_t phase;
  uint32_t pend_lit;      // literal bytes of the current sequence still to copy
  uint32_t pend_ml, pend_off, mdone;   // current match: bytes left, distance, bytes already written
  uint32_t mlnib;
  uint32_t last_match;    // the current match is the stream's last one and its source stays in the ring: no far copy needed
};

// Batched step (see lz4_batch_step in k_decode.hip for the parse): up to 16 sequences whose tokens, literals,
// offsets and at most one length byte lie in the 64 stream bytes at ip, accepted only while their output stays
// inside the slice (`limit`).  Literals go to the ring in one scattered byte store; short matches whose source is in
// the ring, does not wrap and lies before the step's output are copied by 4 lanes each; the rest in stream order.
__device__ __forceinline__ uint32_t lz4_batch_step_ring(const Window& w, const PlaneOut& o, volatile BAMD_LAS uint32_t* scr, uint32_t& ip, uint32_t& op,
                                                        uint32_t cap, uint32_t limit, uint32_t n, int lane BDP_ARG) {
  BDP_T0(tstep);
  const uint32_t B = w.gather_bytes(ip);
  // (a literal length of 15 takes one extension byte, see lz4_batch_step in k_decode.hip)
  const uint32_t ll0 = B >> 4, mlc = B & 15u;
  const uint32_t e_ll = bperm(((uint32_t)lane + 1u) & 63u, B);
  const bool ll_ext = ll0 == 15u;
  const uint32_t ll = ll_ext ? 15u + e_ll : ll0;
  const uint32_t offpos = (uint32_t)lane + 1u + (ll_ext ? 1u : 0u) + ll;
  const uint32_t o_lo = bperm(offpos & 63u, B), o_hi = bperm((offpos + 1u) & 63u, B), e1 = bperm((offpos + 2u) & 63u, B);
  const bool has_ext = mlc == 15u;
  const uint32_t ml = has_ext ? 19u + e1 : mlc + 4u;
  const uint32_t size = 3u + ll + (has_ext ? 1u : 0u) + (ll_ext ? 1u : 0u);
  const bool complete = !(ll_ext && (e_ll == 255u || ip + (uint32_t)lane + 16u >= n)) && !(has_ext && e1 == 255u) && (uint32_t)lane + size <= 64u;
  const uint32_t off = o_lo | (o_hi << 8);
  const uint32_t nxt = complete ? (uint32_t)lane + size : 64u;
  const uint32_t J0 = nxt;
  const uint32_t J1 = hop(J0, J0), J2 = hop(J1, J1), J3 = hop(J2, J2);
  uint32_t c = 0;
  { const uint32_t t = hop(J0, c); c = (lane & 1) ? t : c; }
  { const uint32_t t = hop(J1, c); c = (lane & 2) ? t : c; }
  { const uint32_t t = hop(J2, c); c = (lane & 4) ? t : c; }
  { const uint32_t t = hop(J3, c); c = (lane & 8) ? t : c; }
  const uint32_t pk = bperm(c & 63u, ll | (ml << 9) | ((complete ? 1u : 0u) << 18) | ((ll_ext ? 1u : 0u) << 19) | (nxt << 20));
  const uint32_t off_r = bperm(c & 63u, off);
  const uint32_t ll_r = pk & 0x1ffu, ml_r = (pk >> 9) & 0x1ffu, nxt_r = pk >> 20, ext_r = (pk >> 19) & 1u;
  const bool valid = lane < (int)BATCH_MAXSEQ && c < 64u && ((pk >> 18) & 1u);
  const uint32_t tot_r = valid ? ll_r + ml_r : 0u;
  uint32_t incl = tot_r;
  incl += row_shr<1>(incl); incl += row_shr<2>(incl); incl += row_shr<4>(incl); incl += row_shr<8>(incl);
  const uint32_t excl = incl - tot_r;
  const uint32_t mrel_r = excl + ll_r;
  const bool ok = valid && off_r != 0u && off_r <= op + mrel_r && op + excl + tot_r + 12u <= cap && op + excl + tot_r <= limit;
  const uint32_t okmask = (uint32_t)__ballot(ok) & 0xffffu;
  const uint32_t cnt = (uint32_t)__builtin_ctz(~okmask);
  BDP_LAP(7, tstep);
  if (cnt == 0u) return 0u;
  const uint32_t consumed = (uint32_t)__builtin_amdgcn_readlane((int)nxt_r, (int)(cnt - 1u));
  const uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(cnt - 1u));
  const uint32_t W = op + acc;                                  // ring frontier once this step's literals are out
  // ---- literals of every accepted sequence: one scattered byte store ----
  scr[lane] = 0u;
  if ((uint32_t)lane < cnt) scr[c] = 0x80000000u | excl | (ll_r << 16) | (ext_r << 25);
  const uint64_t mask = __ballot(scr[lane] >> 31);
  {
    const uint64_t below = mask & ((2ull << lane) - 1ull);
    const uint32_t s = 63u - (uint32_t)__builtin_clzll(below | 1ull);
    const uint32_t inf = scr[s];
    const uint32_t xe = (inf >> 25) & 1u;
    const uint32_t k = (uint32_t)lane - s - 1u - xe;
    if ((uint32_t)lane < consumed && (uint32_t)lane > s + xe && k < ((inf >> 16) & 0x1ffu)) o.ring[(op + (inf & 0xffffu) + k) & o.mask] = (uint8_t)B;
  }
  LDS_ORDER();
  // ---- short matches whose source lies before the step's output: 4 lanes each, all of them in one go.  The source is
  //      either in the ring (not overwritten by this step, not wrapping) or entirely in the far copy - the far ones of a
  //      step travel together in ONE global load instruction instead of one round trip each ----
  const uint32_t src_r = op + mrel_r - off_r;                   // source position (valid lanes only)
  const int32_t lo_r = (int32_t)W - (int32_t)o.near;
  const bool near_r = (int32_t)src_r >= lo_r && (src_r & o.mask) + ml_r <= o.mask + 1u;
  const bool far_r = (int32_t)(src_r + ml_r) <= lo_r;
  const bool fast_r = (uint32_t)lane < cnt && ml_r <= 64u && off_r >= mrel_r + ml_r && (near_r || far_r);
  {
    const uint32_t r = (uint32_t)lane >> 2, q = (uint32_t)lane & 3u;
    const uint32_t fA = bperm(r, fast_r ? (ml_r | 0x200u | (mrel_r << 10) | (far_r ? 0x40000000u : 0u)) : 0u);
    const uint32_t fB = bperm(r, off_r);
    const uint32_t mlen = fA & 0x1ffu;
    const bool go = (fA & 0x200u) != 0u, isfar = (fA & 0x40000000u) != 0u;
    const uint32_t dpos = op + ((fA >> 10) & 0xfffffu);
    lu8* d = o.ring + (dpos & o.mask);
    const uint32_t spos = dpos - fB;
    const lu8* sp = o.ring + (spos & o.mask);
    const gu8* sg = o.far + spos;
    const uint32_t np16 = (mlen + 15u) >> 4;
    const bool w16 = go && mlen >= 16u && q < np16;
    const bool w8 = go && mlen >= 8u && mlen < 16u && q < 2u;
    const bool w4 = go && mlen < 8u && q < 2u;
    const uint32_t po16 = (q == np16 - 1u) ? mlen - 16u : 16u * q;
    const uint32_t po8 = q ? mlen - 8u : 0u, po4 = q ? mlen - 4u : 0u;
    uint4 v16 = make_uint4(0, 0, 0, 0); uint64_t v8 = 0; uint32_t v4 = 0;
    if (w16) v16 = isfar ? g_ld16(sg + po16) : l_ld16(sp + po16);
    if (w8) v8 = isfar ? g_ld8(sg + po8) : l_ld8(sp + po8);
    if (w4) v4 = isfar ? g_ld4(sg + po4) : l_ld4(sp + po4);
    if (w16) l_st16(d + po16, v16);
    if (w8) l_st8(d + po8, v8);
    if (w4) l_st4(d + po4, v4);
  }
  LDS_ORDER();
  // ---- everything else in stream order ----
  uint32_t rest = (uint32_t)__ballot((uint32_t)lane < cnt && !fast_r);
#ifdef BAMD_PROFILE_DECODE
  __builtin_amdgcn_s_waitcnt(0);
  bp.c[10] += (uint32_t)(__builtin_amdgcn_s_memtime() - tstep);     // slot 10 (profile build): whole step up to the rest loop
#endif
  BDP_ADD(0, 1); BDP_ADD(1, cnt); BDP_ADD(2, __builtin_popcount(rest));
  BDP_T0(trest);
  while (rest) {
    const int sl = __builtin_ctz(rest);
    rest &= rest - 1u;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)ml_r, sl);
    const uint32_t of = (uint32_t)__builtin_amdgcn_readlane((int)off_r, sl);
    const uint32_t mr = (uint32_t)__builtin_amdgcn_readlane((int)mrel_r, sl);
    ring_match(o, op + mr, of, m, W, lane BDP_PASS);
  }
  BDP_LAP(11, trest);
  ip += consumed;
  op += acc;
  return cnt;
}

__device__ __forceinline__ void lz4_plane_init(PlaneDec& s, const gu8* in, int32_t n, uint32_t cap, int lane) {
  s.in = in; s.n = (uint32_t)(n > 0 ? n : 0); s.cap = cap; s.ip = 0; s.op = 0;
  s.pend_lit = 0; s.pend_ml = 0; s.pend_off = 0; s.mdone = 0; s.mlnib = 0; s.last_match = 0;
  s.phase = n > 0 ? PH_TOKEN : PH_ERROR;
  s.w.init(in, s.n, lane);
}

// decode until op == limit (a slice end, <= cap), the stream ends or an error is found
__device__ __forceinline__ void lz4_plane_run(PlaneDec& s, const PlaneOut& o, volatile BAMD_LAS uint32_t* scr, uint32_t limit, int lane BDP_ARG) {
  const uint32_t n = s.n, cap = s.cap;
  for (;;) {
    if (s.phase >= PH_DONE || s.op >= limit) return;
    if (s.phase == PH_TOKEN) {
      if (s.ip >= n) { s.phase = PH_ERROR; return; }
      BDP_T0(tsk);
      s.w.seek(s.ip);
      const uint32_t hdr = s.w.peek32(s.ip);
#ifdef BAMD_PROFILE_DECODE
      if (s.ip - s.w.base + 72u > 256u) __builtin_amdgcn_s_waitcnt(0);   // attribute the window fetch to slot 6 (the step needs `hi` only then)
      BDP_LAP(6, tsk);
#endif
      if (s.ip + 72u <= n) {
        const uint32_t tk = hdr & 0xffu;
        bool try_batch = true;
        uint32_t ll1 = tk >> 4, tpos = s.ip + 1u;
        if (ll1 == 15u) { const uint32_t e = (hdr >> 8) & 0xffu; try_batch = e != 255u && 17u + e + 3u <= 64u; ll1 = 15u + e; tpos++; }
        if (try_batch && (tk & 15u) == 15u) try_batch = (s.w.peek32(tpos + ll1 + 2u) & 0xffu) != 255u;
        if (try_batch && lz4_batch_step_ring(s.w, o, scr, s.ip, s.op, cap, limit, n, lane BDP_PASS)) continue;
      }
      BDP_ADD(3, 1);
      const uint32_t token = hdr & 0xffu;
      s.ip += 1;
      uint32_t ll = token >> 4;
      if (ll == 15u) {
        if (n < 15u || s.ip >= n - 15u) { s.phase = PH_ERROR; return; }
        lz4_ext_run(s.w, s.ip, ll, cap, lane);
        if (s.ip > n - 15u || ll > cap) { s.phase = PH_ERROR; return; }
      }
      s.pend_lit = ll; s.mlnib = token & 15u;
      if (s.op + ll + 12u > cap || s.ip + ll + 8u > n) {
        if (s.ip + ll != n || s.op + ll > cap) { s.phase = PH_ERROR; return; }
        s.phase = PH_FINAL_LIT;
      } else s.phase = PH_LIT;
      continue;
    }
    if (s.phase == PH_LIT || s.phase == PH_FINAL_LIT) {
      uint32_t c = s.pend_lit < limit - s.op ? s.pend_lit : limit - s.op;
      if (c) {
        if (c <= 64u) {
          s.w.seek(s.ip);
          if (s.ip + c <= s.w.base + 512u) {
            const uint32_t v = s.w.gather_bytes(s.ip);
            if ((uint32_t)lane < c) o.ring[(s.op + (uint32_t)lane) & o.mask] = (uint8_t)v;
          } else ring_put_global(o, s.op, s.in + s.ip, c, lane);
        } else ring_put_global_long(o, s.op, s.in + s.ip, c, lane);
        LDS_ORDER();
        s.ip += c; s.op += c; s.pend_lit -= c;
      }
      if (s.pend_lit) return;                       // slice full
      s.phase = (s.phase == PH_FINAL_LIT) ? PH_DONE : PH_MATCHHDR;
      continue;
    }
    if (s.phase == PH_MATCHHDR) {
      s.w.seek(s.ip);
      const uint32_t t2 = s.w.peek32(s.ip);
      const uint32_t off = t2 & 0xffffu;
      s.ip += 2;
      uint32_t ml = s.mlnib;
      if (ml == 15u) {
        const uint32_t s0 = (t2 >> 16) & 0xffu;
        s.ip++; ml += s0;
        if (s.ip > n - 4u) { s.phase = PH_ERROR; return; }
        if (s0 == 255u) {
          lz4_ext_run(s.w, s.ip, ml, cap, lane);
          if (s.ip > n - 4u || ml > cap) { s.phase = PH_ERROR; return; }
        }
      }
      ml += 4u;
      if (off > s.op || s.op + ml + 5u > cap) { s.phase = PH_ERROR; return; }
      s.pend_ml = ml; s.pend_off = off; s.mdone = 0; s.phase = PH_MATCH;
      // the stream's last match (only the final literal run follows) whose source never leaves the ring: the plane
      // needs no far copy from here on (nothing can ask for older history any more)
      {
        const uint32_t restout = cap - (s.op + ml);
        s.w.seek(s.ip);
        bool last = false;
        if (restout < 15u && s.ip + 1u + restout == n) last = ((s.w.peek32(s.ip) & 0xffu) == (restout << 4));
        s.last_match = (last && off <= o.near - 1024u) ? 1u : 0u;
      }
      continue;
    }
    // PH_MATCH
    {
      const uint32_t c = s.pend_ml < limit - s.op ? s.pend_ml : limit - s.op;
      const uint32_t off = s.pend_off;
      // offset 0: accepted like the reference, bytes unspecified (lz4.c:2356).  A power-of-two period that divides the
      // ring: once R bytes of the match are out the ring already holds every later byte of it - nothing to write.
      const bool idem = (off & (off - 1u)) == 0u && off <= o.mask + 1u && s.mdone >= o.mask + 1u;
      BDP_T0(tm);
      if (off != 0u && !idem) ring_match(o, s.op, off, c, s.op, lane BDP_PASS);
      BDP_LAP(13, tm);
      s.op += c; s.pend_ml -= c; s.mdone += c;
      if (s.pend_ml) return;
      s.phase = PH_TOKEN;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// plane kinds.  Only planes that really need sequential LZ work get a wave and a ring:
//   PK_RING      decoded by a wave into an LDS ring (everything above)
//   PK_RAW       the split was stored raw (csize == neblock, blosc/blosc.c:773-776): the write-out reads the bytes
//                straight from the compressed chunk
//   PK_PERIODIC  the whole stream is "some literals, ONE match with a power-of-two distance <= 256 that runs to the
//                last literals" (constant and short-period byte planes: 4 of the 8 planes of bench19): the plane is a
//                256-byte pattern; the write-out reads the pattern, the bytes in front of the periodic part and the
//                final literals are patched in afterwards
// k_classify_blocks decides per stream (it verifies every byte of such a stream, so the shortcut produces exactly what
// the LZ4 decoder would) and sorts the blocks into two lists by their number of PK_RING planes: <= 4 and more.
// ---------------------------------------------------------------------------------------------
enum : uint32_t { PK_RING = 0, PK_RAW = 1, PK_PERIODIC = 2 };
constexpr uint32_t BD_PAT = 256u;          // pattern bytes per periodic plane: the distance must divide it
// skind word of a PERIODIC plane: kind | log2(distance) << 8 | final literals << 12 | index of the first literal << 16 | literal count << 20
__device__ __forceinline__ uint32_t pk_off(uint32_t k) { return 1u << ((k >> 8) & 15u); }
__device__ __forceinline__ uint32_t pk_r(uint32_t k) { return (k >> 12) & 15u; }
__device__ __forceinline__ uint32_t pk_ls(uint32_t k) { return (k >> 16) & 15u; }
__device__ __forceinline__ uint32_t pk_ll(uint32_t k) { return k >> 20; }

__device__ __forceinline__ uint32_t classify_stream(const gu8* in, int32_t n_, uint32_t neblock) {
  if (n_ == (int32_t)neblock) return PK_RAW;
  if (n_ < 16) return PK_RING;
  const uint32_t n = (uint32_t)n_;
  const uint32_t t0 = in[0];
  if ((t0 & 15u) != 15u) return PK_RING;
  uint32_t ll = t0 >> 4, ls = 1u;                            // literal count (lz4.c:2240-2250), index of the first literal
  if (ll == 15u) {
    for (;;) {
      if (ls >= 5u || ls + 16u > n) return PK_RING;
      const uint32_t e = in[ls++];
      ll += e;
      if (e != 255u) break;
    }
  }
  if (ll == 0u || ll > 1023u || ls + ll + 8u > n) return PK_RING;
  const uint32_t off = (uint32_t)in[ls + ll] | ((uint32_t)in[ls + ll + 1u] << 8);
  if (off == 0u || off > ll || off > BD_PAT || (off & (off - 1u))) return PK_RING;
  const uint32_t e0 = ls + ll + 2u;                          // first match-length extension byte
  for (uint32_t r = 5u; r <= 14u; r++) {                    // final literal run (lz4.c:2423: a match ends >= 5 bytes before the end)
    if (neblock < ll + r + 19u) break;
    const uint32_t ml = neblock - ll - r, k = (ml - 19u) / 255u, x = (ml - 19u) % 255u;
    if (n != e0 + k + 2u + r) continue;
    if (in[e0 + k] != x || in[e0 + k + 1u] != (r << 4)) continue;
    bool all255 = true;
    for (uint32_t i = 0; i < k; i++) if (in[e0 + i] != 255u) { all255 = false; break; }
    if (!all255) continue;
    return PK_PERIODIC | ((uint32_t)__builtin_ctz(off) << 8) | (r << 12) | (ls << 16) | (ll << 20);
  }
  return PK_RING;
}

// one thread per candidate block (blist: the blocks engine.hip marked BLK_LDS); out: skind[stream], lists[2][nlist] and
// their counters cnt[0..1] (cnt[v] blocks in lists + v * nlist)
__global__ void k_classify_blocks(const StreamDesc* __restrict__ streams, const BlockDesc* __restrict__ blocks, const int32_t* __restrict__ blist,
                                  uint32_t nlist, uint32_t* __restrict__ skind, int32_t* __restrict__ lists, uint32_t* __restrict__ cnt) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nlist) return;
  const int32_t gb = blist[t];
  const BlockDesc b = blocks[gb];
  const uint32_t neblock = (uint32_t)b.bsize / (uint32_t)b.nstreams;
  uint32_t nring = 0;
  for (int j = 0; j < b.nstreams; j++) {
    const StreamDesc& sd = streams[b.first_stream + j];
    uint32_t k = PK_RING;
    if (sd.in_size >= 0) k = classify_stream(as_global(sd.in), sd.in_size, neblock);
    skind[b.first_stream + j] = k;
    nring += (k & 255u) == PK_RING;
  }
  const uint32_t v = nring > 4u ? 1u : 0u;
  lists[(size_t)v * nlist + atomicAdd(&cnt[v], 1u)] = gb;
}

// ---------------------------------------------------------------------------------------------
// write-out of one slice: T planes -> element-major destination (blosc/shuffle-generic.h:61-81)
// ---------------------------------------------------------------------------------------------
template <int T>
struct PlaneTab {            // wave-uniform description of the block's planes
  uint32_t lbase[T];         // RING / PERIODIC: LDS byte offset of the ring / the 64-byte pattern
  uint32_t lmask[T];         // ring size - 1 / BD_PAT - 1
  const gu8* raw[T];         // RAW: the bytes in the compressed chunk (nullptr otherwise)
};

// `wo_rank` / `wo_n`: this wave is the wo_rank-th of the wo_n waves that share the write-out (the waves that own a big
// ring - the busy planes - are left out when there are others: they are the block's critical path)
template <int T, int W>
__device__ __forceinline__ void slice_writeout(const PlaneTab<T>& pt, const lu8* lds, gu8* dst, uint32_t p0, uint32_t p1, int wave, int lane, int wo_rank, int wo_n) {
  // positions [p0, p1) of every plane; steps of 256 positions, lane l owns 4 consecutive positions
  const uint32_t nfull = (p1 - p0) >> 8;
  if (wo_rank >= 0)
  for (uint32_t st = (uint32_t)wo_rank; st < nfull; st += (uint32_t)wo_n) {
    const uint32_t p = p0 + (st << 8) + 4u * (uint32_t)lane;
    Rows<T> x;
#pragma unroll
    for (int j = 0; j < T; j++) {
      if (pt.raw[j]) x.r[j] = g_ld4(pt.raw[j] + p);
      else x.r[j] = *(const BAMD_LAS uint32_t*)(lds + pt.lbase[j] + (p & pt.lmask[j]));
    }
    unshuffle_store<T>(dst, p - 4u * (uint32_t)lane, lane, x);
  }
  // fewer than 256 positions left (only when the plane size is not a multiple of 256): byte by byte, first write-out wave
  const uint32_t tail0 = p0 + (nfull << 8);
  if (wo_rank == 0)
    for (uint32_t k = tail0 * T + (uint32_t)lane; k < p1 * T; k += 64u) {
      const uint32_t el = k / T, j = k - el * T;
      uint32_t v = 0;
#pragma unroll
      for (int jj = 0; jj < T; jj++)
        if ((uint32_t)jj == j) v = pt.raw[jj] ? (uint32_t)pt.raw[jj][el] : (uint32_t)lds[pt.lbase[jj] + (el & pt.lmask[jj])];
      dst[k] = (uint8_t)v;
    }
}

// LDS of a workgroup: W rings | W x 256 bytes of step scratch | T x 256 bytes of patterns | control words
template <int W> constexpr uint32_t bd_pool_bytes() { return BD_NBIG * BD_RBIG + ((uint32_t)W - BD_NBIG) * BD_RSMALL; }
template <int T, int W> constexpr uint32_t bd_lds_bytes() { return bd_pool_bytes<W>() + (uint32_t)W * 256u + (uint32_t)T * BD_PAT + 128u; }

// One block by W waves.  Wave w decodes the w-th PK_RING plane (waves beyond the number of such planes only help
// with the write-out).  Slice k occupies ring half k & 1 (R = 2 S), so ONE barrier per slice is enough: a wave that
// has written out its share of slice k goes on decoding slice k + 1 into the other half while slower waves still
// read half k & 1; nobody writes that half again before the barrier of slice k + 1, which every wave reaches only
// after its share of write-out k.  Between blocks the ticket broadcast supplies the barrier.
template <int T, int W>
__device__ __forceinline__ void decode_block(StreamDesc* sds, const uint32_t* skind, const ChunkDesc* c, const BlockDesc* b, int32_t* status,
                                             lu8* lds, volatile BAMD_LAS uint32_t* scr, volatile BAMD_LAS uint32_t* ctl, gu8* far_wg, int wave, int lane BDP_ARG) {
  const uint32_t bsize = uni((uint32_t)b->bsize), neblock = bsize / (uint32_t)T;
  gu8* dst = uni_ptr(as_global(c->dst)) + (size_t)uni((uint32_t)b->blk) * (size_t)uni((uint32_t)c->blocksize);
  // ---- plane table (every wave builds the same one) ----
  // The BD_NBIG ring planes with the most compressed bytes get the big rings: the compressed size is what says, before
  // any decoding, which planes carry the sequences (bench19: 11.6 KB for the two busy planes, 1.4 KB and 0.8 KB for the others).
  PlaneTab<T> pt;
  uint32_t kinds[T];
  int my_plane = -1;
  uint32_t my_ring = 0, my_rbytes = BD_RSMALL;
  {
    uint32_t cs[T];
    uint32_t big1 = 0, big2 = 0;            // the two largest compressed sizes among the ring planes (with their plane index as tie-break)
#pragma unroll
    for (int j = 0; j < T; j++) {
      kinds[j] = uni(skind[j]);
      cs[j] = ((kinds[j] & 255u) == PK_RING) ? ((uni((uint32_t)sds[j].in_size) << 4) | (uint32_t)(15 - j)) : 0u;
      if (cs[j] > big1) { big2 = big1; big1 = cs[j]; } else if (cs[j] > big2) big2 = cs[j];
    }
    uint32_t nr = 0, base = 0;
#pragma unroll
    for (int j = 0; j < T; j++) {
      const uint32_t kd = kinds[j] & 255u;
      pt.raw[j] = nullptr; pt.lbase[j] = 0; pt.lmask[j] = BD_RSMALL - 1u;
      if (kd == PK_RING) {
        const uint32_t rb = (cs[j] != 0u && cs[j] >= big2) ? BD_RBIG : BD_RSMALL;     // at most BD_NBIG = 2 planes qualify
        if ((int)nr == wave) { my_plane = j; my_ring = base; my_rbytes = rb; }
        pt.lbase[j] = base; pt.lmask[j] = rb - 1u;
        base += rb; nr++;
      } else if (kd == PK_RAW) pt.raw[j] = uni_ptr(as_global(sds[j].in));
      else { pt.lbase[j] = bd_pool_bytes<W>() + (uint32_t)W * 256u + (uint32_t)j * BD_PAT; pt.lmask[j] = BD_PAT - 1u; }
    }
  }
  // ---- patterns of the periodic planes: wave j % W writes plane j's 256 bytes ----
#pragma unroll
  for (int j = 0; j < T; j++) {
    if ((kinds[j] & 255u) != PK_PERIODIC || (j % W) != wave) continue;
    const uint32_t ll = pk_ll(kinds[j]), off = pk_off(kinds[j]), ls = pk_ls(kinds[j]);
    const gu8* lit = uni_ptr(as_global(sds[j].in)) + ls;     // L = the literals
    // plane[p] = L[(ll - off) + ((p - (ll - off)) mod off)] for p >= ll - off; pattern index = p mod 256 (off divides 256)
#pragma unroll
    for (uint32_t q = 0; q < BD_PAT; q += 64u) {
      const uint32_t i = q + (uint32_t)lane;
      lds[pt.lbase[j] + i] = lit[(ll - off) + ((i + 1024u - (ll - off)) & (off - 1u))];
    }
  }
  // ---- this wave's plane ----
  PlaneOut o;
  PlaneDec s;
  s.phase = PH_DONE; s.op = neblock; s.last_match = 1;
  if (my_plane >= 0) {
    const StreamDesc* sd = sds + my_plane;
    o.ring = lds + my_ring; o.mask = my_rbytes - 1u; o.near = my_rbytes - BD_NEAR_GAP;
    o.far = far_wg + (size_t)wave * neblock;
    o.flushed = 0;
    lz4_plane_init(s, uni_ptr(as_global(sd->in)), (int32_t)uni((uint32_t)sd->in_size), neblock, lane);
  }
  // who writes out: waves without a big ring (idle waves and the planes with little work), all waves if there are none
  int wo_rank, wo_n;
  {
    uint32_t nbig = 0;
#pragma unroll
    for (int j = 0; j < T; j++) nbig += ((kinds[j] & 255u) == PK_RING && pt.lmask[j] == BD_RBIG - 1u) ? 1u : 0u;
    // ring planes are numbered in plane order = wave order; big-ring waves are those whose plane has a big ring
    uint32_t bigwaves = 0, nr2 = 0;
#pragma unroll
    for (int j = 0; j < T; j++) if ((kinds[j] & 255u) == PK_RING) { if (pt.lmask[j] == BD_RBIG - 1u) bigwaves |= 1u << nr2; nr2++; }
    const uint32_t light = ~bigwaves & ((1u << W) - 1u);
    if (light == 0u) { wo_rank = wave; wo_n = W; }
    else { wo_n = __builtin_popcount(light); wo_rank = (light >> wave) & 1u ? __builtin_popcount(light & ((1u << wave) - 1u)) : -1; }
    (void)nbig;
  }
  // ---- slices.  No lock-step: every wave publishes how far it is in LDS (in-order DS pipeline: a flag written after
  //      the data is seen after the data), a plane may run ahead of the write-out by as many slices as its ring holds
  //      (4 for the big rings), and the write-out of slice k starts when every plane has published k + 1.  With a
  //      barrier per slice the two busy planes of a bench19 block waited for each other a third of the time (a slice is
  //      two to four batched steps; profiles/r02_b_block_decoder.md).
  volatile BAMD_LAS uint32_t* prog = ctl + 4;           // [T] slices decoded per plane (0xffffffff: nothing to decode)
  volatile BAMD_LAS uint32_t* wprog = ctl + 4 + T;      // [W] slices written out per wave (0xffffffff: takes no part)
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < T; j++) if ((j % W) == wave) prog[j] = ((kinds[j] & 255u) == PK_RING) ? 0u : 0xffffffffu;
    wprog[wave] = wo_rank >= 0 ? 0u : 0xffffffffu;
  }
  __syncthreads();
  const uint32_t nsl = (neblock + BD_S - 1u) / BD_S;
  const uint32_t cap_slices = my_rbytes / BD_S;
  for (uint32_t k = 0; k < nsl; k++) {
    const uint32_t p0 = k * BD_S, p1 = p0 + BD_S < neblock ? p0 + BD_S : neblock;
    if (my_plane >= 0) {
      BDP_T0(tb);
      if (k >= cap_slices) {                             // the ring slot of slice k still holds slice k - cap_slices: written out?
        const uint32_t need = k - cap_slices + 1u;
        for (;;) {
          uint32_t m = 0xffffffffu;
#pragma unroll
          for (int w2 = 0; w2 < W; w2++) { const uint32_t v = uni(wprog[w2]); m = v < m ? v : m; }
          if (m >= need) break;
          __builtin_amdgcn_s_sleep(4);
        }
      }
      BDP_LAP(9, tb);
      BDP_T0(td);
      o.flushed = p0;
      lz4_plane_run(s, o, scr, p1, lane BDP_PASS);
      // this wave's own plane, plane-major, for its own far matches of later slices (same wave stores and loads)
      if (!s.last_match && p1 < neblock) {
        const lu8* r = o.ring + (p0 & o.mask);
        for (uint32_t q = 16u * (uint32_t)lane; q < p1 - p0; q += 1024u) g_st16(o.far + p0 + q, l_ld16(r + q));
      }
      LDS_ORDER();
      if (lane == 0) prog[my_plane] = k + 1u;
      BDP_LAP(8, td);
    }
    if (wo_rank >= 0) {
      BDP_T0(tb2);
      for (;;) {
        uint32_t m = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < T; j++) { const uint32_t v = uni(prog[j]); m = v < m ? v : m; }
        if (m >= k + 1u) break;
        __builtin_amdgcn_s_sleep(4);
      }
      BDP_LAP(9, tb2);
      BDP_T0(tw);
      slice_writeout<T, W>(pt, lds, dst, p0, p1, wave, lane, wo_rank, wo_n);
      LDS_ORDER();
      if (lane == 0) wprog[wave] = k + 1u;
      BDP_LAP(4, tw);
    }
  }
  __syncthreads();                         // the block is complete (and every store of it has been performed)
  if (my_plane >= 0) {
    const bool good = s.phase == PH_DONE && s.op == neblock;
    if (lane == 0) {
      sds[my_plane].result = good ? (int32_t)neblock : -1;
      if (!good) atomicMin(&status[uni((uint32_t)sds[my_plane].chunk)], (int32_t)ST_BADCODEC);      // blosc.c:780-782
    }
  }
  // ---- periodic planes: the bytes in front of the periodic part and the final literals (<= 14 + 14 bytes per plane) ----
  bool any_per = false;
#pragma unroll
  for (int j = 0; j < T; j++) any_per |= (kinds[j] & 255u) == PK_PERIODIC;
  if (any_per) {
#pragma unroll
    for (int j = 0; j < T; j++) {
      if ((kinds[j] & 255u) != PK_PERIODIC || (j % W) != wave) continue;
      const uint32_t ll = pk_ll(kinds[j]), off = pk_off(kinds[j]), r = pk_r(kinds[j]), ls = pk_ls(kinds[j]);
      const gu8* in = uni_ptr(as_global(sds[j].in));
      const uint32_t n = uni((uint32_t)sds[j].in_size);
      for (uint32_t i = (uint32_t)lane; i < ll - off; i += 64u) dst[(size_t)i * T + (uint32_t)j] = in[ls + i];
      if ((uint32_t)lane < r) dst[(size_t)(neblock - r + (uint32_t)lane) * T + (uint32_t)j] = in[n - r + (uint32_t)lane];
      if (lane == 0) sds[j].result = (int32_t)neblock;
    }
  }
#pragma unroll
  for (int j = 0; j < T; j++) if ((kinds[j] & 255u) == PK_RAW && (j % W) == wave && lane == 0) sds[j].result = (int32_t)neblock;
}

constexpr int BD_WG4_PER_CU = 5;           // 4-wave workgroups per CU: 27.7 KiB of LDS each
constexpr int BD_WG8_PER_CU = 2;           // 8-wave workgroups per CU: 45 KiB of LDS each, 128 VGPRs

// Persistent workgroups of W waves; `blist` / `nlist_p` = one of the two lists k_classify_blocks filled, `far` one area
// of `far_stride` bytes per workgroup.
template <int T, int W>
__global__ __launch_bounds__(64 * W, W == 4 ? 5 : 4) void k_decode_blocks(StreamDesc* __restrict__ streams, const uint32_t* __restrict__ skind, int32_t* __restrict__ status,
                                                             uint32_t* __restrict__ ticket, const int32_t* __restrict__ blist, const uint32_t* __restrict__ nlist_p,
                                                             const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks,
                                                             uint8_t* __restrict__ far, size_t far_stride, uint32_t* __restrict__ done
#ifdef BAMD_PROFILE_DECODE
                                                             , uint32_t* __restrict__ profbuf
#endif
                                                             ) {
  __shared__ __attribute__((aligned(16))) uint8_t bd_lds[bd_lds_bytes<T, W>()];
  lu8* lds = (lu8*)bd_lds;
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  volatile BAMD_LAS uint32_t* scr = (volatile BAMD_LAS uint32_t*)(lds + bd_pool_bytes<W>() + (uint32_t)wave * 256u);
  volatile BAMD_LAS uint32_t* ctl = (volatile BAMD_LAS uint32_t*)(lds + bd_pool_bytes<W>() + (uint32_t)W * 256u + (uint32_t)T * BD_PAT);
  gu8* far_wg = as_global(far) + (size_t)blockIdx.x * far_stride;
  const uint32_t nlist = uni(*nlist_p);
  uint32_t ndone = 0;
#ifdef BAMD_PROFILE_DECODE
  BdProf bp; for (int i_ = 0; i_ < 16; i_++) bp.c[i_] = 0;
  const uint64_t tk0 = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    if (threadIdx.x == 0) ctl[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t t = uni(ctl[0]);
    __syncthreads();                       // everybody has the ticket (and is done with the previous block's LDS)
    if (t >= nlist) break;
    const uint32_t gb = (uint32_t)blist[t];
    const BlockDesc* b = blocks + gb;
    const ChunkDesc* c = chunks + uni((uint32_t)b->chunk);
    const uint32_t fs = uni((uint32_t)b->first_stream);
    decode_block<T, W>(streams + fs, skind + fs, c, b, status, lds, scr, ctl, far_wg, wave, lane BDP_PASS);
    ndone++;
  }
  if (threadIdx.x == 0 && ndone) atomicAdd(done, ndone);
#ifdef BAMD_PROFILE_DECODE
  bp.c[14] = (uint32_t)(__builtin_amdgcn_s_memtime() - tk0); bp.c[15] = ndone;
  if (profbuf && lane == 0) for (int i_ = 0; i_ < 16; i_++) profbuf[((size_t)blockIdx.x * W + wave) * 16 + i_] = bp.c[i_];
#endif
}

}  // namespace bamd
