// dec_ring.h — the LZ4 block decoder of one wavefront with its recent output in LDS (included by k_decode.hip).
//
// Replaces  LZ4_decompress_safe  (internal-complibs/lz4-1.10.0/lz4.c:2451 -> LZ4_decompress_generic :2023-2445) for one stream
// (= one split of one block, blosc/blosc.c:760-787).
//
// Why (round 4; profiles/r03/r03m_dec_phase_with_unshuffle_cycles.txt): the round-3 decoder wrote every literal and every match piece
// straight to global memory and read match sources back from there.  A batched step of 13 sequences then waits for one L2 round
// trip under load (4 400 cycles per step on reference-written bench19 planes), a sequence of the scalar path for about five of
// them (17 000 cycles: window fetch, length bytes, literals, match load, stores), and 112 such sequences per block were a third
// of the kernel's wave time.  LZ4 on shuffled numeric data reaches back a few KiB (bench19's noisy planes: 93 % of the distances
// are below 8 KiB), so the wave's recent output belongs in LDS:
//   * history ring: the last DR_RING bytes of the plane live in LDS.  Literals are scattered into it, near matches are LDS -> LDS
//     copies (~100-cycle round trips instead of ~3 000), long matches and long literal runs move through it 1 KiB at a time;
//   * rows: every completed 1 KiB row of the ring leaves for global memory as ONE coalesced 16-byte-per-lane store, fire and
//     forget - nothing of the decoder ever waits for a store, and the scratch sees full lines instead of byte scatters
//     (round 3 measured 4.8 GB written for 2.6 GB of plane bytes);
//   * far matches (source older than the ring) read the rows already written - same wave, so program order makes them visible;
//   * input ring: the compressed bytes sit in a 1 KiB LDS ring refilled one 256-byte block ahead of the parse; a step's 64 bytes
//     and the 3 bytes behind each candidate token are two LDS reads per lane (the register window needed five ds_bpermute).
// Acceptance rules are those of the reference's safe loop (lz4.c:2215-2435), see lz4_decode_wave below.
#pragma once

namespace bamd {

#ifndef BAMD_DEC_RING
#define BAMD_DEC_RING 8192          // history ring bytes per wave (power of two, >= 4096)
#endif
constexpr uint32_t DR_RING = BAMD_DEC_RING, DR_MASK = DR_RING - 1u;
constexpr uint32_t DR_ROW = 1024u;                 // a row of the ring goes to global memory when it is complete
constexpr uint32_t DR_STEP_MAX = 2048u;            // output bytes of one batched step at most
#ifndef BAMD_DEC_INBLOCKS
#define BAMD_DEC_INBLOCKS 4
#endif
constexpr uint32_t DR_INB = BAMD_DEC_INBLOCKS;     // input ring: this many blocks of 256 bytes (a power of two >= 2)
constexpr uint32_t DR_IN = 256u * DR_INB;
constexpr uint32_t DR_LDS_BYTES = 256u + DR_IN + DR_RING;   // 64 scratch dwords | input ring | history ring
static_assert((DR_RING & DR_MASK) == 0u && DR_RING >= 4096u, "ring size");
// a step's sources are either in the ring (>= W - DR_RING, W = the step's end) or in rows already written (< W - DR_RING):
// W - DR_RING + longest match must not exceed what has certainly been flushed (op - DR_ROW)
// Long power-of-two matches take everything behind the doubled period out of one register set (dr_match; round 5).  Simpler than the byte-granular
// pieces around the row boundaries it replaces, and NEUTRAL in time: the "- 7 %" of profiles/r05j_* was one slow library instance of two - with
// three copies of each build taking turns old and new forms are equal (profiles/r05r_*).  Their last piece may store up to 15 bytes beyond the
// match: the guard band.
constexpr uint32_t DR_GUARD = 16u;
static_assert(DR_STEP_MAX + 273u + DR_ROW + DR_GUARD <= DR_RING, "far sources must lie in flushed rows");

#ifdef BAMD_WAVE_EMU
inline unsigned long long g_emu_ring_steps = 0;       // emulator only: batched steps executed (tests assert that they run at all)
inline unsigned long long g_emu_ring_far = 0;         // ... and matches served from flushed rows
#endif

#define DR_SYNC() do { LDS_ORDER(); BAMD_LDS_SYNC(); } while (0)
// lane i <- lane i + N of its 16-lane row (0 beyond the row): DPP row_shl
__device__ __forceinline__ uint32_t dpp_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_shl2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_shl3(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x103, 0xf, 0xf, true); }

struct RingIO {
  volatile BAMD_LAS uint32_t* scr;     // 64 dwords: token info of a step on its way back to byte-lane space
  BAMD_LAS uint32_t* in32;             // input ring (DR_IN / 4 dwords)
  lu8* hist;                           // history ring
  const gu8* in; uint32_t n;
  gu8* out; uint32_t cap;
  uint32_t b_hi;                       // input blocks below b_hi (and not older than b_hi - 4) are in the ring
  uint32_t pend, pv;                   // block b_hi has been requested: its dword is on its way in pv (per lane)
  uint32_t flushed;                    // out[0, flushed) is in global memory (or skipped by a periodic span); a multiple of DR_ROW
  uint32_t rfloor;                     // ring positions below this are not valid (behind a periodic span: never written)
  int lane;
#ifdef BAMD_LOO_PLANES
  uint32_t noflush;
#endif
};
#ifdef BAMD_LOO_PLANES
#define DR_FLUSH_ON(io) (!(io).noflush)
#else
#define DR_FLUSH_ON(io) true
#endif

// ---- input ring ----------------------------------------------------------------------------------------------------------
// lane's dword of input block j; bytes beyond the stream read as zero (and are never consumed: every consumer checks n first)
__device__ __forceinline__ uint32_t dr_in_load(const gu8* in, uint32_t n, uint32_t j, int lane) {
  const uint32_t p = 256u * j + 4u * (uint32_t)lane;
  uint32_t v = 0;
  if (p + 4u <= n) v = g_ld4(in + p);
  else if (p < n) {
    if (n >= 4u) v = g_ld4(in + n - 4u) >> (8u * (p + 4u - n));
    else for (uint32_t b = 0; p + b < n; b++) v |= (uint32_t)in[p + b] << (8u * b);
  }
  return v;
}
__device__ __forceinline__ void dr_in_store(RingIO& io, uint32_t j, uint32_t v) { io.in32[64u * (j & (DR_INB - 1u)) + (uint32_t)io.lane] = v; }
// makes stream bytes [ip, ip + 72) readable; ip only ever grows
__device__ __forceinline__ void dr_input(RingIO& io, uint32_t ip) {
  const uint32_t bi = ip >> 8;
  if (bi > io.b_hi) { io.pend = 0u; io.b_hi = bi; }            // jumped over everything present (a long literal run): start again at ip's block
  // the block requested earlier goes in once nothing still needed lives in its slot (its old tenant is block b_hi - 4)
  // (handing it over one, two or four calls later - the first use of pv is where the wave waits for the load - changes nothing: profiles/r05b_*)
  if (io.pend && io.b_hi < bi + DR_INB) { dr_in_store(io, io.b_hi, io.pv); io.b_hi++; io.pend = 0u; }
  if (((ip + 71u) >> 8) >= io.b_hi) {                          // stream start, behind a jump, or the prefetch fell behind: up to three blocks in one round trip
    const uint32_t room = DR_INB - (io.b_hi - bi);             // blocks that may come in without evicting ip's own (b_hi - bi is 0 or 1 here)
    const uint32_t v0 = dr_in_load(io.in, io.n, io.b_hi, io.lane);
    uint32_t v1 = 0u, v2 = 0u;
    if (room >= 2u) v1 = dr_in_load(io.in, io.n, io.b_hi + 1u, io.lane);
    if (room >= 3u) v2 = dr_in_load(io.in, io.n, io.b_hi + 2u, io.lane);
    dr_in_store(io, io.b_hi, v0);
    if (room >= 2u) dr_in_store(io, io.b_hi + 1u, v1);
    if (room >= 3u) dr_in_store(io, io.b_hi + 2u, v2);
    io.b_hi += room >= 3u ? 3u : room; io.pend = 0u;
  }
  if (!io.pend && io.b_hi < bi + DR_INB && 256u * io.b_hi < io.n) { io.pv = dr_in_load(io.in, io.n, io.b_hi, io.lane); io.pend = 1u; }
  DR_SYNC();
}
// the four stream bytes at p (little endian), p per lane or uniform; needs dr_input(q) with q <= p, p + 4 <= q + 72
__device__ __forceinline__ uint32_t dr_in4(const RingIO& io, uint32_t p) {
  const uint32_t i = (p >> 2) & (DR_IN / 4u - 1u);
  return __builtin_amdgcn_alignbyte(io.in32[(i + 1u) & (DR_IN / 4u - 1u)], io.in32[i], p & 3u);
}
__device__ __forceinline__ uint32_t dr_peek32(const RingIO& io, uint32_t p) { return uni(dr_in4(io, p)); }

// ---- history ring --------------------------------------------------------------------------------------------------------
// 16 bytes at plane position pos; a piece across the ring's end goes byte by byte (once per DR_RING bytes of output)
__device__ __forceinline__ uint4 dr_get16(const lu8* hist, uint32_t pos) {
  const uint32_t o = pos & DR_MASK;
  if (o <= DR_RING - 16u) return l_ld16(hist + o);
  uint32_t w0 = 0u, w1 = 0u, w2 = 0u, w3 = 0u;               // (named words and constant shifts: an indexed array would live in scratch memory)
#pragma unroll
  for (uint32_t b = 0; b < 4u; b++) {
    w0 |= (uint32_t)hist[(pos + b) & DR_MASK] << (8u * b); w1 |= (uint32_t)hist[(pos + 4u + b) & DR_MASK] << (8u * b);
    w2 |= (uint32_t)hist[(pos + 8u + b) & DR_MASK] << (8u * b); w3 |= (uint32_t)hist[(pos + 12u + b) & DR_MASK] << (8u * b);
  }
  return make_uint4(w0, w1, w2, w3);
}
__device__ __forceinline__ void dr_put16(lu8* hist, uint32_t pos, uint4 v) {
  const uint32_t o = pos & DR_MASK;
  if (o <= DR_RING - 16u) { l_st16(hist + o, v); return; }
#pragma unroll
  for (uint32_t b = 0; b < 4u; b++) {
    hist[(pos + b) & DR_MASK] = (uint8_t)(v.x >> (8u * b)); hist[(pos + 4u + b) & DR_MASK] = (uint8_t)(v.y >> (8u * b));
    hist[(pos + 8u + b) & DR_MASK] = (uint8_t)(v.z >> (8u * b)); hist[(pos + 12u + b) & DR_MASK] = (uint8_t)(v.w >> (8u * b));
  }
}
// every complete row below op leaves for global memory: 16 bytes per lane, 1 KiB per instruction, nobody waits for it
__device__ __forceinline__ void dr_flush_rows(RingIO& io, uint32_t op) {
  while (op - io.flushed >= DR_ROW) {
    DR_SYNC();
    if (DR_FLUSH_ON(io)) g_st16(io.out + io.flushed + 16u * (uint32_t)io.lane, l_ld16(io.hist + ((io.flushed + 16u * (uint32_t)io.lane) & DR_MASK)));
    io.flushed += DR_ROW;
  }
}
// the stream's end: whatever is left of the last row
__device__ __forceinline__ void dr_flush_tail(RingIO& io, uint32_t op) {
  dr_flush_rows(io, op);
  DR_SYNC();
  const uint32_t r = op - io.flushed, l16 = 16u * (uint32_t)io.lane;
  if (l16 + 16u <= r) g_st16(io.out + io.flushed + l16, l_ld16(io.hist + ((io.flushed + l16) & DR_MASK)));
  const uint32_t t0 = r & ~15u;
  if (t0 + (uint32_t)io.lane < r) io.out[io.flushed + t0 + (uint32_t)io.lane] = io.hist[(io.flushed + t0 + (uint32_t)io.lane) & DR_MASK];
  io.flushed = op;
}
// lowest plane position a copy ending at W may still read from the ring
__device__ __forceinline__ uint32_t dr_near_lo(const RingIO& io, uint32_t W) {
  // (DR_GUARD: the row-register form of long power-of-two matches, dr_match, may store up to 15 bytes beyond the match's end - ring slots of the
  //  oldest 15 positions of the window, which therefore do not count as "in the ring")
  const uint32_t lo = W > DR_RING - DR_GUARD ? W - (DR_RING - DR_GUARD) : 0u;
  return lo > io.rfloor ? lo : io.rfloor;
}

// c <= 1024 bytes, plane positions [src, src + c) -> [pos, pos + c), no overlap (src + c <= pos): out of the ring, or out of the rows
// already written when the source is older than the ring (the caller cuts c so that one of the two holds for the whole piece)
__device__ __forceinline__ void dr_copy_chunk(RingIO& io, uint32_t pos, uint32_t src, uint32_t c, bool far, int lane) {
  const uint32_t n16 = c >> 4, l16 = 16u * (uint32_t)lane, t0 = c & ~15u;
  DR_SYNC();
  if (!far) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u); uint32_t tb = 0u;
    if ((uint32_t)lane < n16) v = dr_get16(io.hist, src + l16);
    if (t0 + (uint32_t)lane < c) tb = io.hist[(src + t0 + (uint32_t)lane) & DR_MASK];
    DR_SYNC();
    if ((uint32_t)lane < n16) dr_put16(io.hist, pos + l16, v);
    if (t0 + (uint32_t)lane < c) io.hist[(pos + t0 + (uint32_t)lane) & DR_MASK] = (uint8_t)tb;
  } else {
    BAMD_MEM_SYNC();
    uint4 v = make_uint4(0u, 0u, 0u, 0u); uint32_t tb = 0u;
    if ((uint32_t)lane < n16) v = g_ld16(io.out + src + l16);
    if (t0 + (uint32_t)lane < c) tb = io.out[src + t0 + (uint32_t)lane];
    if ((uint32_t)lane < n16) dr_put16(io.hist, pos + l16, v);
    if (t0 + (uint32_t)lane < c) io.hist[(pos + t0 + (uint32_t)lane) & DR_MASK] = (uint8_t)tb;
#ifdef BAMD_WAVE_EMU
    if (lane == 0) g_emu_ring_far++;
#endif
  }
  DR_SYNC();
}

// LZ match of any length and distance at op (byte-wise forward semantics, lz4.c:2387-2434 / blosc/fastcopy.c:530-639): through the
// ring in pieces of at most one row, every completed row flushed on the way.  off >= 1, off <= op (checked by the caller).
__device__ __forceinline__ void dr_match(RingIO& io, uint32_t& op, uint32_t off, uint32_t len, int lane) {
  uint32_t done = 0, off_e = off;
  const uint32_t mpos = op;
  if (off < 64u && off < len) {
    // short period: the off bytes in front of the match, replicated from registers, give the first G = off * floor(64 / off) bytes
    DR_SYNC();
    // (off < 64: in the ring - unless the ring is empty down there: right behind a periodic span that ended on a row boundary (rfloor == mpos or
    //  a few bytes below it) those bytes exist only in global memory, where the span's materialisation or the head copy put them.  Round 5: the
    //  reference's linspace chunks at typesize 4 hold exactly that - "2 literals, 32 766 bytes at distance 2, 1 literal, 32 767 bytes at distance 2" -
    //  and the second match was replicated from stale ring bytes: tests/test_gpu_spans.py::test_short_period_match_right_behind_a_span.)
    const uint32_t pnlo = dr_near_lo(io, mpos);
    uint32_t pat = 0u;
    if (mpos - off < pnlo) {                                    // wave-uniform, rare
      BAMD_MEM_SYNC();
      const uint32_t pp = mpos - off + (uint32_t)lane;
      if ((uint32_t)lane < off) pat = pp < pnlo ? (uint32_t)io.out[pp] : (uint32_t)io.hist[pp & DR_MASK];
    } else if ((uint32_t)lane < off) pat = (uint32_t)io.hist[(mpos - off + (uint32_t)lane) & DR_MASK];
    const uint32_t M = 65536u / off + 1u;                 // floor(i / off) == (i * M) >> 16 for i < 64
    const uint32_t G = ((64u * M) >> 16) * off;
    const uint32_t i_mod = (uint32_t)lane - (((uint32_t)lane * M) >> 16) * off;
    const uint32_t val = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i_mod << 2), (int)pat);
    if ((16u % off) == 0u && len >= 64u && len < 2u * DR_ROW) {      // (longer ones: the row-register form below, which also stores the rows itself)
      // Periods 1, 2, 4, 8, 16 (runs and narrow counters: a smooth byte plane is made of them - the reference's linspace chunks hold 256 runs of 511 bytes in one
      // plane, 256 of 255 in another): the 16 bytes every lane would write are the SAME - the first sixteen lanes of `val` - so the whole match goes into the
      // ring a row's worth per store, straight from registers.  Before (round 6): 64 bytes from registers, then three to five copies of doubling stride through
      // the ring, each with its three LDS barriers - 7.6 k cycles per run (profiles/r06m_dec_phase_linspace.txt).
      const uint32_t b4 = val | (dpp_shl1(val) << 8) | (dpp_shl2(val) << 16) | (dpp_shl3(val) << 24);      // lane i: pattern bytes i .. i + 3 (i + 3 < 16 is all that is read)
      const uint4 row = make_uint4((uint32_t)__builtin_amdgcn_readlane((int)b4, 0), (uint32_t)__builtin_amdgcn_readlane((int)b4, 4),
                                   (uint32_t)__builtin_amdgcn_readlane((int)b4, 8), (uint32_t)__builtin_amdgcn_readlane((int)b4, 12));
      DR_SYNC();
      for (uint32_t d = 0; d < len; d += DR_ROW) {             // (every piece starts a multiple of 1 KiB behind the match's start: the pattern's phase is 0 again)
        const uint32_t c = len - d < DR_ROW ? len - d : DR_ROW, pos = mpos + d, n16 = c >> 4, t0 = c & ~15u;
        if ((uint32_t)lane < n16) dr_put16(io.hist, pos + 16u * (uint32_t)lane, row);
        if (t0 + (uint32_t)lane < c) io.hist[(pos + t0 + (uint32_t)lane) & DR_MASK] = (uint8_t)val;      // (lane j < 16: pattern byte j)
        op = pos + c;
        dr_flush_rows(io, op);
      }
      DR_SYNC();
      return;
    }
    const uint32_t head = len < G ? len : G;
    DR_SYNC();
    if ((uint32_t)lane < head) io.hist[(mpos + (uint32_t)lane) & DR_MASK] = (uint8_t)val;
    done = head; off_e = G;
    op = mpos + done;
    dr_flush_rows(io, op);
  }
  // Power-of-two periods up to a row, long matches (byte planes of a few significant bits decode into dozens of 4 KiB runs of period 128;
  // constant planes when spans are off).  The general loop below doubles its stride from `off` up to a row - five dependent LDS copies
  // before the first full row.  Here: ONE copy of the period (so that two periods lie back to back), then every lane reads its 16 bytes of a
  // row straight out of those periods (plane[q] = plane[b + ((q - b) & (per - 1))]: one unaligned 16-byte LDS read) - the same 16 bytes for
  // EVERY row, because the period divides a row - and that one register set is the rest of the first row, every whole row (stored to the
  // ring and to global memory) and the piece behind the last row boundary.
  // (Every ring write counted from the match's own start instead - no piece up to the row boundary - is 9 % slower: 16-byte LDS writes that
  //  are not 16-byte aligned.  profiles/r04/r04zt_*)
  if ((off & (off - 1u)) == 0u && off <= DR_ROW && len >= 2u * DR_ROW && mpos - off >= dr_near_lo(io, mpos + off)) {
    if (off >= 64u) { dr_copy_chunk(io, mpos, mpos - off, off, false, lane); done = off; op = mpos + done; dr_flush_rows(io, op); }
    const uint32_t per = off < 64u ? 32u : off, pm = per - 1u, b0 = mpos - off, l16 = 16u * (uint32_t)lane;     // (off < 64: the head above wrote 64 bytes, off divides 32)
    // 16 bytes of plane position q out of the two periods that end closest in front of `front` (everything written so far is periodic)
    auto base_for = [&](uint32_t front) { const uint32_t t = front - per - 16u; return t - ((t - b0) & pm); };
    uint32_t pos = mpos + done;
    const uint32_t end = mpos + len, mis = pos & (DR_ROW - 1u);
    // Round 5: a lane's 16 bytes of ANY row are the same (the period divides a row), so ONE read out of the two periods serves the rest of the first
    // row, every whole row and the piece behind the last boundary - 7 LDS round trips per match instead of 13 - 15 (bench19's planes 2 / 6 are 32 such
    // matches each, a quarter of the block's wave cycles: profiles/r04/r04zl_*).  Lanes that straddle `pos` rewrite up to 15 bytes in front of it with
    // the same bytes (this match's own output, >= 33 bytes of it exist); the last piece may write up to 15 bytes BEYOND the match (DR_GUARD).
    {
      const uint32_t rb = pos - mis, b = base_for(pos);
      DR_SYNC();
      const uint4 row = dr_get16(io.hist, b + ((rb + l16 - b) & pm));
      DR_SYNC();
      if (mis) {
        if (l16 + 16u > mis) l_st16(io.hist + ((rb + l16) & DR_MASK), row);
        pos = rb + DR_ROW; op = pos;                           // (len >= 2 rows and done <= 1 row: the boundary lies inside the match)
        dr_flush_rows(io, op);
      }
      for (; end - pos >= DR_ROW; pos += DR_ROW) {             // pos is a row boundary and everything below it has been flushed
        l_st16(io.hist + ((pos + l16) & DR_MASK), row);
        if (DR_FLUSH_ON(io)) g_st16(io.out + pos + l16, row);
      }
      io.flushed = pos; op = pos;
      if (end > pos) { if (l16 < end - pos) l_st16(io.hist + ((pos + l16) & DR_MASK), row); op = end; }
      DR_SYNC();
      return;
    }
  }
  while (done < len) {
    const uint32_t pos = mpos + done, rem = len - done;
    while (off_e < DR_ROW && 2u * off_e <= off + done) off_e *= 2u;      // the history written so far is periodic: lengthen the stride
    uint32_t c = rem < DR_ROW ? rem : DR_ROW;
    if (c > off_e) c = off_e;
    const uint32_t src = pos - off_e;
    const bool far = src < dr_near_lo(io, pos + c);
    if (far && src + c > io.flushed) c = io.flushed - src;              // (src < near_lo <= flushed: at least one byte)
    dr_copy_chunk(io, pos, src, c, far, lane);
    done += c;
    op = mpos + done;
    dr_flush_rows(io, op);
  }
}

// `ll` literal bytes from stream position ip to plane position op (both advance); runs that do not sit in the input ring come
// straight from the compressed stream, one row at a time
__device__ __forceinline__ void dr_literals(RingIO& io, uint32_t& ip, uint32_t& op, uint32_t ll, int lane) {
  if (ll <= 64u) {                                            // (dr_input(ip0) with ip <= ip0 + 4 covers ip + 64 + 4)
    const uint32_t v = dr_in4(io, ip + (uint32_t)lane);
    DR_SYNC();
    if ((uint32_t)lane < ll) io.hist[(op + (uint32_t)lane) & DR_MASK] = (uint8_t)v;
    ip += ll; op += ll;
    DR_SYNC();
    dr_flush_rows(io, op);
    return;
  }
  uint32_t done = 0;
  while (done < ll) {
    const uint32_t c = ll - done < DR_ROW ? ll - done : DR_ROW, n16 = c >> 4, l16 = 16u * (uint32_t)lane, t0 = c & ~15u;
    const gu8* s = io.in + ip + done;
    uint4 v = make_uint4(0u, 0u, 0u, 0u); uint32_t tb = 0u;
    if ((uint32_t)lane < n16) v = g_ld16(s + l16);
    if (t0 + (uint32_t)lane < c) tb = s[t0 + (uint32_t)lane];
    DR_SYNC();
    if ((uint32_t)lane < n16) dr_put16(io.hist, op + l16, v);
    if (t0 + (uint32_t)lane < c) io.hist[(op + t0 + (uint32_t)lane) & DR_MASK] = (uint8_t)tb;
    done += c; op += c;
    DR_SYNC();
    dr_flush_rows(io, op);
  }
  ip += ll;
}

// LZ4's 255-run length extension (lz4.c:2240-2250 / :2330-2342), 64 stream bytes per round: value += 255 * (leading 0xFF bytes) + the
// first other byte; ip ends behind that byte.  Bytes beyond the stream read as zero, i.e. as a terminator: the caller's bound check
// on the final ip (monotonic, so equivalent to the reference's per-byte check) catches a run into the end.
__device__ __forceinline__ void dr_ext_run(RingIO& io, uint32_t& ip, uint32_t& value, uint32_t cap, int lane) {
  for (;;) {
    dr_input(io, ip);
    const uint32_t B = dr_in4(io, ip + (uint32_t)lane) & 0xffu;
    const uint64_t m = __ballot(B != 255u);
    if (m) {
      const uint32_t k = (uint32_t)__builtin_ctzll(m);
      value += 255u * k + (uint32_t)__builtin_amdgcn_readlane((int)B, (int)k);
      ip += k + 1u;
      return;
    }
    value += 255u * 64u; ip += 64u;
    if (value > cap) return;                       // the caller rejects; keeps the loop bounded by cap
    // 64 bytes of 0xFF: a long run (the one match of a constant or periodic plane is 130 KB = 513 such bytes).  The rest of it straight from
    // the stream, a dword per lane = 256 bytes per memory round trip instead of 64 per trip through the input ring (reference-written config-2 chunks - 1.5 %,
    // profiles/r04/r04zs_*).  A dword that reaches beyond the stream reads as a terminator at its first byte: ip then ends within four
    // bytes of the end, which every caller rejects (lz4.c:2240-2250, :2330-2342), as it would a run of real 0xFF bytes into the end.
    for (;;) {
      const uint32_t p = ip + 4u * (uint32_t)lane;
      uint32_t w = 0u;
      if (p + 4u <= io.n) w = g_ld4(io.in + p);
      const uint64_t m2 = __ballot(w != 0xffffffffu);
      if (m2) {
        const uint32_t k = (uint32_t)__builtin_ctzll(m2);
        const uint32_t wk = (uint32_t)__builtin_amdgcn_readlane((int)w, (int)k);
        const uint32_t b = (uint32_t)__builtin_ctz(~wk) >> 3;                  // first byte of that dword that is not 0xFF
        const uint32_t nff = 4u * k + b;
        value += 255u * nff + ((wk >> (8u * b)) & 0xffu);
        ip += nff + 1u;
        return;
      }
      value += 255u * 256u; ip += 256u;
      if (value > cap) return;
    }
  }
}

// ---- periodic spans on top of the ring (see SpanCtx in k_decode.hip for what they are) -----------------------------------------
// A match of >= 16 KiB with a power-of-two distance is not written at all: head up to the next row boundary, the 2 KiB pattern
// table (distances <= 2048; out of the ring) or the self-span words, and the bytes behind the last row boundary.  The rows in
// between are marked flushed although nobody wrote them and the ring is declared empty below `hi`.
// (a real call with plain arguments, made once per plane at most: its registers must not count against the hot loop of the caller,
//  and a RingIO handed over by reference would live in memory)
__device__ __attribute__((noinline)) void dr_span_call(volatile uint32_t* lds_, gu8* out_, uint32_t flushed_, uint32_t mpos_, uint32_t off_, uint32_t ml_, gu8* pat_, int lane) {
  RingIO io;
  {
    const uint64_t lv = (uint64_t)lds_;
    volatile uint32_t* lds = (volatile uint32_t*)(((uint64_t)uni((uint32_t)(lv >> 32)) << 32) | uni((uint32_t)lv));
    io.scr = (volatile BAMD_LAS uint32_t*)lds; io.in32 = (BAMD_LAS uint32_t*)lds + 64; io.hist = (lu8*)((BAMD_LAS uint32_t*)lds + 64 + DR_IN / 4u);
  }
  io.in = nullptr; io.n = 0u; io.out = uni_ptr(out_); io.cap = 0u; io.b_hi = 0u; io.pend = 0u; io.pv = 0u; io.flushed = uni(flushed_); io.rfloor = 0u; io.lane = lane;
#ifdef BAMD_LOO_PLANES
  io.noflush = 0u;
#endif
  const uint32_t mpos = uni(mpos_), off = uni(off_), ml = uni(ml_);
  gu8* pat = uni_ptr(pat_);
  const uint32_t lo = (mpos + 1023u) & ~1023u, hi = (mpos + ml) & ~1023u, end = mpos + ml;
  uint32_t op = mpos;
  if (lo > mpos) dr_match(io, op, off, lo - mpos, lane);          // head: through the ring; the rows up to lo are flushed by it
  DR_SYNC();
  const uint32_t base = mpos - off, pm = off - 1u;                // plane[q] = plane[base + ((q - base) & pm)] for q >= base
  if (off > SPAN_PAT) {
    // self span: the unshuffle reads the plane's own period out of the rows in front of the span (flushed above: lo >= mpos)
    const uint32_t ob = (mpos - off + 3u) & ~3u;
    if (lane == 0) { g_st4(pat, ob); g_st4(pat + 4, off); }
    BAMD_MEM_SYNC();
    // behind the span: < 1 KiB, byte by byte through the same mapping out of those rows (all loads first)
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint32_t q = hi + 64u * (uint32_t)k + (uint32_t)lane; v[k] = q < end ? (uint32_t)io.out[ob + ((q - ob) & pm)] : 0u; }
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint32_t q = hi + 64u * (uint32_t)k + (uint32_t)lane; if (q < end) io.hist[q & DR_MASK] = (uint8_t)v[k]; }
  } else {
    // pattern table pat[i] = plane[q], q = i (mod 2048): the period sits in the ring (off <= 2048), four bytes per lane and round
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t i = 4u * ((uint32_t)lane + 64u * (uint32_t)k);
      const uint32_t b0 = io.hist[(base + ((i - base) & pm)) & DR_MASK], b1 = io.hist[(base + ((i + 1u - base) & pm)) & DR_MASK];
      const uint32_t b2 = io.hist[(base + ((i + 2u - base) & pm)) & DR_MASK], b3 = io.hist[(base + ((i + 3u - base) & pm)) & DR_MASK];
      g_st4(pat + i, b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
    }
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint32_t q = hi + 64u * (uint32_t)k + (uint32_t)lane; v[k] = q < end ? (uint32_t)io.hist[(base + ((q - base) & pm)) & DR_MASK] : 0u; }
    DR_SYNC();
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint32_t q = hi + 64u * (uint32_t)k + (uint32_t)lane; if (q < end) io.hist[q & DR_MASK] = (uint8_t)v[k]; }
  }
  DR_SYNC();
}
// returns true when the match was taken as a span (op then stands behind it)
__device__ __forceinline__ bool dr_span_long_match(RingIO& io, uint32_t& op, uint32_t off, uint32_t ml, int lane, SpanCtx& sp) {
  if (!sp.enabled || sp.hi || ml < 16384u || off > 65536u || (off & (off - 1u))) return false;
  const uint32_t mpos = op, lo = (mpos + 1023u) & ~1023u, hi = (mpos + ml) & ~1023u;
  if (hi < lo + 8192u) return false;
  if (off > SPAN_PAT && mpos >= (1u << 24)) return false;         // (see span_long_match: the self-span base travels in 24 bits)
  dr_span_call((volatile uint32_t*)io.scr, io.out, io.flushed, mpos, off, ml, sp.pat, lane);
  io.flushed = hi; io.rfloor = hi;                                // (a span is only taken while rfloor is 0: sp.hi was 0 and sp.enabled still set)
  sp.lo = lo; sp.hi = hi; sp.off = off;
  op = mpos + ml;
  return true;
}
// a later match reaches into a skipped range: the rows are written after all (global memory only: the ring stays empty below rfloor,
// so such sources are read from the rows)
__device__ __forceinline__ void dr_span_materialize(RingIO& io, SpanCtx& sp, int lane) {
  if (sp.hi) span_fill_call(io.out, sp.lo, sp.off, sp.hi - sp.lo, lane);
  sp.lo = 0; sp.hi = 0; sp.enabled = 0;
}

// ---------------------------------------------------------------------------------------------
// Batched step: up to 16 consecutive sequences whose tokens, literals, offsets (and at most one extension byte per length) all
// lie in the 64 stream bytes at ip.
//   1. every lane l reads "its" byte as if it were a token and works out where the next token would be (speculative parse: two LDS
//      reads per lane - the four bytes at ip + l, the three bytes behind the candidate's literals);
//   2. the real token chain is resolved by pointer doubling in rank space (7 ds_bpermute), a DPP row scan places every sequence;
//   3. literal bytes of ALL accepted sequences go into the ring with one scattered byte store;
//   4. matches of <= 64 bytes whose source lies in front of the step's output are copied by 4 lanes each with overlapping
//      4/8/16-byte pieces: LDS -> LDS, or - sources older than the ring - rows in global memory -> LDS (one gather for all of them);
//   5. the rest (longer, reading this step's own output, or touching the ring's end) runs in stream order, byte lanes, out of LDS.
// Returns the number of sequences done; 0 = the token at ip is not one this step can take (hdr then holds the four bytes at ip for
// the caller's one-sequence path).  The caller guarantees ip + 72 <= n (so no accepted sequence can be the stream's final one and
// "literals end >= 8 bytes before the input end", lz4.c:2279, holds) and dr_input(ip).  Output-side rules (lz4.c:2279, :2423) and
// offset validity are checked per sequence; a sequence that breaks one is simply not accepted, so the one-sequence path re-parses it
// and reports the error.
// ---------------------------------------------------------------------------------------------
template <int G>      // G = 0: LZ4's grammar; G = 1: BloscLZ's (blosclz.c:679-789), `ip` then is the position of the current control byte
__device__ __forceinline__ uint32_t dr_step(RingIO& io, uint32_t& ip, uint32_t& op, uint32_t cap, uint32_t n, uint32_t& hdr, SpanCtx& sp, int lane PROF_ARG) {
  // ---- 1. speculative parse ----
  const uint32_t D = dr_in4(io, ip + (uint32_t)lane);
  const uint32_t B = D & 0xffu;
  uint32_t ll, ml, off, size;
  bool complete, ll_ext = false;
  if (G == 0) {
    const uint32_t e_ll = (D >> 8) & 0xffu;
    const uint32_t ll0 = B >> 4, mlc = B & 15u;
    ll_ext = ll0 == 15u;
    ll = ll_ext ? 15u + e_ll : ll0;
    const uint32_t offpos = (uint32_t)lane + 1u + (ll_ext ? 1u : 0u) + ll;   // where this token's offset would start
    const uint32_t O = dr_in4(io, ip + (offpos < 64u ? offpos : 64u));
    off = O & 0xffffu;
    const uint32_t e1 = (O >> 16) & 0xffu;
    const bool has_ext = mlc == 15u;
    ml = has_ext ? 19u + e1 : mlc + 4u;           // <= 273
    size = 3u + ll + (has_ext ? 1u : 0u) + (ll_ext ? 1u : 0u);   // token (+ ext) + literals + offset (+ ext)
    // lz4.c:2240-2250: the length extension may not be read at or behind n - 15
    complete = !(ll_ext && (e_ll == 255u || ip + (uint32_t)lane + 16u >= n)) && !(has_ext && e1 == 255u) && (uint32_t)lane + size <= 64u;
  } else {
    // a control byte < 32 starts a literal run of ctrl + 1 bytes; otherwise a match: len = (ctrl >> 5) - 1 (+ one extension byte when that field
    // is 7; longer extensions are the one-token path's) + 3, distance - 1 = ((ctrl & 31) << 8) + next byte, or a 16-bit big-endian value + 8191
    // behind the escape 31 / 255.  Only called with ip + 72 <= n: every token taken here is followed by more input, so the reference's
    // end-of-input quirks cannot apply.
    const uint32_t b1 = (D >> 8) & 0xffu, b2 = (D >> 16) & 0xffu, b3 = D >> 24, b4 = dr_in4(io, ip + (uint32_t)lane + 4u) & 0xffu;
    const bool is_lit = B < 32u;
    const uint32_t l3 = B >> 5;
    const bool has_ext = l3 == 7u;
    const uint32_t code = has_ext ? b2 : b1;
    const bool far = !is_lit && code == 255u && (B & 31u) == 31u;
    const uint32_t f0 = has_ext ? b3 : b2, f1 = has_ext ? b4 : b3;
    ll = is_lit ? B + 1u : 0u;                                                          // 1 .. 32
    ml = is_lit ? 0u : l3 + 2u + (has_ext ? b1 : 0u);                                   // 3 .. 263
    off = far ? ((f0 << 8) | f1) + 8192u : ((B & 31u) << 8) + code + 1u;                // the true distance
    size = is_lit ? B + 2u : 2u + (has_ext ? 1u : 0u) + (far ? 2u : 0u);
    complete = (is_lit || !(has_ext && b1 == 255u)) && (uint32_t)lane + size <= 64u;
  }
  hdr = (uint32_t)__builtin_amdgcn_readlane((int)D, 0);
  if (!(__ballot(complete) & 1ull)) return 0u;                 // the token at ip itself: left to the one-sequence path
  const uint32_t nxt = complete ? (uint32_t)lane + size : 64u; // position of the following token, 64 = stop here
  // the hop table: "stop" is lane 63 - never a complete token (a sequence is >= 3 bytes), its own entry points at itself, and whoever
  // lands on it is dropped by the `complete` test below - so a hop is one ds_bpermute with no range check behind it
  const uint32_t nxh = nxt < 63u ? nxt : 63u;
  PROF_LAP(8);
  // ---- 2. token chain in rank space: J1 = J0 o J0, J2 = J1 o J1, J3 = J2 o J2; lane r (< 16) finds the r-th token.  (Round 4 also built
  //         a scalar walk - v_readlane with the position in an SGPR, v_writelane into the rank lane, <= 16 hops of ~10 cycles, and the
  //         literal placement through one more ds_bpermute instead of the 64-dword scratch: 2.6 % SLOWER, the kernel is bound by the
  //         number of VALU / SALU instructions it issues and the walk adds to it: profiles/r04/r04g_dec_ab_bisect_walk_pipe_rowfill.txt) ----
  uint32_t c = 0;                                              // ip is a real token by invariant
  {
    const uint32_t J0 = nxh;
    const uint32_t J1 = bperm(J0, J0), J2 = bperm(J1, J1), J3 = bperm(J2, J2);
    { const uint32_t t = bperm(c, J0); c = (lane & 1) ? t : c; }
    { const uint32_t t = bperm(c, J1); c = (lane & 2) ? t : c; }
    { const uint32_t t = bperm(c, J2); c = (lane & 4) ? t : c; }
    { const uint32_t t = bperm(c, J3); c = (lane & 8) ? t : c; }
  }
  const uint32_t pk = bperm(c & 63u, ll | (ml << 9) | ((complete ? 1u : 0u) << 18) | ((ll_ext ? 1u : 0u) << 19) | (nxt << 20));
  const uint32_t off_r = bperm(c & 63u, off);
  const uint32_t ll_r = pk & 0x1ffu, ml_r = (pk >> 9) & 0x1ffu, nxt_r = pk >> 20, ext_r = (pk >> 19) & 1u;
  const bool valid = lane < (int)BATCH_MAXSEQ && c < 64u && ((pk >> 18) & 1u);      // (c = 63, the stop lane, carries complete = 0)
  const uint32_t tot_r = valid ? ll_r + ml_r : 0u;
  uint32_t incl = tot_r;                                       // inclusive prefix sum over the 16 rank lanes (one DPP row)
  incl += row_shr<1>(incl); incl += row_shr<2>(incl); incl += row_shr<4>(incl); incl += row_shr<8>(incl);
  const uint32_t excl = incl - tot_r;                          // output offset of sequence r relative to op
  const uint32_t mrel_r = excl + ll_r;                         // its match start, relative to op
  // acceptance: offset inside the produced data (lz4.c:2356, and offset 0), far enough from the output end that neither lz4.c:2279
  // nor :2423 can apply, and the step within DR_STEP_MAX bytes
  const bool ok = G == 0 ? (valid && off_r != 0u && off_r <= op + mrel_r && op + excl + tot_r + 12u <= cap && incl <= DR_STEP_MAX)
                         : (valid && (ml_r == 0u || off_r <= op + mrel_r) && op + excl + tot_r <= cap && incl <= DR_STEP_MAX);      // blosclz.c:730-735
  const uint32_t okmask = (uint32_t)__ballot(ok) & 0xffffu;
  const uint32_t cnt = (uint32_t)__builtin_ctz(~okmask);       // leading accepted sequences (<= 16)
  PROF_LAP(9);
  if (cnt == 0u) return 0u;
  const bool mine = (uint32_t)lane < cnt;
  const uint32_t src_r = op + mrel_r - off_r;                  // source position (accepted lanes only)
  // a source inside a skipped periodic span: the span is written after all (rare)
  if (sp.hi && __ballot(mine && (G == 0 || ml_r != 0u) && src_r < sp.hi)) dr_span_materialize(io, sp, lane);
  const uint32_t consumed = (uint32_t)__builtin_amdgcn_readlane((int)nxt_r, (int)(cnt - 1u));
  const uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(cnt - 1u));
  const uint32_t W = op + acc, nlo = dr_near_lo(io, W);
  // ---- 3. literals: token info goes back to byte-lane space through the 64-dword scratch, then one scattered byte store ----
  // scratch word of a token: valid | length-extension flag << 25 | literal count << 16 | output offset (<= DR_STEP_MAX)
  {
    io.scr[lane] = 0u;
    BAMD_LDS_SYNC();
    if (mine) io.scr[c] = 0x80000000u | excl | (ll_r << 16) | (ext_r << 25);
    BAMD_LDS_SYNC();
    const uint64_t mask = __ballot(io.scr[lane] >> 31);
    const uint64_t below = mask & ((2ull << lane) - 1ull);     // accepted tokens at or before this byte lane
    const uint32_t s = 63u - (uint32_t)__builtin_clzll(below | 1ull);
    const uint32_t inf = io.scr[s];
    const uint32_t xe = (inf >> 25) & 1u;
    const uint32_t k = (uint32_t)lane - s - 1u - xe;
    if ((uint32_t)lane < consumed && (uint32_t)lane > s + xe && k < ((inf >> 16) & 0x1ffu)) io.hist[(op + (inf & 0xffffu) + k) & DR_MASK] = (uint8_t)B;
  }
  DR_SYNC();
  // ---- 4. short matches whose source lies in front of the step's output: 4 lanes each, overlapping 4/8/16-byte pieces ----
  const uint32_t dpos_r = op + mrel_r;
  // a source is taken from the ring (wholly at or above nlo) or from the rows already written (wholly below nlo: nlo <= flushed, by the
  // static_assert above and because a span's floor is a flushed position); one that straddles nlo goes byte by byte in step 5
  const bool far_r = src_r + ml_r <= nlo;
  const bool fast_r = mine && (G == 0 || ml_r >= 4u) && ml_r <= 64u && off_r >= mrel_r + ml_r && (dpos_r & DR_MASK) + ml_r <= DR_RING && (far_r || (src_r >= nlo && (src_r & DR_MASK) + ml_r <= DR_RING));
  const bool anyfar = __ballot(fast_r && far_r) != 0ull;
  {
    const uint32_t r = (uint32_t)lane >> 2, q = (uint32_t)lane & 3u;
    const uint32_t fA = bperm(r, fast_r ? (ml_r | 0x200u | (mrel_r << 10) | (far_r ? 0x80000000u : 0u)) : 0u);
    const uint32_t fB = bperm(r, off_r);
    const uint32_t mlen = fA & 0x1ffu;
    const bool go = (fA & 0x200u) != 0u, isfar = (fA >> 31) != 0u;
    const uint32_t dp = op + ((fA >> 10) & 0xfffu), spos = dp - fB;
    lu8* d = io.hist + (dp & DR_MASK);
    const lu8* sl = io.hist + (spos & DR_MASK);
    const uint32_t np16 = (mlen + 15u) >> 4;
    const bool w16 = go && mlen >= 16u && q < np16;
    const bool w8 = go && mlen >= 8u && mlen < 16u && q < 2u;
    const bool w4 = go && mlen < 8u && q < 2u;
    const uint32_t po16 = (q == np16 - 1u) ? mlen - 16u : 16u * q;
    const uint32_t po8 = q ? mlen - 8u : 0u, po4 = q ? mlen - 4u : 0u;
    // (Tried in round 4: the far pieces kept in their registers and stored into the ring behind the NEXT step's parse, so that their round
    //  trip runs under it.  5-9 % SLOWER on every data set, typesize 2 included where one step in two has a far source - the state it
    //  carries across steps costs more than the wait: profiles/r04/r04n_dec_ab_deferred_far_pieces_rejected.txt.
    //  Round 5: because the far loads and the ring loads share their registers, the compiler waits for vmcnt(0) - the previous step's row stores -
    //  in front of the ring loads and piece stores of EVERY step; with the far form out of line the ordinary step holds no such wait, and
    //  nothing changes: profiles/r05m_*.  Sections 3 and 4 merged into three LDS round trips instead of six: 6 % SLOWER, profiles/r05l_*.)
    uint4 v16 = make_uint4(0, 0, 0, 0); uint64_t v8 = 0; uint32_t v4 = 0;
    if (anyfar) {                                              // (wave-uniform: a step without far sources never waits for memory)
      const gu8* sg = io.out + spos;
      BAMD_MEM_SYNC();
      if (w16) v16 = isfar ? g_ld16(sg + po16) : l_ld16(sl + po16);
      if (w8) v8 = isfar ? g_ld8(sg + po8) : l_ld8(sl + po8);
      if (w4) v4 = isfar ? g_ld4(sg + po4) : l_ld4(sl + po4);
#ifdef BAMD_WAVE_EMU
      if (lane == 0) g_emu_ring_far++;
#endif
    } else {
      if (w16) v16 = l_ld16(sl + po16);
      if (w8) v8 = l_ld8(sl + po8);
      if (w4) v4 = l_ld4(sl + po4);
    }
    DR_SYNC();
    if (w16) l_st16(d + po16, v16);
    if (w8) l_st8(d + po8, v8);
    if (w4) l_st4(d + po4, v4);
  }
  DR_SYNC();
  PROF_LAP(10);
  // ---- 5. everything else in stream order: byte lanes, the periodic extension of the off bytes in front of the match when it
  //         overlaps itself (every lane then reads only bytes that are already final).  (Bit planes send half of a step's sequences
  //         here, 5.5 dependency levels for 10 sequences per step: taking two neighbours at a time when the second does not read what
  //         the first writes is 2 % SLOWER - the test costs more than the round trips it saves: profiles/r04/r04zx_*, r04zy_*.) ----
  uint32_t rest = (uint32_t)__ballot(mine && !fast_r && (G == 0 || ml_r != 0u));      // (BloscLZ: literal runs are tokens of their own, and matches of 3 bytes go here)
  PROF_ADD(0, 1); PROF_ADD(1, cnt); PROF_ADD(2, __builtin_popcount(rest));
  while (rest) {
    const int sl = __builtin_ctz(rest);
    rest &= rest - 1u;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)ml_r, sl);
    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off_r, sl);
    const uint32_t mr = (uint32_t)__builtin_amdgcn_readlane((int)mrel_r, sl);
    const uint32_t s0 = op + mr - o;
    // (Round 6: every byte read here is final before the sequence begins and none is written by it, so all - up to five - reads of a lane could go out before its
    //  writes, one round trip instead of five (memory trips when the source is older than the ring: the reference's linspace planes take runs of 255 bytes from
    //  65 535 bytes back).  Built: 3 - 12 % SLOWER on linspace, bench19 and the bit planes alike, profiles/r06r_*.  So was dr_match's register fill for runs of more than
    //  64 bytes in here: nothing on linspace, 5 - 9 % slower on the bit planes, whose steps send dozens of sequences through this loop (profiles/r06s_*).  The in-order loops stay.)
    if (s0 >= nlo) {
      // k mod o per lane with a float reciprocal (exact here: the quotient is only needed when o < m <= 273, k < 512; an integer division
      // per match was 23 scalar instructions, k_zstd.hip: zstd_exec16_lds has the story)
      const float ro = o < m ? __builtin_amdgcn_rcpf((float)o) : 0.0f;
      for (uint32_t k = (uint32_t)lane; k < m; k += 64u) {
        uint32_t kk = k;
        if (o < m) { kk = k - (uint32_t)((float)k * ro) * o; kk = kk >= o ? kk - o : kk; }
        io.hist[(op + mr + k) & DR_MASK] = io.hist[(s0 + kk) & DR_MASK];
      }
    } else {
      // (partly) older than the ring: those bytes out of the rows already written (every position below nlo is in them), the others out of the ring
      BAMD_MEM_SYNC();
      const float ro = o < m ? __builtin_amdgcn_rcpf((float)o) : 0.0f;
      for (uint32_t k = (uint32_t)lane; k < m; k += 64u) {
        uint32_t kk = k;
        if (o < m) { kk = k - (uint32_t)((float)k * ro) * o; kk = kk >= o ? kk - o : kk; }
        const uint32_t p = s0 + kk;
        const uint32_t v = p < nlo ? (uint32_t)io.out[p] : (uint32_t)io.hist[p & DR_MASK];
        io.hist[(op + mr + k) & DR_MASK] = (uint8_t)v;
      }
#ifdef BAMD_WAVE_EMU
      if (lane == 0) g_emu_ring_far++;
#endif
    }
    DR_SYNC();
  }
#ifdef BAMD_WAVE_EMU
  if (lane == 0) g_emu_ring_steps++;
#endif
  ip += consumed;
  op = W;
  PROF_LAP(11);
  return cnt;
}

// ---------------------------------------------------------------------------------------------
// LZ4 block decode, one wave.  Returns bytes produced (== cap on success) or a negative number.
// Acceptance rules are those of the reference's safe loop (lz4.c:2215-2435):
//   literal-length extension stops reading at n-15, match-length extension at n-4;
//   a literal run reaching within 12 bytes of the output end or 8 of the input end must be the
//   last one and end exactly at the input end; offset <= bytes produced; a match must end at
//   least 5 bytes before the output end.  Offset 0 is accepted like the reference does (the match bytes
//   are then whatever the output buffer held).
// `lds`: DR_LDS_BYTES of LDS owned by this wave.
// ---------------------------------------------------------------------------------------------
__device__ int lz4_decode_wave(const gu8* __restrict__ in_, int32_t n_, gu8* out_, int32_t cap_, volatile uint32_t* lds, int lane, SpanCtx& sp PROF_ARG) {
  if (cap_ == 0) return (n_ == 1 && in_[0] == 0) ? 0 : -1;
  if (n_ <= 0) return -1;
  const uint32_t n = uni((uint32_t)n_), cap = uni((uint32_t)cap_);
  RingIO io;
  io.scr = (volatile BAMD_LAS uint32_t*)lds;
  io.in32 = (BAMD_LAS uint32_t*)lds + 64;
  io.hist = (lu8*)((BAMD_LAS uint32_t*)lds + 64 + DR_IN / 4u);
  io.in = uni_ptr(in_); io.n = n; io.out = uni_ptr(out_); io.cap = cap;
  io.b_hi = 0u; io.pend = 0u; io.pv = 0u; io.flushed = 0u; io.rfloor = 0u; io.lane = lane;
#ifdef BAMD_LOO_PLANES
  io.noflush = sp.loo;
#endif
  uint32_t ip = 0, op = 0;
  for (;;) {
    // (the rows of the last step could also leave behind dr_input's wait for its prefetched block - vmcnt counts loads and stores in one
    //  order, so that wait sits out the youngest store's round trip: no difference, profiles/r04/r04zq_*)
    dr_input(io, ip);
    uint32_t hdr;
    if (ip + 72u <= n) {
      if (dr_step<0>(io, ip, op, cap, n, hdr, sp, lane PROF_PASS)) { dr_flush_rows(io, op); PROF_LAP(13); continue; }
    } else hdr = dr_peek32(io, ip);
    // ---- one sequence (long runs, long matches, the stream's tail, everything the step refused) ----
    PROF_ADD(3, 1);
    const uint32_t token = hdr & 0xffu;
    ip += 1;
    uint32_t ll = token >> 4;
    if (ll == 15u) {
      if (n < 15u || ip >= n - 15u) return -2;
      dr_ext_run(io, ip, ll, cap, lane);
      if (ip > n - 15u || ll > cap) return -2;
      dr_input(io, ip);
    }
    // ---- literals ----
    if (op + ll + 12u > cap || ip + ll + 8u > n) {
      // must be the final run
      if (ip + ll != n || op + ll > cap) return -3;
      dr_literals(io, ip, op, ll, lane);
      break;
    }
    if (ll) {                                            // (a match right behind a match: nothing to copy, and ip has not moved since the last dr_input)
      dr_literals(io, ip, op, ll, lane);
      dr_input(io, ip);
    }
    const uint32_t t2 = dr_peek32(io, ip);      // (taking these bytes out of `hdr` when there were no literals - one LDS read less per long match of a run plane - changes nothing: profiles/r05n_*)
    const uint32_t off = t2 & 0xffffu;
    ip += 2;
    uint32_t ml = token & 15u;
    if (ml == 15u) {
      const uint32_t s0 = (t2 >> 16) & 0xffu;   // first extension byte is already in the peeked word
      ip++; ml += s0;
      if (ip > n - 4u) return -4;
      if (s0 == 255u) {
        dr_ext_run(io, ip, ml, cap, lane);
        if (ip > n - 4u || ml > cap) return -4;
      }
    }
    ml += 4u;
    const uint32_t mpos = op;
    if (off > mpos) return -5;
    if (mpos + ml + 5u > cap) return -6;
    if (off == 0u) {
      // The reference does not reject offset 0 (lz4.c:2356 only checks the lower bound): it "copies" the match from its own
      // destination, i.e. leaves whatever the output buffer held.  Same here: the bytes the buffer holds pass through the ring.
      uint32_t done = 0;
      while (done < ml) {
        const uint32_t c = ml - done < DR_ROW ? ml - done : DR_ROW;
        const uint32_t n16 = c >> 4, l16 = 16u * (uint32_t)lane, t0 = c & ~15u;
        uint4 v = make_uint4(0u, 0u, 0u, 0u); uint32_t tb = 0u;
        if ((uint32_t)lane < n16) v = g_ld16(io.out + op + l16);
        if (t0 + (uint32_t)lane < c) tb = io.out[op + t0 + (uint32_t)lane];
        DR_SYNC();
        if ((uint32_t)lane < n16) dr_put16(io.hist, op + l16, v);
        if (t0 + (uint32_t)lane < c) io.hist[(op + t0 + (uint32_t)lane) & DR_MASK] = (uint8_t)tb;
        DR_SYNC();
        done += c; op += c;
        dr_flush_rows(io, op);
      }
      continue;
    }
    if (sp.hi && mpos - off < sp.hi) dr_span_materialize(io, sp, lane);
    if (!dr_span_long_match(io, op, off, ml, lane, sp)) dr_match(io, op, off, ml, lane);
    PROF_LAP(12);
  }
  dr_flush_tail(io, op);
  PROF_LAP(12);
  return (int)op;
}

// ---------------------------------------------------------------------------------------------
// BloscLZ block decode on the same ring, one wave (round 4; until then this decoder wrote straight to global memory and read its
// history back from there).  Returns the bytes produced (the caller compares with the expected size), 0 on the errors the reference
// returns 0 for.  Grammar and rules: blosclz.c:679-789 - the first byte is always a literal-run control (its upper three bits are the
// format marker, :688), a length field of 7 is extended by a 255-run, the distance escape 31 / 255 is followed by a 16-bit big-endian
// distance - 8191, `distance > produced` and `produced + len > room` return 0, and a match whose token ends exactly at the input end
// is dropped (:700-736: the loop stops before the copy).
// ---------------------------------------------------------------------------------------------
__device__ int blosclz_decode_wave(const gu8* __restrict__ in_, int32_t n_, gu8* out_, int32_t cap_, volatile uint32_t* lds, int lane, SpanCtx& sp) {
  if (n_ <= 0) return 0;
  const uint32_t n = uni((uint32_t)n_), cap = uni((uint32_t)cap_);
  RingIO io;
  io.scr = (volatile BAMD_LAS uint32_t*)lds;
  io.in32 = (BAMD_LAS uint32_t*)lds + 64;
  io.hist = (lu8*)((BAMD_LAS uint32_t*)lds + 64 + DR_IN / 4u);
  io.in = uni_ptr(in_); io.n = n; io.out = uni_ptr(out_); io.cap = cap;
  io.b_hi = 0u; io.pend = 0u; io.pv = 0u; io.flushed = 0u; io.rfloor = 0u; io.lane = lane;
#ifdef BAMD_LOO_PLANES
  io.noflush = sp.loo;
#endif
  uint32_t tp = 0, op = 0;                      // tp: position of the current control byte
  PROF_DECL                                     // (the instrumented build counts the LZ4 streams only; the step's laps need a place to go)
  for (;;) {
    dr_input(io, tp);
    uint32_t hdr = dr_peek32(io, tp);
    // batched step from the current control byte; not for the stream's first byte (its marker bits are not a length)
    if (tp > 0u && tp + 72u <= n && !((hdr & 0xffu) >= 224u && ((hdr >> 8) & 0xffu) == 255u)) {      // (longer length extensions: below)
      if (dr_step<1>(io, tp, op, cap, n, hdr, sp, lane PROF_PASS)) { dr_flush_rows(io, op); continue; }
    }
    // ---- one token ----
    uint32_t ctrl = hdr & 0xffu;
    if (tp == 0u) ctrl &= 31u;
    uint32_t ip = tp + 1u;
    if (ctrl >= 32u) {
      uint32_t len = (ctrl >> 5) - 1u;
      const uint32_t ofs = (ctrl & 31u) << 8;
      if (len == 6u) {
        // blosclz.c:712-720: every extension byte must have a byte behind it (ip + 1 < n), the sum may not pass the room; both bounds are
        // monotonic in ip, so they are checked once behind the run (bytes beyond the stream read as zero, i.e. end the run)
        dr_ext_run(io, ip, len, cap, lane);
        if (ip >= n || len > cap) return 0;
        dr_input(io, ip);
      } else if (ip + 1u >= n) return 0;
      const uint32_t t = dr_peek32(io, ip);
      const uint32_t code = t & 0xffu;
      ip++;
      len += 3u;
      uint32_t dist = ofs + code;               // distance - 1
      if (code == 255u && ofs == (31u << 8)) {
        if (ip + 1u >= n) return 0;
        dist = ((((t >> 8) & 0xffu) << 8) | ((t >> 16) & 0xffu)) + 8191u;
        ip += 2u;
      }
      if (op + len > cap) return 0;
      if (dist + 1u > op) return 0;             // reference: ref - 1 < output
      if (ip >= n) break;                       // quirk: the pending match is dropped
      tp = ip;
      if (sp.hi && op - (dist + 1u) < sp.hi) dr_span_materialize(io, sp, lane);
      if (!dr_span_long_match(io, op, dist + 1u, len, lane, sp)) dr_match(io, op, dist + 1u, len, lane);
    } else {
      const uint32_t run = ctrl + 1u;           // 1 .. 32 literal bytes
      if (op + run > cap) return 0;
      if (ip + run > n) return 0;
      dr_literals(io, ip, op, run, lane);       // (dr_input(tp) covers ip + 64 + 4)
      if (ip >= n) break;
      tp = ip;
    }
  }
  dr_flush_tail(io, op);
  return (int)op;
}

}  // namespace bamd
