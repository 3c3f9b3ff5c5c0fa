// dec_bulk.h — the steady-state loop of the LZ4 stream decoder (included by k_decode.hip behind lz4_batch_step).
//
// Why it exists (round 3, profiles/r03_dec_phase_before.txt + the ISA of decode_one_stream): a batched step of the round-2 decoder
// cost ~15 000 cycles for 13 sequences on reference-written bench19 planes, and two thirds of that were `s_waitcnt vmcnt(0)`.
// gfx950 has ONE counter for vector loads and stores, completing in issue order, so "this load has arrived" can only be said as
// "at most N younger operations are outstanding" - and the compiler can only say it when N is known at compile time.  The
// register window of the compressed bytes (wave_prims.h: Window) is loaded in one iteration and read a few iterations later,
// with a data-dependent number of stores in between: every read of it became vmcnt(0), i.e. every step first waited for the
// PREVIOUS step's match stores to be acknowledged, then for its own match loads (and the literal store in front of them).
//
// What is different here:
//  * the compressed bytes live in a 512-byte LDS ring (two 256-byte blocks) that is refilled one block at a time: the block's
//    vector load is issued IN FRONT of a step's match loads and written to LDS behind the wait for those, two steps before the
//    parse can reach it - so no instruction ever waits for the input stream on its own, and a step's 64 bytes are one
//    ds_read_u8.  (Tried first: the window through the scalar cache, s_load_dwordx16 from the constant address space.  It keeps
//    vmcnt clean too, but scalar loads share lgkmcnt with the LDS / bpermute traffic of the parse and complete out of order, so
//    every LDS wait became lgkmcnt(0) and waited for the prefetch it was meant to overlap.)
//  * every step issues the same three vector loads (lanes without a piece read a dummy address) and one literal store, so the
//    compiler can wait for a step's match bytes with an exact vmcnt(N) that leaves younger operations in flight;
//  * software pipeline: step k+1 is parsed and the loads of its independent matches (sources in front of step k's output) are
//    issued BEFORE step k's bytes are stored; their round trip runs under step k+1's parse instead of after it.
// Anything unusual - a token the batched parse cannot take, the stream's tail, a source inside a skipped periodic span, a step the
// LDS-assembled form (lz4_step_lds) should take - ends the loop with nothing of that step consumed, and lz4_decode_wave's
// round-2 code carries on from there.  Acceptance rules are exactly lz4_batch_step's (lz4.c:2215-2445, see there).
#pragma once

namespace bamd {

#ifdef BAMD_WAVE_EMU
inline unsigned long long g_emu_bulk_steps = 0;       // emulator only: steps the bulk loop executed (tests assert that it runs at all)
#endif

// ---- the input window: stream bytes [wb + 256 j, wb + 256 (j + 1)) of block j sit in ring half j & 1 ----
typedef volatile __attribute__((address_space(3))) uint8_t lds_vu8;
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
constexpr uint32_t BULK_RING = 512u;
// lane's dword of block j (clamped into the stream: the bulk loop never consumes bytes behind n - 4, see lz4_bulk)
__device__ __forceinline__ uint32_t bulk_block_load(const gu8* in, uint32_t n, uint32_t wb, uint32_t j, int lane) {
  uint32_t p = wb + 256u * j + 4u * (uint32_t)lane;
  if (p > n - 4u) p = n - 4u;
  return g_ld4(in + p);
}
__device__ __forceinline__ void bulk_block_store(lds_vu32* ring, uint32_t j, uint32_t v, int lane) { ring[64u * (j & 1u) + (uint32_t)lane] = v; }
// lane l gets stream byte pos + l
__device__ __forceinline__ uint32_t bulk_bytes(const lds_vu8* ring, uint32_t wb, uint32_t pos, int lane) {
  return ring[(pos - wb + (uint32_t)lane) & (BULK_RING - 1u)];
}

// one parsed step: fields of the r-th sequence in rank lane r (< 16), totals wave-uniform
struct BulkStep {
  uint32_t B;                       // stream byte ip + lane (literal bytes leave from here)
  uint32_t c;                       // byte lane of the r-th token
  uint32_t pk;                      // ll | ml << 9 | length-extension flag << 18
  uint32_t off_r, excl;             // match distance; output offset of the sequence relative to the step's op
  uint32_t cnt, consumed, acc;      // sequences taken, stream bytes consumed, output bytes produced
  uint32_t restmask;                // sequences (rank bits) whose match runs in stream order after the step's stores
  // pieces of the independent matches, in flight: 4 lanes per sequence (see lz4_batch_step, step 4)
  uint32_t fA, fB;
  uint4 v16; uint64_t v8; uint32_t v4;
};

// steps 1-2 of lz4_batch_step + the decision whether the bulk loop may take this step.  `dep`: output bytes in front of `op` that
// are parsed but not stored yet (the previous step's), so sources reaching into them cannot be loaded now.
__device__ __forceinline__ bool bulk_parse(BulkStep& q, uint32_t B, bool in_range, uint32_t ip, uint32_t op, uint32_t cap, uint32_t n, uint32_t dep,
                                           uint32_t span_hi, int lane) {
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
  const bool ok = valid && off_r != 0u && off_r <= op + mrel_r && op + excl + tot_r + 12u <= cap;
  const uint32_t okmask = (uint32_t)__ballot(ok) & 0xffffu;
  const uint32_t cnt0 = (uint32_t)__builtin_ctz(~okmask);
  const bool mine0 = (uint32_t)lane < cnt0;
  // The verdict, without a branch (the caller's loop has ONE exit, see lz4_bulk): nothing accepted; a source inside a skipped
  // periodic span (the round-2 path fills the span in); four or more matches that read this step's own output (the LDS-assembled
  // form is made for that - decided without `dep`); the stream's tail.
  const bool take = in_range && cnt0 != 0u && !(span_hi && __ballot(mine0 && op + mrel_r - off_r < span_hi)) &&
                    !(BAMD_DEC_LDS_STEP && __builtin_popcount((uint32_t)__ballot(mine0 && !(ml_r <= 64u && off_r >= mrel_r + ml_r))) >= LZB_MIN_REST);
  const uint32_t cnt = take ? cnt0 : 0u;
  const bool mine = (uint32_t)lane < cnt;
  // independent of everything not yet in memory: the source ends at or before op - dep
  const bool fast_r = mine && ml_r <= 64u && off_r >= mrel_r + ml_r + dep;
  q.B = B; q.c = c; q.pk = ll_r | (ml_r << 9) | (ext_r << 18); q.off_r = off_r; q.excl = excl;
  q.cnt = cnt; q.restmask = (uint32_t)__ballot(mine && !fast_r);
  const uint32_t last = cnt ? cnt - 1u : 0u;
  q.consumed = cnt ? (uint32_t)__builtin_amdgcn_readlane((int)nxt_r, (int)last) : 0u;
  q.acc = cnt ? (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)last) : 0u;
  // the pieces of the independent matches, 4 lanes per sequence (an empty step has none: its loads read the dummy address)
  {
    const uint32_t r = (uint32_t)lane >> 2;
    q.fA = bperm(r, fast_r ? (ml_r | 0x200u | (mrel_r << 10)) : 0u);
    q.fB = bperm(r, off_r);
  }
  return take;
}

__device__ __forceinline__ void bulk_issue_loads(BulkStep& q, gu8* out, uint32_t op, int lane) {
  const uint32_t qd = (uint32_t)lane & 3u;
  const uint32_t mlen = q.fA & 0x1ffu;
  const bool go = (q.fA & 0x200u) != 0u;
  const uint32_t so = op + (q.fA >> 10) - q.fB;              // source offset in `out` (only meaningful where go)
  const uint32_t np16 = (mlen + 15u) >> 4;
  const bool w16 = go && mlen >= 16u && qd < np16;
  const bool w8 = go && mlen >= 8u && mlen < 16u && qd < 2u;
  const bool w4 = go && mlen < 8u && qd < 2u;
  const uint32_t po16 = (qd == np16 - 1u) ? mlen - 16u : 16u * qd;
  const uint32_t po8 = qd ? mlen - 8u : 0u, po4 = qd ? mlen - 4u : 0u;
  BAMD_MEM_SYNC();
  // one scalar base + a 32-bit lane offset per load (lanes without a piece read the first bytes of the stream's own output:
  // lz4_bulk only runs when the output area has at least 16 bytes)
  q.v16 = g_ld16(out + (w16 ? so + po16 : 0u));
  q.v8 = g_ld8(out + (w8 ? so + po8 : 0u));
  q.v4 = g_ld4(out + (w4 ? so + po4 : 0u));
}

// A match of a step that has to run in stream order (it reads bytes this or the previous step produced, or is longer than 64
// bytes; at most 273 long: the batched parse takes one length-extension byte).  wave_match_copy's semantics (fastcopy.c:530-639)
// in a form that needs a handful of registers - its 4 KiB rows would push the next step's in-flight pieces out of the
// register file: a short period is replicated from registers, a source that ends in front of the match is fetched with all
// its loads in one round trip, anything else moves 64 bytes per round trip.
__device__ __forceinline__ void bulk_match_copy(gu8* out, uint32_t pos, uint32_t off, uint32_t len, int lane) {
  BAMD_MEM_SYNC();
  if (off < 64u && off < len) {
    const uint32_t pat = ((uint32_t)lane < off) ? out[pos - off + (uint32_t)lane] : 0u;
    const uint32_t M = 65536u / off + 1u;                 // floor(i / off) == (i * M) >> 16 for i < 64
    const uint32_t G = ((64u * M) >> 16) * off;           // whole periods per 64 lanes
    const uint32_t i_mod = (uint32_t)lane - (((uint32_t)lane * M) >> 16) * off;
    const uint32_t val = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i_mod << 2), (int)pat);
    for (uint32_t done = 0; done < len; done += G) {
      const uint32_t chunk = len - done < G ? len - done : G;
      if ((uint32_t)lane < chunk) out[pos + done + (uint32_t)lane] = (uint8_t)val;
    }
    return;
  }
  if (off >= len) {                                        // disjoint: <= 5 byte rows, all loads first
    uint32_t v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { const uint32_t i = 64u * (uint32_t)k + (uint32_t)lane; v[k] = i < len ? (uint32_t)out[pos - off + i] : 0u; }
#pragma unroll
    for (int k = 0; k < 5; k++) { const uint32_t i = 64u * (uint32_t)k + (uint32_t)lane; if (i < len) out[pos + i] = (uint8_t)v[k]; }
    return;
  }
  for (uint32_t done = 0; done < len; done += 64u) {       // 64 <= off < len: every 64-byte row reads what earlier rows wrote
    const uint32_t i = done + (uint32_t)lane;
    if (i < len) out[pos + i] = out[pos + i - off];
    BAMD_MEM_SYNC();
  }
}

// literals, the pieces loaded by bulk_issue_loads, then the matches that must run in stream order
__device__ __forceinline__ void bulk_execute(const BulkStep& q, gu8* out, volatile __attribute__((address_space(3))) uint32_t* scr, uint32_t op, int lane) {
  const uint32_t ll_r = q.pk & 0x1ffu, ml_r = (q.pk >> 9) & 0x1ffu, ext_r = (q.pk >> 18) & 1u;
  // ONE wait for the step's three pieces, here, where the only younger operations are the next step's loads (an exact count).
  // Left to their first uses the waits would sit behind the conditional stores below, and the compiler, which must assume a
  // conditional store was skipped, would count too few younger operations - i.e. wait for the next step's loads as well.
#ifndef BAMD_WAVE_EMU
  asm volatile("; pieces ready" ::"v"(q.v16.x), "v"(q.v16.y), "v"(q.v16.z), "v"(q.v16.w), "v"(q.v8), "v"(q.v4));
#endif
  scr[lane] = 0u;
  BAMD_LDS_SYNC();
  if ((uint32_t)lane < q.cnt) scr[q.c] = 0x80000000u | q.excl | (ll_r << 16) | (ext_r << 25);
  BAMD_LDS_SYNC();
  const uint64_t mask = __ballot(scr[lane] >> 31);
  {
    const uint64_t below = mask & ((2ull << lane) - 1ull);
    const uint32_t s = 63u - (uint32_t)__builtin_clzll(below | 1ull);
    const uint32_t inf = scr[s];
    const uint32_t k = (uint32_t)lane - s - 1u - ((inf >> 25) & 1u);
    if ((uint32_t)lane < q.consumed && (uint32_t)lane > s + ((inf >> 25) & 1u) && k < ((inf >> 16) & 0x1ffu)) out[op + (inf & 0xffffu) + k] = (uint8_t)q.B;
  }
  {
    const uint32_t qd = (uint32_t)lane & 3u;
    const uint32_t mlen = q.fA & 0x1ffu;
    const bool go = (q.fA & 0x200u) != 0u;
    const uint32_t dof = op + (q.fA >> 10);
    const uint32_t np16 = (mlen + 15u) >> 4;
    const bool w16 = go && mlen >= 16u && qd < np16;
    const bool w8 = go && mlen >= 8u && mlen < 16u && qd < 2u;
    const bool w4 = go && mlen < 8u && qd < 2u;
    const uint32_t po16 = (qd == np16 - 1u) ? mlen - 16u : 16u * qd;
    const uint32_t po8 = qd ? mlen - 8u : 0u, po4 = qd ? mlen - 4u : 0u;
    if (w16) g_st16(out + (dof + po16), q.v16);
    if (w8) *(BAMD_GAS u64una*)(out + (dof + po8)) = q.v8;
    if (w4) g_st4(out + (dof + po4), q.v4);
  }
  uint32_t rest = q.restmask;
  if (rest) {
    do {
      const int sl = __builtin_ctz(rest);
      rest &= rest - 1u;
      const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)ml_r, sl);
      const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)q.off_r, sl);
      const uint32_t mr = (uint32_t)__builtin_amdgcn_readlane((int)(q.excl + ll_r), sl);
      bulk_match_copy(out, op + mr, o, m, lane);
    } while (rest);
    // Nothing of these copies may stay "possibly in flight" for the compiler's bookkeeping: their loads sit behind correlated
    // exec tests it cannot follow, and the next reuse of one of their registers would get a vmcnt(0) on the MAIN path - in front of
    // every parse, whether or not a copy ran.  (The next step's pieces are older than these loads, so this wait costs nothing more.)
    __builtin_amdgcn_s_waitcnt(0);
  }
}

// Loop state (all wave-uniform except wpend).
struct BulkState {
  uint32_t ip, op;              // the current step's positions
  uint32_t ip1, op1;            // the next step's
  uint32_t wb, hi_blk;          // ring base; highest block loaded or on its way
  uint32_t wpend_blk;           // the block loaded in the previous iteration, to be written to the ring in this one ...
  bool pend;                    // ... if it was a new one
};

// Half an iteration: parse the step behind `cur` into `nx`, issue its loads, THEN store `cur`.  No branch leaves it: a step the
// loop cannot take becomes an EMPTY record (no sequences, no bytes, dummy loads) and the caller ends the loop behind the whole
// iteration.  With exits in the middle the compiler funnels them through one latch block, and the two step records - whose
// piece registers are still being loaded - get copied there: a copy is a use, a use is a wait.  The caller alternates two
// BulkStep objects for the same reason instead of copying nx to cur.
__device__ __forceinline__ bool bulk_half(BulkState& st, const BulkStep& cur, BulkStep& nx, uint32_t& wload, const uint32_t wstore, const gu8* in, uint32_t n, gu8* out,
                                          uint32_t cap, uint32_t span_hi, volatile __attribute__((address_space(3))) uint32_t* scr, lds_vu32* ring32,
                                          const lds_vu8* ring, int lane PROF_ARG) {
  const bool took = bulk_parse(nx, bulk_bytes(ring, st.wb, st.ip1, lane), st.ip1 + 72u <= n, st.ip1, st.op1, cap, n, cur.acc, span_hi, lane);
  const uint32_t ip2 = st.ip1 + nx.consumed;
  // The block written to the ring at the end of THIS half was loaded in the previous one; a block loaded now is readable from
  // the parse after next, which starts at most 64 bytes behind ip2 and reads 64 bytes: keep everything up to ip2 + 131.
  const bool want = ((ip2 + 131u - st.wb) >> 8) > st.hi_blk;
  // in FRONT of the step's match loads, and in every half (the newest block again when no new one is due): a conditional load
  // would make the compiler's count of younger operations a minimum (see bulk_execute)
  wload = bulk_block_load(in, n, st.wb, st.hi_blk + (want ? 1u : 0u), lane);
  bulk_issue_loads(nx, out, st.op1, lane);
  bulk_execute(cur, out, scr, st.op, lane);
  if (st.pend) { bulk_block_store(ring32, st.wpend_blk, wstore, lane); BAMD_LDS_SYNC(); }
  st.pend = want; st.wpend_blk = st.hi_blk + 1u;
  if (want) st.hi_blk++;
  PROF_ADD(0, cur.cnt ? 1 : 0); PROF_ADD(1, cur.cnt); PROF_ADD(2, __builtin_popcount(cur.restmask));
#ifdef BAMD_WAVE_EMU
  if (lane == 0 && cur.cnt) g_emu_bulk_steps++;
#endif
  st.ip = st.ip1; st.op = st.op1;
  st.ip1 = ip2; st.op1 = st.op + nx.acc;
  return took;
}

// Runs steps from (ip, op) while they are plain; returns the number of steps taken, ip / op behind the last one.
// The caller's register window (wave_prims.h: Window, sought to ip: ip - w.base < 256) IS the ring's first two blocks on the way
// in, and is re-made from the ring on the way out: entering and leaving cost LDS traffic, no memory round trip.  (The first
// version filled the ring with two loads of its own and let the caller re-seek afterwards; with a real call around it that was
// three serialised round trips per visit - 40 000 cycles - and streams that alternate between a plain step and a long match got
// slower than without the loop: profiles/r03b_dec_phase_bulk_v1.txt.)
// LDS: scr[0..64) for the literal scatter as in lz4_batch_step; the ring lives in the first 512 bytes of the step buffer behind it
// (lz4_step_lds is never active at the same time: a step that wants it ends this loop).
__device__ __forceinline__ uint32_t lz4_bulk(Window& w, uint32_t n_, gu8* out_, uint32_t cap_, volatile uint32_t* scr_generic,
                                            uint32_t& ip_, uint32_t& op_, uint32_t span_hi_, int lane PROF_ARG) {
  volatile __attribute__((address_space(3))) uint32_t* scr = (volatile __attribute__((address_space(3))) uint32_t*)scr_generic;
  lds_vu32* ring32 = (lds_vu32*)(scr + 64);
  const lds_vu8* ring = (const lds_vu8*)(scr + 64);
  // everything that steers the loop lives in SGPRs
  const gu8* in = uni_ptr(w.in);
  gu8* out = uni_ptr(out_);
  const uint32_t n = uni(n_), cap = uni(cap_), span_hi = uni(span_hi_);
  BulkState st;
  st.ip = uni(ip_); st.op = uni(op_);
  if (st.ip + 72u > n || cap < 16u) return 0u;
  const uint32_t ip0 = st.ip;
  // ---- the ring takes over the window: blocks 0 and 1 at wb = w.base ----
  st.wb = uni(w.base);
  bulk_block_store(ring32, 0u, w.lo, lane); bulk_block_store(ring32, 1u, w.hi, lane);
  st.hi_blk = 1u; st.pend = false; st.wpend_blk = 0u;
  uint32_t wa = 0u, wb_ = 0u;            // the two block registers: a half loads one and stores the other (no copies, see bulk_half)
  BAMD_LDS_SYNC();
  // Two step records used alternately.  The loop is entered with an EMPTY current step, so that each record's loads are
  // issued at exactly one place in the code (a first step parsed and loaded in front of the loop gave the piece registers two
  // definitions each, joined by copies at the loop's back edge).
  BulkStep a, b;
  b.B = 0u; b.c = 0u; b.pk = 0u; b.off_r = 0u; b.excl = 0u; b.cnt = 0u; b.consumed = 0u; b.acc = 0u; b.restmask = 0u; b.fA = 0u; b.fB = 0u;
  b.v16 = make_uint4(0u, 0u, 0u, 0u); b.v8 = 0u; b.v4 = 0u;
  a = b;
  st.ip1 = st.ip; st.op1 = st.op;
  uint32_t steps = 0, nrest = 0;
  bool go;
  do {
    const bool t1 = bulk_half(st, b, a, wb_, wa, in, n, out, cap, span_hi, scr, ring32, ring, lane PROF_PASS);      // stores b, parses a
    const bool t2 = bulk_half(st, a, b, wa, wb_, in, n, out, cap, span_hi, scr, ring32, ring, lane PROF_PASS);      // stores a, parses b
    steps += (t1 ? 1u : 0u) + (t2 ? 1u : 0u);
    nrest += (a.restmask ? 1u : 0u) + (b.restmask ? 1u : 0u);
    go = t1 && t2;
  } while (go);
  // b is empty here: if a was not taken, the same step was offered to b and refused for the same reason
  if (b.cnt) bulk_execute(b, out, scr, st.op, lane);
  // ---- hand the window back: the ring's blocks j, j + 1 around the position reached ----
  if (st.pend) { bulk_block_store(ring32, st.wpend_blk, wa, lane); }       // the block the last half loaded
  BAMD_LDS_SYNC();
  {
    const uint32_t j = (st.ip1 - st.wb) >> 8;                               // hi_blk - 1 or hi_blk (bulk_half keeps ip + 131 covered)
    w.base = st.wb + 256u * j;
    w.lo = ring32[64u * (j & 1u) + (uint32_t)lane];
    if (j < st.hi_blk) w.hi = ring32[64u * ((j + 1u) & 1u) + (uint32_t)lane];
    else w.hi = w.fetch(w.base + 256u);
  }
  ip_ = st.ip1; op_ = st.op1;
  (void)ip0;
  PROF_LAP(10);
  // bit 31: most steps had matches to run in stream order (long ones, or chains through the step's own output) - those wait for
  // everything in flight, the pipeline buys nothing there and the caller should stay with the round-2 step for a while
  return steps | ((2u * nrest > steps) ? 0x80000000u : 0u);
}

}  // namespace bamd
