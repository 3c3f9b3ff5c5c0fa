// enc_lz4p.h — the LZ4 stream encoder of one wavefront with a PARALLEL parse and emit (round 6; included by k_encode.hip inside namespace bamd,
// behind enc_lz.h whose window, table and helper functions it uses).
//
// Replaces  LZ4_compress_fast  (lz4.c:1453 -> LZ4_compress_generic_validated :930-1338) for one stream, like lz_encode_wave<EF_LZ4> (enc_lz.h),
// which it supersedes for LZ4 (BAMD_ENC_PAR=0 brings the old loop back for A/B runs; BloscLZ and the Zstd / zlib front ends still use it).
//
// Why: lz_encode_wave looks at 64 positions per step in parallel, but then SELECTS AND EMITS ONE SEQUENCE AT A TIME: wave-wide maximum, readlanes,
// extension loads, a store, re-selection behind the match - about 160 wave-instructions and two to three dependent round trips per sequence, 2.3
// sequences per step on bench19's noisy planes (profiles/r05a_enc_phase_t8.txt: 6.3 k of a step's 10.3 k cycles; the CU's scalar unit 60 % busy).
// Here the step's whole greedy parse is resolved at once:
//   * every lane packs (length - lane | lane | length) of its candidate into one word; a SUFFIX maximum over the lanes (4 DPP row shifts + the three
//     row totals) tells every position "who wins if the search starts here" - the selection rule of lz_encode_wave, for all start positions at once;
//   * "start -> end of the winner's match" is a next-pointer per lane; the chain from the step's first free position is resolved by pointer
//     doubling in rank space (7 ds_bpermute, the same trick as the decoder's token chain, dec_ring.h): rank lane r holds the r-th sequence;
//   * backward extensions come out of the four bytes fetched in front of every candidate (the sequences whose four all match look at eight more, one
//     round trip for all of them: BAMD_ENC_BACK2), sizes are prefix-summed over the rank lanes (one DPP row scan), and ALL sequences of the step leave
//     together: one byte store for every literal position of the step, one for the tokens, one 2-byte store for the offsets;
//   * only a match that fills all RANK_CAP compared bytes needs memory (its forward extension): the chain stops behind it and picks up again at
//     its true end - one extra round trip for that sequence, as before.
// The parse is the old one's (same ranking, same table insertions) with two simplifications that cost a fraction of a per cent of ratio:
// backward extension stops at twelve bytes (the old loop went on byte by byte), and a search never STARTS at the step's last probing lane (lane 63 is
// the chain's absorbing stop).  Both loops of the step begin by making the lane number opaque to the compiler (BAMD_ENC_LAUNDER): hoisted out of them, the
// two dozen values it derives from the lane number filled the registers and the step's window or a per-lane address lived in scratch memory.
// What round 6 measured and did not keep is listed where it would have gone (extension under the emission, window ahead of a jump) and in HISTORY.md
// (the lanes' own look beyond RANK_CAP bytes, an LDS-staged output).
#ifndef BAMD_ENC_PAR
#define BAMD_ENC_PAR 1
#endif
constexpr int ENC_SCR_BYTES = 512;     // 128 dwords behind the hash table: sequence info on its way from rank lanes to byte lanes (one word per position of a step)
static_assert(ENC_LZ_LDS_WAVES == (160 * 1024) / (ENC_TAB_BYTES + ENC_SCR_BYTES), "enc_lz.h sizes the persistent grid with this scratch in mind");
#ifndef BAMD_ENC_BACK2
#define BAMD_ENC_BACK2 1     // sequences whose four bytes in front all match look at eight more (one more memory round trip for the steps that hold such a sequence)
#endif
#ifndef BAMD_ENC_SKIPCAP
#define BAMD_ENC_SKIPCAP 32u    // the longest distance between the starts of two failed steps, in 64-byte units.  16 until late in round 6; 32 / 64 (profiles/r06zm_*): random bytes
                                // - 11 / - 16 %, random-walk planes - 5 / - 9 %, everything else 0 / - 1 %, no ratio of the test data changes by more than 1.5 %; but noise with islands
                                // of 0.5 - 2 KiB of repeated content (1.03 here at 16, 1.13 in the reference: a step enters only what it probes) keeps its 1.01 - 1.08 at 32 and loses all of it at 64
#endif
#ifndef BAMD_ENC_NEIGHBOUR
#define BAMD_ENC_NEIGHBOUR 4u   // a power of two, or 0: never
#endif
#ifndef BAMD_ENC_KEEPJ
#define BAMD_ENC_KEEPJ 0     // the doubled next-pointers made once per step (three registers across the chain loop) instead of once per chain
#endif
#ifndef BAMD_ENC_EXT2
#define BAMD_ENC_EXT2 1      // long comparisons two rows per trip (wave_common_fwd, enc_lz.h) - 0: one row per trip (wave_common_fwd_lite)
#endif
#ifndef BAMD_ENC_LAUNDER
#define BAMD_ENC_LAUNDER 1   // the lane number opaque at the top of the step and of the chain loop (0: the compiler hoists what it derives from it, and spills)
#endif
#ifndef BAMD_ENC_PREFULL
#define BAMD_ENC_PREFULL 1   // a step that begins with pending literals (behind skipped, match-less steps) extends its first match backwards into them, up to 64 bytes
#endif

template <int N> __device__ __forceinline__ uint32_t dpp_row_shl0(uint32_t v) {   // lane i <- lane i+N of its 16-lane row, 0 outside
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// wave_common_fwd (enc_lz.h) with one row per trip instead of two behind the first 256 bytes: 8 registers in flight instead of 16 - inside the parse of
// lz4_encode_wave_par the rank lanes' state is live around it, and the long comparisons (a run plane's 4 KiB, a 96 KiB period) are few per stream.
// Same contract: leading equal bytes of src[a..] and src[b..], at most maxlen, a > b, never reads at or beyond src + n; uniform arguments and result.
__device__ __forceinline__ uint32_t wave_common_fwd_lite(const gu8* src, uint32_t n, uint32_t a, uint32_t b, uint32_t maxlen, int lane) {
  uint32_t done = 0;
  if (maxlen && a + 256u <= n) {
    const uint32_t x = g_ld4(src + a + 4 * lane) ^ g_ld4(src + b + 4 * lane);
    const uint32_t q = 4u * (uint32_t)lane;
    uint32_t e0 = x ? (uint32_t)(__builtin_ctz(x) >> 3) : 4u;
    const uint32_t r0 = q < maxlen ? maxlen - q : 0u;
    if (e0 > r0) e0 = r0;
    const uint64_t s0 = __ballot(e0 < 4u);
    if (s0) { const int f = __builtin_ctzll(s0); return 4u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)e0, f); }
    done = 256u;
  }
  while (done < maxlen && a + done + 1024u <= n) {
    const uint4 x0 = g_ld16(src + a + done + 16 * lane), y0 = g_ld16(src + b + done + 16 * lane);
    const uint32_t q1 = done + 16u * (uint32_t)lane;
    uint32_t e1 = common16(x0, y0);
    const uint32_t r1 = q1 < maxlen ? maxlen - q1 : 0u;
    if (e1 > r1) e1 = r1;
    const uint64_t s1 = __ballot(e1 < 16u);
    if (s1) { const int f = __builtin_ctzll(s1); return done + 16u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)e1, f); }
    done += 1024u;
  }
  while (done < maxlen) {                                  // the last < 1 KiB of a stream: 8 bytes per lane, byte-safe at the very end
    const uint32_t q = done + 8u * (uint32_t)lane;
    uint32_t vb = 0;
    if (q < maxlen) vb = (maxlen - q < 8u) ? maxlen - q : 8u;
    uint32_t eq = 0;
    if (vb) {
      if (a + q + 8u <= n) {
        const uint64_t x = ld8u(src + a + q) ^ ld8u(src + b + q);
        eq = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
        if (eq > vb) eq = vb;
      } else {
        while (eq < vb && src[a + q + eq] == src[b + q + eq]) eq++;
      }
    }
    const uint64_t stop = __ballot(eq < 8u);
    if (stop) { const int f = __builtin_ctzll(stop); return done + 8u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)eq, f); }
    done += 512u;
  }
  return maxlen;
}

// (Round 6 also asked for the window of a step that starts behind a jump - after a long match, or in the long strides through data that does not match - a
//  step ahead, where the jump is known: profiles/r06z_* show 4 - 5 k of a match-less step's 5.7 k cycles in front of its window, and the position a match-less
//  step goes on to depends only on the count of misses.  3 - 6 % SLOWER on bench19, random-walk data and typesize 2 alike; only pure noise gained 2 %:
//  profiles/r06za_*.  Not kept.)
// SS = log2 of the PROBE STRIDE (round 6).  SS = 1: a step covers 128 positions; lane l probes position ip + 2 l only, but BOTH positions of a lane enter the
// table (so a repeat is found whatever the parity of its distance) and both bytes of a lane leave as literals.  A match that begins on an odd position
// is found one byte late and gets its first byte back from the backward extension (the four bytes in front of every candidate are there anyway); its
// ranked length counts that byte.  Half the steps - probes, candidate fetches, parses, stores - for the same input; the reference trades positions for
// speed the same way (LZ4_compress_fast's acceleration = 10 - clevel, blosc/blosc.c:577-587, lz4.c:1044-1053: at clevel 5 its search advances five
// bytes at a time from the first miss on).
template <int SS>
__device__ uint32_t lz4_encode_wave_par(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap,
                                        int clevel, enc_entry_t* tab_generic, int lane_in EPROF_ARG) {
  static_assert(SS == 0 || SS == 1, "one or two positions per lane");
  const int lane = lane_in;
  constexpr uint32_t SPAN = 64u << SS;            // positions of one step
  EncTable tab;
  tab.init((void*)tab_generic);
  volatile BAMD_LAS uint32_t* scr = (volatile BAMD_LAS uint32_t*)((BAMD_LAS uint8_t*)(void*)tab_generic + ENC_TAB_BYTES);
  if (n < 13u) return 0u;                                     // lz4.c:245-246, :963-964
  const uint32_t last_start = n - 12u;                        // inclusive bound on match starts
  const uint32_t mlimit = n - 5u;                             // matches end at or before this position
  const int accel = 10 - clevel;                              // blosc/blosc.c:577-587
  const uint32_t minlen = clevel >= 9 ? 4u : (clevel >= 6 ? 5u : 7u);      // (enc_lz.h: the effort knob)
  tab.clear(lane);
  EncWindow win;
  win.init(src, n, lane);
  uint32_t ip = 0, anchor = 0, op = 0, nfail = 0;
  bool ins_pending = false;                       // position ip-2 still has to enter the table (lz4.c:1236-1242)
  while (ip <= last_start) {
    // The lane number of this iteration, opaque to the compiler: it hoists everything a step derives from the lane number out of the loop (two dozen
    // masks, shifted copies, comparisons and per-lane pointers), and at 80 registers the loop then keeps its input WINDOW in scratch memory - reloaded at the
    // top of every step behind s_waitcnt vmcnt(0), i.e. behind the stores of the step before.  Recomputing them costs a few VALU instructions per step.
    int lane = lane_in;
#if !defined(BAMD_WAVE_EMU) && BAMD_ENC_LAUNDER
    asm volatile("; lane of this step" : "+v"(lane));
#endif
    const uint32_t lq = (uint32_t)lane << SS;     // this lane's (first) position, relative to ip
    const uint32_t p = ip + lq;
    const bool live = p <= last_start;
    const bool live1 = SS && p + 1u <= last_start;                // (SS = 1) the lane's second position: entered into the table, never probed
    // ---- round 1 (registers + LDS only): own bytes from the window, table probe (as in lz_encode_wave) ----
    win.seek(ip, lane);
    const uint32_t lo = ip >= 4u ? ip - 4u : 0u;
    const uint32_t rb = lo & ~3u;
    const uint32_t D = (rb - win.wbase) >> 2;
    const int gsel = (int)((D + (uint32_t)lane) << 2);
    const uint32_t ra = (uint32_t)__builtin_amdgcn_ds_bpermute(gsel, (int)win.w0);
    const uint32_t rc = (uint32_t)__builtin_amdgcn_ds_bpermute(gsel, (int)win.w1);
    const uint32_t r = (D + (uint32_t)lane < 64u) ? ra : rc;
    const uint32_t bo0 = ip - rb;
    const uint32_t bo = bo0 + lq;                 // (SS = 1: the last lane's bytes end 7 + 126 + 24 bytes behind rb - inside r's 256 and the window's 348)
    const int ksel = (int)((bo >> 2) << 2);
    const uint32_t x0 = (uint32_t)__builtin_amdgcn_ds_bpermute(ksel, (int)r);
    const uint32_t x1 = (uint32_t)__builtin_amdgcn_ds_bpermute(ksel + 4, (int)r);
    const uint32_t x2 = (uint32_t)__builtin_amdgcn_ds_bpermute(ksel + 8, (int)r);
    const uint32_t x3 = (uint32_t)__builtin_amdgcn_ds_bpermute(ksel + 12, (int)r);
    const uint32_t x4 = (uint32_t)__builtin_amdgcn_ds_bpermute(ksel + 16, (int)r);
    const uint32_t x5 = (uint32_t)__builtin_amdgcn_ds_bpermute(ksel + 20, (int)r);
    const uint32_t sh = bo & 3u;
    const uint32_t o0 = __builtin_amdgcn_alignbyte(x1, x0, sh), o1 = __builtin_amdgcn_alignbyte(x2, x1, sh);
    const uint32_t o2 = __builtin_amdgcn_alignbyte(x3, x2, sh), o3 = __builtin_amdgcn_alignbyte(x4, x3, sh);
    const uint32_t o4 = __builtin_amdgcn_alignbyte(x5, x4, sh);
    const uint32_t ownpre = __builtin_amdgcn_alignbyte(x0, (uint32_t)__builtin_amdgcn_ds_bpermute(ksel - 4, (int)r), sh);   // src[p-4 .. p-1] (p >= 4)
    Bytes20 own;
    own.a = ((uint64_t)o1 << 32) | o0; own.b = ((uint64_t)o3 << 32) | o2; own.c = RANK_CAP > 16u ? o4 : 0u;
    const uint64_t r01 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)r, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)r, 0);
    const uint32_t before2 = bo0 >= 2u ? (uint32_t)(r01 >> (8u * (bo0 - 2u))) & 0xffffu : 0u;   // src[ip-2] | src[ip-1] << 8
    uint32_t prev;
    if (SS) prev = ownpre >> 24;                   // src[p-1] whenever p >= 1 (the alignment shift of a lane with p < 4 leaves that byte in x0's part)
    else prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(o0 & 0xffu), 0x138, 0xf, 0xf, false);   // wave_shr:1
    if (lane == 0) prev = ip ? (bo0 >= 2u ? before2 >> 8 : (uint32_t)(r01 >> (8u * (bo0 - 1u))) & 0xffu) : 0x100u;
    if (ins_pending) {
      const uint32_t m2 = enc_mix(before2 | (o0 << 16));
      if (lane == 0) tab.put(enc_slot(m2), enc_entry(m2, ip - 2u));
      ins_pending = false;
    }
    uint32_t cand = 0, limit = 0;
    bool tab_ok = false;
    const uint32_t mix = enc_mix(o0);              // (slot and entry are made from it again where the position enters the table: one register across the step instead of two)
#define PAR_PUT() do { const uint32_t mix_ = enc_mix((uint32_t)own.a); tab.put(enc_slot(mix_), enc_entry(mix_, p)); } while (0)      // (made from the bytes again where a position enters the table: nothing but own.a lives across the step)
#define PAR_PUT1() do { const uint32_t mix1_ = enc_mix((uint32_t)(own.a >> 8)); tab.put(enc_slot(mix1_), enc_entry(mix1_, p + 1u)); } while (0)
    if (live) {
      limit = mlimit - p; if (limit > RANK_CAP) limit = RANK_CAP;
      const uint32_t h = enc_slot(mix), mine = enc_entry(mix, p);
      const uint32_t e = tab.get(h);
      const uint32_t d = (p - e) & 0xffffu;
      if (d != 0u && d <= p && EncTable::tag_equal(e, mine)) { cand = p - d; tab_ok = true; }
    } else {
      prev = 0x100u;
    }
    PROF_LAP(8); PROF_ADD(0, 1);
    // ---- round 2: candidate bytes, exact lengths up to RANK_CAP; nb = equal bytes right in front of the two positions, 0 .. 4 ----
    uint32_t len = 0, nb = 0;
    if (tab_ok) {
      const uint32_t cpre = cand >= 4u ? ld4u(src + cand - 4u) : 0u;
      const Bytes20 cb = load20(src, cand, n);
      if (cand >= 4u) { const uint32_t x = cpre ^ ownpre; nb = x ? (uint32_t)__builtin_clz(x) >> 3 : 4u; }
      len = common20(own, cb);
      if (len > limit) len = limit;
      if (len < 4u || len + (SS && nb ? 1u : 0u) < minlen) len = 0;      // (SS = 1: a match found one byte late counts the byte the backward extension brings back - but the four
                                                                           //  bytes LZ4 asks for, lz4.c:240 MINMATCH, must be there without it: the extension may have no room)
    }
    if (live && prev < 0x100u) {                    // distance 1: run of the previous byte
      uint32_t rl = runlen20(own, prev);
      if (rl > limit) rl = limit;
      // (the bytes in front: src[p-1-i] against src[p-2-i], i.e. how far the run reaches back - three comparisons inside ownpre)
      const uint32_t xr = (ownpre ^ (ownpre << 8)) | 0xffu;
      const uint32_t nbr = p >= 5u ? (uint32_t)__builtin_clz(xr) >> 3 : 0u;
      if (rl >= 4u && rl + (SS && nbr ? 1u : 0u) >= minlen && rl > len) { len = rl; cand = p - 1u; nb = nbr; }
    }
    win.settle(lane);
    PROF_LAP(9);
    // ---- the parse of the whole step ----
    const uint32_t step_end = ip + SPAN;
    // what the rank lanes fetch from a winner: the DISTANCE to its candidate (1 .. 65535: the table holds 16-bit positions) | equal bytes in front << 28.  (The
    // candidate's position itself travelled here until a stream of 266 MiB - forced blocksize, BLOSC_SPLITMODE=NEVER - came out unreadable: positions beyond 2^28 ran
    // into the count's bits.  scripts/dbg_big_stream.py, tests/test_gpu_compress.py::test_one_lz4_stream_beyond_256_mib.)
    const uint32_t cn = (p - cand) | (nb << 28);
    // S[l] = the best key among the lanes at or above l: who wins when the search starts at l
    uint32_t S = len ? (((len + (SS && nb ? 1u : 0u) + SPAN - lq) << 11) | ((63u - (uint32_t)lane) << 5) | len) : 0u;
    S = umax32(S, dpp_row_shl0<1>(S)); S = umax32(S, dpp_row_shl0<2>(S)); S = umax32(S, dpp_row_shl0<4>(S)); S = umax32(S, dpp_row_shl0<8>(S));
    {
      const uint32_t t3 = (uint32_t)__builtin_amdgcn_readlane((int)S, 48);
      const uint32_t t2 = umax32((uint32_t)__builtin_amdgcn_readlane((int)S, 32), t3);
      const uint32_t t1 = umax32((uint32_t)__builtin_amdgcn_readlane((int)S, 16), t2);
      const uint32_t add = lane < 16 ? t1 : (lane < 32 ? t2 : (lane < 48 ? t3 : 0u));
      S = umax32(S, add);
    }
    // next-pointer: the end of the winner's match; 63 = stop (no candidate left, the match leaves the step, or it needs its forward extension
    // first).  Lane 63 is the absorbing stop: its own entry is 63 whatever it holds.
    uint32_t J0;
    {
      const uint32_t wS = 63u - ((S >> 5) & 63u), LS = S & 31u;
      const uint32_t nx = ((wS << SS) + LS + (SS ? 1u : 0u)) >> SS;                // the first probing lane at or behind the match's end
      J0 = (S == 0u || LS >= RANK_CAP || nx > 63u) ? 63u : nx;
    }
#if BAMD_ENC_KEEPJ
    const uint32_t J1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(J0 << 2), (int)J0);
    const uint32_t J2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(J1 << 2), (int)J1);
    const uint32_t J3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(J2 << 2), (int)J2);
#endif
    uint32_t lane_lo = 0;                         // first POSITION of the step (relative to ip) not covered by a sequence emitted so far
    bool any = false, tail_open = false;
    for (;;) {
#if !defined(BAMD_WAVE_EMU) && BAMD_ENC_LAUNDER
      // (again for the chain: hoisted out of THIS loop, the per-lane address of the extension's first loads was spilled at the top of every step and
      //  reloaded - behind s_waitcnt vmcnt(0) - in front of every extension: 3.4 GB of scratch writes per 8 GiB, profiles/r06v_*)
      asm volatile("; lane of this chain" : "+v"(lane));
#endif
      // rank lane r (< 16): c = start of the r-th search of the chain that begins at lane_lo
      // (the doubled pointers are made here, not once per step: they would be live across everything below, and nine steps in ten run this once)
#if !BAMD_ENC_KEEPJ
      const uint32_t J1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(J0 << 2), (int)J0);
      const uint32_t J2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(J1 << 2), (int)J1);
      const uint32_t J3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(J2 << 2), (int)J2);
#endif
      uint32_t c = umin32((lane_lo + (SS ? 1u : 0u)) >> SS, 63u);                 // (lanes: the first probe at or behind lane_lo; behind the last probe: the stop lane)
      { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)J0); c = (lane & 1) ? t : c; }
      { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)J1); c = (lane & 2) ? t : c; }
      { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)J2); c = (lane & 4) ? t : c; }
      { const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)J3); c = (lane & 8) ? t : c; }
      const uint32_t Sr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)S);
      const bool valid = lane < 16 && c < 63u && Sr != 0u;
      const uint32_t nseq = (uint32_t)__builtin_popcountll(__ballot(valid));      // valid ranks are 0 .. nseq - 1 (the stop is absorbing)
      if (nseq == 0u) { tail_open = any; break; }
      any = true;
      const uint32_t last = nseq - 1u;
      const uint32_t w = 63u - ((Sr >> 5) & 63u), L = Sr & 31u;                    // winner lane and its ranked length
      const uint32_t wq = w << SS;                                                  // the winner's position
      const uint32_t cnr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(w << 2), (int)cn);
      const uint32_t dist_r = cnr & 0xffffu, nb_r = cnr >> 28;
      const uint32_t cand_r = ip + wq - dist_r;                                     // the candidate's position
      // cq: the POSITION the r-th search starts at = the end of the match in front of it (every match of the chain but the last has its exact length)
      uint32_t cq = c;
      if (SS) { const uint32_t e_prev = dpp_row_shr0<1>(wq + L); cq = lane == 0 ? lane_lo : e_prev; }
      const uint32_t room = wq - cq;                                                // positions between the search start and the match
      // the last sequence of the chain may need its forward extension (all RANK_CAP bytes equal)
      const uint32_t w_last = (uint32_t)__builtin_amdgcn_readlane((int)wq, (int)last), L_last = (uint32_t)__builtin_amdgcn_readlane((int)L, (int)last);
      const uint32_t pm_last = ip + w_last;
      uint32_t mext = 0;
      if (L_last >= RANK_CAP && pm_last + RANK_CAP < mlimit) {
        const uint32_t c_last = (uint32_t)__builtin_amdgcn_readlane((int)cand_r, (int)last);
        mext = BAMD_ENC_EXT2 ? wave_common_fwd(src, n, pm_last + RANK_CAP, c_last + RANK_CAP, mlimit - (pm_last + RANK_CAP), lane)
                              : wave_common_fwd_lite(src, n, pm_last + RANK_CAP, c_last + RANK_CAP, mlimit - (pm_last + RANK_CAP), lane);
      }
      // (Round 6 also asked for the first 256 bytes of this extension up here and looked at them behind the step's emission - only the last match's length bytes need
      //  them, and they close the step's output: 4 % SLOWER on every data set, profiles/r06e_enc_ab_extension_under_emission_rejected.txt; again at 80 registers, without the spill of r06v in front of every extension: no difference, r06y.)
      uint32_t back = valid ? umin32(nb_r, umin32(room, cand_r)) : 0u;
      if (BAMD_ENC_BACK2) {
        // all four bytes in front equal and room for more: eight more bytes per such sequence, all of them in one round trip (bench19's noisy planes: 5 % of
        // the sequences; without this their streams are 2.2 % larger, with it 0.1 %)
        const bool more = valid && back == 4u && room > 4u && cand_r >= 12u;
        if (__ballot(more)) {                                                       // uniform
          const uint32_t pm = ip + wq;
          const uint64_t a = g_ld8(src + (more ? pm - 12u : 0u)), b = g_ld8(src + (more ? cand_r - 12u : 0u));
          const uint64_t x = a ^ b;
          const uint32_t eq = x ? (uint32_t)__builtin_clzll(x) >> 3 : 8u;
          if (more) back += umin32(eq, room - 4u);
        }
      }
      uint32_t pre = lane_lo == 0u ? ip - anchor : 0u;                              // literals in front of the step (rank 0 only; anchor <= ip there)
      uint32_t pre_ext = 0;                                                         // rank 0's match extended backwards into them
      if (BAMD_ENC_PREFULL && pre) {                                                // uniform, rare
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)back, 0), room0 = (uint32_t)__builtin_amdgcn_readlane((int)room, 0);
        const uint32_t cand0 = (uint32_t)__builtin_amdgcn_readlane((int)cand_r, 0);
        if (b0 == room0 && cand0 > b0) {                                            // everything between the step's start and the match is part of it
          uint32_t maxb = umin32(pre, cand0 - b0);
          if (maxb > 64u) maxb = 64u;
          const uint32_t pm0 = ip + (uint32_t)__builtin_amdgcn_readlane((int)wq, 0) - b0, cm0 = cand0 - b0;
          const bool bl = (uint32_t)lane < maxb;
          const uint32_t bx = src[bl ? pm0 - 1u - (uint32_t)lane : pm0], by = src[bl ? cm0 - 1u - (uint32_t)lane : cm0];
          const uint64_t bm = __ballot(!bl || bx != by);
          pre_ext = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
          pre -= pre_ext;
        }
      }
      PROF_LAP(10); PROF_ADD(2, L_last >= RANK_CAP); PROF_ADD(6, nseq);
      const uint32_t mlen = L + back + ((uint32_t)lane == last ? mext : 0u) + (lane == 0 ? pre_ext : 0u);
      const uint32_t mcode = mlen - 4u;
      const uint32_t inl = room - back;                                             // literals of the sequence that lie inside the step
      const uint32_t ll = inl + (lane == 0 ? pre : 0u);
      const bool r0pre = pre != 0u && lane == 0;                                    // its token, length bytes and outside literals are written apart
      const uint32_t hdr = r0pre ? 0u : (ll >= 15u ? 2u : 1u);                      // (inside the step ll <= 63: one length byte at most)
      const uint32_t nme = mcode >= 15u ? 1u : 0u;                                  // (a second length byte and more: the extended last match only)
      const uint32_t size = valid ? hdr + inl + 2u + nme : 0u;
      uint32_t incl = size;
      incl += dpp_row_shr0<1>(incl); incl += dpp_row_shr0<2>(incl); incl += dpp_row_shr0<4>(incl); incl += dpp_row_shr0<8>(incl);
      const uint32_t excl = incl - size;
      const uint32_t mcode_last = (uint32_t)__builtin_amdgcn_readlane((int)mcode, (int)last);
      const uint32_t extra_last = mcode_last >= 15u + 255u ? (mcode_last - 15u) / 255u : 0u;
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)last) + extra_last;
      uint32_t ll0 = 0, prebytes = 0;
      if (pre) { ll0 = (uint32_t)__builtin_amdgcn_readlane((int)ll, 0); prebytes = 1u + (ll0 >= 15u ? 1u + (ll0 - 15u) / 255u : 0u) + pre; }
      // limitedOutput (lz4.c:1114-1117, :1187-1211): the stream is stored raw when it does not fit
      if (op + prebytes + total + 8u > cap) return 0u;
      if (pre) {                                                                    // uniform, rare (a step that begins with pending literals)
        const uint32_t mc0 = (uint32_t)__builtin_amdgcn_readlane((int)mcode, 0);
        if (lane == 0) ENC_ST1(dst + op, (uint8_t)(((ll0 < 15u ? ll0 : 15u) << 4) | (mc0 < 15u ? mc0 : 15u)));
        op += 1u;
        if (ll0 >= 15u) op += emit_ext255(dst + op, ll0 - 15u, lane);
        wave_copy_disjoint(dst + op, src + anchor, pre, lane);
        op += pre;
        PROF_ADD(3, 1);
      }
      // sequence info from rank lanes to byte lanes: the word lands on the sequence's first position (the search start cq)
      scr[lane] = 0u;
      if (SS) scr[lane + 64] = 0u;
      BAMD_LDS_SYNC();
      if (valid) scr[cq] = 0x80000000u | (excl + hdr) | (inl << 10) | (w << 17) | (L << 23);
      BAMD_LDS_SYNC();
      const uint32_t myw = scr[lq];
      const uint32_t myw1 = SS ? scr[lq + 1u] : 0u;
      const uint64_t fm = __ballot((myw >> 31) != 0u);
      uint32_t s;                                                                   // the last sequence start at or before this lane's first position
      bool has;                                                                     // (positions below lane_lo belong to sequences emitted earlier)
      if (SS) {
        const uint64_t fm1 = __ballot((myw1 >> 31) != 0u);
        const uint64_t be = fm & ((2ull << lane) - 1ull), bo_ = fm1 & ((1ull << lane) - 1ull);      // even starts 2 j <= 2 l, odd starts 2 j + 1 < 2 l
        const uint32_t se = 2u * (63u - (uint32_t)__builtin_clzll(be | 1ull)), so = 2u * (63u - (uint32_t)__builtin_clzll(bo_ | 1ull)) + 1u;
        s = (bo_ != 0ull && (be == 0ull || so > se)) ? so : se;
        has = (be | bo_) != 0ull;
      } else {
        const uint64_t below = fm & ((2ull << lane) - 1ull);
        s = 63u - (uint32_t)__builtin_clzll(below | 1ull);
        has = below != 0ull;
      }
      const uint32_t inf = scr[s];
      BAMD_LDS_SYNC();
      // ---- all stores of the step: tokens, length bytes, offsets (rank lanes), then literals (byte lanes; SS = 1: one position after the other, so that
      //      the fields of the first are dead before those of the second are made) ----
      {
        const uint32_t tpos = op + excl;
        if (valid && hdr) ENC_ST1(dst + tpos, (uint8_t)(((ll < 15u ? ll : 15u) << 4) | (mcode < 15u ? mcode : 15u)));
        if (valid && hdr == 2u) ENC_ST1(dst + tpos + 1u, (uint8_t)(ll - 15u));
        const uint32_t opos = tpos + hdr + inl;
#ifndef BAMD_ENC_NOSTORE
        if (valid) g_st2(dst + opos, dist_r);
#endif
        if (valid && nme) ENC_ST1(dst + opos + 2u, (uint8_t)(mcode - 15u < 255u ? mcode - 15u : 255u));
        if (extra_last) {                                                           // uniform: the 255-run of a long match (lz4.c:1213-1226)
          const uint32_t opos_last = (uint32_t)__builtin_amdgcn_readlane((int)opos, (int)last);
          emit_ext255(dst + opos_last + 2u, mcode_last - 15u, lane);
        }
      }
      // table: like the reference, nothing inside a match is inserted (lz4.c:1236-1242 inserts ip-2 only): positions up to the winner, the position two
      // bytes before the match's end, and what lies behind the chain's last match when nothing more was found
      {
        const uint32_t k = lq - s;
        const uint32_t i_out = inf & 1023u, i_inl = (inf >> 10) & 127u, i_w = ((inf >> 17) & 63u) << SS, i_L = (inf >> 23) & 31u;
        if (has && k < i_inl) ENC_ST1(dst + op + i_out + k, (uint8_t)own.a);
        const bool open_end = i_L < RANK_CAP;                                       // (behind a match that was extended the chain picks up again below)
        const uint32_t e = i_w + i_L;
        if (live && has && (lq <= i_w || (open_end && (lq + 2u == e || lq >= e)))) PAR_PUT();
      }
      if (SS) {                                                                     // the lane's second position: a sequence of its own starts there, or it belongs to the first position's
        const bool own1 = (myw1 >> 31) != 0u;
        const uint32_t inf1 = own1 ? myw1 : inf;
        const bool has1 = has || own1;
        const uint32_t q1 = lq + 1u, k1 = own1 ? 0u : q1 - s;
        const uint32_t j_out = inf1 & 1023u, j_inl = (inf1 >> 10) & 127u, j_w = ((inf1 >> 17) & 63u) << SS, j_L = (inf1 >> 23) & 31u;
        if (has1 && k1 < j_inl) ENC_ST1(dst + op + j_out + k1, (uint8_t)(own.a >> 8));
        const bool open1 = j_L < RANK_CAP;
        const uint32_t e1 = j_w + j_L;
        if (live1 && has1 && (q1 <= j_w || (open1 && (q1 + 2u == e1 || q1 >= e1)))) PAR_PUT1();
      }
      op += total;
      const uint32_t e_last = w_last + L_last + mext;                               // end of the chain's last match, relative to ip
      anchor = ip + e_last;
      PROF_LAP(11);
      if (L_last < RANK_CAP || e_last >= SPAN) break;                               // the chain ended by itself / the match leaves the step
      lane_lo = e_last;
      if (live && lq + 2u == lane_lo) PAR_PUT();                             // (lz4.c:1236-1242)
      if (live1 && lq + 3u == lane_lo) PAR_PUT1();
    }
    if (!any) {
      PROF_ADD(1, 1);
      if (live) PAR_PUT();
      if (live1) PAR_PUT1();
      nfail += 1u << SS;                                     // (counted in 64-position units: the skip grows with the bytes that failed, whatever the stride)
      uint32_t adv = (nfail * (uint32_t)accel) / 16u;        // skip faster through incompressible data
      if (adv > BAMD_ENC_SKIPCAP - (1u << SS)) adv = BAMD_ENC_SKIPCAP - (1u << SS);   // (a step never starts more than 2 KiB behind the one before, as in lz_encode_wave)
      // Every fourth failed step is followed by its NEIGHBOUR.  A step finds only what earlier steps put into the table; once the skip is as long as
      // the data's runs (linspace at typesize 2: runs of 1 KiB with a period of 16 bytes, each run with new content) every step lands in a run no
      // earlier step has seen, fails, and keeps the skip long - the reference, which spreads its probes evenly (lz4.c:1044-1053), gets out at once,
      // and so does a step right behind one whose positions have just entered the table.
      if (BAMD_ENC_NEIGHBOUR && ((nfail >> SS) & (BAMD_ENC_NEIGHBOUR - 1u)) == 0u) adv = 0u;
      ip += SPAN + 64u * adv;
      continue;
    }
    nfail = 0;
    if (anchor >= step_end) {
      ip = anchor;
      ins_pending = true;                         // anchor-2 enters the table at the top of the next step (bytes in registers there)
    } else {
      if (tail_open && live && lq >= lane_lo) PAR_PUT();                   // nothing more to find behind an extended match
      if (tail_open && live1 && lq + 1u >= lane_lo) PAR_PUT1();
      ip = step_end;
    }
  }
#undef PAR_PUT
#undef PAR_PUT1
  op = lz4_emit_tail(dst, op, cap, src + anchor, n - anchor, lane);
  if (op == 0xffffffffu) return 0u;
  PROF_LAP(12);
  return op < n ? op : 0u;
}

// Which stride a compression level gets: every position at the levels that ask for ratio, every other one from BAMD_ENC_STRIDE2_MAXCLEVEL down
// (the benchmark's clevel 5 among them; the reference's own LZ4 acceleration at that level is 5, lz4.c:1044-1053).
#ifndef BAMD_ENC_STRIDE2_MAXCLEVEL
#define BAMD_ENC_STRIDE2_MAXCLEVEL 5
#endif
__device__ __forceinline__ uint32_t lz4_encode_wave_auto(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap,
                                                         int clevel, enc_entry_t* tab_generic, int lane EPROF_ARG) {
  if (clevel <= BAMD_ENC_STRIDE2_MAXCLEVEL) return lz4_encode_wave_par<1>(src, n, dst, cap, clevel, tab_generic, lane EPROF_PASS);
  return lz4_encode_wave_par<0>(src, n, dst, cap, clevel, tab_generic, lane EPROF_PASS);
}
