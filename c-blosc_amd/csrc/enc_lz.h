// enc_lz.h — the wave-parallel LZ match finders of the encode kernels (included by k_encode.hip inside namespace bamd):
// lz_encode_wave (LZ4 / BloscLZ streams, and the front end of the Zstd / zlib writers through their sinks) and hc_encode_wave
// (the LZ4HC-grade search), with the table, window and emit helpers they share.  DESIGN.md 3.3.
// build switches of the round-3 changes (defaults are what bench.py measures; the others are kept for same-session A/B runs, scripts/enc_ab.py)
constexpr int ENC_WAVES = 1;       // one stream per workgroup: a slot frees up as soon as ITS stream is done
// Table entry = position mod 65536 | 16 further hash bits as a tag << 16.  The tag lets a lane reject a
// stale or colliding entry WITHOUT touching memory: untagged, nearly every lane of every step fetched 20
// bytes from a random place in the last 64 KiB (a full cache line each, mostly L2 misses with a thousand
// streams in flight per XCD) - rocprofv3 FETCH_SIZE showed 6.8x the input being read.
#ifndef BAMD_ENC_HASH_BITS
#define BAMD_ENC_HASH_BITS 11
#endif
constexpr int ENC_HASH_BITS = BAMD_ENC_HASH_BITS;   // 2048 entries
constexpr int ENC_TAB = 1 << ENC_HASH_BITS;
typedef uint32_t enc_entry_t;
#ifndef BAMD_ENC_SPLIT_TAB
#define BAMD_ENC_SPLIT_TAB 1   // positions (u16) and 8-bit tags in separate arrays: 6 KiB instead of 8 KiB per wave, 24 instead
                               // of 20 waves per CU (same-session A/B: bench19 10.9 -> 10.3 ms); 0 = one u32 array, 16-bit tags
#endif
constexpr int ENC_TAB_BYTES = BAMD_ENC_SPLIT_TAB ? ENC_TAB * 3 : ENC_TAB * 4;
#ifndef BAMD_ENC_MINWAVES
#define BAMD_ENC_MINWAVES (BAMD_ENC_SPLIT_TAB ? 6 : 5)   // waves per SIMD the register allocator leaves room for (5: + 3 %, 7 = 26 waves per CU at 72 registers: + 1.8 %, profiles/r04/r04zu_*)
#endif
#ifndef BAMD_ENC_SEQ_SKIPCAP
#define BAMD_ENC_SEQ_SKIPCAP 32u    // the longest step of the sequential finder through match-less data, in 64-byte units (see BAMD_ENC_SKIPCAP in enc_lz4p.h)
#endif
#ifndef BAMD_ENC_LZ_MINWAVES
#define BAMD_ENC_LZ_MINWAVES 6   // the LZ4 / BloscLZ kernel.  Round 6: the two-positions-per-lane step of enc_lz4p.h first needed 96 registers (5) - at 80 its input window lived in
                                 // scratch memory and every step reloaded it behind the stores of the step before (profiles/r06c_*: 8.6 ms at 24 waves per CU, 7.5 at 20).  What
                                 // filled the registers were two dozen loop-invariant values derived from the lane number that the compiler hoists out of the step and the chain
                                 // loop; with the lane number made opaque at the top of both (enc_lz4p.h) the step fits 80 registers without a spill: 24 waves, 7.6 ms (r06w)
#endif
constexpr int ENC_LDS_WAVES = (160 * 1024) / ENC_TAB_BYTES;
constexpr int ENC_WAVES_PER_CU = ENC_LDS_WAVES < 4 * BAMD_ENC_MINWAVES ? ENC_LDS_WAVES : 4 * BAMD_ENC_MINWAVES;   // persistent grid size per CU
constexpr int ENC_LZ_LDS_WAVES = (160 * 1024) / (ENC_TAB_BYTES + 512);                                             // (+ the parallel emitter's 128 scratch dwords, enc_lz4p.h)
constexpr int ENC_LZ_WAVES_PER_CU = ENC_LZ_LDS_WAVES < 4 * BAMD_ENC_LZ_MINWAVES ? ENC_LZ_LDS_WAVES : 4 * BAMD_ENC_LZ_MINWAVES;   // ... of the LZ4 / BloscLZ kernel

// the table of one wave (LDS)
struct EncTable {
  __attribute__((address_space(3))) uint32_t* w;     // unified: entry words;  split: unused
  __attribute__((address_space(3))) uint16_t* pos;   // split: positions
  __attribute__((address_space(3))) uint8_t* tag;    // split: top 8 bits of the tag
  __device__ __forceinline__ void init(void* base) {
    w = (__attribute__((address_space(3))) uint32_t*)base;
    pos = (__attribute__((address_space(3))) uint16_t*)base;
    tag = (__attribute__((address_space(3))) uint8_t*)base + 2 * ENC_TAB;
  }
  __device__ __forceinline__ void clear(int lane) {
    for (int k = lane; k < ENC_TAB_BYTES / 4; k += 64) w[k] = 0u;
  }
  __device__ __forceinline__ void put(uint32_t h, uint32_t entry) {
    if (BAMD_ENC_SPLIT_TAB) { pos[h] = (uint16_t)entry; tag[h] = (uint8_t)(entry >> 24); }
    else w[h] = entry;
  }
  // entry with the same layout as enc_entry(); in split mode only the top 8 tag bits are kept
  __device__ __forceinline__ uint32_t get(uint32_t h) const {
    if (BAMD_ENC_SPLIT_TAB) return (uint32_t)pos[h] | ((uint32_t)tag[h] << 24);
    return w[h];
  }
  __device__ __forceinline__ static bool tag_equal(uint32_t a, uint32_t b) {
    return BAMD_ENC_SPLIT_TAB ? ((a ^ b) >> 24) == 0u : ((a ^ b) >> 16) == 0u;
  }
};

__device__ __forceinline__ uint32_t enc_mix(uint32_t seq) { return seq * 2654435761u; }
__device__ __forceinline__ uint32_t enc_slot(uint32_t mix) { return mix >> (32 - ENC_HASH_BITS); }
// table entry for position p whose 4 bytes hash to `mix`: the 16 bits below the slot bits are the tag
__device__ __forceinline__ uint32_t enc_entry(uint32_t mix, uint32_t p) { return ((mix << ENC_HASH_BITS) & 0xffff0000u) | (p & 0xffffu); }

__device__ __forceinline__ uint64_t ld8u(const gu8* p) { return g_ld8(p); }

// leading equal bytes (0..16) of two 16-byte groups
__device__ __forceinline__ uint32_t common16(const uint4& x, const uint4& y) {
  const uint64_t lo = ((uint64_t)(x.y ^ y.y) << 32) | (x.x ^ y.x);
  const uint64_t hi = ((uint64_t)(x.w ^ y.w) << 32) | (x.z ^ y.z);
  if (lo) return (uint32_t)__builtin_ctzll(lo) >> 3;
  return hi ? 8u + ((uint32_t)__builtin_ctzll(hi) >> 3) : 16u;
}

// number of leading equal bytes of src[a..] and src[b..], at most `maxlen`; a > b, wave-uniform
// arguments, wave-uniform result.  Never reads at or beyond src + n.
// Long matches dominate some byte planes (a constant plane is ONE 128 KiB match), so the bulk is
// compared 2 KiB per memory round trip (two 1 KiB rows, all four loads in flight together) with the
// exact mismatch byte found in the same trip; only the last < 2 KiB of a stream go 512 bytes per
// step through byte-safe loads.
// (the part behind the first `done` bytes, which the caller has found equal: done = 256 - the first trip of wave_common_fwd, or of enc_lz4p.h's
//  chain whose first trip travels underneath the step's emission - or 0)
__device__ __forceinline__ uint32_t wave_common_fwd_rest(const gu8* src, uint32_t n, uint32_t a, uint32_t b,
                                                         uint32_t maxlen, uint32_t done, int lane) {
  if (done == 256u && done < maxlen && a + done + 1024u <= n) {      // second trip: 1 KiB
    const uint4 x0 = g_ld16(src + a + done + 16 * lane), y0 = g_ld16(src + b + done + 16 * lane);
    const uint32_t q1 = done + 16u * (uint32_t)lane;
    uint32_t e1 = common16(x0, y0);
    const uint32_t r1 = q1 < maxlen ? maxlen - q1 : 0u;
    if (e1 > r1) e1 = r1;
    const uint64_t s1 = __ballot(e1 < 16u);
    if (s1) { const int f = __builtin_ctzll(s1); return done + 16u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)e1, f); }
    done += 1024u;
  }
  while (done < maxlen && a + done + 2048u <= n) {
    const gu8* pa = src + a + done + 16 * lane;
    const gu8* pb = src + b + done + 16 * lane;
    const uint4 x0 = g_ld16(pa), x1 = g_ld16(pa + 1024);
    const uint4 y0 = g_ld16(pb), y1 = g_ld16(pb + 1024);
    const uint32_t q = done + 16u * (uint32_t)lane;                 // this lane's first byte in row 0 (relative)
    uint32_t e0 = common16(x0, y0), e1 = common16(x1, y1);
    const uint32_t r0 = q < maxlen ? maxlen - q : 0u;               // bytes this lane may count
    const uint32_t r1 = q + 1024u < maxlen ? maxlen - q - 1024u : 0u;
    if (e0 > r0) e0 = r0;
    if (e1 > r1) e1 = r1;
    const uint64_t s0 = __ballot(e0 < 16u);
    if (s0) { const int f = __builtin_ctzll(s0); return done + 16u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)e0, f); }
    const uint64_t s1 = __ballot(e1 < 16u);
    if (s1) { const int f = __builtin_ctzll(s1); return done + 1024u + 16u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)e1, f); }
    done += 2048u;
  }
  while (done < maxlen) {
    const uint32_t q = done + 8u * (uint32_t)lane;     // this lane's first byte (relative)
    uint32_t vb = 0;                                     // bytes this lane may compare
    if (q < maxlen) vb = (maxlen - q < 8u) ? maxlen - q : 8u;
    uint32_t eq = 0;
    if (vb) {
      if (a + q + 8u <= n) {
        uint64_t x = ld8u(src + a + q) ^ ld8u(src + b + q);
        eq = x ? (uint32_t)(__builtin_ctzll(x) >> 3) : 8u;
        if (eq > vb) eq = vb;
      } else {
        while (eq < vb && src[a + q + eq] == src[b + q + eq]) eq++;
      }
    }
    const uint64_t stop = __ballot(eq < 8u);
    if (stop) {
      const int f = __builtin_ctzll(stop);
      return done + 8u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)eq, f);
    }
    done += 512u;
  }
  return maxlen;
}
// the first trip's verdict: x = the lane's dword of src[a..] XOR that of src[b..] (a + 256 <= n); returns the length when the mismatch (or maxlen) lies
// inside these 256 bytes, 0xffffffff when all of them are equal and more may follow
__device__ __forceinline__ uint32_t wave_common_first256(uint32_t x, uint32_t maxlen, int lane) {
  const uint32_t q = 4u * (uint32_t)lane;
  uint32_t e0 = x ? (uint32_t)(__builtin_ctz(x) >> 3) : 4u;
  const uint32_t r0 = q < maxlen ? maxlen - q : 0u;
  if (e0 > r0) e0 = r0;
  const uint64_t s0 = __ballot(e0 < 4u);
  if (s0) { const int f = __builtin_ctzll(s0); return 4u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane((int)e0, f); }
  return 0xffffffffu;
}
__device__ __forceinline__ uint32_t wave_common_fwd(const gu8* src, uint32_t n, uint32_t a, uint32_t b,
                                                    uint32_t maxlen, int lane) {
  uint32_t done = 0;
  // first trips: 256 bytes, then 1 KiB (round 3: 1 KiB / 512 / 256 + 1 KiB first trips cost 5 % / 11 % / 19 % less of the kernel on bench19 than 2 KiB rows at once)
  if (maxlen && a + 256u <= n) {
    const uint32_t r = wave_common_first256(g_ld4(src + a + 4 * lane) ^ g_ld4(src + b + 4 * lane), maxlen, lane);
    if (r != 0xffffffffu) return r;
    done = 256u;
  }
  return wave_common_fwd_rest(src, n, a, b, maxlen, done, lane);
}

#ifdef BAMD_ENC_NOSTORE      // timing-only build (wrong output): what do the encoder's byte stores cost?  (scripts/enc_ab.py does not check the output)
#define ENC_ST1(ptr, val) do { (void)(ptr); (void)(val); } while (0)
#else
#define ENC_ST1(ptr, val) do { *(ptr) = (val); } while (0)
#endif
// (Round 6: making the lane number opaque at the top of lz_encode_wave's loops, as enc_lz4p.h does, takes most spills out of the BloscLZ path - 21 -> 7 scratch
//  accesses in the function - and changes nothing that can be measured: BloscLZ - 1 %, Zstd + 1.5 %, zlib + 0.5 %, profiles/r06x_*; not kept here.)
// write `v` as LZ4's 255-run length extension starting at p; returns bytes written
__device__ __forceinline__ uint32_t emit_ext255(gu8* p, uint32_t v, int lane) {
  const uint32_t n255 = v / 255u, rem = v - n255 * 255u;
  for (uint32_t k = (uint32_t)lane; k < n255; k += 64u) ENC_ST1(p + k, (uint8_t)255u);
  if (lane == 0) ENC_ST1(p + n255, (uint8_t)rem);
  return n255 + 1u;
}

// EF_ZLIB2: Deflate's limits (distance <= 32 KiB) with the Zstd path's sink - the tokens are kept for a second pass (dynamic Huffman codes, enc_zlib.h)
enum { EF_LZ4 = 0, EF_BLOSCLZ = 1, EF_ZSTD = 2, EF_ZLIB = 3, EF_ZLIB2 = 4 };
#ifndef BAMD_ZSTD_MINLEN
#define BAMD_ZSTD_MINLEN 4     // shortest match the Zstd path takes (5 and 6: bench19 ratio and time in DESIGN.md 3.6)
#endif

// Where the match finder puts its findings when the target is a Zstd block (zstd_enc.h): literals go straight to
// their final place in the block being written, (literal length, match length, offset) triples to a scratch of the
// persistent wave; the sequence section is coded afterwards (zs_write_sequences).
struct ZsSink {
  gu8* lit;                       // literal bytes of the block
  uint32_t nlit, litcap;
  BAMD_GAS uint64_t* seq;         // zenc::pack_seq triples
  uint32_t nseq, seqcap;
};
__device__ __forceinline__ uint32_t zs_emit_seq(ZsSink& z, const gu8* lit, uint32_t ll, uint32_t off, uint32_t mlen, int lit_lane0, uint32_t ownbyte, int lane);

// Where the findings go when the target is a zlib stream (deflate_enc.h): straight into the bit stream.  One lane packs one
// symbol (a literal: 8 / 9 bits, a match piece: <= 31 bits); dfl_put_symbols places the symbols of all 64 lanes with a prefix
// sum over their bit counts, ORs them into a 65-dword LDS strip behind the pending bits and stores the full dwords.
struct DflSink {
  gu8* out; uint32_t cap;         // the stream being written and its capacity
  uint32_t pos;                   // bytes written so far
  uint32_t acc, nb;               // pending bits (< 32), wave-uniform
  volatile BAMD_LAS uint32_t* zb; // 65 dwords of this wave
};
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, d, 64); if (lane >= d) v += t; }
  return v;
}
__device__ __forceinline__ bool dfl_put_symbols(DflSink& z, uint32_t bits, uint32_t nbits, int lane) {
  const uint32_t incl = wave_incl_scan_u32(nbits, lane);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  z.zb[lane] = lane == 0 ? z.acc : 0u;
  if (lane == 0) z.zb[64] = 0u;
  BAMD_LDS_SYNC();
  if (nbits) {
    const uint32_t start = z.nb + incl - nbits, w = start >> 5, sh = start & 31u;
    __hip_atomic_fetch_or((BAMD_LAS uint32_t*)z.zb + w, bits << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (sh + nbits > 32u) __hip_atomic_fetch_or((BAMD_LAS uint32_t*)z.zb + w + 1u, bits >> (32u - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  const uint32_t fill = z.nb + total, ndw = fill >> 5;                  // <= 63 full dwords
  if (z.pos + 4u * ndw + 16u > z.cap) return false;
  BAMD_LDS_SYNC();
  const uint32_t mine = z.zb[lane], rest = z.zb[ndw];
  if ((uint32_t)lane < ndw) g_st4(z.out + z.pos + 4u * (uint32_t)lane, mine);
  z.acc = uni(rest); z.nb = fill & 31u; z.pos += 4u * ndw;
  return true;
}
// `ll` literal bytes (from registers when the run lies inside this step, see emit_literals) and, when mlen != 0, one match
__device__ __forceinline__ uint32_t dfl_emit_seq(DflSink& z, const gu8* lit, uint32_t ll, uint32_t dist, uint32_t mlen, int lit_lane0, uint32_t ownbyte, int lane) {
  const bool in_regs = lit_lane0 >= 0 && ll <= 64u && (uint32_t)lit_lane0 + ll <= 64u;
  const uint32_t np = mlen ? dfl::npieces(mlen) : 0u;
  uint32_t ldone = 0, pdone = 0;
  do {
    // literals first, match pieces behind them in the same step when they fit
    const uint32_t lcnt = ll - ldone < 64u ? ll - ldone : 64u;
    uint32_t pcnt = 0;
    if (ldone + lcnt == ll) { pcnt = np - pdone < 64u - lcnt ? np - pdone : 64u - lcnt; }
    // the register gather runs in ALL lanes (a lane that sits out a ds_bpermute cannot be read by the others)
    const uint32_t vreg = in_regs ? (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lit_lane0 + (uint32_t)lane) & 63u) << 2, (int)ownbyte) & 0xffu : 0u;
    dfl::Sym s = {0u, 0u};
    if ((uint32_t)lane < lcnt) s = dfl::literal(in_regs ? vreg : (uint32_t)lit[ldone + (uint32_t)lane]);
    else if ((uint32_t)lane < lcnt + pcnt) s = dfl::match(dfl::piece_len(mlen, pdone + (uint32_t)lane - lcnt, np), dist);
    if (!dfl_put_symbols(z, s.bits, s.nbits, lane)) return 0xffffffffu;
    ldone += lcnt; pdone += pcnt;
  } while (ldone < ll || pdone < np);
  return 0u;
}

// ---- emitters --------------------------------------------------------------------------------
// LZ4 sequence (lz4.c:1111-1226): token | litlen ext | literals | offset LE16 | matchlen ext.
// Returns new op, or 0xffffffff when the limitedOutput budget (lz4.c:1114-1117, :1187-1211) is hit.
// When the literal run lies inside the positions this step has just loaded (lit_lane0 >= 0: lane
// lit_lane0 + k holds literal byte k in `ownbyte`), the bytes are taken from registers with one
// ds_bpermute instead of being re-read from memory (saves a full memory round trip per sequence).
__device__ __forceinline__ void emit_literals(gu8* dst, const gu8* lit, uint32_t ll, int lit_lane0, uint32_t ownbyte, int lane) {
  if (lit_lane0 >= 0 && ll <= 64u && (uint32_t)lit_lane0 + ll <= 64u) {
    const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lit_lane0 + (uint32_t)lane) & 63u) << 2, (int)ownbyte);
    if ((uint32_t)lane < ll) ENC_ST1(dst + lane, (uint8_t)v);
  } else {
    wave_copy_disjoint(dst, lit, ll, lane);
  }
}

// (Round 3 also sent the common sequence shape through a 512-byte LDS ring that left 256 bytes at a time - the cure for k_zstd_seq's store
//  acknowledgements: bit-identical output, 6 % SLOWER here - profiles/r03/r03zj_enc_ab_run_buffer_rejected.txt; removed in round 4.)
__device__ __forceinline__ uint32_t lz4_emit_seq(gu8* dst, uint32_t op, uint32_t cap, const gu8* lit,
                                                 uint32_t ll, uint32_t off, uint32_t mlen, int lit_lane0, uint32_t ownbyte, int lane) {
  if (op + 1u + ll + (2u + 1u + 5u) + ll / 255u > cap) return 0xffffffffu;
  const uint32_t mcode = mlen - 4u;
  if (ll < 15u && mcode < 15u + 255u && (ll == 0u || (lit_lane0 >= 0 && (uint32_t)lit_lane0 + ll <= 64u))) {
    // the common shape (bench19's noisy planes: 1150 of 1163 sequences): token, <= 14 literals out of this step's registers, offset, at most one
    // length byte - lane j holds byte j of the sequence, one store.  Same bytes and the same two budget checks as the general path below.
    if (op + 1u + ll + 2u + (1u + 5u) + (mcode + 240u) / 255u > cap) return 0xffffffffu;
    const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lit_lane0 + (uint32_t)lane - 1u) & 63u) << 2, (int)ownbyte);
    uint32_t b = v;
    b = lane == 0 ? ((ll << 4) | (mcode < 15u ? mcode : 15u)) : b;
    b = (uint32_t)lane == ll + 1u ? off : b;
    b = (uint32_t)lane == ll + 2u ? off >> 8 : b;
    b = (uint32_t)lane == ll + 3u ? mcode - 15u : b;
    const uint32_t total = ll + 3u + (mcode >= 15u ? 1u : 0u);
    if ((uint32_t)lane < total) ENC_ST1(dst + op + lane, (uint8_t)b);
    return op + total;
  }
  const uint32_t tok = ((ll < 15u ? ll : 15u) << 4) | (mcode < 15u ? mcode : 15u);
  if (lane == 0) ENC_ST1(dst + op, (uint8_t)tok);
  op += 1u;
  if (ll >= 15u) op += emit_ext255(dst + op, ll - 15u, lane);
  emit_literals(dst + op, lit, ll, lit_lane0, ownbyte, lane);
  op += ll;
  if (lane < 2) ENC_ST1(dst + op + lane, (uint8_t)(off >> (8 * lane)));
  op += 2u;
  if (op + (1u + 5u) + (mcode + 240u) / 255u > cap) return 0xffffffffu;
  if (mcode >= 15u) op += emit_ext255(dst + op, mcode - 15u, lane);
  return op;
}
// final literal run (lz4.c:1302-1329)
__device__ __forceinline__ uint32_t lz4_emit_tail(gu8* dst, uint32_t op, uint32_t cap, const gu8* lit,
                                                  uint32_t run, int lane) {
  if (op + run + 1u + (run + 255u - 15u) / 255u > cap) return 0xffffffffu;
  if (lane == 0) dst[op] = (uint8_t)((run < 15u ? run : 15u) << 4);
  op += 1u;
  if (run >= 15u) op += emit_ext255(dst + op, run - 15u, lane);
  wave_copy_disjoint(dst + op, lit, run, lane);
  return op + run;
}

// BloscLZ literal run(s) (blosclz.c:246-256): every <= 32 literal bytes are preceded by ctrl = count-1
__device__ __forceinline__ uint32_t blz_emit_literals(gu8* dst, uint32_t op, uint32_t cap, const gu8* lit,
                                                      uint32_t ll, int lane) {
  if (ll == 0u) return op;
  const uint32_t nch = (ll + 31u) >> 5;
  if (op + ll + nch > cap) return 0xffffffffu;
  for (uint32_t c = (uint32_t)lane; c < nch; c += 64u) {
    const uint32_t cnt = (ll - 32u * c < 32u) ? ll - 32u * c : 32u;
    dst[op + 33u * c] = (uint8_t)(cnt - 1u);
  }
  for (uint32_t k = (uint32_t)lane; k < ll; k += 64u) dst[op + k + (k >> 5) + 1u] = lit[k];
  return op + ll + nch;
}
// BloscLZ match (blosclz.c:268-314); `dist` is the true distance (>= 1), `mlen` the true length (>= 3)
__device__ __forceinline__ uint32_t blz_emit_match(gu8* dst, uint32_t op, uint32_t cap, uint32_t dist,
                                                   uint32_t mlen, int lane) {
  const uint32_t L = mlen - 2u;
  uint32_t bd = dist - 1u;
  const bool far = bd >= 8191u;
  if (far) bd -= 8191u;
  const uint32_t hi = far ? 31u : (bd >> 8);
  const uint32_t next = (L >= 7u) ? (L - 7u) / 255u + 1u : 0u;
  const uint32_t total = 1u + next + (far ? 3u : 1u);
  if (op + total + 2u > cap) return 0xffffffffu;   // +2: room for the closing literal run
  if (lane == 0) dst[op] = (uint8_t)(((L < 7u ? L : 7u) << 5) | hi);
  op += 1u;
  if (L >= 7u) op += emit_ext255(dst + op, L - 7u, lane);
  if (far) {
    if (lane == 0) { dst[op] = 255u; dst[op + 1] = (uint8_t)(bd >> 8); dst[op + 2] = (uint8_t)bd; }
    op += 3u;
  } else {
    if (lane == 0) dst[op] = (uint8_t)bd;
    op += 1u;
  }
  return op;
}

// ---------------------------------------------------------------------------------------------
// one stream, one wave.  Returns the compressed size, or 0 when the stream must be stored raw
// (does not fit in `cap`, too small, or — BloscLZ — below the reference's per-clevel ratio floor).
//
// Per step the wave looks at 64 consecutive positions p = ip + lane:
//   round 1  each lane loads its own 20 bytes (+ the byte before) and probes the LDS hash table
//   round 2  each lane loads 20 bytes at its candidate; exact match length up to 20 is known per lane,
//            as is the run length against distance 1 (the one near distance the table cannot give)
//   select   the lane with the largest (length - lane) wins: a long match a few bytes later beats a
//            4-byte match now (tests/tools/enc_model.c: this ranking + the insertion rule below give
//            ratios at or above LZ4_compress_fast's on the SURVEY §8d data with fewer sequences)
//   insert   only lanes up to the winner enter the table — like the reference, nothing inside a match
//            is inserted (lz4.c:1236-1242 inserts ip-2 only), which keeps the START of repeated runs
//            findable
//   round 3  backward + forward extension loads are issued together; then the sequence is emitted.
// ---------------------------------------------------------------------------------------------
#ifndef BAMD_ENC_RANK_CAP
#define BAMD_ENC_RANK_CAP 20
#endif
constexpr uint32_t RANK_CAP = BAMD_ENC_RANK_CAP;   // bytes of a candidate that are compared for ranking (16 instead: more extensions, lower ratio, no faster)

// (Backward extension out of registers - the 8 (or 4) bytes in front of a candidate fetched together with its 20 - was built in round 3 in
//  two forms: bit-identical output, no gain / 3 % slower, profiles/r03/r03l_enc_ab_back8_no_gain.txt, r03x_enc_ab_back4_ext512.txt; removed in round 4.)
struct Bytes20 { uint64_t a, b; uint32_t c; };

// 20 bytes at src[pos..], zero beyond n (only the last step of a stream takes the slow branch)
__device__ __forceinline__ Bytes20 load20(const gu8* src, uint32_t pos, uint32_t n) {
  Bytes20 r;
  if (pos + 20u <= n) { r.a = g_ld8(src + pos); r.b = g_ld8(src + pos + 8u); r.c = RANK_CAP > 16u ? g_ld4(src + pos + 16u) : 0u; }
  else {
    r.a = 0; r.b = 0; r.c = 0;
    for (uint32_t k = 0; k < 20u && pos + k < n; k++) {
      const uint64_t v = src[pos + k];
      if (k < 8u) r.a |= v << (8u * k); else if (k < 16u) r.b |= v << (8u * (k - 8u)); else r.c |= (uint32_t)v << (8u * (k - 16u));
    }
  }
  return r;
}
__device__ __forceinline__ uint32_t common20(const Bytes20& x, const Bytes20& y) {
  uint64_t d = x.a ^ y.a;
  if (d) return (uint32_t)__builtin_ctzll(d) >> 3;
  d = x.b ^ y.b;
  if (d) return 8u + ((uint32_t)__builtin_ctzll(d) >> 3);
  const uint32_t e = x.c ^ y.c;
  return e ? 16u + ((uint32_t)__builtin_ctz(e) >> 3) : 20u;
}
// leading bytes of x equal to byte v (0..20)
__device__ __forceinline__ uint32_t runlen20(const Bytes20& x, uint32_t v) {
  const uint64_t rep = 0x0101010101010101ull * (uint64_t)v;
  Bytes20 y; y.a = rep; y.b = rep; y.c = (uint32_t)rep;
  return common20(x, y);
}
template <int N> __device__ __forceinline__ uint32_t dpp_row_shr0(uint32_t v) {   // lane i <- lane i-N of its 16-lane row, 0 outside
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true);
}
// wave-wide unsigned max, uniform result: four DPP steps inside each 16-lane row, then the four row
// maxima through SGPRs (no LDS-pipe traffic, unlike a butterfly of ds_bpermutes)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  uint32_t t;
  t = dpp_row_shr0<1>(v); v = t > v ? t : v;
  t = dpp_row_shr0<2>(v); v = t > v ? t : v;
  t = dpp_row_shr0<4>(v); v = t > v ? t : v;
  t = dpp_row_shr0<8>(v); v = t > v ? t : v;
  const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
  const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
  const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

// ---------------------------------------------------------------------------------------------
// Input window: 768 bytes of the stream in three VGPRs (lane L of w_i holds the dword at
// wbase + 256 i + 4 L).  The positions a step looks at always lie in w0/w1; w2 is fetched one
// 256-byte stride ahead, so its latency is hidden behind the steps in between.  A step gets its
// own 20 bytes per lane with ds_bpermutes from here instead of a memory round trip.
// ---------------------------------------------------------------------------------------------
struct EncWindow {
  uint32_t w0, w1, w2;     // while `pending`, w2 is RAW: the dword as loaded, not yet corrected for the end of the stream.  settle() corrects it,
                           // and the step calls settle() right behind its wait for the candidate bytes - where every older load has landed anyway.
                           // Correcting it inside seek() made the wave wait for the load it had just issued (the shift is the load's first use): a
                           // full memory round trip every 256 bytes instead of a prefetch; correcting it at the next rotation still waited for
                           // the stores of the step before (the one in-order counter).
  bool pending;            // uniform
  uint32_t wbase;          // uniform, multiple of 256
  const gu8* src;
  uint32_t n;

  // dword at stream offset base + 4*lane; bytes at or beyond n read as zero.  Branch-free: a dword that
  // would cross the end is read at n-4 instead and shifted down (n >= 13 here).
  __device__ __forceinline__ uint32_t fetch_raw(uint32_t base, int lane) const {
    const uint32_t off = base + 4u * (uint32_t)lane;
    const uint32_t a = off < n - 4u ? off : n - 4u;
    return g_ld4(src + a);
  }
  __device__ __forceinline__ uint32_t fix(uint32_t v, uint32_t base, int lane) const {
    const uint32_t off = base + 4u * (uint32_t)lane;
    const uint32_t a = off < n - 4u ? off : n - 4u;
    const uint32_t sh = off - a;                       // 0 in the body of the stream
    return sh < 4u ? v >> (8u * sh) : 0u;
  }
  __device__ __forceinline__ uint32_t fetch(uint32_t base, int lane) const { return fix(fetch_raw(base, lane), base, lane); }
  __device__ __forceinline__ void settle(int lane) {
    // a select, not a branch: w2 must be the result of an ALU instruction on EVERY path through the step, or the next rotation waits for
    // "a load that may still be in flight" - with vmcnt(0), i.e. for the emitter's stores
    const uint32_t fixed = fix(w2, wbase + 512u, lane);
    w2 = pending ? fixed : w2;
    pending = false;
#ifndef BAMD_WAVE_EMU
    asm volatile("; window settled" : "+v"(w2));       // keeps the correction HERE (the compiler sank it to the end of the step, behind the emitter's stores)
#endif
  }
  __device__ __forceinline__ void init(const gu8* s, uint32_t n_, int lane) {
    src = s; n = n_; wbase = 0;
    w0 = fetch(0u, lane); w1 = fetch(256u, lane); w2 = fetch(512u, lane); pending = false;
  }
  // make [lo, lo + 256 + 92) resident in w0/w1, lo = max(ip - 4, 0); ip only moves forward
  __device__ __forceinline__ void seek(uint32_t ip, int lane) {
    const uint32_t lo = ip >= 4u ? ip - 4u : 0u;
    const uint32_t d = lo - wbase;
    if (d < 256u) return;
    // w2 is settled here: every step calls settle() after its seek(), init() leaves it settled
    if (d < 512u) { w0 = w1; w1 = w2; wbase += 256u; }
    else if (d < 768u) { w0 = w2; wbase += 512u; w1 = fetch(wbase + 256u, lane); }
    else { wbase = lo & ~255u; w0 = fetch(wbase, lane); w1 = fetch(wbase + 256u, lane); }
    w2 = fetch_raw(wbase + 512u, lane); pending = true;      // (corrected behind the step's candidate wait: settle())
  }
};

// [start, n): the part of the stream this call covers (Zstd: one block of a frame; the table and earlier positions stay
// valid candidates); LZ4 / BloscLZ always start at 0.  `zs` is only used by EF_ZSTD, which returns the position up to
// which sequences were emitted (the caller appends the literals behind it) or 0xffffffff when the sink is full.
template <int FMT>
__device__ uint32_t lz_encode_wave(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap,
                                   int clevel, enc_entry_t* tab_generic, int lane EPROF_ARG, uint32_t start = 0, ZsSink* zs = nullptr, DflSink* df = nullptr) {
  // the table lives in LDS; say so explicitly (a generic pointer in a non-inlined function would make
  // every probe a flat_load)
  EncTable tab;
  tab.init((void*)tab_generic);
  // stream-end rules.  LZ4: last match starts <= n-12, ends <= n-5 (lz4.c:245-246, :963-964).
  // BloscLZ: matches start < n-12 (blosclz.c:465), stream must end with >= 1 literal (blosclz.c:708-710).
  if (FMT == EF_ZSTD || FMT == EF_ZLIB || FMT == EF_ZLIB2) { if (n < start + 16u) return start; }
  else if (FMT == EF_LZ4 ? (n < 13u) : (n < 16u || cap < 66u)) return 0u;
  const uint32_t last_start = n - 12u;                       // inclusive bound on match starts
  const uint32_t mlimit = (FMT == EF_LZ4) ? n - 5u : n - 2u;  // matches end at or before this position
  const int accel = 10 - clevel;                              // blosc/blosc.c:577-587
  // Effort knob in the spirit of the reference's clevel -> LZ4 acceleration mapping (blosc.c:577-587: lower
  // levels look at fewer positions) and of blosclz's tunable minimum match length (blosclz.c:445-457):
  // below clevel 9 a match must be longer than the format minimum to be taken.  4-byte matches are mostly accidental in noisy planes, save one
  // byte each and cost a full sequence: requiring 6 halved the encode time of noisy float64 data for
  // < 1 % of ratio (bench19: 53.3 -> 48.5, still far above the reference's 36.7 at this clevel); 7 (round 4,
  // profiles/r04/r04zb_enc_ab_min_match_6_7_8.txt): random-walk data another - 11 %, bench19 - 2 % at ratio 47.8; 8 would cost 13 % of bench19's ratio.  LZ4 only: BloscLZ stores a stream raw below its ratio floor (blosclz.c:426-435), which low-entropy noise then misses.
  // (Zstd sequences are cheaper than LZ4's - a repeated distance costs 5 bits - so short matches pay off there.)
  static_assert(EF_ZSTD == 2, "");
  const uint32_t zmin = (uint32_t)__builtin_amdgcn_readfirstlane(BAMD_ZSTD_MINLEN);
  const uint32_t minlen = FMT == EF_ZSTD ? zmin : (clevel >= 9 ? 4u : (clevel >= 6 ? 5u : (FMT == EF_LZ4 ? 7u : 6u)));

  if (start == 0u) tab.clear(lane);

  EncWindow win;
  win.init(src, n, lane);
  uint32_t ip = start, anchor = start, op = 0, nfail = 0;
  bool ins_pending = false;                       // position ip-2 still has to enter the table (lz4.c:1236-1242)
  while (ip <= last_start) {
    const uint32_t p = ip + (uint32_t)lane;
    const bool live = p <= last_start;
    // ---- round 1 (registers + LDS only): own bytes from the window, table probe ----
    win.seek(ip, lane);
    const uint32_t lo = ip >= 4u ? ip - 4u : 0u;
    const uint32_t rb = lo & ~3u;                                   // stream offset of r's lane 0
    const uint32_t D = (rb - win.wbase) >> 2;                       // < 64
    const int gsel = (int)((D + (uint32_t)lane) << 2);
    const uint32_t ra = (uint32_t)__builtin_amdgcn_ds_bpermute(gsel, (int)win.w0);
    const uint32_t rc = (uint32_t)__builtin_amdgcn_ds_bpermute(gsel, (int)win.w1);
    const uint32_t r = (D + (uint32_t)lane < 64u) ? ra : rc;        // lane j: dword at rb + 4j (j <= 23 is all that is used)
    const uint32_t bo0 = ip - rb;                                   // 4..7 (or ip when ip < 4)
    const uint32_t bo = bo0 + (uint32_t)lane;
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
    // the two bytes before ip (uniform): r's lanes 0/1 hold them
    const uint64_t r01 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)r, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)r, 0);
    const uint32_t before2 = bo0 >= 2u ? (uint32_t)(r01 >> (8u * (bo0 - 2u))) & 0xffffu : 0u;   // src[ip-2] | src[ip-1] << 8
    uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(o0 & 0xffu), 0x138, 0xf, 0xf, false);   // wave_shr:1 - the byte of the lane below, without the LDS pipe
    if (lane == 0) prev = ip ? (bo0 >= 2u ? before2 >> 8 : (uint32_t)(r01 >> (8u * (bo0 - 1u))) & 0xffu) : 0x100u;
    if (ins_pending) {
      const uint32_t m2 = enc_mix(before2 | (o0 << 16));
      if (lane == 0) tab.put(enc_slot(m2), enc_entry(m2, ip - 2u));
      ins_pending = false;
    }
    uint32_t h = 0, cand = 0, limit = 0, mine = 0;   // mine: this lane's own table entry
    bool tab_ok = false;
    if (live) {
      limit = mlimit - p; if (limit > RANK_CAP) limit = RANK_CAP;     // bytes of a match starting at p that may be counted
      const uint32_t mix = enc_mix(o0);
      h = enc_slot(mix);
      mine = enc_entry(mix, p);
      const uint32_t e = tab.get(h);
      const uint32_t d = (p - e) & 0xffffu;
      if (d != 0u && d <= p && EncTable::tag_equal(e, mine) && ((FMT != EF_ZLIB && FMT != EF_ZLIB2) || d <= dfl::kMaxDist)) { cand = p - d; tab_ok = true; }
    } else {
      prev = 0x100u;
    }
    PROF_LAP(8); PROF_ADD(0, 1);
    // ---- round 2: candidate bytes, exact lengths up to RANK_CAP ----
    uint32_t len = 0;
    // nb: how many bytes in front of the candidate equal those in front of p, counted backwards, 0 .. 4 (4 = at least four); 7 = not looked at.
    // The four bytes travel with the candidate's 20: most backward extensions are shorter than four bytes (bench19: 455 of 1162 sequences
    // extend backwards, 49 by more than 4), and a sequence that needs neither more backward bytes nor forward rows then costs no memory
    // round trip of its own in the selection loop below.
    uint32_t nb = 7u;
    if (tab_ok) {
      const uint32_t cpre = cand >= 4u ? ld4u(src + cand - 4u) : 0u;
      const Bytes20 cb = load20(src, cand, n);
      if (cand >= 4u) { const uint32_t x = cpre ^ ownpre; nb = x ? (uint32_t)__builtin_clz(x) >> 3 : 4u; }
      len = common20(own, cb);
      if (len > limit) len = limit;
      if (len < minlen) len = 0;
      if (FMT == EF_BLOSCLZ && len < 6u && p - cand - 1u >= 8191u) len = 0;   // far and short (blosclz.c:535)
    }
    if (live && prev < 0x100u) {                    // distance 1: run of the previous byte
      uint32_t rl = runlen20(own, prev);
      if (rl > limit) rl = limit;
      if (rl >= minlen && rl > len) { len = rl; cand = p - 1u; nb = 7u; }
    }
    win.settle(lane);                               // behind the wait for the candidates: free
    // ---- select + emit.  The winner maximises (len - lane), ties to the lower lane.  When its match
    // ends inside this step's 64 positions, the lanes behind it still hold valid candidates: pick
    // again among them instead of paying a new probe + candidate round trip for a short advance. ----
    const uint32_t step_end = ip + 64u;
    uint32_t lane_lo = 0;                         // first lane not covered by a sequence emitted in this step
    bool any = false;
    PROF_LAP(9);
    for (;;) {
      const uint32_t key = (len && (uint32_t)lane >= lane_lo) ? (((len + 64u - (uint32_t)lane) << 6) | (63u - (uint32_t)lane)) : 0u;
      const uint32_t best = wave_max_u32(key);
      if (best == 0u) break;
      any = true;
      const int f = 63 - (int)(best & 63u);
      if (live && (uint32_t)lane >= lane_lo && lane <= f) tab.put(h, mine);
      uint32_t pm = ip + (uint32_t)f;
      uint32_t cm = (uint32_t)__builtin_amdgcn_readlane((int)cand, f);
      const uint32_t len_f = (uint32_t)__builtin_amdgcn_readlane((int)len, f);
      // ---- round 3: extensions ----
      uint32_t maxb = pm - anchor;
      if (cm < maxb) maxb = cm;
      if (maxb > 64u) maxb = 64u;
      uint32_t back = 0;
      bool back_known = maxb == 0u;
      if (FMT != EF_BLOSCLZ && !back_known) {
        const uint32_t nb_f = (uint32_t)__builtin_amdgcn_readlane((int)nb, f);
        const uint32_t q = nb_f < maxb ? nb_f : maxb;
        if (nb_f <= 4u && (q < 4u || q == maxb)) { back = q; back_known = true; }     // a mismatch within four bytes, or no room for more
      }
      // otherwise backward bytes are requested first and looked at last, so that they travel together with the
      // forward rows (one memory round trip for both directions)
      // Every lane loads (lanes >= maxb a harmless byte): a load under a per-lane condition is compared where it is issued - the
      // compiler then waits for it at once (the ISA had "load, load, s_waitcnt vmcnt(0)" here) and the forward rows became a second trip.
      uint32_t bx = 0, by = 1;
      if (!back_known) {                                 // uniform
        const bool bl = (uint32_t)lane < maxb;
        bx = src[bl ? pm - 1u - (uint32_t)lane : pm]; by = src[bl ? cm - 1u - (uint32_t)lane : cm];
      }
      asm volatile("" ::: "memory");
      uint32_t mlen = len_f;
      if (len_f == RANK_CAP && pm + RANK_CAP < mlimit)
        mlen += wave_common_fwd(src, n, pm + RANK_CAP, cm + RANK_CAP, mlimit - (pm + RANK_CAP), lane);
#ifndef BAMD_WAVE_EMU
      asm volatile("; back bytes looked at here" : "+v"(bx), "+v"(by) :: "memory");   // or the comparison is hoisted to the loads, wait included
#endif
      if (!back_known) {
        const uint64_t bm = __ballot((uint32_t)lane >= maxb || bx != by);
        back = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
      }
      PROF_LAP(10); PROF_ADD(2, len_f == RANK_CAP); PROF_ADD(3, anchor < ip && pm > anchor); PROF_ADD(6, 1); PROF_ADD(7, back > 0); if (FMT != EF_ZSTD) { PROF_ADD(4, back > 4); PROF_ADD(5, maxb > 0); }
      pm -= back; cm -= back; mlen += back;
      const uint32_t ll = pm - anchor;
      const uint32_t dist = pm - cm;
      if (FMT == EF_LZ4) {
        op = lz4_emit_seq(dst, op, cap, src + anchor, ll, dist, mlen, anchor >= ip ? (int)(anchor - ip) : -1, (uint32_t)own.a & 0xffu, lane);
        if (op == 0xffffffffu) return 0u;
      } else if (FMT == EF_ZSTD || FMT == EF_ZLIB2) {
        if (zs_emit_seq(*zs, src + anchor, ll, dist, mlen, anchor >= ip ? (int)(anchor - ip) : -1, (uint32_t)own.a & 0xffu, lane) == 0xffffffffu) return 0xffffffffu;
      } else if (FMT == EF_ZLIB) {
        if (dfl_emit_seq(*df, src + anchor, ll, dist, mlen, anchor >= ip ? (int)(anchor - ip) : -1, (uint32_t)own.a & 0xffu, lane) == 0xffffffffu) return 0xffffffffu;
      } else {
        op = blz_emit_literals(dst, op, cap, src + anchor, ll, lane);
        if (op == 0xffffffffu) return 0u;
        op = blz_emit_match(dst, op, cap, dist, mlen, lane);
        if (op == 0xffffffffu) return 0u;
      }
      anchor = pm + mlen;
      PROF_LAP(11);
      if (anchor >= step_end) break;
      lane_lo = anchor - ip;                      // >= 4
      // like the reference, remember the position two bytes before the new anchor (lz4.c:1236-1242)
      if (live && (uint32_t)lane + 2u == lane_lo) tab.put(h, mine);
    }
    if (!any) {
      PROF_ADD(1, 1);
      if (live) tab.put(h, mine);
      nfail++;
      uint32_t adv = 1u + (nfail * (uint32_t)accel) / 16u;   // skip faster through incompressible data
      if (adv > BAMD_ENC_SEQ_SKIPCAP) adv = BAMD_ENC_SEQ_SKIPCAP;
      ip += 64u * adv;
      continue;
    }
    nfail = 0;
    if (anchor >= step_end) {
      ip = anchor;
      ins_pending = true;                         // anchor-2 enters the table at the top of the next step (bytes in registers there)
    } else {
      if (live && (uint32_t)lane >= lane_lo) tab.put(h, mine);   // nothing more to find behind the last match
      ip = step_end;
    }
  }
  // closing literals
  if (FMT == EF_ZSTD || FMT == EF_ZLIB || FMT == EF_ZLIB2) return anchor;
  if (FMT == EF_LZ4) {
    op = lz4_emit_tail(dst, op, cap, src + anchor, n - anchor, lane);
    if (op == 0xffffffffu) return 0u;
  } else {
    op = blz_emit_literals(dst, op, cap, src + anchor, n - anchor, lane);
    if (op == 0xffffffffu) return 0u;
    // first byte is always a literal-run control; set the marker bit (blosclz.c:607).  Lane 0 wrote
    // dst[0] itself, so this same-lane read-modify-write is ordered.
    if (lane == 0) dst[0] |= 0x20u;
    // reference policy: streams that compress worse than the per-clevel floor are stored raw
    // (blosclz.c:426-435, applied there to a probe of the last quarter; here to the real result)
    const float floor_ratio[10] = {0.f, 2.f, 1.5f, 1.2f, 1.2f, 1.2f, 1.2f, 1.15f, 1.1f, 1.0f};
    if ((float)n < floor_ratio[clevel] * (float)op) return 0u;
  }
  PROF_LAP(12);
  return op < n ? op : 0u;
}

// ---------------------------------------------------------------------------------------------
// LZ4HC-grade search ("lz4hc", blosc/blosc.c:422-433 -> LZ4_compress_HC, lz4hc.c): the same LZ4 block format and the same
// step as above, with the two things the reference's chain search has over a single table probe, in the form a wave can
// afford (tests/tools/enc_model2.c measures each on the CPU: SURVEY 8d planes, ratio against LZ4_compress_HC level 9):
//   * several candidates per position: 4-way buckets (FIFO) instead of one entry - positions (4 x u16) and tags (4 x u8) of a
//     bucket come with one 8-byte and one 4-byte LDS read, all candidates of a lane are fetched together;
//   * candidates ranked by their TRUE length, not by their first 20 bytes: lanes whose candidates share a distance look
//     at the same match, so one wave-wide comparison (wave_common_fwd) gives every one of them its exact length;
//     at most HC_GROUPS such comparisons per step, what is left ranks with 20 bytes as before.
// What enters the table, the (length - lane) choice and the re-selection behind a short match are those of
// lz_encode_wave: the model says the reference's other ingredients (every position in the chain, deeper chains, an optimal parse of
// the step) add nothing on this data once the ranking is exact.
// ---------------------------------------------------------------------------------------------
constexpr int HC_HASH_BITS = 11;
constexpr int HC_SLOTS = 1 << HC_HASH_BITS;
constexpr int HC_TAB_BYTES = HC_SLOTS * 12;            // 24 KiB per wave: 6 waves per CU
constexpr int HC_WAVES_PER_CU = (160 * 1024) / HC_TAB_BYTES;
constexpr int HC_GROUPS = 8;
constexpr uint32_t HC_RANK_MAX = 1u << 20;             // lengths beyond this rank alike (keeps the selection key inside 32 bits)

struct HcTable {
  BAMD_LAS uint64_t* pos;     // bucket h: four positions mod 65536, newest in the low 16 bits
  BAMD_LAS uint32_t* tag;     // bucket h: their four 8-bit tags, newest in the low byte
  __device__ __forceinline__ void init(void* base) {
    pos = (BAMD_LAS uint64_t*)base;
    tag = (BAMD_LAS uint32_t*)((BAMD_LAS uint8_t*)base + 8 * HC_SLOTS);
  }
  __device__ __forceinline__ void clear(int lane) {
    BAMD_LAS uint32_t* w = (BAMD_LAS uint32_t*)pos;
    for (int k = lane; k < HC_TAB_BYTES / 4; k += 64) w[k] = 0u;
  }
  // Lanes of one call that share a bucket all shift the same old content; one of them lands.  Whatever a bucket
  // holds is only ever a hint: every candidate is compared byte by byte before it is used.
  __device__ __forceinline__ void put(uint32_t h, uint32_t p, uint32_t t8) {
    const uint64_t pp = pos[h];
    const uint32_t tt = tag[h];
    pos[h] = (pp << 16) | (uint64_t)(p & 0xffffu);
    tag[h] = (tt << 8) | (t8 & 0xffu);
  }
};
__device__ __forceinline__ uint32_t hc_slot(uint32_t mix) { return mix >> (32 - HC_HASH_BITS); }
__device__ __forceinline__ uint32_t hc_tag(uint32_t mix) { return (mix << HC_HASH_BITS) >> 24; }

// FMT: EF_LZ4 (the "lz4hc" compressor), or EF_ZSTD / EF_ZLIB: the same search in front of the Zstd / zlib writers (their sinks, the
// part [start, n) of the stream and the return value as in lz_encode_wave)
template <int FMT>
__device__ uint32_t hc_encode_wave(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap,
                                   enc_entry_t* tab_generic, int lane, uint32_t start = 0, ZsSink* zs = nullptr, DflSink* df = nullptr) {
  HcTable tab;
  tab.init((void*)tab_generic);
  if (FMT == EF_LZ4) { if (n < 13u) return 0u; }              // lz4.c:245-246, :963-964 as in lz_encode_wave
  else if (n < start + 16u) return start;
  const uint32_t last_start = n - 12u;
  const uint32_t mlimit = (FMT == EF_LZ4) ? n - 5u : n - 2u;
  const uint32_t minlen = 4u;
  if (start == 0u) tab.clear(lane);
  EncWindow win;
  win.init(src, n, lane);
  uint32_t ip = start, anchor = start, op = 0, nfail = 0;
  bool ins_pending = false;
  while (ip <= last_start) {
    const uint32_t p = ip + (uint32_t)lane;
    const bool live = p <= last_start;
    // ---- own bytes out of the register window (as in lz_encode_wave) ----
    win.seek(ip, lane);
    const uint32_t lo = ip >= 4u ? ip - 4u : 0u;
    const uint32_t rb = lo & ~3u;
    const uint32_t D = (rb - win.wbase) >> 2;
    const int gsel = (int)((D + (uint32_t)lane) << 2);
    const uint32_t ra = (uint32_t)__builtin_amdgcn_ds_bpermute(gsel, (int)win.w0);
    const uint32_t rc = (uint32_t)__builtin_amdgcn_ds_bpermute(gsel, (int)win.w1);
    const uint32_t r = (D + (uint32_t)lane < 64u) ? ra : rc;
    const uint32_t bo0 = ip - rb;
    const uint32_t bo = bo0 + (uint32_t)lane;
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
    Bytes20 own;
    own.a = ((uint64_t)o1 << 32) | o0; own.b = ((uint64_t)o3 << 32) | o2; own.c = o4;
    const uint64_t r01 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)r, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)r, 0);
    const uint32_t before2 = bo0 >= 2u ? (uint32_t)(r01 >> (8u * (bo0 - 2u))) & 0xffffu : 0u;   // src[ip-2] | src[ip-1] << 8
    uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(o0 & 0xffu), 0x138, 0xf, 0xf, false);   // wave_shr:1 - the byte of the lane below, without the LDS pipe
    if (lane == 0) prev = ip ? (bo0 >= 2u ? before2 >> 8 : (uint32_t)(r01 >> (8u * (bo0 - 1u))) & 0xffu) : 0x100u;
    if (!live) prev = 0x100u;
    if (ins_pending) {
      const uint32_t m2 = enc_mix(before2 | (o0 << 16));
      if (lane == 0) tab.put(hc_slot(m2), ip - 2u, hc_tag(m2));
      ins_pending = false;
    }
    // ---- candidates: four bucket entries + distance 1, all fetched together, lengths up to RANK_CAP ----
    const uint32_t room = live ? mlimit - p : 0u;                  // bytes a match starting at p may have
    const uint32_t limit = room > RANK_CAP ? RANK_CAP : room;
    const uint32_t mix = enc_mix(o0);
    const uint32_t h = hc_slot(mix), mytag = hc_tag(mix);
    uint32_t cw[5], lw[5];
    uint32_t okm = 0;                                              // ways that hold a usable candidate
    {
      uint64_t pp = 0; uint32_t tt = 0;
      if (live) { pp = tab.pos[h]; tt = tab.tag[h]; }
      uint32_t dw[4];
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const uint32_t e = (uint32_t)(pp >> (16 * w)) & 0xffffu;
        const uint32_t d = (p - e) & 0xffffu;
        bool ok = live && d != 0u && d <= p && ((tt >> (8 * w)) & 0xffu) == mytag && ((FMT != EF_ZLIB && FMT != EF_ZLIB2) || d <= dfl::kMaxDist);
#pragma unroll
        for (int v = 0; v < w; v++) ok = ok && !(((okm >> v) & 1u) && dw[v] == d);   // the same position twice in a bucket
        dw[w] = d;
        cw[w] = ok ? p - d : 0u;
        okm |= ok ? (1u << w) : 0u;
      }
    }
    Bytes20 cb0 = load20(src, cw[0], n), cb1 = load20(src, cw[1], n), cb2 = load20(src, cw[2], n), cb3 = load20(src, cw[3], n);
    {
      uint32_t l0 = common20(own, cb0), l1 = common20(own, cb1), l2 = common20(own, cb2), l3 = common20(own, cb3);
      l0 = l0 > limit ? limit : l0; l1 = l1 > limit ? limit : l1; l2 = l2 > limit ? limit : l2; l3 = l3 > limit ? limit : l3;
      lw[0] = ((okm & 1u) && l0 >= minlen) ? l0 : 0u;
      lw[1] = ((okm & 2u) && l1 >= minlen) ? l1 : 0u;
      lw[2] = ((okm & 4u) && l2 >= minlen) ? l2 : 0u;
      lw[3] = ((okm & 8u) && l3 >= minlen) ? l3 : 0u;
    }
    win.settle(lane);
    cw[4] = 0u; lw[4] = 0u;
    if (prev < 0x100u) {                                           // distance 1: run of the previous byte
      uint32_t rl = runlen20(own, prev);
      if (rl > limit) rl = limit;
      if (rl >= minlen) { lw[4] = rl; cw[4] = p - 1u; }
    }
    // ---- exact lengths for candidates that ran into RANK_CAP: one wave-wide comparison per distance ----
    uint32_t um = 0;                                               // ways still ranked by their first RANK_CAP bytes only
#pragma unroll
    for (int w = 0; w < 5; w++) um |= (lw[w] == RANK_CAP && room > RANK_CAP) ? (1u << w) : 0u;
    for (int g = 0; g < HC_GROUPS; g++) {
      const uint64_t open = __ballot(um != 0u);
      if (open == 0ull) break;
      const int l0 = __builtin_ctzll(open);
      const uint32_t um0 = (uint32_t)__builtin_amdgcn_readlane((int)um, l0);
      const uint32_t w0 = (uint32_t)__builtin_ctz(um0);            // uniform
      const uint32_t csel = w0 == 0u ? cw[0] : (w0 == 1u ? cw[1] : (w0 == 2u ? cw[2] : (w0 == 3u ? cw[3] : cw[4])));
      const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)csel, l0);
      const uint32_t p0 = ip + (uint32_t)l0;
      const uint32_t d0 = p0 - c0;
      // lane l0 has room > RANK_CAP, so there is at least one more byte to compare
      const uint32_t L = RANK_CAP + wave_common_fwd(src, n, p0 + RANK_CAP, c0 + RANK_CAP, mlimit - (p0 + RANK_CAP), lane);
      // lane l0 + j with the same distance sees the same match from j bytes further in: length L - j, as long as
      // that is still what its own 20-byte comparison said (>= RANK_CAP)
      const uint32_t j = (uint32_t)lane - (uint32_t)l0;
      const bool inside = lane >= l0 && j + RANK_CAP <= L;
#pragma unroll
      for (int w = 0; w < 5; w++) {
        if (((um >> w) & 1u) && inside && p - cw[w] == d0) { lw[w] = L - j; um &= ~(1u << w); }
      }
    }
    // ---- this lane's best candidate ----
    uint32_t len = 0, cand = 0;
    bool exact = true;                                             // len is the whole match (no forward extension needed)
#pragma unroll
    for (int w = 0; w < 5; w++) {
      if (lw[w] > len) { len = lw[w]; cand = cw[w]; exact = ((um >> w) & 1u) == 0u; }
    }
    if (len > HC_RANK_MAX) { len = HC_RANK_MAX; exact = false; }
    // ---- select + emit (as in lz_encode_wave) ----
    const uint32_t step_end = ip + 64u;
    uint32_t lane_lo = 0;
    bool any = false;
    for (;;) {
      const uint32_t key = (len && (uint32_t)lane >= lane_lo) ? (((len + 64u - (uint32_t)lane) << 6) | (63u - (uint32_t)lane)) : 0u;
      const uint32_t best = wave_max_u32(key);
      if (best == 0u) break;
      any = true;
      const int f = 63 - (int)(best & 63u);
      if (live && (uint32_t)lane >= lane_lo && lane <= f) tab.put(h, p, mytag);
      uint32_t pm = ip + (uint32_t)f;
      uint32_t cm = (uint32_t)__builtin_amdgcn_readlane((int)cand, f);
      const uint32_t len_f = (uint32_t)__builtin_amdgcn_readlane((int)len, f);
      const bool exact_f = __builtin_amdgcn_readlane((int)(exact ? 1 : 0), f) != 0;
      uint32_t maxb = pm - anchor;
      if (cm < maxb) maxb = cm;
      if (maxb > 64u) maxb = 64u;
      uint32_t bx = 0, by = 1;
      if (maxb) {                                        // uniform; every lane loads (see lz_encode_wave)
        const bool bl = (uint32_t)lane < maxb;
        bx = src[bl ? pm - 1u - (uint32_t)lane : pm]; by = src[bl ? cm - 1u - (uint32_t)lane : cm];
      }
      asm volatile("" ::: "memory");
      uint32_t mlen = len_f;
      if (!exact_f && pm + len_f < mlimit)
        mlen += wave_common_fwd(src, n, pm + len_f, cm + len_f, mlimit - (pm + len_f), lane);
#ifndef BAMD_WAVE_EMU
      asm volatile("; back bytes looked at here" : "+v"(bx), "+v"(by) :: "memory");
#endif
      const uint64_t bm = __ballot((uint32_t)lane >= maxb || bx != by);
      const uint32_t back = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
      pm -= back; cm -= back; mlen += back;
      if (FMT == EF_LZ4) {
        op = lz4_emit_seq(dst, op, cap, src + anchor, pm - anchor, pm - cm, mlen, anchor >= ip ? (int)(anchor - ip) : -1, (uint32_t)own.a & 0xffu, lane);
        if (op == 0xffffffffu) return 0u;
      } else if (FMT == EF_ZSTD || FMT == EF_ZLIB2) {
        if (zs_emit_seq(*zs, src + anchor, pm - anchor, pm - cm, mlen, anchor >= ip ? (int)(anchor - ip) : -1, (uint32_t)own.a & 0xffu, lane) == 0xffffffffu) return 0xffffffffu;
      } else {
        if (dfl_emit_seq(*df, src + anchor, pm - anchor, pm - cm, mlen, anchor >= ip ? (int)(anchor - ip) : -1, (uint32_t)own.a & 0xffu, lane) == 0xffffffffu) return 0xffffffffu;
      }
      anchor = pm + mlen;
      if (anchor >= step_end) break;
      lane_lo = anchor - ip;
      if (live && (uint32_t)lane + 2u == lane_lo) tab.put(h, p, mytag);
    }
    if (!any) {
      if (live) tab.put(h, p, mytag);
      nfail++;
      uint32_t adv = 1u + nfail / 16u;
      if (adv > 16u) adv = 16u;
      ip += 64u * adv;
      continue;
    }
    nfail = 0;
    if (anchor >= step_end) {
      ip = anchor;
      ins_pending = true;
    } else {
      if (live && (uint32_t)lane >= lane_lo) tab.put(h, p, mytag);
      ip = step_end;
    }
  }
  if (FMT != EF_LZ4) return anchor;
  op = lz4_emit_tail(dst, op, cap, src + anchor, n - anchor, lane);
  if (op == 0xffffffffu) return 0u;
  return op < n ? op : 0u;
}
__device__ uint32_t lz4hc_encode_wave(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap, enc_entry_t* tab_generic, int lane) {
  return hc_encode_wave<EF_LZ4>(src, n, dst, cap, tab_generic, lane);
}

