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
#define BAMD_ENC_MINWAVES (BAMD_ENC_SPLIT_TAB ? 6 : 5)   // waves per SIMD the register allocator leaves room for
#endif
constexpr int ENC_LDS_WAVES = (160 * 1024) / ENC_TAB_BYTES;
constexpr int ENC_WAVES_PER_CU = ENC_LDS_WAVES < 4 * BAMD_ENC_MINWAVES ? ENC_LDS_WAVES : 4 * BAMD_ENC_MINWAVES;   // persistent grid size per CU

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
__device__ __forceinline__ uint32_t wave_common_fwd(const gu8* src, uint32_t n, uint32_t a, uint32_t b,
                                                    uint32_t maxlen, int lane) {
  uint32_t done = 0;
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

// write `v` as LZ4's 255-run length extension starting at p; returns bytes written
__device__ __forceinline__ uint32_t emit_ext255(gu8* p, uint32_t v, int lane) {
  const uint32_t n255 = v / 255u, rem = v - n255 * 255u;
  for (uint32_t k = (uint32_t)lane; k < n255; k += 64u) p[k] = 255u;
  if (lane == 0) p[n255] = (uint8_t)rem;
  return n255 + 1u;
}

enum { EF_LZ4 = 0, EF_BLOSCLZ = 1, EF_ZSTD = 2, EF_ZLIB = 3 };
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
    if ((uint32_t)lane < ll) dst[lane] = (uint8_t)v;
  } else {
    wave_copy_disjoint(dst, lit, ll, lane);
  }
}

__device__ __forceinline__ uint32_t lz4_emit_seq(gu8* dst, uint32_t op, uint32_t cap, const gu8* lit,
                                                 uint32_t ll, uint32_t off, uint32_t mlen, int lit_lane0, uint32_t ownbyte, int lane) {
  if (op + 1u + ll + (2u + 1u + 5u) + ll / 255u > cap) return 0xffffffffu;
  const uint32_t mcode = mlen - 4u;
  const uint32_t tok = ((ll < 15u ? ll : 15u) << 4) | (mcode < 15u ? mcode : 15u);
  if (lane == 0) dst[op] = (uint8_t)tok;
  op += 1u;
  if (ll >= 15u) op += emit_ext255(dst + op, ll - 15u, lane);
  emit_literals(dst + op, lit, ll, lit_lane0, ownbyte, lane);
  op += ll;
  if (lane < 2) dst[op + lane] = (uint8_t)(off >> (8 * lane));
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
constexpr uint32_t RANK_CAP = 20u;   // bytes of a candidate that are compared for ranking (16 instead: more extensions, lower ratio, no faster)

struct Bytes20 { uint64_t a, b; uint32_t c; };

// 20 bytes at src[pos..], zero beyond n (only the last step of a stream takes the slow branch)
__device__ __forceinline__ Bytes20 load20(const gu8* src, uint32_t pos, uint32_t n) {
  Bytes20 r;
  if (pos + 20u <= n) { r.a = g_ld8(src + pos); r.b = g_ld8(src + pos + 8u); r.c = g_ld4(src + pos + 16u); }
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
  uint32_t w0, w1, w2;
  uint32_t wbase;          // uniform, multiple of 256
  const gu8* src;
  uint32_t n;

  // dword at stream offset base + 4*lane; bytes at or beyond n read as zero.  Branch-free: a dword that
  // would cross the end is read at n-4 instead and shifted down (n >= 13 here).
  __device__ __forceinline__ uint32_t fetch(uint32_t base, int lane) const {
    const uint32_t off = base + 4u * (uint32_t)lane;
    const uint32_t a = off < n - 4u ? off : n - 4u;
    const uint32_t sh = off - a;                       // 0 in the body of the stream
    const uint32_t v = g_ld4(src + a);
    return sh < 4u ? v >> (8u * sh) : 0u;
  }
  __device__ __forceinline__ void init(const gu8* s, uint32_t n_, int lane) {
    src = s; n = n_; wbase = 0;
    w0 = fetch(0u, lane); w1 = fetch(256u, lane); w2 = fetch(512u, lane);
  }
  // make [lo, lo + 256 + 92) resident in w0/w1, lo = max(ip - 4, 0); ip only moves forward
  __device__ __forceinline__ void seek(uint32_t ip, int lane) {
    const uint32_t lo = ip >= 4u ? ip - 4u : 0u;
    const uint32_t d = lo - wbase;
    if (d < 256u) return;
    if (d < 512u) { w0 = w1; w1 = w2; wbase += 256u; w2 = fetch(wbase + 512u, lane); }
    else if (d < 768u) { w0 = w2; wbase += 512u; w1 = fetch(wbase + 256u, lane); w2 = fetch(wbase + 512u, lane); }
    else { wbase = lo & ~255u; w0 = fetch(wbase, lane); w1 = fetch(wbase + 256u, lane); w2 = fetch(wbase + 512u, lane); }
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
  if (FMT == EF_ZSTD || FMT == EF_ZLIB) { if (n < start + 16u) return start; }
  else if (FMT == EF_LZ4 ? (n < 13u) : (n < 16u || cap < 66u)) return 0u;
  const uint32_t last_start = n - 12u;                       // inclusive bound on match starts
  const uint32_t mlimit = (FMT == EF_LZ4) ? n - 5u : n - 2u;  // matches end at or before this position
  const int accel = 10 - clevel;                              // blosc/blosc.c:577-587
  // Effort knob in the spirit of the reference's clevel -> LZ4 acceleration mapping (blosc.c:577-587: lower
  // levels look at fewer positions) and of blosclz's tunable minimum match length (blosclz.c:445-457):
  // below clevel 9 a match must be longer than the format minimum to be taken.  4-byte matches are mostly accidental in noisy planes, save one
  // byte each and cost a full sequence: requiring 6 halves the encode time of noisy float64 data for
  // < 1 % of ratio (bench19: 53.3 -> 48.5, still far above the reference's 36.7 at this clevel).
  // (Zstd sequences are cheaper than LZ4's - a repeated distance costs 5 bits - so short matches pay off there.)
  static_assert(EF_ZSTD == 2, "");
  const uint32_t zmin = (uint32_t)__builtin_amdgcn_readfirstlane(BAMD_ZSTD_MINLEN);
  const uint32_t minlen = FMT == EF_ZSTD ? zmin : (clevel >= 9 ? 4u : (clevel >= 6 ? 5u : 6u));

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
    Bytes20 own;
    own.a = ((uint64_t)o1 << 32) | o0; own.b = ((uint64_t)o3 << 32) | o2; own.c = o4;
    // the two bytes before ip (uniform): r's lanes 0/1 hold them
    const uint64_t r01 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)r, 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)r, 0);
    const uint32_t before2 = bo0 >= 2u ? (uint32_t)(r01 >> (8u * (bo0 - 2u))) & 0xffffu : 0u;   // src[ip-2] | src[ip-1] << 8
    uint32_t prev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane - 1u) & 63u) << 2, (int)(o0 & 0xffu));
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
      if (d != 0u && d <= p && EncTable::tag_equal(e, mine) && (FMT != EF_ZLIB || d <= dfl::kMaxDist)) { cand = p - d; tab_ok = true; }
    } else {
      prev = 0x100u;
    }
    PROF_LAP(8); PROF_ADD(0, 1);
    // ---- round 2: candidate bytes, exact lengths up to RANK_CAP ----
    uint32_t len = 0;
    if (tab_ok) {
      const Bytes20 cb = load20(src, cand, n);
      len = common20(own, cb);
      if (len > limit) len = limit;
      if (len < minlen) len = 0;
      if (FMT == EF_BLOSCLZ && len < 6u && p - cand - 1u >= 8191u) len = 0;   // far and short (blosclz.c:535)
    }
    if (live && prev < 0x100u) {                    // distance 1: run of the previous byte
      uint32_t rl = runlen20(own, prev);
      if (rl > limit) rl = limit;
      if (rl >= minlen && rl > len) { len = rl; cand = p - 1u; }
    }
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
      // backward bytes are requested first and looked at last, so that they travel together with the
      // forward rows (one memory round trip for both directions)
      uint32_t bx = 0, by = 1;
      if ((uint32_t)lane < maxb) { bx = src[pm - 1u - (uint32_t)lane]; by = src[cm - 1u - (uint32_t)lane]; }
      asm volatile("" ::: "memory");
      uint32_t mlen = len_f;
      if (len_f == RANK_CAP && pm + RANK_CAP < mlimit)
        mlen += wave_common_fwd(src, n, pm + RANK_CAP, cm + RANK_CAP, mlimit - (pm + RANK_CAP), lane);
      const uint64_t bm = __ballot(bx != by);          // lanes >= maxb always vote "differs"
      const uint32_t back = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
      PROF_LAP(10); PROF_ADD(2, len_f == RANK_CAP); PROF_ADD(3, anchor < ip && pm > anchor); PROF_ADD(6, 1); PROF_ADD(7, back > 0); if (FMT != EF_ZSTD) { PROF_ADD(4, back > 4); PROF_ADD(5, maxb > 0); }
      pm -= back; cm -= back; mlen += back;
      const uint32_t ll = pm - anchor;
      const uint32_t dist = pm - cm;
      if (FMT == EF_LZ4) {
        op = lz4_emit_seq(dst, op, cap, src + anchor, ll, dist, mlen, anchor >= ip ? (int)(anchor - ip) : -1, (uint32_t)own.a & 0xffu, lane);
        if (op == 0xffffffffu) return 0u;
      } else if (FMT == EF_ZSTD) {
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
      if (adv > 16u) adv = 16u;
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
  if (FMT == EF_ZSTD || FMT == EF_ZLIB) return anchor;
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
    uint32_t prev = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane - 1u) & 63u) << 2, (int)(o0 & 0xffu));
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
        bool ok = live && d != 0u && d <= p && ((tt >> (8 * w)) & 0xffu) == mytag && (FMT != EF_ZLIB || d <= dfl::kMaxDist);
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
      if ((uint32_t)lane < maxb) { bx = src[pm - 1u - (uint32_t)lane]; by = src[cm - 1u - (uint32_t)lane]; }
      asm volatile("" ::: "memory");
      uint32_t mlen = len_f;
      if (!exact_f && pm + len_f < mlimit)
        mlen += wave_common_fwd(src, n, pm + len_f, cm + len_f, mlimit - (pm + len_f), lane);
      const uint64_t bm = __ballot(bx != by);          // lanes >= maxb always vote "differs"
      const uint32_t back = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
      pm -= back; cm -= back; mlen += back;
      if (FMT == EF_LZ4) {
        op = lz4_emit_seq(dst, op, cap, src + anchor, pm - anchor, pm - cm, mlen, anchor >= ip ? (int)(anchor - ip) : -1, (uint32_t)own.a & 0xffu, lane);
        if (op == 0xffffffffu) return 0u;
      } else if (FMT == EF_ZSTD) {
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

// ---------------------------------------------------------------------------------------------
// Zstd frames (zstd_enc.h has the format; this is its wave-parallel use).  One frame per stream, blocks of at most
// 128 KiB; per block the match finder above fills a ZsSink, then the sequence section is coded: code numbers and extra
// bits of 64 sequences at a time in the lanes, the three FSE state chains and the bit writer as a wave-uniform
// (scalar) loop over them, tables in the wave's LDS (the hash table is rebuilt per stream anyway).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zs_emit_seq(ZsSink& z, const gu8* lit, uint32_t ll, uint32_t off, uint32_t mlen, int lit_lane0, uint32_t ownbyte, int lane) {
  if (z.nlit + ll > z.litcap || z.nseq >= z.seqcap) return 0xffffffffu;
  emit_literals(z.lit + z.nlit, lit, ll, lit_lane0, ownbyte, lane);
  if (lane == 0) z.seq[z.nseq] = zenc::pack_seq(ll, mlen, off);
  z.nlit += ll; z.nseq++;
  return 0u;
}

// distances -> Offset_Values (repeat codes, zstd_enc.h: rep_value), in stream order: 64 sequences per load, the history
// as wave-uniform state
__device__ __forceinline__ void zs_assign_offset_values(BAMD_GAS uint64_t* seqs, uint32_t nseq, zenc::RepState& rep, int lane) {
  for (uint32_t base = 0; base < nseq; base += 64u) {
    const uint32_t cnt = nseq - base < 64u ? nseq - base : 64u;
    const uint64_t q = (uint32_t)lane < cnt ? seqs[base + (uint32_t)lane] : 0ull;
    const uint32_t off_l = zenc::seq_off(q), ll_l = zenc::seq_ll(q);
    uint32_t val_l = 0;
    for (uint32_t k = 0; k < cnt; k++) {
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)off_l, (int)k), ll = (uint32_t)__builtin_amdgcn_readlane((int)ll_l, (int)k);
      const uint32_t v = zenc::rep_value(rep, off, ll);
      val_l = (uint32_t)lane == k ? v : val_l;
    }
    if ((uint32_t)lane < cnt) seqs[base + (uint32_t)lane] = zenc::pack_seq(ll_l, zenc::seq_ml(q), val_l);
  }
}

typedef BAMD_LAS const zenc::CTab LdsCTab;
__device__ __forceinline__ uint32_t zs_tab_u32(const BAMD_LAS uint32_t* p, uint32_t i) { return uni(p[i]); }

// out: start of the Sequences_Section, room: bytes available.  Returns the section size or 0xffffffff.
__device__ __forceinline__ uint32_t zs_write_sequences(gu8* out, uint32_t room, const BAMD_GAS uint64_t* seqs, uint32_t nseq,
                                                       const BAMD_LAS zenc::CTabs* T, int lane) {
  if (room < 8u) return 0xffffffffu;
  uint32_t pos = 0;
  if (nseq == 0u) { if (lane == 0) out[0] = 0; return 1u; }
  if (nseq < 128u) { if (lane == 0) out[0] = (uint8_t)nseq; pos = 1; }
  else if (nseq < 0x7f00u) { if (lane == 0) { out[0] = (uint8_t)((nseq >> 8) + 128u); out[1] = (uint8_t)nseq; } pos = 2; }
  else { if (lane == 0) { out[0] = 255u; out[1] = (uint8_t)(nseq - 0x7f00u); out[2] = (uint8_t)((nseq - 0x7f00u) >> 8); } pos = 3; }
  if (lane == 0) out[pos] = 0;                               // three predefined tables
  pos += 1;
  const BAMD_LAS uint32_t* ll_dnb = (const BAMD_LAS uint32_t*)T->ll.dnb; const BAMD_LAS int32_t* ll_dfs = (const BAMD_LAS int32_t*)T->ll.dfs; const BAMD_LAS uint16_t* ll_st = (const BAMD_LAS uint16_t*)T->ll.st;
  const BAMD_LAS uint32_t* ml_dnb = (const BAMD_LAS uint32_t*)T->ml.dnb; const BAMD_LAS int32_t* ml_dfs = (const BAMD_LAS int32_t*)T->ml.dfs; const BAMD_LAS uint16_t* ml_st = (const BAMD_LAS uint16_t*)T->ml.st;
  const BAMD_LAS uint32_t* of_dnb = (const BAMD_LAS uint32_t*)T->of.dnb; const BAMD_LAS int32_t* of_dfs = (const BAMD_LAS int32_t*)T->of.dfs; const BAMD_LAS uint16_t* of_st = (const BAMD_LAS uint16_t*)T->of.st;
  uint64_t acc = 0; uint32_t nb = 0; bool ovf = false;
  auto add = [&](uint32_t v, uint32_t n) {
    acc |= (uint64_t)v << nb; nb += n;
    if (nb >= 32u) {
      if (pos + 4u > room) ovf = true; else if (lane == 0) g_st4(out + pos, (uint32_t)acc);
      pos += 4u; acc >>= 32; nb -= 32u;
    }
  };
  uint32_t sll = 0, sml = 0, sof = 0;
  bool first = true;
  for (uint32_t base = ((nseq - 1u) >> 6) << 6;; base -= 64u) {
    const uint32_t cnt = nseq - base < 64u ? nseq - base : 64u;
    const uint64_t q = (uint32_t)lane < cnt ? seqs[base + (uint32_t)lane] : zenc::pack_seq(0, 3, 4);
    const zenc::Code l = zenc::ll_code(zenc::seq_ll(q)), m = zenc::ml_code(zenc::seq_ml(q)), o = zenc::of_code_value(zenc::seq_off(q));
    // everything that does not depend on the FSE states is prepared per lane, 64 sequences at once: the table rows of the
    // sequence's three codes and its extra bits as ONE field (literal-length | match-length | offset bits, <= 49 bits).
    // The serial loop below - a scalar program, and all waves of a CU share one scalar unit - only walks the states.
    const uint32_t dl_v = ll_dnb[l.code], dm_v = ml_dnb[m.code], do_v = of_dnb[o.code];
    const uint32_t nbx_v = l.bits + m.bits + o.bits;
    const uint32_t fpk_v = ((uint32_t)ll_dfs[l.code] & 0xffu) | (((uint32_t)ml_dfs[m.code] & 0xffu) << 8) | (((uint32_t)of_dfs[o.code] & 0xffu) << 16) | (nbx_v << 24);
    const uint64_t ext_v = (uint64_t)l.extra | ((uint64_t)m.extra << l.bits) | ((uint64_t)o.extra << (l.bits + m.bits));
    const uint32_t exl_v = (uint32_t)ext_v, exh_v = (uint32_t)(ext_v >> 32);
    for (int k = (int)cnt - 1; k >= 0; k--) {
      const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)dl_v, k), dm = (uint32_t)__builtin_amdgcn_readlane((int)dm_v, k);
      const uint32_t dO = (uint32_t)__builtin_amdgcn_readlane((int)do_v, k), fpk = (uint32_t)__builtin_amdgcn_readlane((int)fpk_v, k);
      const uint32_t exl = (uint32_t)__builtin_amdgcn_readlane((int)exl_v, k), exh = (uint32_t)__builtin_amdgcn_readlane((int)exh_v, k);
      const int32_t fl = (int32_t)(int8_t)(fpk & 0xffu), fm = (int32_t)(int8_t)((fpk >> 8) & 0xffu), fo = (int32_t)(int8_t)((fpk >> 16) & 0xffu);
      const uint32_t nbx = fpk >> 24;
      if (first) {
        first = false;
        const uint32_t nm = (dm + (1u << 15)) >> 16, nO = (dO + (1u << 15)) >> 16, nl = (dl + (1u << 15)) >> 16;
        sml = uni((uint32_t)ml_st[(int32_t)(((nm << 16) - dm) >> nm) + fm]);
        sof = uni((uint32_t)of_st[(int32_t)(((nO << 16) - dO) >> nO) + fo]);
        sll = uni((uint32_t)ll_st[(int32_t)(((nl << 16) - dl) >> nl) + fl]);
      } else {
        // the three state transitions: their bits (<= 5 + 6 + 6) go out as one field
        const uint32_t nO = (sof + dO) >> 16, nm = (sml + dm) >> 16, nl = (sll + dl) >> 16;
        const uint32_t bitsv = (sof & ((1u << nO) - 1u)) | ((sml & ((1u << nm) - 1u)) << nO) | ((sll & ((1u << nl) - 1u)) << (nO + nm));
        const uint32_t nxo = uni((uint32_t)of_st[(int32_t)(sof >> nO) + fo]), nxm = uni((uint32_t)ml_st[(int32_t)(sml >> nm) + fm]);
        const uint32_t nxl = uni((uint32_t)ll_st[(int32_t)(sll >> nl) + fl]);
        add(bitsv, nO + nm + nl);
        sof = nxo; sml = nxm; sll = nxl;
      }
      if (nbx > 24u) { add(exl & 0xffffffu, 24u); add((exl >> 24) | (exh << 8), nbx - 24u); }      // <= 49 bits: two pieces of <= 25
      else add(exl, nbx);
    }
    if (base == 0u) break;
  }
  add(sml & 63u, (uint32_t)zenc::kMLLog); add(sof & 31u, (uint32_t)zenc::kOFLog); add(sll & 63u, (uint32_t)zenc::kLLLog);
  add(1u, 1u);
  while (nb > 0u) {
    if (pos >= room) { ovf = true; break; }
    if (lane == 0) out[pos] = (uint8_t)acc;
    pos++; acc >>= 8; nb = nb > 8u ? nb - 8u : 0u;
  }
  return ovf ? 0xffffffffu : pos;
}

// ---------------------------------------------------------------------------------------------
// The same section, written without the scalar unit (BAMD_ZSTD_VSEQ, default).  The scalar version above costs ~60 SALU
// instructions per sequence, and the 20 waves of a CU share ONE scalar unit: 262 M sequences per 8 GiB of bench19 were
// ~30 of the kernel's 42 ms.  Here, per batch of 64 sequences:
//   * every lane prepares its sequence (codes, table rows, extra bits) as before;
//   * the three FSE state chains run on THREE LANES (0: literal lengths, 1: match lengths, 2: offsets), vector code: per
//     step one dependent LDS read (the next state); the bits each transition emits go to an LDS array;
//   * all 64 lanes then place their sequence's bits - state bits, then extra bits, <= 66 per sequence - with a prefix sum
//     over the bit counts (in stream order: last sequence first), OR them into an LDS strip and store the full dwords.
// Bit for bit the output of zenc::write_sequences (tests: every frame is read by ZSTD_decompress and by the oracle).
// Scratch: 64 x 3 transition words + a 136-dword strip, taken from the START of this wave's hash table - the section is
// written after the block's match finding; a later block of the same stream finds some stale entries there, which the
// candidate check (tag, then bytes) rejects like any other stale entry.
// ---------------------------------------------------------------------------------------------
#ifndef BAMD_ZSTD_VSEQ
#define BAMD_ZSTD_VSEQ 1
#endif
constexpr uint32_t ZV_STRIP = 136u;

// OR the low n (<= 49) bits of v into the strip at bit position bitpos
__device__ __forceinline__ void zv_or_bits(volatile BAMD_LAS uint32_t* strip, uint32_t bitpos, uint64_t v, uint32_t n) {
  if (n == 0u) return;
  const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
  const uint64_t lo = v << sh;
  __hip_atomic_fetch_or((BAMD_LAS uint32_t*)strip + w, (uint32_t)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  if (sh + n > 32u) __hip_atomic_fetch_or((BAMD_LAS uint32_t*)strip + w + 1u, (uint32_t)(lo >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  if (sh + n > 64u) __hip_atomic_fetch_or((BAMD_LAS uint32_t*)strip + w + 2u, (uint32_t)(v >> (64u - sh)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// ---------------------------------------------------------------------------------------------
// Per-block sequence tables (zstd_enc.h: "per-block tables"; kernel k_encode_streams_t<ENC_ZSTD_T>).  The predefined
// distributions are made for text-like sequences; byte planes of numeric data have a handful of literal-length and match-length
// codes and two or three offset codes, and coding them with tables made for the block is worth +29 % ratio on the SURVEY 8d
// planes (+67 % on linspace; tests/tools/zstd_enc_cpu.cpp measures it on the CPU with the same format functions).
// Per block, after the match finder: histogram of the three code alphabets (LDS atomics), then per alphabet with one lane
// per symbol: RLE when a single code occurs; else probabilities normalised to 64 (zenc::fse_normalize's rule), the table
// description written with a prefix sum over the field widths, its cost compared with the predefined table's, and the
// encoder table built with one lane per CELL - with Accuracy_Log 6 and no "less than 1" probabilities the k-th cell
// handed out lies at (43 k) & 63, so the spread, its inverse (k = 3 u & 63) and every cell's rank inside its symbol need no
// serial pass.  Everything lives in the scratch at the start of this wave's hash table, like the writer's strips.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ZT_HIST = 328u, ZT_CTABS = 520u, ZT_DESC = 934u, ZT_END = 970u;     // dword offsets in the wave's scratch
static_assert(sizeof(zenc::CTabs) == 1656 && ZT_CTABS + sizeof(zenc::CTabs) / 4 == ZT_DESC && ZT_END * 4u <= (uint32_t)ENC_TAB_BYTES, "scratch layout");
static_assert(zenc::kCustomLog == 6, "one lane per cell");
struct ZsTabs {                  // wave-uniform; index 0 literal lengths, 1 offsets, 2 match lengths (the order of the modes byte)
  uint32_t mode[3], rle[3], log[3], desc_len[3];
};
// the predefined distributions (RFC 8878 3.1.1.3.2.2.1), one row per alphabet in ZsTabs order, -1 = "less than 1"
__device__ const int8_t kZtPredef[3][64] = {
  {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1},
  {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1},
  {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1}};
// -log2(n / 2^log) in 1/256 bits, as zenc::fse_cost256 counts it
__device__ __forceinline__ uint32_t zt_bits256(uint32_t n, uint32_t log) {
  const uint32_t hbit = 31u - (uint32_t)__builtin_clz(n);
  const uint32_t frac = ((n << 8) >> hbit) - 256u;
  return ((log - hbit) << 8) - frac;
}
// One alphabet.  `c`: this lane's symbol count (0 for lanes beyond the alphabet).  Fills zt.* for index a and, in FSE mode, the
// encoder table `ct` and the description at desc[0 .. 12).  Returns nothing; all decisions are wave-uniform.
__device__ __forceinline__ void zt_make_table(uint32_t c, uint32_t total, int a, uint32_t predef_log, ZsTabs& zt, BAMD_LAS zenc::CTab* ct,
                                              volatile BAMD_LAS uint32_t* desc, int lane) {
  zt.mode[a] = zenc::kModePredefined; zt.log[a] = predef_log; zt.rle[a] = 0u; zt.desc_len[a] = 0u;
  const uint64_t present = __ballot(c != 0u);
  if (present == 0ull) return;
  if ((present & (present - 1ull)) == 0ull) { zt.mode[a] = zenc::kModeRLE; zt.rle[a] = (uint32_t)__builtin_ctzll(present); zt.log[a] = 0u; return; }
  // ---- probabilities: floor of the proportional share, at least 1; the rest to the most frequent symbol ----
  uint32_t v = c ? (c << 6) / total : 0u;
  if (c && v == 0u) v = 1u;
  const uint32_t sum = wave_sum_u32(v);
  const int big = 63 - (int)(wave_max_u32((c << 6) | (63u - (uint32_t)lane)) & 63u);       // first lane among the largest counts
  if (sum <= 64u) { if (lane == big) v += 64u - sum; }
  else {
    for (uint32_t over = sum - 64u; over > 0u; over--) {
      const uint32_t best = wave_max_u32((v << 6) | (63u - (uint32_t)lane));
      if ((best >> 6) < 2u) return;                                  // cannot be normalised: predefined
      if (lane == 63 - (int)(best & 63u)) v--;
    }
  }
  const uint64_t nz = __ballot(v != 0u);
  const int last = 63 - __builtin_clzll(nz);
  const uint32_t incl = wave_incl_scan_u32(v, lane);
  const uint32_t excl = incl - v;                                    // cells handed out before this symbol
  // ---- cost with this table against the predefined one ----
  const int32_t pn = (int32_t)kZtPredef[a][lane];
  const uint32_t cost_pre = wave_sum_u32(c ? c * zt_bits256(pn < 0 ? 1u : (uint32_t)pn, predef_log) : 0u);
  uint32_t cost_new = wave_sum_u32(c ? c * zt_bits256(v, 6u) : 0u);
  // ---- the description: 4 bits log - 5, then per symbol up to `last` a field whose width follows from the points still left ----
  uint64_t field = 0; uint32_t width = 0;
  if (lane <= last) {
    const uint32_t remaining = 64u - excl;
    const uint32_t value = v + 1u;
    const uint32_t bits = (31u - (uint32_t)__builtin_clz(remaining + 1u)) + 1u;
    const uint32_t low = (1u << bits) - 1u - (remaining + 1u);
    if (value < low) { field = value; width = bits - 1u; }
    else if (value < (1u << (bits - 1u))) { field = value; width = bits; }
    else { field = value + low; width = bits; }
  }
  const uint32_t vprev = (uint32_t)__shfl_up((int)v, 1, 64);
  if (lane <= last && v == 0u) {
    if (lane != 0 && vprev == 0u) { field = 0; width = 0; }          // inside a run of absent symbols: counted by its first one
    else {
      const uint32_t next = (uint32_t)lane + 1u + (uint32_t)__builtin_ctzll(nz >> ((uint32_t)lane + 1u));   // lane < last here
      const uint32_t z = next - (uint32_t)lane - 1u;                 // further zeros behind this one
      const uint32_t threes = z / 3u;
      const uint64_t flags = ((1ull << (2u * threes)) - 1ull) | ((uint64_t)(z - 3u * threes) << (2u * threes));
      field |= flags << width; width += 2u * (threes + 1u);
    }
  }
  if (lane == 0) { field = (field << 4) | (uint64_t)(6u - 5u); width += 4u; }
  const uint32_t wincl = wave_incl_scan_u32(width, lane);
  const uint32_t dbits = (uint32_t)__builtin_amdgcn_readlane((int)wincl, 63);
  const uint32_t dlen = (dbits + 7u) >> 3;
  cost_new += dlen << 11;
  if (cost_new >= cost_pre) return;                                  // the predefined table is cheaper (short blocks)
  if (lane < 12) desc[lane] = 0u;
  BAMD_LDS_SYNC();
  zv_or_bits(desc, wincl - width, field, width);
  // ---- encoder table (zenc::build_ctab): per symbol deltaNbBits / deltaFindState ... ----
  BAMD_LAS uint32_t* dnb = (BAMD_LAS uint32_t*)ct->dnb; BAMD_LAS int32_t* dfs = (BAMD_LAS int32_t*)ct->dfs; BAMD_LAS uint16_t* st = (BAMD_LAS uint16_t*)ct->st;
  if (lane < 53) {
    if (v == 0u) { dnb[lane] = (7u << 16) - 64u; dfs[lane] = 0; }
    else if (v == 1u) { dnb[lane] = (6u << 16) - 64u; dfs[lane] = (int32_t)excl - 1; }
    else {
      const uint32_t maxbits = 6u - (31u - (uint32_t)__builtin_clz(v - 1u));
      dnb[lane] = (maxbits << 16) - (v << maxbits);
      dfs[lane] = (int32_t)excl - (int32_t)v;
    }
  }
  // ---- ... and per cell the coded state: lane u is cell u; it was the k-th cell handed out, k = 3 u & 63 (43 * 3 = 129) ----
  // symbol of hand-out index k: the last symbol whose first index is <= k
  uint32_t mark = 0;
  {
    // lane k learns the symbol that starts at index k (if any): scatter through the cross-lane network, one symbol at a time
    // would be serial; instead every lane k counts the symbols whose range starts at or before k
    uint32_t sy = 0;
    for (int sidx = 0; sidx <= last; sidx++) {
      const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)excl, sidx), n = (uint32_t)__builtin_amdgcn_readlane((int)v, sidx);
      if (n != 0u && e <= (uint32_t)lane) sy = (uint32_t)sidx;
    }
    mark = sy;                                                       // symbol of hand-out index `lane`
  }
  const uint32_t k = (3u * (uint32_t)lane) & 63u;
  const uint32_t S = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(k << 2), (int)mark);                 // symbol in cell `lane`
  const uint32_t cS = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(S << 2), (int)excl);
  const uint32_t nS = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(S << 2), (int)v);
  uint32_t rank = 0;
  for (uint32_t j = 0; j < nS; j++) rank += ((((cS + j) * 43u) & 63u) < (uint32_t)lane) ? 1u : 0u;
  st[cS + rank] = (uint16_t)(64u + (uint32_t)lane);
  BAMD_LDS_SYNC();
  zt.mode[a] = zenc::kModeFSE; zt.log[a] = 6u; zt.desc_len[a] = dlen;
}

// histogram + the three tables of one block; `pre`: the predefined tables, `scr`: the wave's scratch
__device__ __forceinline__ void zt_make_tables(const BAMD_GAS uint64_t* seqs, uint32_t nseq, volatile BAMD_LAS uint32_t* scr, ZsTabs& zt, int lane) {
  volatile BAMD_LAS uint32_t* hist = scr + ZT_HIST;
  BAMD_LDS_SYNC();                                     // the scratch overlays the match finder's table: every lane is done with that
  hist[lane] = 0u; hist[lane + 64] = 0u; hist[lane + 128] = 0u;
  BAMD_LDS_SYNC();
  for (uint32_t base = 0; base < nseq; base += 64u) {
    if (base + (uint32_t)lane < nseq) {
      const uint64_t q = seqs[base + (uint32_t)lane];
      __hip_atomic_fetch_add((BAMD_LAS uint32_t*)hist + zenc::ll_code(zenc::seq_ll(q)).code, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add((BAMD_LAS uint32_t*)hist + 64u + zenc::of_code_value(zenc::seq_off(q)).code, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add((BAMD_LAS uint32_t*)hist + 128u + zenc::ml_code(zenc::seq_ml(q)).code, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  BAMD_LDS_SYNC();
  BAMD_LAS zenc::CTabs* C = (BAMD_LAS zenc::CTabs*)((BAMD_LAS uint32_t*)scr + ZT_CTABS);
  zt_make_table(lane < zenc::kLLSyms ? hist[lane] : 0u, nseq, 0, (uint32_t)zenc::kLLLog, zt, &C->ll, scr + ZT_DESC, lane);
  zt_make_table(lane < 32 ? hist[64 + lane] : 0u, nseq, 1, (uint32_t)zenc::kOFLog, zt, &C->of, scr + ZT_DESC + 12, lane);
  zt_make_table(lane < zenc::kMLSyms ? hist[128 + lane] : 0u, nseq, 2, (uint32_t)zenc::kMLLog, zt, &C->ml, scr + ZT_DESC + 24, lane);
}
template <bool TABLES>
__device__ __forceinline__ uint32_t zs_write_sequences_v(gu8* out, uint32_t room, const BAMD_GAS uint64_t* seqs, uint32_t nseq,
                                                         const BAMD_LAS zenc::CTabs* T, volatile BAMD_LAS uint32_t* scr, int lane, const ZsTabs* ztp = nullptr) {
  if (room < 8u) return 0xffffffffu;
  uint32_t pos = 0;
  if (nseq == 0u) { if (lane == 0) out[0] = 0; return 1u; }
  if (nseq < 128u) { if (lane == 0) out[0] = (uint8_t)nseq; pos = 1; }
  else if (nseq < 0x7f00u) { if (lane == 0) { out[0] = (uint8_t)((nseq >> 8) + 128u); out[1] = (uint8_t)nseq; } pos = 2; }
  else { if (lane == 0) { out[0] = 255u; out[1] = (uint8_t)(nseq - 0x7f00u); out[2] = (uint8_t)((nseq - 0x7f00u) >> 8); } pos = 3; }
  // the block's three tables: the predefined ones, or (TABLES) what zt_make_tables chose - index 0 literal lengths, 1 offsets, 2 match lengths
  const BAMD_LAS zenc::CTab* tl = &T->ll; const BAMD_LAS zenc::CTab* tm = &T->ml; const BAMD_LAS zenc::CTab* to = &T->of;
  uint32_t log_l = (uint32_t)zenc::kLLLog, log_m = (uint32_t)zenc::kMLLog, log_o = (uint32_t)zenc::kOFLog;
  bool my_rle = false;                                       // this chain lane's alphabet has one symbol: no state, no bits
  if (TABLES) {
    const ZsTabs& zt = *ztp;
    if (lane == 0) out[pos] = (uint8_t)((zt.mode[0] << 6) | (zt.mode[1] << 4) | (zt.mode[2] << 2));      // Symbol_Compression_Modes
    pos += 1;
    const BAMD_LAS zenc::CTabs* C = (const BAMD_LAS zenc::CTabs*)((BAMD_LAS uint32_t*)scr + ZT_CTABS);
    uint32_t extra = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) extra += zt.mode[a] == zenc::kModeRLE ? 1u : (zt.mode[a] == zenc::kModeFSE ? zt.desc_len[a] : 0u);
    if (pos + extra + 8u > room) return 0xffffffffu;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      if (zt.mode[a] == zenc::kModeRLE) { if (lane == 0) out[pos] = (uint8_t)zt.rle[a]; pos += 1u; }
      else if (zt.mode[a] == zenc::kModeFSE) {
        const uint32_t w = scr[ZT_DESC + 12u * (uint32_t)a + ((uint32_t)lane >> 2)];
        if ((uint32_t)lane < zt.desc_len[a]) out[pos + (uint32_t)lane] = (uint8_t)(w >> (8u * ((uint32_t)lane & 3u)));
        pos += zt.desc_len[a];
      }
    }
    if (zt.mode[0] == zenc::kModeFSE) tl = &C->ll;
    if (zt.mode[1] == zenc::kModeFSE) to = &C->of;
    if (zt.mode[2] == zenc::kModeFSE) tm = &C->ml;
    log_l = zt.log[0]; log_o = zt.log[1]; log_m = zt.log[2];
    my_rle = (lane == 0 ? zt.mode[0] : (lane == 1 ? zt.mode[2] : zt.mode[1])) == zenc::kModeRLE;
  } else {
    if (lane == 0) out[pos] = 0;                             // three predefined tables
    pos += 1;
  }
  const BAMD_LAS uint32_t* ll_dnb = (const BAMD_LAS uint32_t*)tl->dnb; const BAMD_LAS int32_t* ll_dfs = (const BAMD_LAS int32_t*)tl->dfs;
  const BAMD_LAS uint32_t* ml_dnb = (const BAMD_LAS uint32_t*)tm->dnb; const BAMD_LAS int32_t* ml_dfs = (const BAMD_LAS int32_t*)tm->dfs;
  const BAMD_LAS uint32_t* of_dnb = (const BAMD_LAS uint32_t*)to->dnb; const BAMD_LAS int32_t* of_dfs = (const BAMD_LAS int32_t*)to->dfs;
  // this lane's chain (lanes 0 / 1 / 2): its state table
  const BAMD_LAS uint16_t* my_st = lane == 0 ? (const BAMD_LAS uint16_t*)tl->st : (lane == 1 ? (const BAMD_LAS uint16_t*)tm->st : (const BAMD_LAS uint16_t*)to->st);
  volatile BAMD_LAS uint32_t* trans = scr;                   // [64][3]: bits | count << 16 of every state transition
  volatile BAMD_LAS uint32_t* strip = scr + 192;             // [ZV_STRIP]
  uint32_t state = 0;                                        // lanes 0..2
  uint32_t pend = 0, npend = 0;                              // bits not yet stored (< 32), wave-uniform
  bool first = true, ovf = false;
  for (uint32_t base = ((nseq - 1u) >> 6) << 6;; base -= 64u) {
    const uint32_t cnt = nseq - base < 64u ? nseq - base : 64u;
    const uint64_t q = (uint32_t)lane < cnt ? seqs[base + (uint32_t)lane] : zenc::pack_seq(0, 3, 4);
    const zenc::Code l = zenc::ll_code(zenc::seq_ll(q)), m = zenc::ml_code(zenc::seq_ml(q)), o = zenc::of_code_value(zenc::seq_off(q));
    const uint32_t dl_v = ll_dnb[l.code], dm_v = ml_dnb[m.code], do_v = of_dnb[o.code];
    const uint32_t nbx = l.bits + m.bits + o.bits;
    const uint32_t fpk_v = ((uint32_t)ll_dfs[l.code] & 0xffu) | (((uint32_t)ml_dfs[m.code] & 0xffu) << 8) | (((uint32_t)of_dfs[o.code] & 0xffu) << 16);
    const uint64_t ext = (uint64_t)l.extra | ((uint64_t)m.extra << l.bits) | ((uint64_t)o.extra << (l.bits + m.bits));
    // ---- the three chains, last sequence of the batch first ----
    for (int k = (int)cnt - 1; k >= 0; k--) {
      const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)dl_v, k), dm = (uint32_t)__builtin_amdgcn_readlane((int)dm_v, k);
      const uint32_t dO = (uint32_t)__builtin_amdgcn_readlane((int)do_v, k), fpk = (uint32_t)__builtin_amdgcn_readlane((int)fpk_v, k);
      if (TABLES && my_rle) { if (lane < 3) trans[3 * k + lane] = 0u; }
      else if (lane < 3) {
        const uint32_t d = lane == 0 ? dl : (lane == 1 ? dm : dO);
        const int32_t f = (int32_t)(int8_t)((fpk >> (8u * (uint32_t)lane)) & 0xffu);
        if (first) {
          const uint32_t nb = (d + (1u << 15)) >> 16;
          state = my_st[(int32_t)(((nb << 16) - d) >> nb) + f];
          trans[3 * k + lane] = 0u;
        } else {
          const uint32_t nb = (state + d) >> 16;
          trans[3 * k + lane] = (state & ((1u << nb) - 1u)) | (nb << 16);
          state = my_st[(int32_t)(state >> nb) + f];
        }
      }
      first = false;
    }
    // ---- placement: per sequence [offset-state bits | match-length-state bits | literal-length-state bits | extra bits] ----
    BAMD_LDS_SYNC();
    uint32_t a_bits = 0, a_n = 0;
    if ((uint32_t)lane < cnt) {
      const uint32_t tl = trans[3 * lane], tm = trans[3 * lane + 1], to = trans[3 * lane + 2];
      const uint32_t nO = to >> 16, nm = tm >> 16, nl = tl >> 16;
      a_bits = (to & 0xffffu) | ((tm & 0xffffu) << nO) | ((tl & 0xffffu) << (nO + nm));
      a_n = nO + nm + nl;
    }
    const uint32_t mylen = (uint32_t)lane < cnt ? a_n + nbx : 0u;
    const uint32_t incl = wave_incl_scan_u32(mylen, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    strip[lane] = lane == 0 ? pend : 0u; strip[lane + 64] = 0u;
    if (lane < (int)ZV_STRIP - 128) strip[lane + 128] = 0u;
    BAMD_LDS_SYNC();
    if (mylen) {
      const uint32_t at = npend + (total - incl);            // stream order: the batch's last sequence first
      zv_or_bits(strip, at, (uint64_t)a_bits, a_n);
      zv_or_bits(strip, at + a_n, ext, nbx);
    }
    const uint32_t fill = npend + total, ndw = fill >> 5;     // <= 133 full dwords
    if (pos + 4u * ndw + 8u > room) { ovf = true; break; }
    BAMD_LDS_SYNC();
#pragma unroll
    for (uint32_t i = 0; i < 3u; i++) { const uint32_t w = (uint32_t)lane + 64u * i; if (w < ndw) g_st4(out + pos + 4u * w, strip[w]); }
    pend = uni(strip[ndw]); npend = fill & 31u; pos += 4u * ndw;
    if (base == 0u) break;
  }
  if (ovf) return 0xffffffffu;
  // final states (match length, offset, literal length), the mark bit, the pending bytes
  const uint32_t sll = (uint32_t)__builtin_amdgcn_readlane((int)state, 0), sml = (uint32_t)__builtin_amdgcn_readlane((int)state, 1), sof = (uint32_t)__builtin_amdgcn_readlane((int)state, 2);
  uint64_t acc = (uint64_t)pend; uint32_t nb = npend;
  if (TABLES) {
    acc |= (uint64_t)(sml & ((1u << log_m) - 1u)) << nb; nb += log_m;
    acc |= (uint64_t)(sof & ((1u << log_o) - 1u)) << nb; nb += log_o;
    acc |= (uint64_t)(sll & ((1u << log_l) - 1u)) << nb; nb += log_l;
  } else {
  acc |= (uint64_t)(sml & 63u) << nb; nb += (uint32_t)zenc::kMLLog;
  acc |= (uint64_t)(sof & 31u) << nb; nb += (uint32_t)zenc::kOFLog;
  acc |= (uint64_t)(sll & 63u) << nb; nb += (uint32_t)zenc::kLLLog;
  }
  acc |= 1ull << nb; nb += 1u;
  const uint32_t nbytes = (nb + 7u) >> 3;                    // <= 7
  if (pos + nbytes > room) return 0xffffffffu;
  if ((uint32_t)lane < nbytes) out[pos + (uint32_t)lane] = (uint8_t)(acc >> (8u * (uint32_t)lane));
  return pos + nbytes;
}

// One stream -> one frame.  Returns the frame size, or 0 when it would not be smaller than the input (the split is
// then stored raw by blosc's own rule, blosc.c:703-717).  `seqbuf`: zenc::kBlockMax / 4 entries of this wave.
constexpr uint32_t ZS_SEQCAP = zenc::kBlockMax / 4u;
constexpr int ZS_LDS_BYTES = (int)((sizeof(zenc::CTabs) + 15) / 16 * 16);
template <bool TABLES = false, bool HC = false>      // HC: the LZ4HC-grade search (hc_encode_wave) as the match finder, its 24 KiB table in front of the FSE tables
__device__ uint32_t zstd_encode_wave(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap, int clevel,
                                     enc_entry_t* tab_generic, BAMD_GAS uint64_t* seqbuf, int lane EPROF_ARG) {
  if (n < 32u || cap < 64u) return 0u;
  if (lane == 0) {
    uint8_t h[16];
    zenc::write_frame_header(h, n);
    for (int i = 0; i < 9; i++) dst[i] = h[i];
  }
  uint32_t op = zenc::kFrameHeader;
  zenc::RepState rep;
  zenc::rep_init(rep);
  for (uint32_t s0 = 0; s0 < n; s0 += zenc::kBlockMax) {
    const uint32_t s1 = s0 + zenc::kBlockMax < n ? s0 + zenc::kBlockMax : n;
    const bool last = s1 == n;
    const uint32_t seg = s1 - s0;
    if (op + zenc::kBlockHeader + zenc::kLitHeader + 16u >= cap) return 0u;
    gu8* bh = dst + op;
    ZsSink z;
    z.lit = bh + zenc::kBlockHeader + zenc::kLitHeader; z.nlit = 0;
    z.litcap = cap - (op + zenc::kBlockHeader + zenc::kLitHeader);
    z.seq = seqbuf; z.nseq = 0; z.seqcap = ZS_SEQCAP;
    const uint32_t covered = HC ? hc_encode_wave<EF_ZSTD>(src, s1, dst, cap, tab_generic, lane, s0, &z)
                                : lz_encode_wave<EF_ZSTD>(src, s1, dst, cap, clevel, tab_generic, lane EPROF_PASS, s0, &z);
    uint32_t bsize = 0xffffffffu;
    const zenc::RepState rep_before = rep;
    if (covered != 0xffffffffu && z.nlit + (s1 - covered) <= z.litcap) {
      wave_copy_disjoint(z.lit + z.nlit, src + covered, s1 - covered, lane);
      z.nlit += s1 - covered;
      if (lane == 0) { uint8_t h[4]; zenc::write_raw_literals_header(h, z.nlit); bh[3] = h[0]; bh[4] = h[1]; bh[5] = h[2]; }
      // the FSE tables sit behind this wave's hash table in LDS (k_encode_streams_t<true> puts them there once)
      const BAMD_LAS zenc::CTabs* T = (const BAMD_LAS zenc::CTabs*)((BAMD_LAS uint8_t*)(void*)tab_generic + (HC ? HC_TAB_BYTES : ENC_TAB_BYTES));
      __builtin_amdgcn_s_waitcnt(0);      // this wave's sequence triples are in memory before other lanes load them
      zs_assign_offset_values(seqbuf, z.nseq, rep, lane);
      __builtin_amdgcn_s_waitcnt(0);
      PROF_LAP(4);                        // Zstd: slot 4 = tail literals + offset values, slot 5 = sequences section
      uint32_t ss;
      if (TABLES) {
        // per-block tables (zt_make_tables): worth their description from a few dozen sequences on
        volatile BAMD_LAS uint32_t* scr = (volatile BAMD_LAS uint32_t*)(BAMD_LAS uint8_t*)(void*)tab_generic;
        ZsTabs zt;
        zt_make_tables(seqbuf, z.nseq, scr, zt, lane);
        ss = zs_write_sequences_v<true>(z.lit + z.nlit, z.litcap - z.nlit, seqbuf, z.nseq, T, scr, lane, &zt);
      } else
      ss = BAMD_ZSTD_VSEQ ? zs_write_sequences_v<false>(z.lit + z.nlit, z.litcap - z.nlit, seqbuf, z.nseq, T, (volatile BAMD_LAS uint32_t*)(BAMD_LAS uint8_t*)(void*)tab_generic, lane)
                                         : zs_write_sequences(z.lit + z.nlit, z.litcap - z.nlit, seqbuf, z.nseq, T, lane);
      PROF_LAP(5);
      if (ss != 0xffffffffu) bsize = zenc::kLitHeader + z.nlit + ss;
    }
    if (bsize >= seg) {                   // no gain: Raw_Block
      if (op + zenc::kBlockHeader + seg >= cap) return 0u;
      BAMD_MEM_SYNC();                    // the copy overwrites what other lanes have just written of the compressed form
      wave_copy_disjoint(bh + zenc::kBlockHeader, src + s0, seg, lane);
      bsize = seg;
      rep = rep_before;                   // a raw block leaves the decoder's repeat offsets alone
      if (lane == 0) { uint8_t h[4]; zenc::write_block_header(h, last, 0u, bsize); bh[0] = h[0]; bh[1] = h[1]; bh[2] = h[2]; }
    } else if (lane == 0) { uint8_t h[4]; zenc::write_block_header(h, last, 2u, bsize); bh[0] = h[0]; bh[1] = h[1]; bh[2] = h[2]; }
    op += zenc::kBlockHeader + bsize;
  }
  return op < n ? op : 0u;
}

// ---------------------------------------------------------------------------------------------
// zlib streams (deflate_enc.h has the format).  One stream per blosc stream: header, ONE final block with the fixed
// Huffman codes whose symbols the match finder above emits as it goes (dfl_emit_seq), end-of-block, Adler-32.
// Returns the stream size, or 0 when it would not be smaller than the input (blosc then stores the split raw).
// ---------------------------------------------------------------------------------------------
constexpr int DFL_LDS_BYTES = 65 * 4 + 12;     // the 65-dword strip of dfl_put_symbols, rounded to 16 bytes
template <bool HC = false>
__device__ uint32_t zlib_encode_wave(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap, int clevel,
                                     enc_entry_t* tab_generic, int lane EPROF_ARG) {
  if (n < 16u || cap < 64u) return 0u;
  DflSink z;
  z.out = dst; z.cap = cap; z.pos = dfl::kHeader;
  z.zb = (volatile BAMD_LAS uint32_t*)((BAMD_LAS uint8_t*)(void*)tab_generic + (HC ? HC_TAB_BYTES : ENC_TAB_BYTES));
  if (lane == 0) { uint8_t h[2]; dfl::write_header(h); dst[0] = h[0]; dst[1] = h[1]; }
  const dfl::Sym bh = dfl::block_header();
  z.acc = bh.bits; z.nb = bh.nbits;
  const uint32_t covered = HC ? hc_encode_wave<EF_ZLIB>(src, n, dst, cap, tab_generic, lane, 0u, nullptr, &z)
                              : lz_encode_wave<EF_ZLIB>(src, n, dst, cap, clevel, tab_generic, lane EPROF_PASS, 0u, nullptr, &z);
  if (covered == 0xffffffffu) return 0u;
  // the literals behind the last match, then the end-of-block symbol.  A literal costs at least 8 bits: when what is left
  // cannot fit below n any more (incompressible planes end here with everything still pending), skip the packing
  if (z.pos + (n - covered) + 8u >= n) return 0u;
  if (dfl_emit_seq(z, src + covered, n - covered, 0u, 0u, -1, 0u, lane) == 0xffffffffu) return 0u;
  const dfl::Sym eob = dfl::end_of_block();
  if (!dfl_put_symbols(z, lane == 0 ? eob.bits : 0u, lane == 0 ? eob.nbits : 0u, lane)) return 0u;
  const uint32_t tailbytes = (z.nb + 7u) >> 3;                   // <= 4, zero padding up to the byte boundary
  if ((uint32_t)lane < tailbytes) dst[z.pos + (uint32_t)lane] = (uint8_t)(z.acc >> (8u * (uint32_t)lane));
  z.pos += tailbytes;
  const uint32_t ad = wave_adler32(src, n, lane);
  if (lane < 4) dst[z.pos + (uint32_t)lane] = (uint8_t)(ad >> (24u - 8u * (uint32_t)lane));
  z.pos += dfl::kTrailer;
  return z.pos < n ? z.pos : 0u;
}

// ---------------------------------------------------------------------------------------------
// Fused byte shuffle of one block by ONE wavefront (typesize 4 or 8): element-major source -> plane-major
// scratch (blosc/shuffle-generic.h:27-58).  The mirror image of unshuffle_block_wave in k_decode.hip: per
// step lane l loads the T*4 contiguous bytes of elements e+4l..e+4l+3 (coalesced 16-byte loads), transposes
// bytes in registers and stores 4 bytes into every plane (each wave store writes 256 contiguous bytes).
// ---------------------------------------------------------------------------------------------
#ifndef BAMD_SHUF_LD_NT
#define BAMD_SHUF_LD_NT 0     // 1: the fused shuffle reads the source (read once) with non-temporal loads
#endif
template <int T>
struct ElemRows { uint4 a, b; };

template <int T>
__device__ __forceinline__ ElemRows<T> shuffle_load(const gu8* src, uint32_t e, int lane) {
  ElemRows<T> x;
  const gu8* in = src + (size_t)(e + 4u * (uint32_t)lane) * T;
  x.a = BAMD_SHUF_LD_NT ? g_ld16_nt(in) : g_ld16(in);
  if (T == 8) x.b = BAMD_SHUF_LD_NT ? g_ld16_nt(in + 16) : g_ld16(in + 16); else x.b = make_uint4(0, 0, 0, 0);
  return x;
}
template <int T>
__device__ __forceinline__ void shuffle_store(gu8* dst, uint32_t N, uint32_t e, int lane, const ElemRows<T>& x) {
  gu8* o = dst + e + 4u * (uint32_t)lane;
  uint32_t r0, r1, r2, r3;
  if (T == 8) {
    // a = (lo0, hi0, lo1, hi1), b = (lo2, hi2, lo3, hi3): low / high dword of elements 0..3
    transpose4x4(x.a.x, x.a.z, x.b.x, x.b.z, r0, r1, r2, r3);
    g_st4(o, r0); g_st4(o + (size_t)N, r1); g_st4(o + 2 * (size_t)N, r2); g_st4(o + 3 * (size_t)N, r3);
    transpose4x4(x.a.y, x.a.w, x.b.y, x.b.w, r0, r1, r2, r3);
    g_st4(o + 4 * (size_t)N, r0); g_st4(o + 5 * (size_t)N, r1); g_st4(o + 6 * (size_t)N, r2); g_st4(o + 7 * (size_t)N, r3);
  } else {
    transpose4x4(x.a.x, x.a.y, x.a.z, x.a.w, r0, r1, r2, r3);
    g_st4(o, r0); g_st4(o + (size_t)N, r1); g_st4(o + 2 * (size_t)N, r2); g_st4(o + 3 * (size_t)N, r3);
  }
}
template <int T>
__device__ void shuffle_block_wave_T(const gu8* src, gu8* dst, uint32_t bsize, int lane) {
  const uint32_t N = bsize / T;
  uint32_t e = 0;
  for (; e + 1024u <= N; e += 1024u) {   // 4 steps per iteration: all loads are issued before the first store
    const ElemRows<T> a = shuffle_load<T>(src, e, lane), b = shuffle_load<T>(src, e + 256u, lane);
    const ElemRows<T> c = shuffle_load<T>(src, e + 512u, lane), d = shuffle_load<T>(src, e + 768u, lane);
    shuffle_store<T>(dst, N, e, lane, a); shuffle_store<T>(dst, N, e + 256u, lane, b);
    shuffle_store<T>(dst, N, e + 512u, lane, c); shuffle_store<T>(dst, N, e + 768u, lane, d);
  }
  for (; e + 256u <= N; e += 256u) shuffle_store<T>(dst, N, e, lane, shuffle_load<T>(src, e, lane));
  // tail: fewer than 256 elements, then the bytes that do not form a whole element (copied as they are)
  for (uint32_t k = e * T + (uint32_t)lane; k < N * T; k += 64u) { const uint32_t el = k / T, j = k - el * T; dst[(size_t)j * N + el] = src[k]; }
  for (uint32_t k = N * T + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}

// ---------------------------------------------------------------------------------------------
// Periodic planes.  A byte plane whose every 256-byte row equals its first row - a constant byte, a counter's low
// byte, the zero top bytes of small integers: the planes shuffling exists to produce - needs no match finder and no
// trip through the scratch: its stream is "first period as literals + one match over the rest".  The shuffle wave
// notices them for free (it holds each row in a register): as long as a plane's rows keep repeating, nothing is
// stored; the first row that differs back-fills the rows skipped so far (copies of row 0) and the plane is an
// ordinary one from there on.  A plane that stays periodic to the end gets its first row stored (the literals' source)
// and its period p (smallest power of two, 1..256) left in its stream's `result` as -p; encode_one_stream turns that
// into the stream (emit_periodic_stream).  The mirror image of the decoder's periodic spans (k_decode.hip).
// Only whole-row blocks (N % 256 == 0, N >= 1024) whose planes are streams of their own (split blocks).
// ---------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void shuffle_rows(const ElemRows<T>& x, uint32_t (&r)[8]) {
  if (T == 8) {
    transpose4x4(x.a.x, x.a.z, x.b.x, x.b.z, r[0], r[1], r[2], r[3]);
    transpose4x4(x.a.y, x.a.w, x.b.y, x.b.w, r[4], r[5], r[6], r[7]);
  } else {
    transpose4x4(x.a.x, x.a.y, x.a.z, x.a.w, r[0], r[1], r[2], r[3]);
    r[4] = r[5] = r[6] = r[7] = 0;
  }
}
// smallest power-of-two period (bytes) of a 256-byte row held one dword per lane; 256 when there is none below
__device__ __forceinline__ uint32_t row_period(uint32_t w, int lane) {
  uint32_t p = 256u;
  for (uint32_t sh = 32u; sh >= 1u; sh >>= 1) {
    const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane + sh) & 63u) << 2), (int)w);
    if (__ballot(other != w) != 0ull) return p;
    p = 4u * sh;
  }
  const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);       // all lanes hold the same dword here
  if ((w0 >> 16) != (w0 & 0xffffu)) return 4u;
  return ((w0 >> 8) & 0xffu) == (w0 & 0xffu) ? 1u : 2u;
}
// returns the mask of planes that stayed periodic; their period goes to period[k]
template <int T>
__device__ uint32_t shuffle_block_wave_detect(const gu8* src, gu8* dst, uint32_t bsize, int lane, uint32_t (&period)[8]) {
  const uint32_t N = bsize / T;
  uint32_t row0[8], r[8];
  uint32_t per = (1u << T) - 1u;                       // wave-uniform: planes whose rows all equalled row 0 so far
  shuffle_rows<T>(shuffle_load<T>(src, 0u, lane), row0);
  auto step = [&](const ElemRows<T>& x, uint32_t e) {
    if (per == 0u) { shuffle_store<T>(dst, N, e, lane, x); return; }
    shuffle_rows<T>(x, r);
    gu8* o = dst + e + 4u * (uint32_t)lane;
#pragma unroll
    for (int k = 0; k < T; k++) {
      if (per & (1u << k)) {
        if (__ballot(r[k] != row0[k]) == 0ull) continue;
        per &= ~(1u << k);
        for (uint32_t t = 0; t < e; t += 256u) g_st4(dst + (size_t)k * N + t + 4u * (uint32_t)lane, row0[k]);
      }
      g_st4(o + (size_t)k * N, r[k]);
    }
  };
  uint32_t e = 256u;
  for (; e + 1024u <= N; e += 1024u) {
    const ElemRows<T> a = shuffle_load<T>(src, e, lane), b = shuffle_load<T>(src, e + 256u, lane);
    const ElemRows<T> c = shuffle_load<T>(src, e + 512u, lane), d = shuffle_load<T>(src, e + 768u, lane);
    step(a, e); step(b, e + 256u); step(c, e + 512u); step(d, e + 768u);
  }
  for (; e + 256u <= N; e += 256u) step(shuffle_load<T>(src, e, lane), e);
#pragma unroll
  for (int k = 0; k < T; k++) {
    period[k] = 0u;
    if (per & (1u << k)) {
      g_st4(dst + (size_t)k * N + 4u * (uint32_t)lane, row0[k]);
      period[k] = row_period(row0[k], lane);
    }
  }
  return per;
}

// queue task "shuffle block gb": afterwards the block's flag tells the encoders of its streams to go ahead.
// Producer and consumers run on the same XCD (per-XCD queues), so the hand-off goes through that XCD's L2:
// drain the stores, then a relaxed agent-scope flag store - no L2 write-back needed.
template <int T>
__device__ __forceinline__ void shuffle_block_detect_T(const gu8* src, gu8* dst, uint32_t bsize, StreamDesc* planes, int lane) {
  uint32_t period[8];
  const uint32_t per = shuffle_block_wave_detect<T>(src, dst, bsize, lane, period);
#pragma unroll
  for (int k = 0; k < T; k++)
    if ((per & (1u << k)) && lane == 0) planes[k].result = -(int32_t)period[k];
}
__device__ __attribute__((noinline)) void shuffle_block_task(const ChunkDesc* chunks, const BlockDesc* blocks, uint32_t gb,
                                                             uint32_t* blk_ready, StreamDesc* streams, int detect, int lane) {
  const BlockDesc* b = blocks + gb;
  const ChunkDesc* c = chunks + uni((uint32_t)b->chunk);
  const uint32_t blk = uni((uint32_t)b->blk), bsize = uni((uint32_t)b->bsize), bs = uni((uint32_t)c->blocksize);
  const gu8* src = uni_ptr(as_global(c->src)) + (size_t)blk * bs;
  gu8* dst = uni_ptr(as_global(c->filt)) + (size_t)blk * bs;
  const uint32_t T = uni((uint32_t)c->typesize), N = bsize / T;
  // periodic planes (see above): whole rows only, planes that are streams of their own, LZ4 / BloscLZ streams
  const bool det = detect && uni((uint32_t)b->nstreams) == T && N * T == bsize && (N & 255u) == 0u && N >= 1024u &&
                   (uni((uint32_t)c->fmt) == (uint32_t)FMT_LZ4 || uni((uint32_t)c->fmt) == (uint32_t)FMT_BLOSCLZ);
  StreamDesc* planes = streams + uni((uint32_t)b->first_stream);
  if (T == 8u) { if (det) shuffle_block_detect_T<8>(src, dst, bsize, planes, lane); else shuffle_block_wave_T<8>(src, dst, bsize, lane); }
  else { if (det) shuffle_block_detect_T<4>(src, dst, bsize, planes, lane); else shuffle_block_wave_T<4>(src, dst, bsize, lane); }
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): every store of this wave has reached L2
  if (lane == 0) __hip_atomic_store(&blk_ready[gb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The stream of a plane of n bytes (n % 256 == 0, n >= 1024) that repeats with period p <= 256; `in` holds its first 256
// bytes.  LZ4: p literals, one match at distance p up to n - 5, the last five bytes as literals (lz4.c:245-246 end
// rules).  BloscLZ: the same with the match ending at n - 2 (the encoder's own limit above) and the marker bit of
// blosclz.c:607.  Always smaller than n, so the "store raw" fallback - which would need the plane in memory - cannot hit.
__device__ __forceinline__ uint32_t emit_periodic_stream(const gu8* in, uint32_t n, gu8* out, uint32_t cap, uint32_t p, bool lz4, int lane) {
  uint32_t op;
  if (lz4) {
    op = lz4_emit_seq(out, 0u, cap, in, p, p, n - 5u - p, -1, 0u, lane);
    if (op == 0xffffffffu) return 0u;
    op = lz4_emit_tail(out, op, cap, in + 251u, 5u, lane);
    return op == 0xffffffffu ? 0u : op;
  }
  op = blz_emit_literals(out, 0u, cap, in, p, lane);
  if (op == 0xffffffffu) return 0u;
  op = blz_emit_match(out, op, cap, p, n - 2u - p, lane);
  if (op == 0xffffffffu) return 0u;
  op = blz_emit_literals(out, op, cap, in + 254u, 2u, lane);
  if (op == 0xffffffffu) return 0u;
  if (lane == 0) out[0] |= 0x20u;
  return op;
}

// one stream, not inlined into the queue loop (see decode_one_stream in k_decode.hip for why)
// MODE: 0 = LZ4 / BloscLZ, 1 = Zstd, 2 = Zlib, 3 = LZ4 with the LZ4HC-grade search, 4 = Zstd with per-block sequence tables -
// a batch has ONE codec, so every kernel carries only its own code path
// 5 = Zstd with per-block tables behind the LZ4HC-grade search, 6 = zlib behind the LZ4HC-grade search
enum { ENC_LZ = 0, ENC_ZSTD = 1, ENC_ZLIB = 2, ENC_HC = 3, ENC_ZSTD_T = 4, ENC_ZSTD_HC = 5, ENC_ZLIB_HC = 6 };
constexpr bool enc_mode_hc(int mode) { return mode == ENC_HC || mode == ENC_ZSTD_HC || mode == ENC_ZLIB_HC; }
template <int MODE>
__device__ __attribute__((noinline)) void encode_one_stream(StreamDesc* sd, enc_entry_t* tab, const ChunkDesc* chunks, uint32_t* blk_ready, int lane,
                                                            const BlockDesc* blocks, uint32_t sid, uint32_t* plane_cost, uint64_t* seqbuf
#ifdef BAMD_PROFILE_DECODE
                                                            , uint32_t* profslot
#endif
                                                            ) {
  PROF_DECL
#ifdef BAMD_PROFILE_DECODE
  prof_.c[4] = 0; prof_.c[5] = 0;
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
  else if (MODE == ENC_ZSTD_HC) r = seqbuf ? zstd_encode_wave<true, true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZLIB_HC) r = zlib_encode_wave<true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, lane EPROF_PASS);
  else if (MODE == ENC_ZSTD_T) r = seqbuf ? zstd_encode_wave<true>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, (BAMD_GAS uint64_t*)seqbuf, lane EPROF_PASS) : 0u;
  else if (MODE == ENC_ZLIB) r = zlib_encode_wave(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, lane EPROF_PASS);
  else if (hint < 0) r = emit_periodic_stream(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, (uint32_t)-hint, uni((uint32_t)sd->fmt) == (uint32_t)FMT_LZ4, lane);
  else if (MODE == ENC_HC) r = lz4hc_encode_wave(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, tab, lane);
  else if (sd->fmt == FMT_LZ4) r = lz_encode_wave<EF_LZ4>(uni_ptr(as_global(sd->in)), n, uni_ptr(as_global(sd->out)), cap, clevel, tab, lane EPROF_PASS);
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
__global__ __launch_bounds__(64 * ENC_WAVES, enc_mode_hc(MODE) ? 2 : (MODE == ENC_ZSTD_T ? 5 : BAMD_ENC_MINWAVES)) void k_encode_streams_t(
    StreamDesc* __restrict__ streams, uint32_t* __restrict__ tickets /*[8]*/, const int32_t* __restrict__ qlist,
    const int32_t* __restrict__ qoff /*[9]*/, const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks,
    uint32_t* __restrict__ blk_ready, uint32_t* __restrict__ plane_cost, int single_queue,
    uint64_t* __restrict__ seqbufs, const zenc::CTabs* __restrict__ ctabs, int detect_periodic
#ifdef BAMD_PROFILE_DECODE
    , uint32_t* __restrict__ profbuf
#endif
    ) {
  constexpr bool ZSTD = MODE == ENC_ZSTD || MODE == ENC_ZSTD_T || MODE == ENC_ZSTD_HC;
  constexpr int TABBYTES = enc_mode_hc(MODE) ? HC_TAB_BYTES : ENC_TAB_BYTES;      // the match finder's table; the writers' LDS sits behind it
  __shared__ __attribute__((aligned(16))) enc_entry_t tabs[ENC_WAVES][(TABBYTES + (ZSTD ? ZS_LDS_BYTES : ((MODE == ENC_ZLIB || MODE == ENC_ZLIB_HC) ? DFL_LDS_BYTES : 0))) / 4];
  static_assert(ENC_WAVES == 1, "one stream per wave, one wave per workgroup");
  const int lane = threadIdx.x & 63;
  uint64_t* seqbuf = nullptr;
  if (ZSTD) {       // the predefined FSE tables of the sequence coder, once per persistent wave
    const uint32_t* g = (const uint32_t*)ctabs;
    for (uint32_t k = (uint32_t)lane; k < sizeof(zenc::CTabs) / 4u; k += 64u) tabs[0][TABBYTES / 4 + k] = g[k];
    seqbuf = seqbufs + (size_t)blockIdx.x * ZS_SEQCAP;
  }
  // HW_REG_XCC_ID[3:0]; queue 0 for everybody in the single-queue fallback (no in-kernel hand-offs there)
  const uint32_t xcc = single_queue ? 0u : (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
  const uint32_t qbase = (uint32_t)qoff[xcc], qlen = (uint32_t)qoff[xcc + 1] - qbase;
  uint32_t t = take_ticket(tickets + xcc, lane);
  uint32_t ndone = 0;      // tasks this wave took: summed into plane_cost[256], the host checks the total
  while (t < qlen) {
    const int32_t task = (int32_t)uni((uint32_t)qlist[qbase + t]);
    if (task < 0) {
      shuffle_block_task(chunks, blocks, (uint32_t)(-(task + 1)), blk_ready, streams, detect_periodic, lane);
    } else {
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
