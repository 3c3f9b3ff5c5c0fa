// inflate_serial.h — the inherently SERIAL parts of reading a zlib stream (codec row "Zlib" of SURVEY §8f-3:
// zlib_wrap_decompress blosc/blosc.c:484-495 -> uncompress -> inflate, internal-complibs/zlib-1.3.1/inflate.c:590-1270,
// inftrees.c:32-299): the LSB-first bit reader, the zlib header, block headers, the code-length code, canonical Huffman
// table builds and the symbol decoder.  Written from the formats (RFC 1950, RFC 1951) with the reference's acceptance
// rules where the RFC leaves room:
//   * an over-subscribed set of code lengths is an error; an incomplete one is an error too, EXCEPT a literal/length or
//     distance code whose longest code is one bit long (inftrees.c:130-135), and a distance code without any symbol;
//   * the code-length code must be complete (inftrees.c:134, type == CODES);
//   * a dynamic block must give symbol 256 a code (inflate.c:1003-1007); HLIT > 286 or HDIST > 30 is an error (:929-935);
//   * literal/length symbols 286 / 287 and distance symbols 30 / 31 are invalid when they are decoded (fixed tables);
//   * bits missing at the end of the input are an error (uncompress turns Z_BUF_ERROR into a failure, uncompr.c:79-84).
// No malloc, every read bounded by the stream, all tables in caller-provided memory (LDS on the GPU).
//
// Plain C++ for BOTH sides (like zstd_serial.h): k_zlib.hip runs these functions wave-uniformly (every lane computes the
// same values) and does the byte moving wave-parallel; tests/tools/inflate_serial_stream.cpp compiles the same code with
// g++ and is compared with the real zlib of the reference on valid and corrupted streams (tests/test_inflate_serial_cpu.py).
#pragma once
#include <stdint.h>
#ifndef BAMD_LDS_SYNC          // wave_prims.h: where lanes hand data to each other through LDS (a rendezvous for the wavefront emulator only)
#define BAMD_LDS_SYNC() ((void)0)
#endif

#if defined(__HIPCC__)
#define ZI_FN __device__ __forceinline__
#define ZI_MFN __device__ __forceinline__
#define ZI_COLD __device__ __attribute__((noinline))   // once per block: real calls keep the symbol loop small (and its registers free)
#define ZI_NOUNROLL _Pragma("nounroll")
#else
#define ZI_FN static inline
#define ZI_MFN inline
#define ZI_COLD static inline
#define ZI_NOUNROLL
#endif

namespace zi {

// ---------------- bit reader: LSB first, never touches a byte outside the stream ----------------
// `Src` supplies the bytes: fetch32(pos) = the four bytes at pos, little endian, zero beyond the end.  MemSrc reads memory
// (CPU build); the GPU kernel supplies a 512-byte register window of the stream (k_zlib.hip: WinSrc).
struct MemSrc {
  const uint8_t* p; uint32_t n;
  ZI_MFN uint32_t fetch32(uint32_t pos) const {
    if (pos + 4u <= n) { uint32_t v; __builtin_memcpy(&v, p + pos, 4); return v; }
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4u && pos + k < n; k++) v |= (uint32_t)p[pos + k] << (8u * k);
    return v;
  }
};
template <class Src>
struct BitsT { Src s; uint32_t n; uint32_t pos; uint64_t acc; uint32_t nacc; uint32_t used; };   // used: bits consumed so far
typedef BitsT<MemSrc> Bits;
template <class B> ZI_FN void bits_start(B& b, uint32_t n) { b.n = n; b.pos = 0; b.acc = 0; b.nacc = 0; b.used = 0; }
ZI_FN void bits_init(Bits& b, const uint8_t* p, uint32_t n) { b.s.p = p; b.s.n = n; bits_start(b, n); }
// at least 32 valid bits in the accumulator (zeros beyond the end of the input: the callers compare `used` with 8 n)
template <class B> ZI_FN void bits_fill(B& b) {
  if (b.nacc >= 32u) return;
  b.acc |= (uint64_t)b.s.fetch32(b.pos) << b.nacc; b.nacc += 32u; b.pos += 4u;
}
template <class B> ZI_FN uint32_t bits_peek(B& b, uint32_t k) { bits_fill(b); return (uint32_t)b.acc & ((1u << k) - 1u); }     // k <= 16
template <class B> ZI_FN void bits_drop(B& b, uint32_t k) { b.acc >>= k; b.nacc -= k; b.used += k; }
template <class B> ZI_FN uint32_t bits_get(B& b, uint32_t k) { const uint32_t v = bits_peek(b, k); bits_drop(b, k); return v; }
template <class B> ZI_FN bool bits_overrun(const B& b) { return b.used > 8u * b.n; }
template <class B> ZI_FN void bits_align(B& b) { const uint32_t k = (8u - (b.used & 7u)) & 7u; bits_drop(b, k); }   // the accumulator holds >= 7 bits here (bits_peek before)
template <class B> ZI_FN uint32_t bits_bytepos(const B& b) { return b.used >> 3; }                                      // only meaningful when aligned

// ---------------- canonical Huffman codes ----------------
// count[len] = number of symbols with that code length, sym[] = symbols ordered by (length, symbol), fast[] = direct
// lookup for codes of at most FB bits: symbol << 4 | length, 0 = longer code (or no code).
// On the GPU every lane of the wave runs this code with the same values (wave-uniform); the tables live in LDS, are written
// by ONE lane (writer()) and read by all: volatile, so that no lane works with a value it has kept in a register across
// another lane's store.
constexpr int kLitFast = 9, kDistFast = 7, kMaxBits = 15;
#if defined(__HIPCC__) || defined(BAMD_WAVE_EMU)      // (the wavefront emulator of tests/tools/wave_emu runs the device form on the host)
#define ZI_TAB volatile
ZI_FN bool writer() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u; }
#else
#define ZI_TAB
ZI_FN bool writer() { return true; }
#endif
typedef ZI_TAB uint16_t tab16;
typedef ZI_TAB uint8_t tab8;
// On the device the tables are LDS, and the symbol loop must SAY so: through `const Tabs&` the pointers are generic, the address-space
// inference leaves volatile accesses alone, and every table lookup of k_zlib_streams was a flat_load (96 of them, not one ds_read;
// 0.49 G flat loads per 8 GiB, profiles/r03/r03zp_zlib_decode_sq_counters.txt) - twice the latency of an LDS read in a loop that is one
// dependent lookup after the other.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__) && !defined(BAMD_WAVE_EMU)
#define ZI_LDS __attribute__((address_space(3)))
#else
#define ZI_LDS
#endif
typedef const ZI_LDS tab16* ltab16;
struct Tabs {
  tab16 lcount[16], lsym[288], lfast[1 << kLitFast];
  tab16 dcount[16], dsym[32], dfast[1 << kDistFast];
  tab8 lens[320];                     // code lengths of the block being set up (literal/length, then distance)
  tab16 wcnt[16], woffs[16];          // huff_build's counters: in the table memory (LDS) - as arrays of the function they would be
                                      // dynamically indexed private memory, i.e. scratch in HBM on the GPU (1.7 ms per block, measured)
};
ZI_FN uint32_t rev_bits(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; i++) { r = (r << 1) | (v & 1u); v >>= 1; } return r; }
// returns: 0 complete, > 0 incomplete (bits of code space left), < 0 over-subscribed; *maxlen = longest code (0: no symbol)
ZI_COLD int huff_build(tab16* count, tab16* sym, tab16* fast, int fastbits, const tab8* lens, int n, int* maxlen, tab16* cnt, tab16* offs) {
  const bool w = writer();
  BAMD_LDS_SYNC();          // the counters are shared by the three tables of a block: every lane is done with the previous build
  ZI_NOUNROLL for (int l = 0; l <= kMaxBits; l++) if (w) cnt[l] = 0;
  ZI_NOUNROLL for (int s = 0; s < n; s++) { const int l = lens[s]; const uint16_t c = cnt[l]; if (w) cnt[l] = (uint16_t)(c + 1); }
  int left = 1, mx = 0;
  ZI_NOUNROLL for (int l = 1; l <= kMaxBits; l++) { const int c = cnt[l]; left <<= 1; left -= c; if (left < 0) { BAMD_LDS_SYNC(); return left; } if (c) mx = l; }
  *maxlen = mx;
  ZI_NOUNROLL for (int l = 0; l <= kMaxBits; l++) { const uint16_t c = cnt[l]; if (w) count[l] = c; }
  if (w) offs[1] = 0;
  ZI_NOUNROLL for (int l = 1; l < kMaxBits; l++) { const uint16_t o = (uint16_t)(offs[l] + cnt[l]); if (w) offs[l + 1] = o; }
  ZI_NOUNROLL for (int s = 0; s < n; s++) { const int l = lens[s]; if (l) { const uint16_t o = offs[l]; if (w) { sym[o] = (uint16_t)s; offs[l] = (uint16_t)(o + 1); } } }
  ZI_NOUNROLL for (int k = 0; k < (1 << fastbits); k++) if (w) fast[k] = 0;
  uint32_t code = 0; int idx = 0;
  ZI_NOUNROLL for (int l = 1; l <= fastbits; l++) {
    const int c_l = cnt[l];
    ZI_NOUNROLL for (int c = 0; c < c_l; c++, code++, idx++) {
      const uint16_t e = (uint16_t)((sym[idx] << 4) | l);
      ZI_NOUNROLL for (uint32_t k = rev_bits(code, l); k < (1u << fastbits); k += 1u << l) if (w) fast[k] = e;
    }
    code <<= 1;
  }
  BAMD_LDS_SYNC();
  return left;
}
// next symbol, or -1 when the bits match no code
template <class B, class PT> ZI_FN int huff_decode(B& b, PT count, PT sym, PT fast, int fastbits) {      // PT: const tab16* or its LDS-qualified form
  const uint32_t v = bits_peek(b, kMaxBits);
  const uint32_t e = fast[v & ((1u << fastbits) - 1u)];
  if (e) { bits_drop(b, e & 15u); return (int)(e >> 4); }
  int code = 0, first = 0, index = 0;
  for (int l = 1; l <= kMaxBits; l++) {
    code |= (int)((v >> (l - 1)) & 1u);
    const int c = (int)count[l];
    if (code - c < first) { bits_drop(b, (uint32_t)l); return (int)sym[index + (code - first)]; }
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -1;
}

// ---------------- stream / block structure ----------------
enum { BLK_STORED = 0, BLK_CODED = 1, BLK_ERROR = -1 };
enum { OP_LIT = 0, OP_MATCH = 1, OP_EOB = 2, OP_ERROR = -1 };
struct Op { uint32_t len, dist; };     // OP_LIT: len = the byte

// zlib header (RFC 1950; inflate.c:672-707 with the default 15 window bits and no dictionary)
template <class B> ZI_FN bool zlib_header(B& b) {
  if (b.n < 2u) return false;
  const uint32_t cmf = bits_get(b, 8), flg = bits_get(b, 8);
  if (((cmf << 8) | flg) % 31u) return false;
  if ((cmf & 15u) != 8u || (cmf >> 4) > 7u) return false;
  if (flg & 0x20u) return false;                                   // preset dictionary: Z_NEED_DICT -> failure
  return true;
}

ZI_COLD void fixed_lengths(tab8* lens) {
  if (!writer()) return;
  ZI_NOUNROLL for (int s = 0; s < 144; s++) lens[s] = 8;
  ZI_NOUNROLL for (int s = 144; s < 256; s++) lens[s] = 9;
  ZI_NOUNROLL for (int s = 256; s < 280; s++) lens[s] = 7;
  ZI_NOUNROLL for (int s = 280; s < 288; s++) lens[s] = 8;
  ZI_NOUNROLL for (int s = 0; s < 32; s++) lens[288 + s] = 5;      // 30 and 31 have codes but are invalid when decoded (next_op)
}

// reads one block header; *final = BFINAL.  BLK_STORED: *stored_len bytes follow at byte bits_bytepos(b) (the caller copies
// them and calls bits_skip_bytes); BLK_CODED: the tables in `t` are ready for next_op().
template <class B> ZI_FN int block_begin(B& b, Tabs& t, int* final, uint32_t* stored_len) {
  *final = (int)bits_get(b, 1);
  const uint32_t type = bits_get(b, 2);
  if (bits_overrun(b)) return BLK_ERROR;
  if (type == 0u) {
    bits_peek(b, 8); bits_align(b);
    const uint32_t len = bits_get(b, 16), nlen = bits_get(b, 16);
    if (bits_overrun(b) || len != (nlen ^ 0xffffu)) return BLK_ERROR;
    if ((uint64_t)bits_bytepos(b) + len > b.n) return BLK_ERROR;
    *stored_len = len;
    return BLK_STORED;
  }
  if (type == 3u) return BLK_ERROR;
  int nlen = 288, ndist = 32;
  if (type == 1u) fixed_lengths(t.lens);
  else {
    nlen = (int)bits_get(b, 5) + 257; ndist = (int)bits_get(b, 5) + 1;
    const int ncode = (int)bits_get(b, 4) + 4;
    if (nlen > 286 || ndist > 30) return BLK_ERROR;
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    // the 19 lengths of the code-length code: scratch behind the distance lengths' slot (rewritten below)
    tab8* cl = t.lens + 296;
    const bool w = writer();
    ZI_NOUNROLL for (int i = 0; i < 19; i++) if (w) cl[i] = 0;
    ZI_NOUNROLL for (int i = 0; i < ncode; i++) { const uint32_t v = bits_get(b, 3); if (w) cl[order[i]] = (uint8_t)v; }
    if (bits_overrun(b)) return BLK_ERROR;
    int mx;
    // the code-length code uses the distance arrays as scratch (they are rebuilt below)
    const int left = huff_build(t.dcount, t.dsym, t.dfast, kDistFast, cl, 19, &mx, t.wcnt, t.woffs);
    if (left < 0 || (left > 0 && mx != 0)) return BLK_ERROR;   // must be complete (mx == 0: no code at all - every read below fails)
    int have = 0;
    ZI_NOUNROLL while (have < nlen + ndist) {
      const int s = mx ? huff_decode(b, t.dcount, t.dsym, t.dfast, kDistFast) : -1;
      if (s < 0 || bits_overrun(b)) return BLK_ERROR;
      if (s < 16) { if (w) t.lens[have] = (uint8_t)s; have++; continue; }
      int rep, val = 0;
      if (s == 16) { if (have == 0) return BLK_ERROR; val = t.lens[have - 1]; rep = 3 + (int)bits_get(b, 2); }
      else if (s == 17) rep = 3 + (int)bits_get(b, 3);
      else rep = 11 + (int)bits_get(b, 7);
      if (have + rep > nlen + ndist) return BLK_ERROR;
      ZI_NOUNROLL while (rep--) { if (w) t.lens[have] = (uint8_t)val; have++; }
    }
    if (bits_overrun(b)) return BLK_ERROR;
    if (t.lens[256] == 0) return BLK_ERROR;
    // distance lengths to their fixed place behind the 288 literal/length slots
    ZI_NOUNROLL for (int s = ndist - 1; s >= 0; s--) { const uint8_t v = t.lens[nlen + s]; if (w) t.lens[288 + s] = v; }
  }
  int mx;
  int left = huff_build(t.lcount, t.lsym, t.lfast, kLitFast, t.lens, nlen, &mx, t.wcnt, t.woffs);
  if (left < 0 || (left > 0 && mx != 1)) return BLK_ERROR;
  left = huff_build(t.dcount, t.dsym, t.dfast, kDistFast, t.lens + 288, ndist, &mx, t.wcnt, t.woffs);
  if (left < 0 || (left > 0 && mx > 1)) return BLK_ERROR;   // mx == 0: no distance code (fine until one is needed)
  return BLK_CODED;
}
template <class B> ZI_FN void bits_skip_bytes(B& b, uint32_t nbytes) {     // behind a stored block's header: the reader is byte-aligned
  const uint32_t at = bits_bytepos(b) + nbytes;
  b.pos = at; b.acc = 0; b.nacc = 0; b.used = 8u * at;
}

ZI_FN uint32_t len_base(int c) {      // c = symbol - 257, 0..28
  return c < 8 ? 3u + (uint32_t)c : (c == 28 ? 258u : 3u + ((4u + ((uint32_t)c & 3u)) << ((uint32_t)(c >> 2) - 1u)));
}
ZI_FN uint32_t len_extra(int c) { return (c < 8 || c == 28) ? 0u : (uint32_t)(c >> 2) - 1u; }
ZI_FN uint32_t dist_base(int c) { return c < 4 ? 1u + (uint32_t)c : 1u + ((2u + ((uint32_t)c & 1u)) << ((uint32_t)(c >> 1) - 1u)); }
ZI_FN uint32_t dist_extra(int c) { return c < 4 ? 0u : (uint32_t)(c >> 1) - 1u; }

// next literal / match / end-of-block of a coded block
template <class B> ZI_FN int next_op(B& b, const Tabs& t, Op& op) {
  const int s = huff_decode(b, (ltab16)t.lcount, (ltab16)t.lsym, (ltab16)t.lfast, kLitFast);
  if (s < 0 || bits_overrun(b)) return OP_ERROR;
  if (s < 256) { op.len = (uint32_t)s; return OP_LIT; }
  if (s == 256) return OP_EOB;
  if (s > 285) return OP_ERROR;
  const int lc = s - 257;
  op.len = len_base(lc) + bits_get(b, len_extra(lc));
  const int d = huff_decode(b, (ltab16)t.dcount, (ltab16)t.dsym, (ltab16)t.dfast, kDistFast);
  if (d < 0 || d > 29) return OP_ERROR;
  const uint32_t de = dist_extra(d);
  op.dist = dist_base(d) + bits_get(b, de);
  if (bits_overrun(b)) return OP_ERROR;
  return OP_MATCH;
}

// Adler-32 of the plain bytes follows the last block, big-endian, on a byte boundary (RFC 1950)
template <class B> ZI_FN bool read_adler(B& b, uint32_t* value) {
  bits_peek(b, 8); bits_align(b);
  uint32_t v = 0;
  for (int i = 0; i < 4; i++) v = (v << 8) | bits_get(b, 8);
  *value = v;
  return !bits_overrun(b);
}

}  // namespace zi
