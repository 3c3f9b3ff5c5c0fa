// deflate_enc.h — the format-writing half of the GPU Zlib encoder (codec row "Zlib" of SURVEY §8f-3, compress direction:
// zlib_wrap_compress blosc/blosc.c:472-482 -> compress2).  A writer of VALID zlib streams, not a port of deflate.c: one
// stream per blosc split = zlib header (RFC 1950) + ONE final block with the FIXED Huffman codes (RFC 1951 3.2.6) +
// Adler-32.  What the match finder of k_encode.hip hands over - literal runs and (length, distance) matches - is turned
// into code words here: a literal is 8 or 9 bits, a match piece (length 3..258, distance 1..32768) at most 31 bits, so one
// lane of the wave packs one symbol and a prefix sum over the bit counts places 64 of them at once (k_encode.hip:
// dfl_put_symbols).  Matches longer than 258 bytes are cut into pieces of the same distance.
// Plain C++ for both sides: tests/tools/deflate_enc_cpu.cpp writes streams with exactly these functions behind a greedy
// matcher and has the reference's own `uncompress` (oracle/_ref) and the oracle read them (tests/test_deflate_enc_cpu.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DF_FN __device__ __forceinline__
#else
#define DF_FN static inline
#endif

namespace bamd { namespace dfl {

constexpr uint32_t kMaxDist = 32768u, kMaxLen = 258u, kMinLen = 3u;
constexpr uint32_t kHeader = 2u, kTrailer = 4u;
struct Sym { uint32_t bits, nbits; };     // `bits` in stream order (LSB first), nbits <= 31

// Huffman codes enter the stream MSB first: the low n bits of v, reversed
DF_FN uint32_t rev(uint32_t v, uint32_t n) {
#if defined(__HIPCC__)
  return __builtin_bitreverse32(v) >> (32u - n);
#else
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1); v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4); v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
  v = (v >> 16) | (v << 16);
  return v >> (32u - n);
#endif
}
DF_FN uint32_t hb(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }                      // v != 0

// zlib header: deflate, 32 KiB window, "fastest" level hint, no dictionary; (CMF * 256 + FLG) % 31 == 0
DF_FN void write_header(uint8_t* h) { h[0] = 0x78; h[1] = 0x01; }
// first three bits of the stream: BFINAL = 1, BTYPE = 01 (fixed codes)
DF_FN Sym block_header() { Sym s = {1u | (1u << 1), 3u}; return s; }

// fixed literal/length code (RFC 1951 3.2.6)
DF_FN Sym litlen_code(uint32_t sym) {
  Sym s;
  if (sym < 144u) { s.bits = rev(0x30u + sym, 8u); s.nbits = 8u; }
  else if (sym < 256u) { s.bits = rev(0x190u + (sym - 144u), 9u); s.nbits = 9u; }
  else if (sym < 280u) { s.bits = rev(sym - 256u, 7u); s.nbits = 7u; }
  else { s.bits = rev(0xc0u + (sym - 280u), 8u); s.nbits = 8u; }
  return s;
}
DF_FN Sym literal(uint32_t byte) { return litlen_code(byte); }
DF_FN Sym end_of_block() { return litlen_code(256u); }

// one match piece: length code + extra bits + 5-bit distance code + extra bits (3 <= len <= 258, 1 <= dist <= 32768)
DF_FN Sym match(uint32_t len, uint32_t dist) {
  uint32_t lc, le = 0, lx = 0;
  const uint32_t l = len - 3u;
  if (len == 258u) lc = 28u;
  else if (l < 8u) lc = l;
  else { const uint32_t h = hb(l); le = h - 2u; lc = 4u * (h - 1u) + ((l >> le) & 3u); lx = l & ((1u << le) - 1u); }
  uint32_t dc, de = 0, dx = 0;
  const uint32_t d = dist - 1u;
  if (d < 4u) dc = d;
  else { const uint32_t h = hb(d); de = h - 1u; dc = 2u * h + ((d >> de) & 1u); dx = d & ((1u << de) - 1u); }
  const Sym c = litlen_code(257u + lc);
  Sym s;
  s.bits = c.bits | (lx << c.nbits) | (rev(dc, 5u) << (c.nbits + le)) | (dx << (c.nbits + le + 5u));
  s.nbits = c.nbits + le + 5u + de;
  return s;
}

// ---- dynamic Huffman blocks (RFC 1951 3.2.7): codes made for the block, their lengths in front of it ----
// The symbol numbers and extra bits of match(), separately: a length symbol 257..285, a distance symbol 0..29.
constexpr int kLitLenSyms = 286, kDistSyms = 30, kCodeLenSyms = 19, kMaxCodeBits = 15, kMaxCodeLenBits = 7;
struct MatchSyms { uint32_t lsym, lextra, lbits, dsym, dextra, dbits; };
DF_FN MatchSyms match_symbols(uint32_t len, uint32_t dist) {
  MatchSyms m;
  const uint32_t l = len - 3u;
  m.lextra = 0; m.lbits = 0;
  if (len == 258u) m.lsym = 285u;
  else if (l < 8u) m.lsym = 257u + l;
  else { const uint32_t h = hb(l); m.lbits = h - 2u; m.lsym = 257u + 4u * (h - 1u) + ((l >> m.lbits) & 3u); m.lextra = l & ((1u << m.lbits) - 1u); }
  const uint32_t d = dist - 1u;
  m.dextra = 0; m.dbits = 0;
  if (d < 4u) m.dsym = d;
  else { const uint32_t h = hb(d); m.dbits = h - 1u; m.dsym = 2u * h + ((d >> m.dbits) & 1u); m.dextra = d & ((1u << m.dbits) - 1u); }
  return m;
}
// canonical codes from code lengths (RFC 1951 3.2.2: shorter codes first, within a length by symbol), kept bit-reversed - the way
// they enter the LSB-first stream
DF_FN void assign_codes(const uint8_t* len, int n, uint16_t* code) {
  uint32_t count[kMaxCodeBits + 1] = {0}, next[kMaxCodeBits + 2];
  for (int s = 0; s < n; s++) count[len[s]]++;
  count[0] = 0;
  uint32_t c = 0;
  for (int b = 1; b <= kMaxCodeBits; b++) { c = (c + count[b - 1]) << 1; next[b] = c; }
  for (int s = 0; s < n; s++) code[s] = len[s] ? (uint16_t)rev(next[len[s]]++, len[s]) : 0;
}
struct DynCodes {
  uint8_t llen[kLitLenSyms], dlen[kDistSyms];
  uint16_t lcode[kLitLenSyms], dcode[kDistSyms];
};
DF_FN Sym dyn_literal(const DynCodes& c, uint32_t byte) { Sym s = {c.lcode[byte], c.llen[byte]}; return s; }
DF_FN Sym dyn_end_of_block(const DynCodes& c) { Sym s = {c.lcode[256], c.llen[256]}; return s; }
// one match piece: <= 15 + 5 + 15 + 13 = 48 bits, in two parts (length symbol + extra, distance symbol + extra)
DF_FN void dyn_match(const DynCodes& c, uint32_t len, uint32_t dist, Sym& lpart, Sym& dpart) {
  const MatchSyms m = match_symbols(len, dist);
  lpart.bits = c.lcode[m.lsym] | (m.lextra << c.llen[m.lsym]); lpart.nbits = c.llen[m.lsym] + m.lbits;
  dpart.bits = c.dcode[m.dsym] | (m.dextra << c.dlen[m.dsym]); dpart.nbits = c.dlen[m.dsym] + m.dbits;
}
// The code lengths of both alphabets as the block header carries them: run-length symbols 16 (repeat the previous length 3-6
// times), 17 (3-10 zeros), 18 (11-138 zeros) over llen[0..nlit) followed by dlen[0..ndist).  Fills sym[] / extra[] (extra bits'
// value; their width follows from the symbol), returns the count.
DF_FN int code_length_symbols(const uint8_t* lens, int n, uint8_t* sym, uint8_t* extra) {
  int k = 0;
  for (int i = 0; i < n;) {
    const uint8_t v = lens[i];
    int r = 1;
    while (i + r < n && lens[i + r] == v) r++;
    i += r;
    if (v == 0) {
      while (r >= 11) { const int t = r < 138 ? r : 138; sym[k] = 18; extra[k++] = (uint8_t)(t - 11); r -= t; }
      if (r >= 3) { sym[k] = 17; extra[k++] = (uint8_t)(r - 3); r = 0; }
      while (r-- > 0) { sym[k] = 0; extra[k++] = 0; }
    } else {
      sym[k] = v; extra[k++] = 0; r--;
      while (r >= 3) { const int t = r < 6 ? r : 6; sym[k] = 16; extra[k++] = (uint8_t)(t - 3); r -= t; }
      while (r-- > 0) { sym[k] = v; extra[k++] = 0; }
    }
  }
  return k;
}
constexpr uint8_t kCodeLenOrder[kCodeLenSyms] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// a match of `mlen` >= 3 bytes as pieces of at most 258: all but the last two are 258 long, the last is never shorter than 3
DF_FN uint32_t npieces(uint32_t mlen) { return (mlen + kMaxLen - 1u) / kMaxLen; }
DF_FN uint32_t piece_len(uint32_t mlen, uint32_t k, uint32_t np) {
  const uint32_t last = mlen - kMaxLen * (np - 1u);            // 1..258
  const uint32_t steal = (np > 1u && last < kMinLen) ? kMinLen - last : 0u;
  if (k + 1u == np) return last + steal;
  if (k + 2u == np) return kMaxLen - steal;
  return kMaxLen;
}

}}  // namespace bamd::dfl
