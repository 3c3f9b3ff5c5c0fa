// zstd_enc.h — the format-writing half of the Zstd encoder (row K8 of SURVEY §8a, compress direction).
//
// Replaces  zstd_wrap_compress -> ZSTD_compress   (blosc/blosc.c:499-511, zstd-1.5.6/lib/compress/zstd_compress.c:5372)
// with a writer of VALID frames, not a port of the reference's encoder: one frame per blosc stream
// (magic, FHD 0xA0 = single segment + 4-byte content size, blocks of at most 128 KiB), literals stored raw, sequences
// coded with the PREDEFINED FSE tables.  Everything here is written from the format specification (RFC 8878):
//   section 3.1.1.3.2.1.1  sequence codes and extra bits        -> ll_code / ml_code tables below
//   section 3.1.1.3.2.2.1  predefined distributions             -> kLLNorm / kMLNorm / kOFNorm
//   section 4.1            FSE table construction                -> build_ctab (the encoder's view of the same table)
//   section 3.1.1.3.2.1.2  order of the bits in the sequence bitstream -> SeqBitWriter users
// The file is plain C++ usable on the host (g++, tests/tools/zstd_enc_cpu.cpp checks the frames with the reference's
// own ZSTD_decompress) and in device code (k_encode.hip).  The match finder is not here: k_encode.hip's wave-parallel
// LZ finder produces (literal length, match length, offset) triples.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define ZE_FN __host__ __device__ inline
#else
#define ZE_FN inline
#endif

namespace bamd {
namespace zenc {

constexpr uint32_t kMagic = 0xFD2FB528u;
constexpr uint32_t kBlockMax = 128u * 1024u;     // Block_Maximum_Size
constexpr int kLLLog = 6, kMLLog = 6, kOFLog = 5;
constexpr int kLLSyms = 36, kMLSyms = 53, kOFSyms = 29;

// ---- encoder view of an FSE table: per symbol deltaNbBits / deltaFindState, per rank the coded state ----
struct CTab {
  uint32_t dnb[53];
  int32_t dfs[53];
  uint16_t st[64];
};
struct CTabs { CTab ll, ml, of; };               // what the device keeps in LDS while it writes a sequence section

ZE_FN int highbit(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

// RFC 8878 section 4.1: symbols with probability "less than 1" take the highest cells, the others are spread with
// step (size >> 1) + (size >> 3) + 3; the decoder numbers the cells of a symbol in cell order, so rank k of symbol s
// is its (k+1)-th cell.  The encoder keeps states in [size, 2 size).
ZE_FN void build_ctab(CTab& t, const int16_t* norm, int nsym, int log) {
  const int size = 1 << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
  uint8_t cell[64];
  int cumul[54];
  int high = size - 1;
  cumul[0] = 0;
  for (int s = 0; s < nsym; s++) {
    if (norm[s] == -1) { cumul[s + 1] = cumul[s] + 1; cell[high--] = (uint8_t)s; }
    else cumul[s + 1] = cumul[s] + norm[s];
  }
  int pos = 0;
  for (int s = 0; s < nsym; s++)
    for (int i = 0; i < norm[s]; i++) {
      cell[pos] = (uint8_t)s;
      do pos = (pos + step) & mask; while (pos > high);
    }
  for (int u = 0; u < size; u++) { const int s = cell[u]; t.st[cumul[s]++] = (uint16_t)(size + u); }
  int total = 0;
  for (int s = 0; s < nsym; s++) {
    const int n = norm[s];
    if (n == 0) { t.dnb[s] = (uint32_t)(((log + 1) << 16) - size); t.dfs[s] = 0; }
    else if (n == -1 || n == 1) { t.dnb[s] = (uint32_t)((log << 16) - size); t.dfs[s] = total - 1; total++; }
    else {
      const int maxbits = log - highbit((uint32_t)(n - 1));
      t.dnb[s] = (uint32_t)((maxbits << 16) - (n << maxbits));
      t.dfs[s] = total - n;
      total += n;
    }
  }
}

// RFC 8878 section 3.1.1.3.2.2.1
ZE_FN void build_predefined(CTabs& T) {
  const int16_t ll[kLLSyms] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
  const int16_t ml[kMLSyms] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                               -1, -1, -1, -1, -1, -1, -1};
  const int16_t of[kOFSyms] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
  build_ctab(T.ll, ll, kLLSyms, kLLLog);
  build_ctab(T.ml, ml, kMLSyms, kMLLog);
  build_ctab(T.of, of, kOFSyms, kOFLog);
}

// ---- sequence codes (RFC 8878 section 3.1.1.3.2.1.1): code, number of extra bits, extra value ----
struct Code { uint32_t code, bits, extra; };
ZE_FN Code ll_code(uint32_t ll) {          // literal length 0 .. 131071
  if (ll < 16u) return Code{ll, 0u, 0u};
  if (ll < 24u) return Code{16u + ((ll - 16u) >> 1), 1u, (ll - 16u) & 1u};
  if (ll < 32u) return Code{20u + ((ll - 24u) >> 2), 2u, (ll - 24u) & 3u};
  if (ll < 48u) return Code{22u + ((ll - 32u) >> 3), 3u, (ll - 32u) & 7u};
  if (ll < 64u) return Code{24u, 4u, ll - 48u};
  const uint32_t hb = (uint32_t)highbit(ll);                    // 64 -> 6 (code 25, 6 bits) ... 65536 -> 16 (code 35, 16 bits)
  return Code{19u + hb, hb, ll - (1u << hb)};
}
ZE_FN Code ml_code(uint32_t ml) {          // match length 3 .. 131074
  const uint32_t b = ml - 3u;
  if (b < 32u) return Code{b, 0u, 0u};
  if (b < 40u) return Code{32u + ((b - 32u) >> 1), 1u, (b - 32u) & 1u};     // 35,37,39,41
  if (b < 48u) return Code{36u + ((b - 40u) >> 2), 2u, (b - 40u) & 3u};     // 43,47
  if (b < 64u) return Code{38u + ((b - 48u) >> 3), 3u, (b - 48u) & 7u};     // 51,59
  if (b < 96u) return Code{40u + ((b - 64u) >> 4), 4u, (b - 64u) & 15u};    // 67,83
  if (b < 128u) return Code{42u, 5u, b - 96u};                              // 99
  const uint32_t hb = (uint32_t)highbit(b);                     // 128 -> 7 (code 43, 7 bits; baseline 131) ... 65536 -> 16 (code 52)
  return Code{36u + hb, hb, b - (1u << hb)};
}
// Offset_Value (RFC 8878 sections 3.1.1.3.2.1.1 and 3.1.1.5): distance + 3, or one of the three repeat codes.  With
// literals in front of the match 1 / 2 / 3 mean Repeated_Offset 1 / 2 / 3; without literals they mean Repeated_Offset 2 /
// 3 / (Repeated_Offset1 - 1).  A used repeat offset moves to the front of the history, a new distance is pushed onto it.
// (On shuffled numeric data nine of ten sequences follow their predecessor without literals and a quarter of them
// return to the distance before the last one: 5 bits instead of 16.)
struct RepState { uint32_t r1, r2, r3; };
ZE_FN void rep_init(RepState& r) { r.r1 = 1u; r.r2 = 4u; r.r3 = 8u; }          // start of a frame
ZE_FN uint32_t rep_value(RepState& r, uint32_t off, uint32_t ll) {
  uint32_t v;
  if (ll != 0u) {
    if (off == r.r1) return 1u;
    if (off == r.r2) { v = 2u; r.r2 = r.r1; r.r1 = off; return v; }
    if (off == r.r3) { v = 3u; r.r3 = r.r2; r.r2 = r.r1; r.r1 = off; return v; }
  } else {
    if (off == r.r2) { v = 1u; r.r2 = r.r1; r.r1 = off; return v; }
    if (off == r.r3) { v = 2u; r.r3 = r.r2; r.r2 = r.r1; r.r1 = off; return v; }
    if (r.r1 > 1u && off == r.r1 - 1u) { v = 3u; r.r3 = r.r2; r.r2 = r.r1; r.r1 = off; return v; }
  }
  r.r3 = r.r2; r.r2 = r.r1; r.r1 = off;
  return off + 3u;
}
ZE_FN Code of_code_value(uint32_t v) {
  const uint32_t hb = (uint32_t)highbit(v);
  return Code{hb, hb, v - (1u << hb)};
}
// one sequence as the match finder hands it over; after assign_offset_values the third field holds the Offset_Value
ZE_FN uint64_t pack_seq(uint32_t ll, uint32_t ml, uint32_t off) { return (uint64_t)ll | ((uint64_t)ml << 18) | ((uint64_t)off << 36); }
ZE_FN uint32_t seq_ll(uint64_t q) { return (uint32_t)q & 0x3ffffu; }
ZE_FN uint32_t seq_ml(uint64_t q) { return (uint32_t)(q >> 18) & 0x3ffffu; }
ZE_FN uint32_t seq_off(uint64_t q) { return (uint32_t)(q >> 36); }
ZE_FN void assign_offset_values(uint64_t* seqs, uint32_t nseq, RepState& r) {
  for (uint32_t n = 0; n < nseq; n++) seqs[n] = pack_seq(seq_ll(seqs[n]), seq_ml(seqs[n]), rep_value(r, seq_off(seqs[n]), seq_ll(seqs[n])));
}

// ---- forward bit writer (the decoder reads the section backwards from the final mark bit) ----
struct BitWriter {
  uint64_t acc;
  uint32_t nbits;
  uint8_t* out;        // next byte to write
  uint8_t* end;
  bool overflow;
  ZE_FN void init(uint8_t* o, uint8_t* e) { acc = 0; nbits = 0; out = o; end = e; overflow = false; }
  ZE_FN void add(uint32_t value, uint32_t n) {              // n <= 31, value < 2^n
    acc |= (uint64_t)value << nbits;
    nbits += n;
    if (nbits >= 32u) {
      if (out + 4 > end) { overflow = true; nbits -= 32u; acc >>= 32; return; }
      out[0] = (uint8_t)acc; out[1] = (uint8_t)(acc >> 8); out[2] = (uint8_t)(acc >> 16); out[3] = (uint8_t)(acc >> 24);
      out += 4; acc >>= 32; nbits -= 32u;
    }
  }
  ZE_FN uint8_t* close() {                                   // final mark bit, then the pending bytes
    add(1u, 1u);
    while (nbits > 0u) {
      if (out >= end) { overflow = true; break; }
      *out++ = (uint8_t)acc; acc >>= 8; nbits = nbits > 8u ? nbits - 8u : 0u;
    }
    return out;
  }
};

struct FseState { uint32_t v; };
ZE_FN void fse_init(FseState& s, const CTab& t, uint32_t sym) {
  const uint32_t nb = (t.dnb[sym] + (1u << 15)) >> 16;
  const uint32_t value = (nb << 16) - t.dnb[sym];
  s.v = t.st[(value >> nb) + (uint32_t)t.dfs[sym]];
}
ZE_FN void fse_encode(FseState& s, const CTab& t, uint32_t sym, BitWriter& bw) {
  const uint32_t nb = (s.v + t.dnb[sym]) >> 16;
  bw.add(s.v & ((1u << nb) - 1u), nb);
  s.v = t.st[(int32_t)(s.v >> nb) + t.dfs[sym]];
}

// Sequences_Section of one block (RFC 8878 section 3.1.1.3.2): count, compression-modes byte 0 (three predefined
// tables), bitstream.  `seqs` hold Offset_Values already (assign_offset_values).  Returns the end of the section, or
// nullptr when it does not fit.
ZE_FN uint8_t* write_sequences(uint8_t* out, uint8_t* end, const uint64_t* seqs, uint32_t nseq, const CTabs& T) {
  if (out + 4 > end) return nullptr;
  if (nseq == 0u) { *out++ = 0; return out; }
  if (nseq < 128u) *out++ = (uint8_t)nseq;
  else if (nseq < 0x7f00u) { *out++ = (uint8_t)((nseq >> 8) + 128u); *out++ = (uint8_t)nseq; }
  else { *out++ = 255u; *out++ = (uint8_t)(nseq - 0x7f00u); *out++ = (uint8_t)((nseq - 0x7f00u) >> 8); }
  *out++ = 0;                                           // Symbol_Compression_Modes: predefined x 3
  BitWriter bw;
  bw.init(out, end);
  FseState sll, sml, sof;
  {
    const uint64_t q = seqs[nseq - 1u];
    const Code l = ll_code(seq_ll(q)), m = ml_code(seq_ml(q)), o = of_code_value(seq_off(q));
    fse_init(sml, T.ml, m.code); fse_init(sof, T.of, o.code); fse_init(sll, T.ll, l.code);
    bw.add(l.extra, l.bits); bw.add(m.extra, m.bits);
    if (o.bits > 24u) { bw.add(o.extra & 0xffffffu, 24u); bw.add(o.extra >> 24, o.bits - 24u); } else bw.add(o.extra, o.bits);
  }
  for (uint32_t n = nseq - 1u; n-- > 0u;) {
    const uint64_t q = seqs[n];
    const Code l = ll_code(seq_ll(q)), m = ml_code(seq_ml(q)), o = of_code_value(seq_off(q));
    fse_encode(sof, T.of, o.code, bw); fse_encode(sml, T.ml, m.code, bw); fse_encode(sll, T.ll, l.code, bw);
    bw.add(l.extra, l.bits); bw.add(m.extra, m.bits);
    if (o.bits > 24u) { bw.add(o.extra & 0xffffffu, 24u); bw.add(o.extra >> 24, o.bits - 24u); } else bw.add(o.extra, o.bits);
  }
  bw.add(sml.v & ((1u << kMLLog) - 1u), kMLLog);
  bw.add(sof.v & ((1u << kOFLog) - 1u), kOFLog);
  bw.add(sll.v & ((1u << kLLLog) - 1u), kLLLog);
  uint8_t* e = bw.close();
  return bw.overflow ? nullptr : e;
}

// frame header: magic, FHD 0xA0 (single segment, 4-byte content size), content size
ZE_FN uint32_t write_frame_header(uint8_t* out, uint32_t content) {
  out[0] = (uint8_t)kMagic; out[1] = (uint8_t)(kMagic >> 8); out[2] = (uint8_t)(kMagic >> 16); out[3] = (uint8_t)(kMagic >> 24);
  out[4] = 0xA0;
  out[5] = (uint8_t)content; out[6] = (uint8_t)(content >> 8); out[7] = (uint8_t)(content >> 16); out[8] = (uint8_t)(content >> 24);
  return 9u;
}
// block header: last flag, type (0 raw, 1 RLE, 2 compressed), size
ZE_FN void write_block_header(uint8_t* out, bool last, uint32_t type, uint32_t size) {
  const uint32_t h = (last ? 1u : 0u) | (type << 1) | (size << 3);
  out[0] = (uint8_t)h; out[1] = (uint8_t)(h >> 8); out[2] = (uint8_t)(h >> 16);
}
// raw literals section header, always the 3-byte form (Size_Format 3: 20-bit regenerated size)
ZE_FN void write_raw_literals_header(uint8_t* out, uint32_t nlit) {
  const uint32_t h = 0u | (3u << 2) | (nlit << 4);
  out[0] = (uint8_t)h; out[1] = (uint8_t)(h >> 8); out[2] = (uint8_t)(h >> 16);
}
constexpr uint32_t kFrameHeader = 9u, kBlockHeader = 3u, kLitHeader = 3u;

}  // namespace zenc
}  // namespace bamd
