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
constexpr uint32_t kLitHeaderRaw = 3u;            // the raw literals header this writer always uses (write_raw_literals_header)
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

// ---- per-block tables (RFC 8878 section 3.1.1.3.2.1: Symbol_Compression_Modes; section 4.1.1: FSE table description) ----
// A table of a block is one of: the predefined distribution (mode 0), one symbol for every sequence (RLE_Mode 1: one byte
// names it, the state needs no bits), or a distribution made for this block (FSE_Compressed_Mode 2: its description
// travels in front of the bitstream).  The block-made tables here always have Accuracy_Log kCustomLog = 6 - the encoder
// table stays 64 cells, like the predefined ones - and no "less than 1" probabilities, so cell k of the spread is simply
// (k * step) & mask.
enum : uint32_t { kModePredefined = 0u, kModeRLE = 1u, kModeFSE = 2u };
constexpr int kCustomLog = 6;
constexpr uint32_t kMaxNCountBytes = 48u;         // 4 + 53 symbols x <= 7 bits, rounded up generously

// counts -> probabilities that sum to 1 << log, every present symbol >= 1.  Proportional shares rounded down (at least 1),
// what is left over goes to the most frequent symbol; when the floor of 1 has handed out too much, the largest shares
// give it back one by one.  false: fewer than two symbols present (RLE_Mode's case) or more symbols than cells.
ZE_FN bool fse_normalize(const uint32_t* count, int nsym, uint32_t total, int log, int16_t* norm) {
  const int size = 1 << log;
  int present = 0, sum = 0, big = 0;
  for (int s = 0; s < nsym; s++) {
    int v = 0;
    if (count[s]) { present++; v = (int)(((uint64_t)count[s] << log) / total); if (v < 1) v = 1; }
    norm[s] = (int16_t)v; sum += v;
    if (count[s] > count[big]) big = s;
  }
  if (present < 2 || present > size) return false;
  if (sum <= size) { norm[big] = (int16_t)(norm[big] + size - sum); return true; }
  for (int over = sum - size; over > 0; over--) {
    int m = 0;
    for (int s = 1; s < nsym; s++) if (norm[s] > norm[m]) m = s;
    if (norm[m] < 2) return false;
    norm[m]--;
  }
  return true;
}
// FSE table description (RFC 8878 section 4.1.1) of `norm` (no -1 entries), Accuracy_Log `log`; returns its length in bytes.
// Fields are little-endian bit fields in a forward bitstream: 4 bits log - 5, then per symbol value = probability + 1 in
// a field whose width depends on how many points are still to be given out; a probability of 0 is followed by 2-bit
// repeat flags that count further zeros.
ZE_FN uint32_t fse_write_ncount(uint8_t* out, const int16_t* norm, int nsym, int log) {
  struct Bits {
    uint8_t* out; uint64_t acc; uint32_t nb, pos;
    ZE_FN void put(uint32_t v, uint32_t n) { acc |= (uint64_t)v << nb; nb += n; while (nb >= 8u) { out[pos++] = (uint8_t)acc; acc >>= 8; nb -= 8u; } }
  } b = {out, 0, 0, 0};
  b.put((uint32_t)(log - 5), 4u);
  int remaining = 1 << log;                    // points still to be given out
  int s = 0;
  while (remaining > 0 && s < nsym) {
    const uint32_t value = (uint32_t)(norm[s] + 1);
    const uint32_t bits = (uint32_t)highbit((uint32_t)remaining + 1u) + 1u;       // enough for 0 .. remaining + 1
    const uint32_t low = (1u << bits) - 1u - ((uint32_t)remaining + 1u);          // this many small values take one bit less
    if (value < low) b.put(value, bits - 1u);
    else if (value < (1u << (bits - 1u))) b.put(value, bits);
    else b.put(value + low, bits);
    remaining -= norm[s];
    s++;
    if (value == 1u) {                          // probability 0: how many more zeros follow, 2 bits at a time
      int z = 0;
      while (s + z < nsym && norm[s + z] == 0) z++;
      s += z;
      for (; z >= 3; z -= 3) b.put(3u, 2u);
      b.put((uint32_t)z, 2u);
    }
  }
  if (b.nb) out[b.pos++] = (uint8_t)b.acc;
  return b.pos;
}
// bits the sequences' codes of one alphabet cost with table `norm` / log (for choosing a mode; 1/256 bit units)
ZE_FN uint64_t fse_cost256(const uint32_t* count, int nsym, const int16_t* norm, int log) {
  uint64_t c = 0;
  for (int s = 0; s < nsym; s++) if (count[s]) {
    const int n = norm[s] == -1 ? 1 : norm[s];
    if (n <= 0) return ~0ull >> 1;               // a symbol the table cannot code
    // -log2(n / size) in 1/256 bits: integer part from the high bit, fraction linearly between powers of two
    const int hbit = highbit((uint32_t)n);
    const uint32_t frac = ((uint32_t)n << 8 >> hbit) - 256u;                        // 0..255
    const uint32_t bits256 = (uint32_t)((log - hbit) << 8) - frac;                   // log2(1 + x) ~ x
    c += (uint64_t)count[s] * bits256;
  }
  return c;
}

struct SeqTables {              // what write_sequences needs to know about the three tables of a block
  uint32_t mode[3];             // kMode*: literal lengths, offsets, match lengths (the order of the modes byte)
  uint32_t log[3];
  uint32_t rle[3];              // RLE_Mode: the symbol
  const uint8_t* desc[3];       // FSE_Compressed_Mode: table description bytes
  uint32_t desc_len[3];
};
ZE_FN void seq_tables_predefined(SeqTables& st) {
  for (int k = 0; k < 3; k++) { st.mode[k] = kModePredefined; st.rle[k] = 0; st.desc[k] = nullptr; st.desc_len[k] = 0; }
  st.log[0] = kLLLog; st.log[1] = kOFLog; st.log[2] = kMLLog;
}

// Sequences_Section of one block (RFC 8878 section 3.1.1.3.2): count, compression-modes byte, table descriptions in the order
// literal lengths / offsets / match lengths, bitstream.  `seqs` hold Offset_Values already (assign_offset_values); T holds
// the encoder tables that go with `st` (an RLE table is not consulted).  Returns the end of the section, or nullptr when it
// does not fit.
ZE_FN uint8_t* write_sequences(uint8_t* out, uint8_t* end, const uint64_t* seqs, uint32_t nseq, const CTabs& T, const SeqTables* stp = nullptr) {
  SeqTables st0;
  if (!stp) { seq_tables_predefined(st0); stp = &st0; }
  const SeqTables& st = *stp;
  if (out + 4 > end) return nullptr;
  if (nseq == 0u) { *out++ = 0; return out; }
  if (nseq < 128u) *out++ = (uint8_t)nseq;
  else if (nseq < 0x7f00u) { *out++ = (uint8_t)((nseq >> 8) + 128u); *out++ = (uint8_t)nseq; }
  else { *out++ = 255u; *out++ = (uint8_t)(nseq - 0x7f00u); *out++ = (uint8_t)((nseq - 0x7f00u) >> 8); }
  *out++ = (uint8_t)((st.mode[0] << 6) | (st.mode[1] << 4) | (st.mode[2] << 2));   // Symbol_Compression_Modes
  for (int k = 0; k < 3; k++) {
    if (st.mode[k] == kModeRLE) { if (out + 1 > end) return nullptr; *out++ = (uint8_t)st.rle[k]; }
    else if (st.mode[k] == kModeFSE) { if (out + st.desc_len[k] > end) return nullptr; memcpy(out, st.desc[k], st.desc_len[k]); out += st.desc_len[k]; }
  }
  const bool fll = st.mode[0] != kModeRLE, fof = st.mode[1] != kModeRLE, fml = st.mode[2] != kModeRLE;
  BitWriter bw;
  bw.init(out, end);
  FseState sll, sml, sof;
  sll.v = sml.v = sof.v = 0;
  {
    const uint64_t q = seqs[nseq - 1u];
    const Code l = ll_code(seq_ll(q)), m = ml_code(seq_ml(q)), o = of_code_value(seq_off(q));
    if (fml) fse_init(sml, T.ml, m.code);
    if (fof) fse_init(sof, T.of, o.code);
    if (fll) fse_init(sll, T.ll, l.code);
    bw.add(l.extra, l.bits); bw.add(m.extra, m.bits);
    if (o.bits > 24u) { bw.add(o.extra & 0xffffffu, 24u); bw.add(o.extra >> 24, o.bits - 24u); } else bw.add(o.extra, o.bits);
  }
  for (uint32_t n = nseq - 1u; n-- > 0u;) {
    const uint64_t q = seqs[n];
    const Code l = ll_code(seq_ll(q)), m = ml_code(seq_ml(q)), o = of_code_value(seq_off(q));
    if (fof) fse_encode(sof, T.of, o.code, bw);
    if (fml) fse_encode(sml, T.ml, m.code, bw);
    if (fll) fse_encode(sll, T.ll, l.code, bw);
    bw.add(l.extra, l.bits); bw.add(m.extra, m.bits);
    if (o.bits > 24u) { bw.add(o.extra & 0xffffffu, 24u); bw.add(o.extra >> 24, o.bits - 24u); } else bw.add(o.extra, o.bits);
  }
  if (fml) bw.add(sml.v & ((1u << st.log[2]) - 1u), st.log[2]);
  if (fof) bw.add(sof.v & ((1u << st.log[1]) - 1u), st.log[1]);
  if (fll) bw.add(sll.v & ((1u << st.log[0]) - 1u), st.log[0]);
  uint8_t* e = bw.close();
  return bw.overflow ? nullptr : e;
}

// ---- Huffman-coded literals (RFC 8878 sections 3.1.1.3.1 and 4.2) ----
// A prefix code over the literal bytes, code lengths 1..kHufMaxBits, given as `nbits[256]` (0 = byte does not occur) and
// COMPLETE (the Kraft sum is exactly 1): the format has no way to say otherwise - it transmits weights
// w = maxbits + 1 - nbits for all bytes but the last present one, whose weight is whatever completes a power of two.
// Codes are canonical in the format's own order: within a length by byte value, and longer codes BELOW shorter ones
// (the decoder's table starts with the longest codes).
constexpr int kHufMaxBits = 11;
struct HufCode { uint16_t code[256]; uint8_t nbits[256]; int maxbits; int last; };     // last = highest byte value present
// false: not a complete code of at least two symbols within the length limit
ZE_FN bool huf_assign_codes(HufCode& h) {
  uint32_t count[kHufMaxBits + 2] = {0};
  int maxbits = 0, present = 0; h.last = -1;
  for (int s = 0; s < 256; s++) if (h.nbits[s]) { if (h.nbits[s] > kHufMaxBits) return false; count[h.nbits[s]]++; if (h.nbits[s] > maxbits) maxbits = h.nbits[s]; present++; h.last = s; }
  if (present < 2) return false;
  uint32_t kraft = 0;
  for (int b = 1; b <= maxbits; b++) kraft += count[b] << (maxbits - b);
  if (kraft != (1u << maxbits)) return false;
  // first table index of every length: the longest codes come first
  uint32_t start[kHufMaxBits + 2]; uint32_t at = 0;
  for (int b = maxbits; b >= 1; b--) { start[b] = at; at += count[b] << (maxbits - b); }
  for (int s = 0; s < 256; s++) {
    const int b = h.nbits[s];
    if (!b) { h.code[s] = 0; continue; }
    h.code[s] = (uint16_t)(start[b] >> (maxbits - b));
    start[b] += 1u << (maxbits - b);
  }
  h.maxbits = maxbits;
  return true;
}
// Huffman_Tree_Description: weights of bytes 0 .. last-1, direct (4 bits each, up to 128 of them) or FSE-compressed (two
// interleaved states).  Returns its size, 0 when neither form can carry these weights (the caller stores the literals raw).
ZE_FN uint32_t huf_write_tree(uint8_t* out, uint32_t room, const HufCode& h) {
  const int nw = h.last;                                     // the last present byte's weight is implied
  uint8_t w[256];
  for (int s = 0; s < nw; s++) w[s] = h.nbits[s] ? (uint8_t)(h.maxbits + 1 - h.nbits[s]) : 0;
  // ---- FSE-compressed: alphabet 0..11, Accuracy_Log <= 6 ----
  uint32_t cnt[12] = {0};
  for (int s = 0; s < nw; s++) cnt[w[s]]++;
  int16_t norm[16]; int maxsym = 0;
  for (int v = 0; v < 12; v++) if (cnt[v]) maxsym = v;
  uint32_t fse_len = 0;
  uint8_t tmp[160];
  if (nw >= 2 && fse_normalize(cnt, maxsym + 1, (uint32_t)nw, 6, norm)) {
    CTab t;
    build_ctab(t, norm, maxsym + 1, 6);
    const uint32_t hl = fse_write_ncount(tmp, norm, maxsym + 1, 6);
    BitWriter bw; bw.init(tmp + hl, tmp + sizeof tmp);
    // weight k is decoded from state 1 when k is even, from state 2 when odd; the two last weights only initialise
    // the states (with the state that needs the most bits next: the decoder must run dry right behind them)
    FseState s1, s2; s1.v = s2.v = 0;
    bool i1 = false, i2 = false;
    for (int k = nw - 1; k >= 0; k--) {
      FseState& st = (k & 1) ? s2 : s1; bool& inited = (k & 1) ? i2 : i1;
      if (!inited) { fse_init(st, t, w[k]); inited = true; } else fse_encode(st, t, w[k], bw);
    }
    bw.add(s2.v & 63u, 6u); bw.add(s1.v & 63u, 6u);
    uint8_t* e = bw.close();
    if (!bw.overflow && (uint32_t)(e - tmp) < 128u) fse_len = (uint32_t)(e - tmp);
  }
  const uint32_t direct_len = nw <= 128 ? (uint32_t)((nw + 1) / 2) : 0u;
  if (fse_len && (!direct_len || fse_len < direct_len)) {
    if (1u + fse_len > room) return 0u;
    out[0] = (uint8_t)fse_len; memcpy(out + 1, tmp, fse_len);
    return 1u + fse_len;
  }
  if (!direct_len || 1u + direct_len > room) return 0u;
  out[0] = (uint8_t)(127 + nw);
  for (int k = 0; k < nw; k += 2) out[1 + k / 2] = (uint8_t)((w[k] << 4) | (k + 1 < nw ? w[k + 1] : 0));
  return 1u + direct_len;
}
// one Huffman stream: the decoder reads it backwards from the final mark bit and meets the FIRST byte's code first
ZE_FN uint8_t* huf_write_stream(uint8_t* out, uint8_t* end, const uint8_t* lit, uint32_t n, const HufCode& h) {
  BitWriter bw; bw.init(out, end);
  for (uint32_t k = n; k-- > 0u;) bw.add(h.code[lit[k]], h.nbits[lit[k]]);
  uint8_t* e = bw.close();
  return bw.overflow ? nullptr : e;
}
// Literals_Section with Literals_Block_Type 2 (Compressed): header (3 / 4 / 5 bytes by size), tree, and ONE stream (up to 1023
// literal bytes, if it fits the 10-bit sizes) or FOUR with their jump table.  Returns the end of the section, nullptr when it
// does not fit or is not smaller than the raw form (3 + n bytes).
ZE_FN uint8_t* write_huffman_literals(uint8_t* out, uint8_t* end, const uint8_t* lit, uint32_t n, const HufCode& h) {
  if (n < 8u || out + 5 + 6 > end) return nullptr;
  const bool single = n < 256u;                                  // short runs: one stream, no jump table
  const uint32_t hdr = (single || n < 1024u) ? 3u : (n < 16384u ? 4u : 5u);
  uint8_t* p = out + hdr;
  const uint32_t tl = huf_write_tree(p, (uint32_t)(end - p), h);
  if (!tl) return nullptr;
  p += tl;
  uint32_t csize;
  if (single) {
    uint8_t* e = huf_write_stream(p, end, lit, n, h);
    if (!e) return nullptr;
    csize = (uint32_t)(e - (out + hdr));
    p = e;
  } else {
    if (p + 6 > end) return nullptr;
    uint8_t* jump = p; p += 6;
    const uint32_t q = (n + 3u) / 4u;
    for (int k = 0; k < 4; k++) {
      const uint32_t from = q * (uint32_t)k, len = k < 3 ? q : n - 3u * q;
      uint8_t* e = huf_write_stream(p, end, lit + from, len, h);
      if (!e) return nullptr;
      const uint32_t sz = (uint32_t)(e - p);
      if (k < 3) { if (sz > 0xffffu) return nullptr; jump[2 * k] = (uint8_t)sz; jump[2 * k + 1] = (uint8_t)(sz >> 8); }
      p = e;
    }
    csize = (uint32_t)(p - (out + hdr));
  }
  if (csize + hdr >= n + kLitHeaderRaw) return nullptr;
  // header: type 2 | Size_Format << 2 | regenerated size | compressed size (10 / 10, 14 / 14 or 18 / 18 bits)
  if (hdr == 3u) {
    if (csize >= 1024u) return nullptr;
    const uint32_t v = 2u | ((single ? 0u : 1u) << 2) | (n << 4) | (csize << 14);
    out[0] = (uint8_t)v; out[1] = (uint8_t)(v >> 8); out[2] = (uint8_t)(v >> 16);
  } else if (hdr == 4u) {
    if (csize >= 16384u) return nullptr;
    const uint32_t v = 2u | (2u << 2) | (n << 4) | (csize << 18);
    out[0] = (uint8_t)v; out[1] = (uint8_t)(v >> 8); out[2] = (uint8_t)(v >> 16); out[3] = (uint8_t)(v >> 24);
  } else {
    if (n >= (1u << 18) || csize >= (1u << 18)) return nullptr;
    const uint64_t v = 2u | (3u << 2) | ((uint64_t)n << 4) | ((uint64_t)csize << 22);
    for (int k = 0; k < 5; k++) out[k] = (uint8_t)(v >> (8 * k));
  }
  return p;
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
