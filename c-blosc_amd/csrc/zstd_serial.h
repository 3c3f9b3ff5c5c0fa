// zstd_serial.h — the inherently SERIAL parts of a Zstandard frame decode (codec row K8, SURVEY §8a:
// zstd_wrap_decompress blosc/blosc.c:515-522 -> ZSTD_decompress): bit readers, FSE table description and
// table build (zstd_decompress_block.c:485-592), Huffman table description, table build and stream decode
// (HUF_readStats / HUF_decompress4X1), literals / sequences section headers (:134, :695) and the sequence
// decoder with its repeat-offset history (:1229-1301).  Written from the format (RFC 8878), like
// oracle/zstd_oracle.c, but without malloc, with every read bounded by the stream, and with all tables in
// caller-provided memory (LDS on the GPU).
//
// The file is plain C++ and compiles for BOTH sides: k_zstd.hip runs these functions on one lane (four for
// the Huffman streams) of the wave that owns the frame and does the sequence EXECUTION wave-parallel;
// tests/tools/zstd_serial_frame.cpp compiles the same code with g++ and decodes whole frames with it on the
// CPU, so everything here is checked against the oracle and the reference without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZD_FN __device__ __forceinline__
#else
#define ZD_FN static inline
#endif

namespace zd {

ZD_FN int hb32(uint32_t v) { return 31 - __builtin_clz(v); }   // v != 0

// ---------------- bit readers: never touch a byte outside [p, p + len) ----------------
// forward (FSE table descriptions): bit k of the stream = bit (k & 7) of byte k >> 3
struct Fwd { const uint8_t* p; int len; int pos; };   // pos in bits
ZD_FN uint32_t fwd_read(Fwd& b, int n) {
  uint32_t v = 0;
  for (int k = 0; k < n; k++) {
    const int bit = b.pos + k, byte = bit >> 3;
    const uint32_t x = byte < b.len ? (b.p[byte] >> (bit & 7)) & 1u : 0u;
    v |= x << k;
  }
  b.pos += n;
  return v;
}

// backward (Huffman and FSE payloads): starts just below the end marker, runs towards bit 0; bits below
// bit 0 read as zero and make `off` negative (the callers' corruption checks look at `off`)
struct Back { const uint8_t* p; int bytepos; uint64_t acc; int nacc; int off; };
ZD_FN bool back_init(Back& b, const uint8_t* p, int len) {
  if (len <= 0 || p[len - 1] == 0) return false;
  const int top = hb32(p[len - 1]);                 // marker bit position in the last byte
  b.p = p; b.bytepos = len - 1;
  b.acc = p[len - 1] & ((1u << top) - 1u); b.nacc = top;
  b.off = (len - 1) * 8 + top;
  return true;
}
ZD_FN uint32_t back_read(Back& b, int n) {          // n <= 32
  if (n == 0) return 0u;
  if (b.nacc < n && b.bytepos >= 4) {                 // refill four bytes at a time (one unaligned load) ...
    uint32_t v; __builtin_memcpy(&v, b.p + b.bytepos - 4, 4);
    b.bytepos -= 4; b.acc = (b.acc << 32) | v; b.nacc += 32;
  }
  while (b.nacc < n && b.bytepos > 0) { b.bytepos--; b.acc = (b.acc << 8) | b.p[b.bytepos]; b.nacc += 8; }   // ... single bytes near the start
  if (b.nacc < n) { b.acc <<= (n - b.nacc); b.nacc = n; }       // past the start: zero bits
  const uint32_t v = (uint32_t)(b.acc >> (b.nacc - n)) & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u));
  b.nacc -= n;
  b.acc &= (b.nacc ? ((1ull << b.nacc) - 1ull) : 0ull);
  b.off -= n;
  return v;
}

// ---------------- FSE ----------------
// table entry: symbol | nbBits << 8 | baseline << 16; at most 512 entries
struct Fse { uint32_t* e; int al; };
ZD_FN int fse_sym(uint32_t e) { return (int)(e & 0xffu); }
ZD_FN int fse_nb(uint32_t e) { return (int)((e >> 8) & 0xffu); }
ZD_FN uint32_t fse_base(uint32_t e) { return e >> 16; }

// norm: normalized counts (-1 = "less than one"); `next` is scratch for 256 uint16
// compact (optional): the same table as 16-bit cells symbol | x << 6, x = the cell's next-state counter (nbBits = al - floor(log2 x),
// baseline = (x << nbBits) - 2^al): what the one-lane-per-frame sequence kernel reads (k_zstd2.hip: k_zstd_seq)
ZD_FN bool fse_build(Fse& t, const int16_t* norm, int nsym, int al, uint16_t* next, uint16_t* compact = nullptr) {
  const int size = 1 << al;
  int high = size - 1;
  t.al = al;
  for (int s = 0; s < nsym; s++) {
    if (norm[s] == -1) { t.e[high--] = (uint32_t)s; next[s] = 1; }
    else next[s] = (uint16_t)norm[s];
  }
  const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
  int pos = 0;
  for (int s = 0; s < nsym; s++) {
    if (norm[s] <= 0) continue;
    for (int i = 0; i < norm[s]; i++) {
      t.e[pos] = (uint32_t)s;
      do { pos = (pos + step) & mask; } while (pos > high);
    }
  }
  if (pos != 0) return false;
  for (int i = 0; i < size; i++) {
    const uint32_t s = t.e[i] & 0xffu;
    const uint32_t x = next[s]++;
    const uint32_t nb = (uint32_t)(al - hb32(x));
    t.e[i] = s | (nb << 8) | ((((x << nb) - (uint32_t)size) & 0xffffu) << 16);
    if (compact) compact[i] = (uint16_t)(s | (x << 6));
  }
  return true;
}
ZD_FN void fse_rle(Fse& t, int sym) { t.al = 0; t.e[0] = (uint32_t)sym; }

// FSE table description: bytes consumed, or -1
ZD_FN int fse_read_ncount(const uint8_t* src, int srcsize, int max_al, int max_sym, int16_t* norm, int* nsym_out, int* al_out) {
  Fwd b = {src, srcsize, 0};
  const int al = 5 + (int)fwd_read(b, 4);
  if (al > max_al) return -1;
  int remaining = 1 << al, s = 0;
  while (remaining > 0 && s <= max_sym) {
    const int nbits = hb32((uint32_t)remaining + 1u) + 1;
    uint32_t val = fwd_read(b, nbits);
    const uint32_t lower = (1u << (nbits - 1)) - 1u;
    const uint32_t thresh = (1u << nbits) - 1u - ((uint32_t)remaining + 1u);
    if ((val & lower) < thresh) { b.pos--; val &= lower; }
    else if (val > lower) val -= thresh;
    const int proba = (int)val - 1;
    remaining -= proba < 0 ? -proba : proba;
    norm[s++] = (int16_t)proba;
    if (proba == 0) {
      int rep = (int)fwd_read(b, 2);
      for (;;) {
        for (int i = 0; i < rep && s <= max_sym; i++) norm[s++] = 0;
        if (rep != 3) break;
        rep = (int)fwd_read(b, 2);
      }
    }
    if (b.pos > srcsize * 8) return -1;
  }
  if (remaining != 0 || s > max_sym + 1) return -1;
  *nsym_out = s; *al_out = al;
  const int used = (b.pos + 7) >> 3;
  return used <= srcsize ? used : -1;
}

// ---------------- Huffman ----------------
// table entry: symbol | nbBits << 8; 2^maxbits <= 2048 entries
struct Huf { uint16_t* e; int maxbits; };

// weights[0..nw): the last weight is implied.  `w` is scratch for 256 bytes.
ZD_FN bool huf_build(Huf& t, uint8_t* w, int nw) {
  uint32_t total = 0;
  if (nw < 1 || nw > 255) return false;
  for (int i = 0; i < nw; i++) { if (w[i] > 11) return false; if (w[i]) total += 1u << (w[i] - 1); }
  if (total == 0) return false;
  const int maxbits = hb32(total) + 1;
  if (maxbits > 11) return false;
  const uint32_t left = (1u << maxbits) - total;
  if (left & (left - 1u)) return false;
  w[nw] = (uint8_t)(hb32(left) + 1);
  const int n = nw + 1;
  uint32_t rank_count[13], rank_idx[13];
  for (int i = 0; i < 13; i++) rank_count[i] = 0;
  for (int i = 0; i < n; i++) rank_count[w[i] ? maxbits + 1 - w[i] : 0]++;
  rank_idx[maxbits] = 0;
  for (int i = maxbits; i >= 1; i--) rank_idx[i - 1] = rank_idx[i] + rank_count[i] * (1u << (maxbits - i));
  if (rank_idx[0] != (1u << maxbits)) return false;
  for (int i = 0; i < n; i++) {
    if (!w[i]) continue;
    const int bits = maxbits + 1 - w[i];
    const uint32_t len = 1u << (maxbits - bits);
    const uint16_t ent = (uint16_t)(i | (bits << 8));
    for (uint32_t k = 0; k < len; k++) t.e[rank_idx[bits] + k] = ent;
    rank_idx[bits] += len;
  }
  t.maxbits = maxbits;
  return true;
}

// Huffman tree description: bytes consumed, or -1.  `w` scratch 256 bytes, `ftab` scratch for 64 FSE entries,
// `next` scratch 256 uint16, `norm` scratch 16 int16.
ZD_FN int huf_read_table(Huf& t, const uint8_t* src, int srcsize, uint8_t* w, uint32_t* ftab, uint16_t* next, int16_t* norm) {
  if (srcsize < 1) return -1;
  const int hb = src[0];
  int nw = 0, used;
  if (hb >= 128) {
    nw = hb - 127;
    used = 1 + (nw + 1) / 2;
    if (used > srcsize) return -1;
    for (int i = 0; i < nw; i++) w[i] = (i & 1) ? (uint8_t)(src[1 + i / 2] & 15) : (uint8_t)(src[1 + i / 2] >> 4);
  } else {
    used = 1 + hb;
    if (hb == 0 || used > srcsize) return -1;
    int nsym, al;
    const int h = fse_read_ncount(src + 1, hb, 6, 12, norm, &nsym, &al);
    if (h < 0) return -1;
    Fse ft = {ftab, 0};
    if (!fse_build(ft, norm, nsym, al, next)) return -1;
    Back b;
    if (!back_init(b, src + 1 + h, hb - h)) return -1;
    uint32_t s1 = back_read(b, al), s2 = back_read(b, al);
    for (;;) {
      if (nw >= 254) return -1;
      w[nw++] = (uint8_t)fse_sym(ft.e[s1]);
      s1 = fse_base(ft.e[s1]) + back_read(b, fse_nb(ft.e[s1]));
      if (b.off < 0) { w[nw++] = (uint8_t)fse_sym(ft.e[s2]); break; }
      if (nw >= 254) return -1;
      w[nw++] = (uint8_t)fse_sym(ft.e[s2]);
      s2 = fse_base(ft.e[s2]) + back_read(b, fse_nb(ft.e[s2]));
      if (b.off < 0) { w[nw++] = (uint8_t)fse_sym(ft.e[s1]); break; }
    }
  }
  if (!huf_build(t, w, nw)) return -1;
  return used;
}

// one Huffman stream -> exactly n symbols
ZD_FN bool huf_decode_stream(const Huf& t, const uint8_t* src, int len, uint8_t* out, int n) {
  Back b;
  if (!back_init(b, src, len)) return false;
  const int mb = t.maxbits; const uint32_t mask = (1u << mb) - 1u;
  uint32_t state = back_read(b, mb);
  int i = 0;
  for (; i < n && b.off > -mb; i++) {
    const uint16_t e = t.e[state];
    out[i] = (uint8_t)e;
    const int nb = e >> 8;
    state = ((state << nb) & mask) | back_read(b, nb);
  }
  return i == n && b.off == -mb;
}

// ---------------- section headers ----------------
struct LitHdr { int type, regen, csize, nstreams, hdr; };   // type 0 raw, 1 RLE, 2 compressed, 3 treeless
ZD_FN bool lit_header(const uint8_t* src, int size, LitHdr& h) {
  if (size < 1) return false;
  h.type = src[0] & 3; const int sf = (src[0] >> 2) & 3;
  h.csize = 0; h.nstreams = 1;
  if (h.type < 2) {
    if (sf == 0 || sf == 2) { h.hdr = 1; h.regen = src[0] >> 3; }
    else if (sf == 1) { if (size < 2) return false; h.hdr = 2; h.regen = (src[0] >> 4) | (src[1] << 4); }
    else { if (size < 3) return false; h.hdr = 3; h.regen = (src[0] >> 4) | (src[1] << 4) | (src[2] << 12); }
  } else {
    if (size < 3) return false;
    uint64_t v = 0;
    for (int k = 0; k < 5 && k < size; k++) v |= (uint64_t)src[k] << (8 * k);
    if (sf == 0 || sf == 1) { h.hdr = 3; h.regen = (int)((v >> 4) & 0x3ff); h.csize = (int)((v >> 14) & 0x3ff); h.nstreams = sf == 0 ? 1 : 4; }
    else if (sf == 2) { h.hdr = 4; h.regen = (int)((v >> 4) & 0x3fff); h.csize = (int)((v >> 18) & 0x3fff); h.nstreams = 4; }
    else { h.hdr = 5; h.regen = (int)((v >> 4) & 0x3ffff); h.csize = (int)((v >> 22) & 0x3ffff); h.nstreams = 4; }
    if (h.hdr > size) return false;
  }
  return h.regen <= (1 << 17);
}

// number of sequences; returns bytes consumed or -1
ZD_FN int seq_count(const uint8_t* src, int size, int* nseq) {
  if (size < 1) return -1;
  int n = src[0], used = 1;
  if (n >= 128) {
    if (n == 255) { if (size < 3) return -1; n = src[1] + (src[2] << 8) + 0x7f00; used = 3; }
    else { if (size < 2) return -1; n = ((n - 128) << 8) + src[1]; used = 2; }
  }
  *nseq = n;
  return used;
}

// ---------------- sequences ----------------
// RFC 8878 3.1.1.3.2.1.1: code -> baseline, number of extra bits
ZD_FN uint32_t ll_base(int c) { return c < 16 ? (uint32_t)c : (c < 20 ? 16u + 2u * (uint32_t)(c - 16) : (c < 22 ? 24u + 4u * (uint32_t)(c - 20) : (c < 24 ? 32u + 8u * (uint32_t)(c - 22) : (c == 24 ? 48u : 64u << (c - 25))))); }
ZD_FN int ll_bits(int c) { return c < 16 ? 0 : (c < 20 ? 1 : (c < 22 ? 2 : (c < 24 ? 3 : (c == 24 ? 4 : c - 19)))); }
ZD_FN uint32_t ml_base(int c) {
  if (c < 32) return (uint32_t)c + 3u;
  if (c < 36) return 35u + 2u * (uint32_t)(c - 32);
  if (c < 38) return 43u + 4u * (uint32_t)(c - 36);
  if (c < 40) return 51u + 8u * (uint32_t)(c - 38);
  if (c < 42) return 67u + 16u * (uint32_t)(c - 40);
  if (c == 42) return 99u;
  return 3u + (128u << (c - 43));
}
ZD_FN int ml_bits(int c) { return c < 32 ? 0 : (c < 36 ? 1 : (c < 38 ? 2 : (c < 40 ? 3 : (c < 42 ? 4 : (c == 42 ? 5 : c - 36))))); }

ZD_FN int16_t ll_default(int s) { return s == 0 ? 4 : ((s == 1 || s == 25) ? 3 : ((s >= 32) ? -1 : (((s >= 13 && s <= 15) || (s >= 27 && s <= 31)) ? 1 : 2))); }
ZD_FN int16_t ml_default(int s) { return s == 0 ? 1 : (s == 1 ? 4 : (s == 2 ? 3 : (s <= 8 ? 2 : (s >= 46 ? -1 : 1)))); }
ZD_FN int16_t of_default(int s) { return (s >= 6 && s <= 8) ? 2 : (s >= 24 ? -1 : 1); }

struct SeqTabs { Fse ll, of, ml; bool have_ll, have_of, have_ml; };

// table of one kind (0 LL, 1 OF, 2 ML) in the given mode; bytes consumed or -1.  norm: scratch 64 int16.
ZD_FN int seq_table(Fse& t, bool& have, int kind, int mode, const uint8_t* src, int srcsize, int16_t* norm, uint16_t* next) {
  const int max_sym = kind == 0 ? 35 : (kind == 1 ? 31 : 52), max_al = kind == 1 ? 8 : 9;
  if (mode == 0) {
    const int n = kind == 0 ? 36 : (kind == 1 ? 29 : 53), al = kind == 1 ? 5 : 6;
    for (int s = 0; s < n; s++) norm[s] = kind == 0 ? ll_default(s) : (kind == 1 ? of_default(s) : ml_default(s));
    if (!fse_build(t, norm, n, al, next)) return -1;
    have = true; return 0;
  }
  if (mode == 1) { if (srcsize < 1 || src[0] > max_sym) return -1; fse_rle(t, src[0]); have = true; return 1; }
  if (mode == 2) {
    int nsym, al;
    const int h = fse_read_ncount(src, srcsize, max_al, max_sym, norm, &nsym, &al);
    if (h < 0 || !fse_build(t, norm, nsym, al, next)) return -1;
    have = true; return h;
  }
  return have ? 0 : -1;
}

struct Seq { uint32_t ll, ml, off; };
struct SeqState { Back b; uint32_t sl, so, sm; uint32_t rep[3]; };

ZD_FN bool seq_begin(SeqState& st, const SeqTabs& tb, const uint8_t* src, int len) {
  if (!back_init(st.b, src, len)) return false;
  st.sl = back_read(st.b, tb.ll.al); st.so = back_read(st.b, tb.of.al); st.sm = back_read(st.b, tb.ml.al);
  return true;
}
// next sequence; `last` = no state update behind it.  false on corruption.
ZD_FN bool seq_next(SeqState& st, const SeqTabs& tb, bool last, Seq& q) {
  const uint32_t el = tb.ll.e[st.sl], eo = tb.of.e[st.so], em = tb.ml.e[st.sm];
  const int lc = fse_sym(el), oc = fse_sym(eo), mc = fse_sym(em);
  if (oc > 31 || mc > 52 || lc > 35) return false;
  const uint32_t ov = (1u << oc) + back_read(st.b, oc);              // offset bits first, then match, then literal length
  q.ml = ml_base(mc) + back_read(st.b, ml_bits(mc));
  q.ll = ll_base(lc) + back_read(st.b, ll_bits(lc));
  if (!last) {                                                       // state updates: LL, ML, OF
    st.sl = fse_base(el) + back_read(st.b, fse_nb(el));
    st.sm = fse_base(em) + back_read(st.b, fse_nb(em));
    st.so = fse_base(eo) + back_read(st.b, fse_nb(eo));
  }
  if (st.b.off < 0) return false;
  if (ov > 3) { q.off = ov - 3u; st.rep[2] = st.rep[1]; st.rep[1] = st.rep[0]; st.rep[0] = q.off; }
  else {
    uint32_t idx = ov - 1u;
    if (q.ll == 0) idx++;
    if (idx == 0) q.off = st.rep[0];
    else {
      q.off = idx < 3 ? st.rep[idx] : st.rep[0] - 1u;
      if (q.off == 0) return false;
      if (idx > 1) st.rep[2] = st.rep[1];
      st.rep[1] = st.rep[0]; st.rep[0] = q.off;
    }
  }
  return true;
}

// frame header: bytes consumed or -1; content size in *fcs (-1 if absent)
ZD_FN int frame_header(const uint8_t* src, int srcsize, long long* fcs, bool* checksum) {
  if (srcsize < 6) return -1;
  if (!(src[0] == 0x28 && src[1] == 0xB5 && src[2] == 0x2F && src[3] == 0xFD)) return -1;
  const int fhd = src[4];
  const int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
  if ((fhd & 0x08) || did) return -1;
  *checksum = ((fhd >> 2) & 1) != 0;
  int ip = 5 + (single ? 0 : 1);
  const int nb = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
  if (ip + nb > srcsize) return -1;
  unsigned long long v = 0;
  for (int k = 0; k < nb; k++) v |= (unsigned long long)src[ip + k] << (8 * k);
  if (nb == 2) v += 256;
  *fcs = nb ? (long long)v : -1;
  return ip + nb;
}

// ---------------- content checksum ----------------
// XXH64 (seed 0) from the published algorithm; a frame's Content_Checksum is its low 32 bits (RFC 8878 3.1.1)
ZD_FN uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
ZD_FN uint64_t rd64(const uint8_t* p) { uint64_t v = 0; for (int k = 7; k >= 0; k--) v = (v << 8) | p[k]; return v; }
ZD_FN uint64_t xxh64(const uint8_t* p, uint32_t len) {
  const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                 P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
  const uint8_t* end = p + len;
  uint64_t h;
  if (len >= 32u) {
    uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
    while (p + 32 <= end) {
      v1 = rotl64(v1 + rd64(p) * P2, 31) * P1; v2 = rotl64(v2 + rd64(p + 8) * P2, 31) * P1;
      v3 = rotl64(v3 + rd64(p + 16) * P2, 31) * P1; v4 = rotl64(v4 + rd64(p + 24) * P2, 31) * P1;
      p += 32;
    }
    h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = (h ^ (rotl64(v1 * P2, 31) * P1)) * P1 + P4; h = (h ^ (rotl64(v2 * P2, 31) * P1)) * P1 + P4;
    h = (h ^ (rotl64(v3 * P2, 31) * P1)) * P1 + P4; h = (h ^ (rotl64(v4 * P2, 31) * P1)) * P1 + P4;
  } else h = P5;
  h += (uint64_t)len;
  while (p + 8 <= end) { h ^= rotl64(rd64(p) * P2, 31) * P1; h = rotl64(h, 27) * P1 + P4; p += 8; }
  if (p + 4 <= end) { h ^= (uint64_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
  while (p < end) { h ^= (uint64_t)(*p++) * P5; h = rotl64(h, 11) * P1; }
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

}  // namespace zd
