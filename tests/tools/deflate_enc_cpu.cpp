// deflate_enc_cpu.cpp — host-side check of csrc/deflate_enc.h (the format-writing half of the GPU Zlib encoder): a plain
// greedy hash matcher stands in for the wave-parallel match finder, the stream is written with exactly the functions the
// device uses (symbol by symbol; the device packs 64 symbols per step, the bits are the same).  tests/test_deflate_enc_cpu.py
// feeds the streams to the reference's own `uncompress` (oracle/_ref) and to the oracle's decoder.
// dfl_cpu_compress2(..., dynamic = 1): ONE final block with Huffman codes made for the stream (RFC 1951 3.2.7) - two passes: the
// tokens first, then code lengths (a plain Huffman tree cut to 15 / 7 bits and repaired to a complete code), the header, the symbols.
// Build: g++ -O2 -shared -fPIC -o tests/tools/libdeflate_enc_cpu.so tests/tools/deflate_enc_cpu.cpp
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../c-blosc_amd/csrc/deflate_enc.h"

using namespace bamd::dfl;

struct BW { uint8_t* p; int cap; int pos; uint64_t acc; int n; bool ovf; };
static void put(BW& w, Sym s) {
  w.acc |= (uint64_t)s.bits << w.n; w.n += (int)s.nbits;
  while (w.n >= 8) { if (w.pos >= w.cap) { w.ovf = true; w.n -= 8; w.acc >>= 8; continue; } w.p[w.pos++] = (uint8_t)w.acc; w.acc >>= 8; w.n -= 8; }
}

extern "C" int dfl_cpu_compress(const uint8_t* src, int n, uint8_t* dst, int cap, int minmatch, int maxdist) {
  if (cap < 16) return 0;
  write_header(dst);
  BW w = {dst, cap - 4, 2, 0, 0, false};
  put(w, block_header());
  std::vector<int32_t> head(1 << 16, -1);
  int ip = 0, anchor = 0;
  while (ip + 8 < n) {
    uint32_t v; memcpy(&v, src + ip, 4);
    const uint32_t h = (v * 2654435761u) >> 16;
    const int32_t c = head[h];
    head[h] = ip;
    int ml = 0;
    if (c >= 0 && ip - c <= maxdist) while (ip + ml < n && src[c + ml] == src[ip + ml]) ml++;
    if (ml >= minmatch) {
      for (int k = anchor; k < ip; k++) put(w, literal(src[k]));
      const uint32_t np = npieces((uint32_t)ml);
      for (uint32_t k = 0; k < np; k++) put(w, match(piece_len((uint32_t)ml, k, np), (uint32_t)(ip - c)));
      ip += ml; anchor = ip;
    } else ip++;
  }
  for (int k = anchor; k < n; k++) put(w, literal(src[k]));
  put(w, end_of_block());
  if (w.n) { Sym pad = {0u, (uint32_t)(8 - w.n)}; put(w, pad); }
  if (w.ovf) return 0;
  uint32_t a = 1, b = 0;
  for (int i = 0; i < n; i++) { a = (a + src[i]) % 65521u; b = (b + a) % 65521u; }
  const uint32_t ad = (b << 16) | a;
  for (int i = 0; i < 4; i++) dst[w.pos + i] = (uint8_t)(ad >> (24 - 8 * i));
  return w.pos + 4;
}

// ---- dynamic Huffman blocks ----
namespace {
// code lengths <= limit for counts cnt[0..n); a complete code when at least two symbols occur, length 1 for a lone one
void huff_lengths(const uint32_t* cnt, int n, int limit, uint8_t* len) {
  std::vector<int> sym;
  for (int s = 0; s < n; s++) { len[s] = 0; if (cnt[s]) sym.push_back(s); }
  const int m = (int)sym.size();
  if (m == 0) return;
  if (m == 1) { len[sym[0]] = 1; return; }
  std::vector<uint64_t> wgt(2 * m); std::vector<int> parent(2 * m, -1); std::vector<char> live(2 * m, 0);
  for (int i = 0; i < m; i++) { wgt[i] = cnt[sym[i]]; live[i] = 1; }
  int nodes = m;
  for (int k = 0; k < m - 1; k++) {
    int a = -1, b = -1;
    for (int i = 0; i < nodes; i++) if (live[i]) { if (a < 0 || wgt[i] < wgt[a]) { b = a; a = i; } else if (b < 0 || wgt[i] < wgt[b]) b = i; }
    wgt[nodes] = wgt[a] + wgt[b]; live[nodes] = 1; live[a] = live[b] = 0; parent[a] = parent[b] = nodes; nodes++;
  }
  std::vector<int> l(m);
  const int L = limit;
  for (int i = 0; i < m; i++) { int d = 0; for (int j = i; parent[j] >= 0; j = parent[j]) d++; l[i] = d > L ? L : d; }
  long kraft = 0;
  for (int i = 0; i < m; i++) kraft += 1L << (L - l[i]);
  while (kraft > (1L << L)) {
    int best = -1;
    for (int i = 0; i < m; i++) if (l[i] < L && (best < 0 || l[i] > l[best] || (l[i] == l[best] && wgt[i] < wgt[best]))) best = i;
    kraft -= 1L << (L - l[best] - 1); l[best]++;
  }
  long slack = (1L << L) - kraft;
  while (slack > 0) {
    int best = -1;
    for (int i = 0; i < m; i++) if (l[i] > 1 && (1L << (L - l[i])) <= slack && (best < 0 || l[i] < l[best] || (l[i] == l[best] && wgt[i] > wgt[best]))) best = i;
    if (best < 0) break;
    slack -= 1L << (L - l[best]); l[best]--;
  }
  for (int i = 0; i < m; i++) len[sym[i]] = (uint8_t)l[i];
}
struct Tok { uint32_t lit_or_len, dist; };       // dist == 0: a literal byte
}  // namespace

extern "C" int dfl_cpu_compress2(const uint8_t* src, int n, uint8_t* dst, int cap, int minmatch, int maxdist, int dynamic) {
  if (!dynamic) return dfl_cpu_compress(src, n, dst, cap, minmatch, maxdist);
  if (cap < 16) return 0;
  // ---- pass 1: tokens ----
  std::vector<Tok> toks;
  std::vector<int32_t> head(1 << 16, -1);
  int ip = 0, anchor = 0;
  auto lits = [&](int a, int b) { for (int k = a; k < b; k++) toks.push_back(Tok{src[k], 0u}); };
  while (ip + 8 < n) {
    uint32_t v; memcpy(&v, src + ip, 4);
    const uint32_t h = (v * 2654435761u) >> 16;
    const int32_t c = head[h];
    head[h] = ip;
    int ml = 0;
    if (c >= 0 && ip - c <= maxdist) while (ip + ml < n && src[c + ml] == src[ip + ml]) ml++;
    if (ml >= minmatch) {
      lits(anchor, ip);
      const uint32_t np = npieces((uint32_t)ml);
      for (uint32_t k = 0; k < np; k++) toks.push_back(Tok{piece_len((uint32_t)ml, k, np), (uint32_t)(ip - c)});
      ip += ml; anchor = ip;
    } else ip++;
  }
  lits(anchor, n);
  // ---- the two codes ----
  uint32_t lc[kLitLenSyms] = {0}, dc[kDistSyms] = {0};
  for (const Tok& t : toks) { if (t.dist) { const MatchSyms m = match_symbols(t.lit_or_len, t.dist); lc[m.lsym]++; dc[m.dsym]++; } else lc[t.lit_or_len]++; }
  lc[256] = 1;
  DynCodes C;
  huff_lengths(lc, kLitLenSyms, kMaxCodeBits, C.llen);
  huff_lengths(dc, kDistSyms, kMaxCodeBits, C.dlen);
  assign_codes(C.llen, kLitLenSyms, C.lcode);
  assign_codes(C.dlen, kDistSyms, C.dcode);
  int nlit = kLitLenSyms; while (nlit > 257 && C.llen[nlit - 1] == 0) nlit--;
  int ndist = kDistSyms; while (ndist > 1 && C.dlen[ndist - 1] == 0) ndist--;
  // ---- the header: code lengths, run-length coded, under a code of their own ----
  uint8_t all[kLitLenSyms + kDistSyms], cs[kLitLenSyms + kDistSyms], ce[kLitLenSyms + kDistSyms];
  memcpy(all, C.llen, (size_t)nlit); memcpy(all + nlit, C.dlen, (size_t)ndist);
  const int ncs = code_length_symbols(all, nlit + ndist, cs, ce);
  uint32_t cc[kCodeLenSyms] = {0};
  for (int k = 0; k < ncs; k++) cc[cs[k]]++;
  uint8_t cl[kCodeLenSyms]; uint16_t ccode[kCodeLenSyms];
  huff_lengths(cc, kCodeLenSyms, kMaxCodeLenBits, cl);
  { int used = 0, only = 0; for (int s2 = 0; s2 < kCodeLenSyms; s2++) if (cl[s2]) { used++; only = s2; }
    if (used == 1) cl[only == 0 ? 1 : 0] = 1; }                          // inflate wants this code complete: a second 1-bit code nobody uses
  assign_codes(cl, kCodeLenSyms, ccode);
  int ncl = kCodeLenSyms; while (ncl > 4 && cl[kCodeLenOrder[ncl - 1]] == 0) ncl--;
  write_header(dst);
  BW w = {dst, cap - 4, 2, 0, 0, false};
  put(w, Sym{1u | (2u << 1), 3u});                                        // BFINAL = 1, BTYPE = 10
  put(w, Sym{(uint32_t)(nlit - 257), 5u}); put(w, Sym{(uint32_t)(ndist - 1), 5u}); put(w, Sym{(uint32_t)(ncl - 4), 4u});
  for (int k = 0; k < ncl; k++) put(w, Sym{cl[kCodeLenOrder[k]], 3u});
  for (int k = 0; k < ncs; k++) {
    put(w, Sym{ccode[cs[k]], cl[cs[k]]});
    if (cs[k] == 16) put(w, Sym{ce[k], 2u}); else if (cs[k] == 17) put(w, Sym{ce[k], 3u}); else if (cs[k] == 18) put(w, Sym{ce[k], 7u});
  }
  // ---- pass 2: the symbols ----
  for (const Tok& t : toks) {
    if (!t.dist) { put(w, dyn_literal(C, t.lit_or_len)); continue; }
    Sym a, b; dyn_match(C, t.lit_or_len, t.dist, a, b);
    put(w, a); put(w, b);
  }
  put(w, dyn_end_of_block(C));
  if (w.n) { Sym pad = {0u, (uint32_t)(8 - w.n)}; put(w, pad); }
  if (w.ovf) return 0;
  uint32_t a = 1, b = 0;
  for (int i = 0; i < n; i++) { a = (a + src[i]) % 65521u; b = (b + a) % 65521u; }
  const uint32_t ad = (b << 16) | a;
  for (int i = 0; i < 4; i++) dst[w.pos + i] = (uint8_t)(ad >> (24 - 8 * i));
  return w.pos + 4;
}
