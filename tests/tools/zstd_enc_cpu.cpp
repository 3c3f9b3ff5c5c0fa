// zstd_enc_cpu.cpp — host-side check of csrc/zstd_enc.h (the format-writing half of the GPU Zstd encoder): a plain
// greedy hash matcher stands in for the wave-parallel match finder, the frame is written with exactly the functions
// the device uses.  tests/test_zstd_enc_cpu.py feeds the frames to the reference's ZSTD_decompress (oracle/_ref) and to
// the oracle's decoder.   Build: g++ -O2 -shared -fPIC -o tests/tools/libzstd_enc_cpu.so tests/tools/zstd_enc_cpu.cpp
//   tables: bit 0 = per block and per alphabet the cheapest of predefined / RLE_Mode / a distribution made for the block
//   (FSE_Compressed_Mode, Accuracy_Log 6) instead of the predefined FSE tables; bit 1 = Huffman-coded literals where that is
//   smaller than the raw form (code lengths: a plain Huffman tree, cut to 11 bits and repaired to a complete code)
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../c-blosc_amd/csrc/zstd_enc.h"

using namespace bamd::zenc;

namespace {
// one alphabet of one block: counts -> mode, encoder table, description
void choose_table(const uint32_t* count, int nsym, uint32_t nseq, const int16_t* predef_norm, int predef_log, const CTab& predef_tab,
                  CTab& tab, SeqTables& st, int k, uint8_t* desc) {
  int present = 0, only = 0;
  for (int s = 0; s < nsym; s++) if (count[s]) { present++; only = s; }
  st.mode[k] = kModePredefined; st.log[k] = (uint32_t)predef_log; tab = predef_tab; st.desc[k] = nullptr; st.desc_len[k] = 0;
  if (present == 1) { st.mode[k] = kModeRLE; st.rle[k] = (uint32_t)only; return; }
  int16_t norm[53];
  if (!fse_normalize(count, nsym, nseq, kCustomLog, norm)) return;
  int last = 0;
  for (int s = 0; s < nsym; s++) if (norm[s]) last = s;
  const uint32_t len = fse_write_ncount(desc, norm, last + 1, kCustomLog);
  const uint64_t cost_new = fse_cost256(count, nsym, norm, kCustomLog) + ((uint64_t)len << 11);
  const uint64_t cost_pre = fse_cost256(count, nsym, predef_norm, predef_log);
  if (cost_new >= cost_pre) return;
  build_ctab(tab, norm, last + 1, kCustomLog);
  st.mode[k] = kModeFSE; st.log[k] = kCustomLog; st.desc[k] = desc; st.desc_len[k] = len;
}
// ---- code lengths for Huffman-coded literals: classic tree, then the length limit ----
bool huf_lengths(const uint32_t* cnt, uint8_t* nbits) {
  int sym[256], n = 0;
  for (int s = 0; s < 256; s++) { nbits[s] = 0; if (cnt[s]) sym[n++] = s; }
  if (n < 2) return false;
  // nodes 0..n-1 leaves, n.. internal; repeatedly join the two lightest live roots
  uint64_t wgt[512]; int parent[512]; bool live[512];
  for (int i = 0; i < n; i++) { wgt[i] = cnt[sym[i]]; parent[i] = -1; live[i] = true; }
  int nodes = n;
  for (int m = 0; m < n - 1; m++) {
    int a = -1, b = -1;
    for (int i = 0; i < nodes; i++) if (live[i]) { if (a < 0 || wgt[i] < wgt[a]) { b = a; a = i; } else if (b < 0 || wgt[i] < wgt[b]) b = i; }
    wgt[nodes] = wgt[a] + wgt[b]; parent[nodes] = -1; live[nodes] = true; live[a] = live[b] = false; parent[a] = parent[b] = nodes; nodes++;
  }
  const int L = kHufMaxBits;
  int len[256];
  for (int i = 0; i < n; i++) { int d = 0; for (int j = i; parent[j] >= 0; j = parent[j]) d++; len[i] = d > L ? L : d; }
  // the cut may have over-subscribed the code space: lengthen the cheapest codes until it fits, then hand back what is left
  long kraft = 0;
  for (int i = 0; i < n; i++) kraft += 1L << (L - len[i]);
  while (kraft > (1L << L)) {
    int best = -1;
    for (int i = 0; i < n; i++) if (len[i] < L && (best < 0 || len[i] > len[best] || (len[i] == len[best] && wgt[i] < wgt[best]))) best = i;
    if (best < 0) return false;
    kraft -= 1L << (L - len[best] - 1); len[best]++;
  }
  long slack = (1L << L) - kraft;
  while (slack > 0) {
    int best = -1;
    for (int i = 0; i < n; i++) if (len[i] > 1 && (1L << (L - len[i])) <= slack && (best < 0 || len[i] < len[best] || (len[i] == len[best] && wgt[i] > wgt[best]))) best = i;
    if (best < 0) return false;
    slack -= 1L << (L - len[best]); len[best]--;
  }
  for (int i = 0; i < n; i++) nbits[sym[i]] = (uint8_t)len[i];
  return true;
}
const int16_t kLLNorm[kLLSyms] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
const int16_t kMLNorm[kMLSyms] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                  -1, -1, -1, -1, -1, -1, -1};
const int16_t kOFNorm[kOFSyms] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
}  // namespace

extern "C" int zenc_cpu_compress2(const uint8_t* src, int n, uint8_t* dst, int cap, int minmatch, int tables, const uint32_t* ext_seqs, int ext_nseq) {
  CTabs P;
  build_predefined(P);
  if (cap < 32) return 0;
  uint8_t* end = dst + cap;
  uint32_t op = write_frame_header(dst, (uint32_t)n);
  std::vector<int32_t> head(1 << 16, -1);
  std::vector<uint64_t> seqs;
  RepState rep; rep_init(rep);
  uint32_t ext_i = 0, ext_pos = 0;            // sequences handed in from outside (a match finder model): consumed block by block
  for (uint32_t s0 = 0; s0 < (uint32_t)n || s0 == 0; s0 += kBlockMax) {
    const uint32_t s1 = s0 + kBlockMax < (uint32_t)n ? s0 + kBlockMax : (uint32_t)n;
    const bool last = s1 == (uint32_t)n;
    if (op + kBlockHeader + kLitHeader + (s1 - s0) + 8 > (uint32_t)cap) return 0;
    uint8_t* bh = dst + op;
    uint8_t* lit = bh + kBlockHeader + kLitHeader;
    uint32_t nlit = 0, anchor = s0, ip = s0;
    seqs.clear();
    if (ext_seqs) {
      if (s0 != 0 || !last) return -1;        // the model hands over one block
      for (; ext_i < (uint32_t)ext_nseq; ext_i++) {
        const uint32_t ll = ext_seqs[3 * ext_i], ml = ext_seqs[3 * ext_i + 1], off = ext_seqs[3 * ext_i + 2];
        memcpy(lit + nlit, src + ext_pos, ll); nlit += ll;
        seqs.push_back(pack_seq(ll, ml, off));
        ext_pos += ll + ml;
      }
      anchor = ext_pos;
    } else
    while (ip + 8 < s1) {
      uint32_t v; memcpy(&v, src + ip, 4);
      const uint32_t h = (v * 2654435761u) >> 16;
      const int32_t c = head[h];
      head[h] = (int32_t)ip;
      uint32_t ml = 0;
      if (c >= 0 && ip - (uint32_t)c < (1u << 17)) while (ip + ml < s1 && src[c + ml] == src[ip + ml]) ml++;
      if (ml >= (uint32_t)minmatch) {
        const uint32_t ll = ip - anchor;
        memcpy(lit + nlit, src + anchor, ll); nlit += ll;
        seqs.push_back(pack_seq(ll, ml, ip - (uint32_t)c));
        ip += ml; anchor = ip;
      } else ip++;
    }
    memcpy(lit + nlit, src + anchor, s1 - anchor); nlit += s1 - anchor;
    write_raw_literals_header(bh + kBlockHeader, nlit);
    uint8_t* lit_end = lit + nlit;                 // where the sequences section starts
    if ((tables & 2) && nlit >= 32) {
      uint32_t cnt[256] = {0};
      for (uint32_t k = 0; k < nlit; k++) cnt[lit[k]]++;
      HufCode h;
      if (huf_lengths(cnt, h.nbits) && huf_assign_codes(h)) {
        std::vector<uint8_t> raw(lit, lit + nlit), tmp(nlit + 64);
        uint8_t* e = write_huffman_literals(tmp.data(), tmp.data() + nlit + 2, raw.data(), nlit, h);
        if (e) { memcpy(bh + kBlockHeader, tmp.data(), (size_t)(e - tmp.data())); lit_end = bh + kBlockHeader + (e - tmp.data()); }
      }
    }
    const RepState rep_before = rep;
    assign_offset_values(seqs.data(), (uint32_t)seqs.size(), rep);
    CTabs T = P;
    SeqTables st; seq_tables_predefined(st);
    uint8_t desc[3][kMaxNCountBytes];
    if ((tables & 1) && !seqs.empty()) {
      uint32_t cl[kLLSyms] = {0}, cm[kMLSyms] = {0}, co[32] = {0};
      for (uint64_t q : seqs) { cl[ll_code(seq_ll(q)).code]++; cm[ml_code(seq_ml(q)).code]++; co[of_code_value(seq_off(q)).code]++; }
      choose_table(cl, kLLSyms, (uint32_t)seqs.size(), kLLNorm, kLLLog, P.ll, T.ll, st, 0, desc[0]);
      choose_table(co, kOFSyms, (uint32_t)seqs.size(), kOFNorm, kOFLog, P.of, T.of, st, 1, desc[1]);
      choose_table(cm, kMLSyms, (uint32_t)seqs.size(), kMLNorm, kMLLog, P.ml, T.ml, st, 2, desc[2]);
    }
    uint8_t* e = write_sequences(lit_end, end, seqs.data(), (uint32_t)seqs.size(), T, &st);
    uint32_t bsize = e ? (uint32_t)(e - (bh + kBlockHeader)) : 0xffffffffu;
    if (!e || bsize >= s1 - s0) {              // not worth it: Raw_Block
      if (op + kBlockHeader + (s1 - s0) > (uint32_t)cap) return 0;
      memcpy(bh + kBlockHeader, src + s0, s1 - s0);
      bsize = s1 - s0;
      rep = rep_before;                        // a raw block leaves the decoder's repeat offsets alone
      write_block_header(bh, last, 0, bsize);
    } else write_block_header(bh, last, 2, bsize);
    op += kBlockHeader + bsize;
    if (last) break;
  }
  return (int)op;
}

extern "C" int zenc_cpu_compress(const uint8_t* src, int n, uint8_t* dst, int cap, int minmatch) {
  return zenc_cpu_compress2(src, n, dst, cap, minmatch, 0, nullptr, 0);
}
