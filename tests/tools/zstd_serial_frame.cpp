// tests/tools/zstd_serial_frame.cpp — whole-frame Zstd decode on the CPU built ONLY from the serial primitives
// of c-blosc_amd/csrc/zstd_serial.h (the code the GPU kernel runs on single lanes) plus a byte-wise sequence
// execution.  tests/test_zstd_serial_cpu.py compiles this with g++ into a shared object and compares it with
// the oracle / the reference on every frame of the fixtures.  Test infrastructure, not product.
#include <stdlib.h>
#include <string.h>
#include "zstd_serial.h"

extern "C" int zs_decompress(const uint8_t* src, int srcsize, uint8_t* dst, int cap) {
  long long fcs; bool checksum;
  int ip = zd::frame_header(src, srcsize, &fcs, &checksum);
  if (ip < 0 || (fcs >= 0 && fcs > cap)) return 0;
  static thread_local uint16_t huf_e[2048]; static thread_local uint32_t ll_e[512], of_e[512], ml_e[512], ftab[64];
  static thread_local uint16_t next[256]; static thread_local int16_t norm[64]; static thread_local uint8_t w[256];
  zd::Huf huf = {huf_e, 0}; bool huf_valid = false;
  zd::SeqTabs tb = {{ll_e, 0}, {of_e, 0}, {ml_e, 0}, false, false, false};
  uint32_t rep[3] = {1, 4, 8};
  uint8_t* lit = (uint8_t*)malloc((1 << 17) + 8);
  int op = 0; bool ok = false;
  for (;;) {
    if (ip + 3 > srcsize) break;
    const uint32_t bh = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16);
    ip += 3;
    const int last = bh & 1, type = (bh >> 1) & 3, bsize = (int)(bh >> 3);
    if (type == 0) { if (ip + bsize > srcsize || op + bsize > cap) break; memcpy(dst + op, src + ip, (size_t)bsize); op += bsize; ip += bsize; }
    else if (type == 1) { if (ip + 1 > srcsize || op + bsize > cap) break; memset(dst + op, src[ip], (size_t)bsize); op += bsize; ip += 1; }
    else if (type == 2) {
      if (bsize > (1 << 17) || ip + bsize > srcsize) break;
      const uint8_t* b = src + ip; const int size = bsize;
      zd::LitHdr lh;
      if (!zd::lit_header(b, size, lh)) break;
      int p = lh.hdr;
      if (lh.type == 0) { if (p + lh.regen > size) break; memcpy(lit, b + p, (size_t)lh.regen); p += lh.regen; }
      else if (lh.type == 1) { if (p + 1 > size) break; memset(lit, b[p], (size_t)lh.regen); p += 1; }
      else {
        if (p + lh.csize > size) break;
        const uint8_t* hs = b + p; int hlen = lh.csize;
        if (lh.type == 2) { const int used = zd::huf_read_table(huf, hs, hlen, w, ftab, next, norm); if (used < 0) break; hs += used; hlen -= used; huf_valid = true; }
        else if (!huf_valid) break;
        if (lh.nstreams == 1) { if (!zd::huf_decode_stream(huf, hs, hlen, lit, lh.regen)) break; }
        else {
          if (hlen < 6) break;
          const int s1 = hs[0] | (hs[1] << 8), s2 = hs[2] | (hs[3] << 8), s3 = hs[4] | (hs[5] << 8), s4 = hlen - 6 - s1 - s2 - s3;
          const int q = (lh.regen + 3) / 4;
          if (s4 < 1 || 3 * q > lh.regen) break;
          const uint8_t* q0 = hs + 6;
          if (!zd::huf_decode_stream(huf, q0, s1, lit, q) || !zd::huf_decode_stream(huf, q0 + s1, s2, lit + q, q) ||
              !zd::huf_decode_stream(huf, q0 + s1 + s2, s3, lit + 2 * q, q) || !zd::huf_decode_stream(huf, q0 + s1 + s2 + s3, s4, lit + 3 * q, lh.regen - 3 * q)) break;
        }
        p += lh.csize;
      }
      int nseq; const int u0 = zd::seq_count(b + p, size - p, &nseq);
      if (u0 < 0) break;
      p += u0;
      int lp = 0; bool bad = false;
      if (nseq == 0 && p != size) break;
      if (nseq > 0) {
        if (p >= size) break;
        const int modes = b[p++];
        if (modes & 3) break;
        int u;
        if ((u = zd::seq_table(tb.ll, tb.have_ll, 0, modes >> 6, b + p, size - p, norm, next)) < 0) break;
        p += u;
        if ((u = zd::seq_table(tb.of, tb.have_of, 1, (modes >> 4) & 3, b + p, size - p, norm, next)) < 0) break;
        p += u;
        if ((u = zd::seq_table(tb.ml, tb.have_ml, 2, (modes >> 2) & 3, b + p, size - p, norm, next)) < 0) break;
        p += u;
        zd::SeqState st; st.rep[0] = rep[0]; st.rep[1] = rep[1]; st.rep[2] = rep[2];
        if (size - p < 1 || !zd::seq_begin(st, tb, b + p, size - p)) break;
        for (int i = 0; i < nseq; i++) {
          zd::Seq q;
          if (!zd::seq_next(st, tb, i + 1 == nseq, q)) { bad = true; break; }
          if ((unsigned long long)lp + q.ll > (unsigned long long)lh.regen || (unsigned long long)op + q.ll + q.ml > (unsigned long long)cap || q.off > (uint32_t)op + q.ll) { bad = true; break; }
          memcpy(dst + op, lit + lp, q.ll); op += (int)q.ll; lp += (int)q.ll;
          for (uint32_t k = 0; k < q.ml; k++) dst[op + k] = dst[op + k - q.off];
          op += (int)q.ml;
        }
        if (!bad && st.b.off != 0) bad = true;
        if (bad) break;
        rep[0] = st.rep[0]; rep[1] = st.rep[1]; rep[2] = st.rep[2];
      }
      if (op + (lh.regen - lp) > cap) break;
      memcpy(dst + op, lit + lp, (size_t)(lh.regen - lp)); op += lh.regen - lp;
      ip += bsize;
    } else break;
    if (last) { ok = true; break; }
  }
  free(lit);
  if (!ok) return 0;
  if (checksum) {
    if (ip + 4 > srcsize) return 0;
    const uint32_t want = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16) | ((uint32_t)src[ip + 3] << 24);
    if ((uint32_t)zd::xxh64(dst, (uint32_t)op) != want) return 0;
    ip += 4;
  }
  if (ip != srcsize) return 0;
  if (fcs >= 0 && fcs != op) return 0;
  return op;
}
