// deflate_enc_cpu.cpp — host-side check of csrc/deflate_enc.h (the format-writing half of the GPU Zlib encoder): a plain
// greedy hash matcher stands in for the wave-parallel match finder, the stream is written with exactly the functions the
// device uses (symbol by symbol; the device packs 64 symbols per step, the bits are the same).  tests/test_deflate_enc_cpu.py
// feeds the streams to the reference's own `uncompress` (oracle/_ref) and to the oracle's decoder.
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
