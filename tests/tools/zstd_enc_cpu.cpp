// zstd_enc_cpu.cpp — host-side check of csrc/zstd_enc.h (the format-writing half of the GPU Zstd encoder): a plain
// greedy hash matcher stands in for the wave-parallel match finder, the frame is written with exactly the functions
// the device uses.  tests/test_zstd_enc_cpu.py feeds the frames to the reference's ZSTD_decompress (oracle/_ref) and to
// the oracle's decoder.   Build: g++ -O2 -shared -fPIC -o tests/tools/libzstd_enc_cpu.so tests/tools/zstd_enc_cpu.cpp
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../c-blosc_amd/csrc/zstd_enc.h"

using namespace bamd::zenc;

extern "C" int zenc_cpu_compress(const uint8_t* src, int n, uint8_t* dst, int cap, int minmatch) {
  CTabs T;
  build_predefined(T);
  if (cap < 32) return 0;
  uint8_t* end = dst + cap;
  uint32_t op = write_frame_header(dst, (uint32_t)n);
  std::vector<int32_t> head(1 << 16, -1);
  std::vector<uint64_t> seqs;
  RepState rep; rep_init(rep);
  for (uint32_t s0 = 0; s0 < (uint32_t)n || s0 == 0; s0 += kBlockMax) {
    const uint32_t s1 = s0 + kBlockMax < (uint32_t)n ? s0 + kBlockMax : (uint32_t)n;
    const bool last = s1 == (uint32_t)n;
    if (op + kBlockHeader + kLitHeader + (s1 - s0) + 8 > (uint32_t)cap) return 0;
    uint8_t* bh = dst + op;
    uint8_t* lit = bh + kBlockHeader + kLitHeader;
    uint32_t nlit = 0, anchor = s0, ip = s0;
    seqs.clear();
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
    const RepState rep_before = rep;
    assign_offset_values(seqs.data(), (uint32_t)seqs.size(), rep);
    uint8_t* e = write_sequences(lit + nlit, end, seqs.data(), (uint32_t)seqs.size(), T);
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
