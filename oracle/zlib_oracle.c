/* oracle/zlib_oracle.c — CPU oracle for the zlib streams inside Zlib blosc chunks.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement, from RFC 1950 / RFC 1951, of what blosc's zlib_wrap_decompress (blosc/blosc.c:484-495) gets from
 * uncompress() (internal-complibs/zlib-1.3.1/uncompr.c:27-85 -> inflate(), inflate.c:590-1270; code tables inftrees.c:32-299):
 * a bit-at-a-time canonical Huffman decoder (no lookup tables - deliberately not the code the product runs,
 * c-blosc_amd/csrc/inflate_serial.h) with the reference's acceptance rules:
 *   over-subscribed code -> error; incomplete code -> error unless its longest code is one bit long (literal/length and
 *   distance codes only, inftrees.c:130-135) or it is a distance code without symbols; the code-length code must be complete;
 *   a dynamic block needs a code for symbol 256 (inflate.c:1003); HLIT > 286 / HDIST > 30 (inflate.c:929);
 *   symbols 286 / 287 and distance symbols 30 / 31 are errors where they are decoded; a distance may not reach before the
 *   start of the output (inflate.c:1175); input that ends early and a wrong Adler-32 (inflate.c:1215) are errors; bytes
 *   behind the stream are ignored.
 * Parity status: PINNED against the reference's own zlib (oracle/_ref, `uncompress`) by tests/test_oracle_zlib.py: streams
 * written by zlib at every level / strategy, hand-built legal and illegal blocks (tests/deflate_builder.py), bit flips,
 * truncations - verdict and bytes; and the five Zlib compat vectors of the reference (tests/golden/compat) decode to
 * arange(1e6, int32) through orc_decompress.
 * Return value like the wrapper's: bytes written, 0 on any failure. */
#include <stdint.h>
#include <string.h>
#include "blosc_oracle.h"

typedef struct { const uint8_t* p; long n; long bit; int bad; } zo_bits;   /* bit: position in bits; bad: ran off the end */
static unsigned zo_get(zo_bits* b, int k) {
  unsigned v = 0;
  for (int i = 0; i < k; i++) {
    const long byte = b->bit >> 3;
    if (byte >= b->n) { b->bad = 1; b->bit++; continue; }
    v |= (unsigned)((b->p[byte] >> (b->bit & 7)) & 1) << i;
    b->bit++;
  }
  return v;
}

typedef struct { short count[16]; short symbol[288]; } zo_huff;
/* 0 complete, > 0 incomplete, < 0 over-subscribed; *longest = longest code length in use */
static int zo_build(zo_huff* h, const uint8_t* len, int n, int* longest) {
  short offs[16];
  memset(h->count, 0, sizeof h->count);
  for (int s = 0; s < n; s++) h->count[len[s]]++;
  *longest = 0;
  int left = 1;
  for (int l = 1; l <= 15; l++) {
    left = 2 * left - h->count[l];
    if (left < 0) return left;
    if (h->count[l]) *longest = l;
  }
  offs[1] = 0;
  for (int l = 1; l < 15; l++) offs[l + 1] = (short)(offs[l] + h->count[l]);
  for (int s = 0; s < n; s++) if (len[s]) h->symbol[offs[len[s]]++] = (short)s;
  return left;
}
static int zo_decode(zo_bits* b, const zo_huff* h) {
  int code = 0, first = 0, index = 0;
  for (int l = 1; l <= 15; l++) {
    code |= (int)zo_get(b, 1);
    const int c = h->count[l];
    if (code - c < first) return h->symbol[index + (code - first)];
    index += c; first = (first + c) << 1; code <<= 1;
  }
  return -1;
}

static const short zo_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const short zo_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const short zo_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const short zo_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

int orc_zlib_decompress(const void* src_, int srcsize, void* dst_, int dstcap) {
  const uint8_t* src = (const uint8_t*)src_;
  uint8_t* dst = (uint8_t*)dst_;
  if (srcsize < 2) return 0;
  /* RFC 1950 header: method 8, window <= 32 KiB, check bits, no preset dictionary */
  const unsigned cmf = src[0], flg = src[1];
  if (((cmf << 8) | flg) % 31 || (cmf & 15) != 8 || (cmf >> 4) > 7 || (flg & 0x20)) return 0;
  zo_bits b = {src, srcsize, 16, 0};
  long op = 0;
  int last;
  do {
    last = (int)zo_get(&b, 1);
    const unsigned type = zo_get(&b, 2);
    if (b.bad) return 0;
    if (type == 0) {                                   /* stored */
      b.bit = (b.bit + 7) & ~7L;
      const unsigned len = zo_get(&b, 16), nlen = zo_get(&b, 16);
      if (b.bad || len != (~nlen & 0xffffu)) return 0;
      if ((b.bit >> 3) + (long)len > b.n || op + (long)len > dstcap) return 0;
      memcpy(dst + op, src + (b.bit >> 3), len);
      op += len; b.bit += 8L * len;
      continue;
    }
    if (type == 3) return 0;
    uint8_t lengths[320];
    zo_huff lit, dist;
    int nlen = 288, ndist = 32, longest;
    if (type == 1) {                                   /* fixed codes: symbols 286, 287 / 30, 31 have codes and are invalid */
      for (int s = 0; s < 288; s++) lengths[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
      for (int s = 0; s < 32; s++) lengths[288 + s] = 5;
    } else {
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      nlen = (int)zo_get(&b, 5) + 257; ndist = (int)zo_get(&b, 5) + 1;
      const int ncode = (int)zo_get(&b, 4) + 4;
      if (b.bad || nlen > 286 || ndist > 30) return 0;
      uint8_t cl[19]; memset(cl, 0, sizeof cl);
      for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)zo_get(&b, 3);
      zo_huff clh;
      const int left = zo_build(&clh, cl, 19, &longest);
      if (left < 0 || (left > 0 && longest != 0)) return 0;
      if (longest == 0) return 0;                      /* no code-length code at all: nothing can follow (the reference runs on and fails on the missing symbol 256) */
      uint8_t all[320];
      int have = 0;
      while (have < nlen + ndist) {
        const int s = zo_decode(&b, &clh);
        if (s < 0 || b.bad) return 0;
        if (s < 16) { all[have++] = (uint8_t)s; continue; }
        int rep; uint8_t val = 0;
        if (s == 16) { if (!have) return 0; val = all[have - 1]; rep = 3 + (int)zo_get(&b, 2); }
        else if (s == 17) rep = 3 + (int)zo_get(&b, 3);
        else rep = 11 + (int)zo_get(&b, 7);
        if (have + rep > nlen + ndist) return 0;
        while (rep--) all[have++] = val;
      }
      if (b.bad || all[256] == 0) return 0;
      memcpy(lengths, all, (size_t)nlen);
      memcpy(lengths + 288, all + nlen, (size_t)ndist);
    }
    int left = zo_build(&lit, lengths, nlen, &longest);
    if (left < 0 || (left > 0 && longest != 1)) return 0;
    left = zo_build(&dist, lengths + 288, ndist, &longest);
    if (left < 0 || (left > 0 && longest > 1)) return 0;
    for (;;) {
      const int s = zo_decode(&b, &lit);
      if (s < 0 || b.bad) return 0;
      if (s < 256) { if (op >= dstcap) return 0; dst[op++] = (uint8_t)s; continue; }
      if (s == 256) break;
      if (s > 285) return 0;
      const long len = zo_lbase[s - 257] + (long)zo_get(&b, zo_lext[s - 257]);
      const int d = zo_decode(&b, &dist);
      if (d < 0 || d > 29) return 0;
      const long dd = zo_dbase[d] + (long)zo_get(&b, zo_dext[d]);
      if (b.bad || dd > op || op + len > dstcap) return 0;
      for (long i = 0; i < len; i++) dst[op + i] = dst[op - dd + i];
      op += len;
    }
  } while (!last);
  /* Adler-32 of the plain bytes, big endian, on the next byte boundary */
  b.bit = (b.bit + 7) & ~7L;
  if ((b.bit >> 3) + 4 > b.n) return 0;
  const uint8_t* q = src + (b.bit >> 3);
  const uint32_t want = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
  uint32_t a = 1, c = 0;
  for (long i = 0; i < op; i++) { a = (a + dst[i]) % 65521u; c = (c + a) % 65521u; }
  if (want != ((c << 16) | a)) return 0;
  return (int)op;
}
