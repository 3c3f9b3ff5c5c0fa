// tests/tools/inflate_serial_stream.cpp — whole-stream zlib decode on the CPU built ONLY from the serial primitives of
// c-blosc_amd/csrc/inflate_serial.h (the code the GPU kernel runs wave-uniformly) plus byte-wise copies.
// tests/test_inflate_serial_cpu.py compiles this with g++ and compares it with the reference's own zlib (uncompress) on
// valid and corrupted streams.  Return value like blosc's zlib_wrap_decompress (blosc/blosc.c:484-495): bytes written, 0 on
// any failure.  Test infrastructure, not product.
#include <string.h>
#include "inflate_serial.h"

static uint32_t adler32_bytes(const uint8_t* p, uint32_t n) {
  uint32_t a = 1, b = 0;
  for (uint32_t i = 0; i < n; i++) { a = (a + p[i]) % 65521u; b = (b + a) % 65521u; }
  return (b << 16) | a;
}

extern "C" int zi_uncompress(const uint8_t* src, int srcsize, uint8_t* dst, int cap) {
  if (srcsize <= 0) return 0;
  static thread_local zi::Tabs t;
  zi::Bits b;
  zi::bits_init(b, src, (uint32_t)srcsize);
  if (!zi::zlib_header(b)) return 0;
  uint32_t op = 0;
  for (;;) {
    int final = 0; uint32_t slen = 0;
    const int kind = zi::block_begin(b, t, &final, &slen);
    if (kind == zi::BLK_ERROR) return 0;
    if (kind == zi::BLK_STORED) {
      if ((uint64_t)op + slen > (uint64_t)cap) return 0;
      memcpy(dst + op, src + zi::bits_bytepos(b), slen);
      op += slen;
      zi::bits_skip_bytes(b, slen);
    } else {
      for (;;) {
        zi::Op o;
        const int k = zi::next_op(b, t, o);
        if (k == zi::OP_ERROR) return 0;
        if (k == zi::OP_EOB) break;
        if (k == zi::OP_LIT) { if (op >= (uint32_t)cap) return 0; dst[op++] = (uint8_t)o.len; }
        else {
          if (o.dist > op || (uint64_t)op + o.len > (uint64_t)cap) return 0;
          for (uint32_t i = 0; i < o.len; i++) dst[op + i] = dst[op - o.dist + i];
          op += o.len;
        }
      }
    }
    if (final) break;
  }
  uint32_t want = 0;
  if (!zi::read_adler(b, &want)) return 0;
  if (want != adler32_bytes(dst, op)) return 0;
  return (int)op;
}
