// k_decode.hip — per-stream LZ4 / BloscLZ decoders + the block "plan" kernel (rows K5, K7, D1 of
// SURVEY §8a).
//
// Replaces  blosc_d's split loop (blosc/blosc.c:760-787) and the codecs it calls:
//   LZ4_decompress_safe   (internal-complibs/lz4-1.10.0/lz4.c:2451 -> LZ4_decompress_generic :2023-2445)
//   blosclz_decompress    (blosc/blosclz.c:679-789)
// One wavefront decodes one stream (= one split of one block).  The sequence parse is a scalar
// (SGPR) program over a 512-byte register window of the compressed bytes; literal and match bytes
// move with all 64 lanes.  Output goes straight to global memory (the plane-major scratch, or the
// destination when the chunk has no filter); history reads hit L1/L2.
//
// Algorithmic HBM bytes per stream: csize read + neblock written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"
#include "wave_prims.h"

namespace bamd {

constexpr int DEC_WAVES = 1;  // one stream per workgroup: streams differ 1000x in sequence count, a
                              // multi-wave workgroup would hold its slots until the slowest is done

// ---------------------------------------------------------------------------------------------
// plan: walk every block's csize chain once (blosc/blosc.c:760-771), emit StreamDesc entries
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t ld_i32(const uint8_t* p) { return g_ld_i32le(as_global(p)); }

__global__ void k_decode_plan(const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks,
                              StreamDesc* __restrict__ streams, int32_t* __restrict__ status, int nblocks_total) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nblocks_total) return;
  const BlockDesc b = blocks[g];
  const ChunkDesc& c = chunks[b.chunk];
  const int32_t bsize = b.bsize;
  const int32_t neblock = bsize / b.nstreams;
  uint8_t* out = ((c.mode & (CH_SHUFFLE | CH_BITSHUFFLE)) ? c.filt : c.dst) + (size_t)b.blk * c.blocksize;
  int32_t off = ld_i32(c.src + 16 + 4 * (size_t)b.blk);
  bool bad = false;
  for (int j = 0; j < b.nstreams; j++) {
    StreamDesc s;
    s.chunk = b.chunk; s.fmt = c.fmt; s.aux = 0; s.result = 0;
    s.out = out + (size_t)j * neblock; s.out_size = neblock;
    s.in = nullptr; s.in_size = -1;  // -1: nothing to decode (chain broken)
    if (!bad) {
      if (off < 0 || off > c.cbytes - 4) bad = true;
      else {
        int32_t cs = ld_i32(c.src + off);
        off += 4;
        if (cs < 0 || cs > c.cbytes - off) bad = true;
        else { s.in = c.src + off; s.in_size = cs; off += cs; }
      }
    }
    streams[b.first_stream + j] = s;
  }
  if (bad) atomicMin(&status[b.chunk], (int32_t)ST_BADCHAIN);
}

// ---------------------------------------------------------------------------------------------
// LZ4 block decode, one wave.  Returns bytes produced (== cap on success) or a negative number.
// Acceptance rules are those of the reference's safe loop (lz4.c:2215-2435):
//   literal-length extension stops reading at n-15, match-length extension at n-4;
//   a literal run reaching within 12 bytes of the output end or 8 of the input end must be the
//   last one and end exactly at the input end; offset <= bytes produced; a match must end at
//   least 5 bytes before the output end.  Offset 0 (accepted by the reference with unspecified
//   output) is rejected here.
// ---------------------------------------------------------------------------------------------
__device__ int lz4_decode_wave(const gu8* __restrict__ in, int32_t n_, gu8* out, int32_t cap_, int lane) {
  if (cap_ == 0) return (n_ == 1 && in[0] == 0) ? 0 : -1;
  if (n_ <= 0) return -1;
  const uint32_t n = (uint32_t)n_, cap = (uint32_t)cap_;
  Window w;
  w.init(in, n, lane);
  uint32_t ip = 0, op = 0;
  for (;;) {
    w.seek(ip);
    const uint32_t hdr = w.peek32(ip);
    const uint32_t token = hdr & 0xffu;
    ip += 1;
    uint32_t ll = token >> 4;
    if (ll == 15u) {
      if (n < 15u || ip >= n - 15u) return -2;
      uint32_t s;
      do { s = w.byte_at(ip); ip++; ll += s; if (ip > n - 15u || ll > cap) return -2; } while (s == 255u);
    }
    // ---- literals ----
    if (op + ll + 12u > cap || ip + ll + 8u > n) {
      // must be the final run
      if (ip + ll != n || op + ll > cap) return -3;
      wave_copy_disjoint(out + op, in + ip, ll, lane);
      op += ll;
      break;
    }
    const uint32_t lit_src = ip;
    // literal bytes are taken out of the register window BEFORE it may slide for the offset
    const bool lit_in_win = ll <= 64u && lit_src + ll <= w.base + 512u;
    uint32_t litv = 0;
    if (ll && lit_in_win) litv = w.gather_bytes(lit_src);
    ip += ll;
    w.seek(ip);
    uint32_t t2 = w.peek32(ip);
    const uint32_t off = t2 & 0xffffu;
    ip += 2;
    uint32_t ml = token & 15u;
    if (ml == 15u) {
      uint32_t s = (t2 >> 16) & 0xffu;   // first extension byte is already in the peeked word
      ip++; ml += s;
      if (ip > n - 4u) return -4;
      while (s == 255u) {
        s = w.byte_at(ip); ip++; ml += s;
        if (ip > n - 4u || ml > cap) return -4;
      }
    }
    ml += 4u;
    const uint32_t mpos = op + ll;
    if (off == 0u || off > mpos) return -5;
    if (mpos + ml + 5u > cap) return -6;

    if (lit_in_win && ll + ml <= 64u && off >= ml && (ll == 0u || off >= ll + ml)) {
      // short sequence whose match cannot see its own literals: one gather, one 64-lane store
      uint32_t v = litv;
      const uint32_t k = (uint32_t)lane - ll;
      if (k < ml) v = out[mpos - off + k];
      if ((uint32_t)lane < ll + ml) out[op + lane] = (uint8_t)v;
    } else {
      if (ll) {
        if (lit_in_win) { if ((uint32_t)lane < ll) out[op + lane] = (uint8_t)litv; }
        else wave_copy_disjoint(out + op, in + lit_src, ll, lane);
      }
      wave_match_copy(out, mpos, off, ml, lane);
    }
    op = mpos + ml;
  }
  return (int)op;
}

// ---------------------------------------------------------------------------------------------
// BloscLZ decode, one wave (blosclz.c:679-789).  Returns bytes produced; 0 on any violation,
// like the reference.  Kept quirks: the first control byte is masked with 31; a match is executed
// only if at least one more input byte follows it (otherwise decoding stops BEFORE the copy).
// ---------------------------------------------------------------------------------------------
__device__ int blosclz_decode_wave(const gu8* __restrict__ in, int32_t n_, gu8* out, int32_t cap_, int lane) {
  if (n_ <= 0) return 0;
  const uint32_t n = (uint32_t)n_, cap = (uint32_t)cap_;
  Window w;
  w.init(in, n, lane);
  uint32_t ip = 1, op = 0;
  uint32_t ctrl = w.peek32(0) & 31u;
  for (;;) {
    if (ctrl >= 32u) {
      uint32_t len = (ctrl >> 5) - 1u;
      uint32_t ofs = (ctrl & 31u) << 8;
      uint32_t code;
      if (len == 6u) {
        do {
          if (ip + 1u >= n) return 0;
          code = w.byte_at(ip); ip++;
          len += code;
          if (len > cap) return 0;
        } while (code == 255u);
      } else if (ip + 1u >= n) return 0;
      code = w.byte_at(ip); ip++;
      len += 3u;
      uint32_t dist = ofs + code;             // distance - 1
      if (code == 255u && ofs == (31u << 8)) {
        if (ip + 1u >= n) return 0;
        w.seek(ip);
        uint32_t t = w.peek32(ip);
        dist = (((t & 0xffu) << 8) | ((t >> 8) & 0xffu)) + 8191u;
        ip += 2;
      }
      if (op + len > cap) return 0;
      if (dist + 1u > op) return 0;           // reference: ref - 1 < output
      if (ip >= n) break;                     // quirk: the pending match is dropped
      ctrl = w.byte_at(ip); ip++;
      wave_match_copy(out, op, dist + 1u, len, lane);
      op += len;
    } else {
      const uint32_t run = ctrl + 1u;         // 1..32 literal bytes
      if (op + run > cap) return 0;
      if (ip + run > n) return 0;
      w.seek(ip);
      uint32_t v = w.gather_bytes(ip);        // run <= 32 and seek => inside the window
      if ((uint32_t)lane < run) out[op + lane] = (uint8_t)v;
      op += run; ip += run;
      if (ip >= n) break;
      ctrl = w.byte_at(ip); ip++;
    }
  }
  return (int)op;
}

// ---------------------------------------------------------------------------------------------
// decode kernel: grid = ceil(nstreams / DEC_WAVES), block = 64 * DEC_WAVES
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * DEC_WAVES) void k_decode_streams(StreamDesc* __restrict__ streams,
                                                                   int32_t* __restrict__ status, int nstreams) {
  const int lane = threadIdx.x & 63;
  static_assert(DEC_WAVES == 1, "xcd_spread assumes one stream per workgroup");
  if ((int)blockIdx.x >= nstreams) return;
  const int sid = (int)uni(xcd_spread(blockIdx.x, (uint32_t)nstreams));
  StreamDesc* sd = streams + sid;
  const gu8* in = as_global(sd->in);
  const int32_t csize = (int32_t)uni((uint32_t)sd->in_size);
  const int32_t want = (int32_t)uni((uint32_t)sd->out_size);
  gu8* out = as_global(sd->out);
  if (csize < 0) return;  // chain error already recorded by the plan kernel
  int got;
  if (csize == want) {    // split stored raw (blosc/blosc.c:773-776)
    wave_copy_disjoint(out, in, (uint32_t)want, lane);
    got = want;
  } else if (sd->fmt == FMT_LZ4) {
    got = lz4_decode_wave(in, csize, out, want, lane);
  } else {
    got = blosclz_decode_wave(in, csize, out, want, lane);
  }
  if (lane == 0) {
    sd->result = got;
    if (got != want) atomicMin(&status[sd->chunk], (int32_t)ST_BADCODEC);  // blosc.c:780-782
  }
}

}  // namespace bamd
