// k_zlib.hip — zlib streams on the GPU, decode direction (codec row "Zlib" of SURVEY §8f-3): zlib_wrap_decompress
// (blosc/blosc.c:484-495) -> uncompress -> inflate (internal-complibs/zlib-1.3.1/inflate.c:590-1270) for every split of a
// Zlib chunk.  One wavefront per stream, persistent waves + ticket queue, in a kernel of its own (like Zstd's):
//   * the bit stream is serial: every lane runs the plain-C++ primitives of inflate_serial.h with the SAME values
//     (wave-uniform; tests/test_inflate_serial_cpu.py checks exactly that code on the CPU against the reference's own zlib).
//     The stream bytes come out of a 512-byte register window (wave_prims.h; the LZ4 / BloscLZ decoders used it until they moved onto the LDS ring) (one coalesced load per 256
//     bytes, fetched one slide ahead), the code tables sit in LDS (2.3 KiB per wave), written by lane 0;
//   * bytes move wave-parallel: literals are collected one per lane (with their final position) and leave 64 at a time,
//     matches are executed 16 at a time (independent ones by their own lanes in one round trip, the rest through
//     wave_match_copy with its byte-exact overlap semantics), stored blocks through wave_copy_disjoint;
//   * the Adler-32 of the output is computed by all lanes at the end (two weighted sums, RFC 1950) and compared with the
//     stream's - a stream the reference would reject with Z_DATA_ERROR is rejected here.
// Chunks of this codec are never "fused": k_unshuffle / k_bitunshuffle run afterwards as kernels of their own.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"
#include "wave_prims.h"
#include "inflate_serial.h"

namespace bamd {

// byte source of the bit reader: the register window (wave_prims.h: Window); positions only move forward
struct WinSrc {
  Window w;
  // Called from real (non-inlined) functions too, where the compiler takes every value for divergent: the position is made
  // an SGPR value explicitly (v_readlane needs a scalar lane index).
  __device__ __forceinline__ uint32_t fetch32(uint32_t pos_) {
    const uint32_t pos = uni(pos_);
    if (pos >= uni(w.in_size)) return 0u;
    w.base = uni(w.base);
    w.seek(pos);
    const uint32_t r = uni(pos - w.base), i0 = r >> 2, i1 = (i0 + 1u) & 127u;     // r < 256 after the seek
    const uint32_t a = i0 < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)w.lo, (int)i0) : (uint32_t)__builtin_amdgcn_readlane((int)w.hi, (int)(i0 - 64u));
    const uint32_t b = i1 < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)w.lo, (int)i1) : (uint32_t)__builtin_amdgcn_readlane((int)w.hi, (int)(i1 - 64u));
    return (uint32_t)((((uint64_t)b << 32) | a) >> ((r & 3u) * 8u));   // bytes beyond the end of the stream read as zero (Window::fetch)
  }
};
typedef zi::BitsT<WinSrc> WBits;

// Pending work of the symbol loop.  Literals: lane i holds the i-th pending literal and its final position (a literal's place
// is known when it is decoded), 64 leave in one scattered byte store.  Matches: lane r holds the r-th pending match; 16 are
// executed together - those whose source lies before the batch's first match are independent of the batch and are copied by
// their own lanes at once (one memory round trip for all of them), the others (long, overlapping themselves, or reading what
// this batch produces) follow in stream order through wave_match_copy.  A match per round trip, as in the first version, was
// 600 ns per symbol on reference-written bench19 streams.
struct ZlPend { uint32_t lit, lpos, nl; uint32_t mpos, mlen, moff, nm; };
constexpr uint32_t ZL_MATCH_BATCH = 16u, ZL_LANE_COPY_MAX = 64u;

__device__ __forceinline__ void zl_flush_literals(gu8* out, ZlPend& p, int lane) {
  if (p.nl == 0u) return;
  if ((uint32_t)lane < p.nl) out[p.lpos] = (uint8_t)p.lit;
  p.nl = 0u;
}
__device__ __forceinline__ void zl_exec_matches(gu8* out, ZlPend& p, int lane) {
  if (p.nm == 0u) return;
  const bool mine = (uint32_t)lane < p.nm;
  const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)p.mpos, 0);
  const bool indep = mine && p.mlen <= ZL_LANE_COPY_MAX && p.moff >= p.mlen && p.mpos - p.moff + p.mlen <= first;
  if (indep) { gu8* d = out + p.mpos; lane_copy_disjoint(d, d - p.moff, p.mlen); }
  uint32_t rest = (uint32_t)__ballot(mine && !indep);
  while (rest) {
    const int sl = __builtin_ctz(rest);
    rest &= rest - 1u;
    wave_match_copy(out, (uint32_t)__builtin_amdgcn_readlane((int)p.mpos, sl), (uint32_t)__builtin_amdgcn_readlane((int)p.moff, sl),
                    (uint32_t)__builtin_amdgcn_readlane((int)p.mlen, sl), lane);
  }
  p.nm = 0u;
}

// (Round 3 also assembled the match batches in LDS, the way zstd_exec16_lds of k_zstd.hip does: bit-exact, and no faster - 33.7 against 33.8 ms
//  per 8 GiB of reference-written bench19 streams, profiles/r03/r03ze_zlib_lds_batches_no_gain_removed.txt.  What this kernel waits for is the
//  wave-uniform symbol program, ~0.5 us per symbol, not its copies; the code was taken out again.)
// one stream -> out[0..cap); returns bytes produced, 0 on any error (zlib_wrap_decompress's contract)
__device__ __forceinline__ int zlib_decode_wave(const uint8_t* in_, int n_, uint8_t* out_, int cap_, zi::Tabs& T, int lane) {
  if (n_ <= 0) return 0;
  const uint32_t n = (uint32_t)n_, cap = (uint32_t)cap_;
  gu8* out = uni_ptr(as_global(out_));
  const gu8* in = uni_ptr(as_global(in_));
  WBits b;
  b.s.w.init(in, n, lane);
  zi::bits_start(b, n);
  if (!zi::zlib_header(b)) return 0;
  uint32_t op = 0;                        // bytes produced so far (logically: literals and matches may still be pending)
  ZlPend p = {0u, 0u, 0u, 0u, 0u, 1u, 0u};
  uint32_t last_end = 0, last_off = 0;    // end position and distance of the newest pending match (wave-uniform)
  for (;;) {
    int final = 0; uint32_t slen = 0;
    const int kind = zi::block_begin(b, T, &final, &slen);
    if (kind == zi::BLK_ERROR) return 0;
    if (kind == zi::BLK_STORED) {
      if ((uint64_t)op + slen > (uint64_t)cap) return 0;
      zl_flush_literals(out, p, lane); zl_exec_matches(out, p, lane);
      wave_copy_disjoint(out + op, in + zi::bits_bytepos(b), slen, lane);
      op += slen;
      zi::bits_skip_bytes(b, slen);
    } else {
      for (;;) {
        zi::Op o;
        const int k = zi::next_op(b, T, o);
        if (k == zi::OP_ERROR) return 0;
        if (k == zi::OP_LIT) {
          if (op >= cap) return 0;
          if ((uint32_t)lane == p.nl) { p.lit = o.len; p.lpos = op; }
          op++;
          if (++p.nl == 64u) zl_flush_literals(out, p, lane);
          continue;
        }
        if (k == zi::OP_EOB) break;
        if (o.dist > op || (uint64_t)op + o.len > (uint64_t)cap) return 0;
        // deflate cuts long matches into pieces of at most 258 bytes (a constant byte plane of 128 KiB is 508 of them): a
        // piece that continues the pending match right behind it with the same distance is the same copy, only longer
        if (p.nm && o.dist == last_off && op == last_end) {
          if ((uint32_t)lane + 1u == p.nm) p.mlen += o.len;
          op += o.len; last_end = op;
          continue;
        }
        if ((uint32_t)lane == p.nm) { p.mpos = op; p.mlen = o.len; p.moff = o.dist; }
        op += o.len; last_end = op; last_off = o.dist;
        if (++p.nm == ZL_MATCH_BATCH) { zl_flush_literals(out, p, lane); zl_exec_matches(out, p, lane); }
      }
    }
    if (final) break;
  }
  zl_flush_literals(out, p, lane); zl_exec_matches(out, p, lane);
  uint32_t want = 0;
  if (!zi::read_adler(b, &want)) return 0;
  BAMD_MEM_SYNC();                        // the checksum reads what all lanes have stored
  if (want != wave_adler32(out, op, lane)) return 0;
  return (int)op;
}

// Persistent waves over the streams of the launch's zlib chunks, splits stored raw included (k_decode_streams leaves them alone).
constexpr int ZLIB_WAVES_PER_CU = 16;
// Round 3: per-XCD queues like k_decode_streams (qoff[9] | stream indices; the host deals whole BLOCKS to the XCDs,
// queue_order.h: build_xcd_queues with BLK_ZLIB).  zlib chunks are split like LZ4's (blosc/blosc.c:929-959: only Zstd is not), so a block
// is typesize streams on typesize waves; with all of them on one XCD the wave that completes the last one can transpose the block
// out of that XCD's L2 (the hand-off of decode_one_stream: stores drained, relaxed counter, acquire = L1 invalidate) and the
// stand-alone k_unshuffle pass (3.4 ms per 8 GiB behind a 20 - 33 ms kernel) is gone.
__global__ __launch_bounds__(64, 4) void k_zlib_streams(StreamDesc* __restrict__ streams, int32_t* __restrict__ status,
                                                     uint32_t* __restrict__ tickets /*[8]*/, const int32_t* __restrict__ qlist, const int32_t* __restrict__ qoff /*[9]*/,
                                                     uint32_t* __restrict__ done, const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks,
                                                     uint32_t* __restrict__ blk_done, int single_queue) {
  __shared__ zi::Tabs tabs;
  const int lane = threadIdx.x & 63;
  const uint32_t xcc = single_queue ? 0u : (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
  const uint32_t qbase = (uint32_t)qoff[xcc], qlen = (uint32_t)qoff[xcc + 1] - qbase;
  uint32_t t = take_ticket(tickets + xcc, lane);
  uint32_t ndone = 0;
  while (t < qlen) {
    const uint32_t sid = (uint32_t)qlist[qbase + t];
    StreamDesc* sd = streams + sid;
    const int32_t csize = (int32_t)uni((uint32_t)sd->in_size), want = (int32_t)uni((uint32_t)sd->out_size);
    if (uni((uint32_t)sd->fmt) == (uint32_t)FMT_ZLIB && csize >= 0) {
      int got;
      if (csize == want) {    // split stored raw (blosc/blosc.c:773-776)
        wave_copy_disjoint(uni_ptr(as_global(sd->out)), uni_ptr(as_global(sd->in)), (uint32_t)want, lane);
        got = want;
      } else got = zlib_decode_wave(sd->in, csize, sd->out, want, tabs, lane);
      if (lane == 0) {
        sd->result = got;
        if (got != want) atomicMin(&status[sd->chunk], (int32_t)ST_BADCODEC);   // blosc.c:780-782
      }
      const ChunkDesc* c = chunks + uni((uint32_t)sd->chunk);
      const uint32_t gb = uni((uint32_t)sd->aux);
      const BlockDesc* b = blocks + gb;
      const uint32_t nstreams = uni((uint32_t)b->nstreams);
      if (got == want && (uni(c->mode) & CH_FUSED_UNSHUF)) {
        if (nstreams == 1u) fused_unshuffle_own_block(c, b, lane);
        else {
          BAMD_WAIT_STORES();
          uint32_t old = 0;
          if (lane == 0) old = __hip_atomic_fetch_add(&blk_done[gb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          old = (uint32_t)__builtin_amdgcn_readlane((int)old, 0);
          if (old + 1u == nstreams) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint32_t blk = uni((uint32_t)b->blk), bsize = uni((uint32_t)b->bsize);
            unshuffle_block_wave(c->filt + (size_t)blk * filt_block_stride(*c), c->dst + (size_t)blk * (size_t)uni((uint32_t)c->blocksize), bsize,
                                 (int)uni((uint32_t)c->typesize), lane, nullptr, nullptr, nullptr, filt_plane_stride(*c, bsize, (int)nstreams));
          }
        }
      }
    }
    ndone++;
    t = take_ticket(tickets + xcc, lane);
  }
  if (lane == 0 && ndone) atomicAdd(done, ndone);
}

}  // namespace bamd
