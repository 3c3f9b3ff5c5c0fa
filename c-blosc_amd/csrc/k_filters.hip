// k_filters.hip — byte-shuffle / bit-shuffle filters for gfx950 (rows K1, K2, K3 of SURVEY §8a).
//
// Replaces blosc_internal_{shuffle,unshuffle,bitshuffle,bitunshuffle} (blosc/shuffle.c:367-443;
// permutation spec blosc/shuffle-generic.h:32-81, bit layout blosc/bitshuffle-generic.c:125-139).
//
// Data layout.  A block of `bsize` bytes = N = bsize/T elements of T bytes:
//   element-major ("natural")  : byte (e, j) at  e*T + j
//   plane-major  ("shuffled")  : byte (e, j) at  j*N + e        (+ bsize - N*T tail bytes, verbatim)
//   bit-plane    ("bitshuffled"): bit b of byte (e, j) is bit (e & 7) of byte (8j+b)*(N/8) + e/8
// HBM traffic is 2 * bsize per block (read once, write once): every global access below is a
// 16-byte-per-lane access whose lanes are consecutive in memory (1 KiB per wave instruction); the
// transposition happens in an LDS tile with the narrow accesses on the LDS side only.
//
// Launch: grid = (total blocks of the batch, tiles of the largest block), 256 threads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"
#include "mem_prims.h"

namespace bamd {

constexpr int FT_THREADS = 256;
constexpr int FT_TILE_BYTES = 32768;  // byte-shuffle LDS tile (payload)
constexpr int FT_PAD = 16;            // row padding, keeps rows 16-byte aligned and staggers banks

typedef uint4 u128;   // 16-byte aligned: LDS side uses ds_read_b128 / ds_write_b128

__device__ __forceinline__ u128 ld16_unaligned(const gu8* p) { return g_ld16(p); }
__device__ __forceinline__ void st16_unaligned(gu8* p, const u128& v) { g_st16(p, v); }

// elements per tile for typesize T: at most 2048, tile payload <= FT_TILE_BYTES, multiple of 64
__host__ __device__ inline int shuffle_tile_elems(int T) {
  int e = FT_TILE_BYTES / T;
  if (e > 2048) e = 2048;
  e &= ~63;
  if (e < 64) e = 64;  // T <= 255 -> 64 * 255 = 16320 bytes
  return e;
}


// ---------------------------------------------------------------------------------------------
// byte unshuffle: plane-major (src) -> element-major (dst)
// ---------------------------------------------------------------------------------------------
template <int TT>
__device__ __forceinline__ void unshuffle_emit(const uint8_t* lds, int row, int T, int q, u128& v);

// generic: 16 single-byte LDS reads
template <>
__device__ __forceinline__ void unshuffle_emit<0>(const uint8_t* lds, int row, int T, int q, u128& v) {
  uint32_t k = 16u * (uint32_t)q;
  uint32_t e = k / (uint32_t)T, j = k - e * (uint32_t)T;
  uint32_t w[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t acc = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      acc |= (uint32_t)lds[j * row + e] << (8 * b);
      if (++j == (uint32_t)T) { j = 0; e++; }
    }
    w[d] = acc;
  }
  v = make_uint4(w[0], w[1], w[2], w[3]);
}
// T = 2: eight elements per 16-byte chunk; one 8-byte read per plane
template <>
__device__ __forceinline__ void unshuffle_emit<2>(const uint8_t* lds, int row, int, int q, u128& v) {
  uint64_t a = *(const uint64_t*)(lds + 8 * q);
  uint64_t b = *(const uint64_t*)(lds + row + 8 * q);
  uint32_t w[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint32_t x = (uint32_t)(a >> (16 * d)) & 0xffffu, y = (uint32_t)(b >> (16 * d)) & 0xffffu;
    w[d] = (x & 0xff) | ((y & 0xff) << 8) | ((x >> 8) << 16) | ((y >> 8) << 24);
  }
  v = make_uint4(w[0], w[1], w[2], w[3]);
}
// T = 4: four elements per chunk; one 4-byte read per plane
template <>
__device__ __forceinline__ void unshuffle_emit<4>(const uint8_t* lds, int row, int, int q, u128& v) {
  uint32_t r0 = *(const uint32_t*)(lds + 4 * q), r1 = *(const uint32_t*)(lds + row + 4 * q);
  uint32_t r2 = *(const uint32_t*)(lds + 2 * row + 4 * q), r3 = *(const uint32_t*)(lds + 3 * row + 4 * q);
  uint32_t w[4];
#pragma unroll
  for (int d = 0; d < 4; d++)
    w[d] = ((r0 >> (8 * d)) & 0xff) | (((r1 >> (8 * d)) & 0xff) << 8) | (((r2 >> (8 * d)) & 0xff) << 16) |
           (((r3 >> (8 * d)) & 0xff) << 24);
  v = make_uint4(w[0], w[1], w[2], w[3]);
}
// T = 8: two elements per chunk; one 2-byte read per plane
template <>
__device__ __forceinline__ void unshuffle_emit<8>(const uint8_t* lds, int row, int, int q, u128& v) {
  uint32_t r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = *(const uint16_t*)(lds + j * row + 2 * q);
  v.x = (r[0] & 0xff) | ((r[1] & 0xff) << 8) | ((r[2] & 0xff) << 16) | ((r[3] & 0xff) << 24);
  v.y = (r[4] & 0xff) | ((r[5] & 0xff) << 8) | ((r[6] & 0xff) << 16) | ((r[7] & 0xff) << 24);
  v.z = (r[0] >> 8) | ((r[1] >> 8) << 8) | ((r[2] >> 8) << 16) | ((r[3] >> 8) << 24);
  v.w = (r[4] >> 8) | ((r[5] >> 8) << 8) | ((r[6] >> 8) << 16) | ((r[7] >> 8) << 24);
}
// T = 16: one element per chunk
template <>
__device__ __forceinline__ void unshuffle_emit<16>(const uint8_t* lds, int row, int, int q, u128& v) {
  uint32_t w[4];
#pragma unroll
  for (int d = 0; d < 4; d++)
    w[d] = (uint32_t)lds[(4 * d) * row + q] | ((uint32_t)lds[(4 * d + 1) * row + q] << 8) |
           ((uint32_t)lds[(4 * d + 2) * row + q] << 16) | ((uint32_t)lds[(4 * d + 3) * row + q] << 24);
  v = make_uint4(w[0], w[1], w[2], w[3]);
}

template <int TT>
__device__ void unshuffle_tile(uint8_t* lds, const gu8* src, gu8* dst, int T, int N, int e0, int ne) {
  const int tid = threadIdx.x;
  const int E = shuffle_tile_elems(T);
  const int row = E + FT_PAD;
  // phase 1: planes -> LDS rows
  const int cpr = (ne + 15) >> 4;  // 16-byte chunks per row
  for (int idx = tid; idx < cpr * T; idx += FT_THREADS) {
    int j = idx / cpr, c = idx - j * cpr;
    const gu8* g = src + (size_t)j * N + e0 + 16 * c;
    uint8_t* l = lds + j * row + 16 * c;
    if (16 * c + 16 <= ne) {
      *(u128*)l = ld16_unaligned(g);
    } else {
      for (int b = 0; b < ne - 16 * c; b++) l[b] = g[b];
    }
  }
  __syncthreads();
  // phase 2: LDS -> element-major 16-byte chunks
  const int nb = ne * T, nq = (nb + 15) >> 4;
  gu8* out = dst + (size_t)e0 * T;
  for (int q = tid; q < nq; q += FT_THREADS) {
    if (16 * q + 16 <= nb) {
      u128 v;
      unshuffle_emit<TT>(lds, row, T, q, v);
      st16_unaligned(out + 16 * q, v);
    } else {
      for (int k = 16 * q; k < nb; k++) { int e = k / T, j = k - e * T; out[k] = lds[j * row + e]; }
    }
  }
}

__global__ __launch_bounds__(FT_THREADS) void k_unshuffle(const ChunkDesc* __restrict__ chunks,
                                                         const BlockDesc* __restrict__ blocks) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[FT_TILE_BYTES + 255 * FT_PAD];
  const BlockDesc b = blocks[blockIdx.x];
  const ChunkDesc& c = chunks[b.chunk];
  if (!(c.mode & CH_SHUFFLE) || (c.mode & (CH_MEMCPYED | CH_SKIP | CH_FUSED_UNSHUF))) return;
  const int bsize = b.bsize;
  const int T = c.typesize, N = bsize / T;
  const int E = shuffle_tile_elems(T);
  const int e0 = blockIdx.y * E;
  const gu8* src = as_global(c.filt) + (size_t)b.blk * c.blocksize;
  gu8* dst = as_global(c.dst) + (size_t)b.blk * c.blocksize;
  if (blockIdx.y == 0) {  // trailing bytes that do not form a whole element
    for (int k = N * T + threadIdx.x; k < bsize; k += FT_THREADS) dst[k] = src[k];
  }
  if (e0 >= N) return;
  const int ne = min(E, N - e0);
  switch (T) {
    case 2: unshuffle_tile<2>(lds, src, dst, T, N, e0, ne); break;
    case 4: unshuffle_tile<4>(lds, src, dst, T, N, e0, ne); break;
    case 8: unshuffle_tile<8>(lds, src, dst, T, N, e0, ne); break;
    case 16: unshuffle_tile<16>(lds, src, dst, T, N, e0, ne); break;
    default: unshuffle_tile<0>(lds, src, dst, T, N, e0, ne); break;
  }
}

// ---------------------------------------------------------------------------------------------
// byte shuffle: element-major (src) -> plane-major (dst)
// ---------------------------------------------------------------------------------------------
template <int TT>
__device__ __forceinline__ void shuffle_scatter(uint8_t* lds, int row, int T, int q, const u128& v);

template <>
__device__ __forceinline__ void shuffle_scatter<0>(uint8_t* lds, int row, int T, int q, const u128& v) {
  uint32_t k = 16u * (uint32_t)q;
  uint32_t e = k / (uint32_t)T, j = k - e * (uint32_t)T;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int d = 0; d < 4; d++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      lds[j * row + e] = (uint8_t)(w[d] >> (8 * b));
      if (++j == (uint32_t)T) { j = 0; e++; }
    }
  }
}
template <>
__device__ __forceinline__ void shuffle_scatter<2>(uint8_t* lds, int row, int, int q, const u128& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint64_t a = 0, b = 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    uint64_t x = (w[d] & 0xff) | (((w[d] >> 16) & 0xff) << 8);
    uint64_t y = ((w[d] >> 8) & 0xff) | ((w[d] >> 24) << 8);
    a |= x << (16 * d); b |= y << (16 * d);
  }
  *(uint64_t*)(lds + 8 * q) = a;
  *(uint64_t*)(lds + row + 8 * q) = b;
}
template <>
__device__ __forceinline__ void shuffle_scatter<4>(uint8_t* lds, int row, int, int q, const u128& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t r = ((w[0] >> (8 * j)) & 0xff) | (((w[1] >> (8 * j)) & 0xff) << 8) |
                 (((w[2] >> (8 * j)) & 0xff) << 16) | (((w[3] >> (8 * j)) & 0xff) << 24);
    *(uint32_t*)(lds + j * row + 4 * q) = r;
  }
}
template <>
__device__ __forceinline__ void shuffle_scatter<8>(uint8_t* lds, int row, int, int q, const u128& v) {
  // element 0 = (v.x, v.y), element 1 = (v.z, v.w)
#pragma unroll
  for (int j = 0; j < 4; j++) {
    *(uint16_t*)(lds + j * row + 2 * q) = (uint16_t)(((v.x >> (8 * j)) & 0xff) | (((v.z >> (8 * j)) & 0xff) << 8));
    *(uint16_t*)(lds + (j + 4) * row + 2 * q) = (uint16_t)(((v.y >> (8 * j)) & 0xff) | (((v.w >> (8 * j)) & 0xff) << 8));
  }
}
template <>
__device__ __forceinline__ void shuffle_scatter<16>(uint8_t* lds, int row, int, int q, const u128& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int d = 0; d < 4; d++)
#pragma unroll
    for (int b = 0; b < 4; b++) lds[(4 * d + b) * row + q] = (uint8_t)(w[d] >> (8 * b));
}

template <int TT>
__device__ void shuffle_tile(uint8_t* lds, const gu8* src, gu8* dst, int T, int N, int e0, int ne) {
  const int tid = threadIdx.x;
  const int E = shuffle_tile_elems(T);
  const int row = E + FT_PAD;
  const int nb = ne * T, nq = (nb + 15) >> 4;
  const gu8* in = src + (size_t)e0 * T;
  for (int q = tid; q < nq; q += FT_THREADS) {
    if (16 * q + 16 <= nb) {
      shuffle_scatter<TT>(lds, row, T, q, ld16_unaligned(in + 16 * q));
    } else {
      for (int k = 16 * q; k < nb; k++) { int e = k / T, j = k - e * T; lds[j * row + e] = in[k]; }
    }
  }
  __syncthreads();
  const int cpr = (ne + 15) >> 4;
  for (int idx = tid; idx < cpr * T; idx += FT_THREADS) {
    int j = idx / cpr, c = idx - j * cpr;
    gu8* g = dst + (size_t)j * N + e0 + 16 * c;
    const uint8_t* l = lds + j * row + 16 * c;
    if (16 * c + 16 <= ne) st16_unaligned(g, *(const u128*)l);
    else for (int b = 0; b < ne - 16 * c; b++) g[b] = l[b];
  }
}

__global__ __launch_bounds__(FT_THREADS) void k_shuffle(const ChunkDesc* __restrict__ chunks,
                                                       const BlockDesc* __restrict__ blocks) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[FT_TILE_BYTES + 255 * FT_PAD];
  const BlockDesc b = blocks[blockIdx.x];
  const ChunkDesc& c = chunks[b.chunk];
  if (!(c.mode & CH_SHUFFLE) || (c.mode & (CH_MEMCPYED | CH_SKIP | CH_FUSED_SHUF))) return;
  const int bsize = b.bsize;
  const int T = c.typesize, N = bsize / T;
  const int E = shuffle_tile_elems(T);
  const int e0 = blockIdx.y * E;
  const gu8* src = as_global(c.src) + (size_t)b.blk * c.blocksize;
  gu8* dst = as_global(c.filt) + (size_t)b.blk * c.blocksize;
  if (blockIdx.y == 0) {
    for (int k = N * T + threadIdx.x; k < bsize; k += FT_THREADS) dst[k] = src[k];
  }
  if (e0 >= N) return;
  const int ne = min(E, N - e0);
  switch (T) {
    case 2: shuffle_tile<2>(lds, src, dst, T, N, e0, ne); break;
    case 4: shuffle_tile<4>(lds, src, dst, T, N, e0, ne); break;
    case 8: shuffle_tile<8>(lds, src, dst, T, N, e0, ne); break;
    case 16: shuffle_tile<16>(lds, src, dst, T, N, e0, ne); break;
    default: shuffle_tile<0>(lds, src, dst, T, N, e0, ne); break;
  }
}

// ---------------------------------------------------------------------------------------------
// bit shuffle / unshuffle
// ---------------------------------------------------------------------------------------------
constexpr int BT_TILE_BYTES = 24576;  // element-major tile payload; the bit-row tile is the same size

__host__ __device__ inline int bitshuffle_tile_elems(int T) {
  int e = BT_TILE_BYTES / T;
  if (e > 8192) e = 8192;
  e &= ~127;              // rows of E/8 bytes are multiples of 16 bytes
  if (e < 128) e = 128;   // T <= 255 -> 32640 bytes: see BT_LDS below
  return e;
}
constexpr int BT_A_BYTES = 32768;                 // element-major tile
constexpr int BT_B_BYTES = 32768 + 128 * FT_PAD;  // bit-row tile (rows padded when there are <= 128 of them)
__device__ __forceinline__ int bt_rowstride(int T, int E) { return (E >> 3) + ((8 * T <= 128) ? FT_PAD : 0); }

// 8x8 bit-matrix transpose of a 64-bit word (byte k, bit b  <->  byte b, bit k), three
// delta-swap rounds.
__device__ __forceinline__ uint64_t bit_transpose8(uint64_t x) {
  const uint64_t masks[3] = {0x00AA00AA00AA00AAull, 0x0000CCCC0000CCCCull, 0x00000000F0F0F0F0ull};
#pragma unroll
  for (int s = 0; s < 3; s++) {
    const int sh = 7 << s;
    uint64_t t = (x ^ (x >> sh)) & masks[s];
    x ^= t ^ (t << sh);
  }
  return x;
}

// dir = 0: bitshuffle (src element-major -> dst bit rows); dir = 1: inverse.
template <int DIR>
__device__ void bit_tile(uint8_t* A, uint8_t* B, const gu8* src, gu8* dst, int T, int N, int e0, int ne) {
  const int tid = threadIdx.x;
  const int E = bitshuffle_tile_elems(T);
  const int rs = bt_rowstride(T, E);
  const int rowlen = N >> 3;         // bytes per bit row in global memory
  const int m0 = e0 >> 3, nm = ne >> 3;  // this tile's byte range inside every row (ne % 8 == 0)
  const int nb = ne * T, nq = (nb + 15) >> 4;
  const int cpr = (nm + 15) >> 4, nrows = 8 * T;
  if (DIR == 0) {
    const gu8* in = src + (size_t)e0 * T;
    for (int q = tid; q < nq; q += FT_THREADS) {
      if (16 * q + 16 <= nb) *(u128*)(A + 16 * q) = ld16_unaligned(in + 16 * q);
      else for (int k = 16 * q; k < nb; k++) A[k] = in[k];
    }
  } else {
    for (int idx = tid; idx < cpr * nrows; idx += FT_THREADS) {
      int r = idx / cpr, c = idx - r * cpr;
      const gu8* g = src + (size_t)r * rowlen + m0 + 16 * c;
      uint8_t* l = B + r * rs + 16 * c;
      if (16 * c + 16 <= nm) *(u128*)l = ld16_unaligned(g);
      else for (int b = 0; b < nm - 16 * c; b++) l[b] = g[b];
    }
  }
  __syncthreads();
  // one work item = byte plane j of eight consecutive elements (8m .. 8m+7)
  for (int idx = tid; idx < nm * T; idx += FT_THREADS) {
    int m = idx / T, j = idx - m * T;
    uint64_t x = 0;
    if (DIR == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) x |= (uint64_t)A[(8 * m + k) * T + j] << (8 * k);
      x = bit_transpose8(x);
#pragma unroll
      for (int b = 0; b < 8; b++) B[(8 * j + b) * rs + m] = (uint8_t)(x >> (8 * b));
    } else {
#pragma unroll
      for (int b = 0; b < 8; b++) x |= (uint64_t)B[(8 * j + b) * rs + m] << (8 * b);
      x = bit_transpose8(x);
#pragma unroll
      for (int k = 0; k < 8; k++) A[(8 * m + k) * T + j] = (uint8_t)(x >> (8 * k));
    }
  }
  __syncthreads();
  if (DIR == 0) {
    for (int idx = tid; idx < cpr * nrows; idx += FT_THREADS) {
      int r = idx / cpr, c = idx - r * cpr;
      gu8* g = dst + (size_t)r * rowlen + m0 + 16 * c;
      const uint8_t* l = B + r * rs + 16 * c;
      if (16 * c + 16 <= nm) st16_unaligned(g, *(const u128*)l);
      else for (int b = 0; b < nm - 16 * c; b++) g[b] = l[b];
    }
  } else {
    gu8* out = dst + (size_t)e0 * T;
    for (int q = tid; q < nq; q += FT_THREADS) {
      if (16 * q + 16 <= nb) st16_unaligned(out + 16 * q, *(const u128*)(A + 16 * q));
      else for (int k = 16 * q; k < nb; k++) out[k] = A[k];
    }
  }
}

__host__ __device__ inline bool bit_fast_T(int T) { return T == 1 || T == 2 || T == 4 || T == 8; }

template <int DIR>
__device__ void bitfilter_block(const ChunkDesc* chunks, const BlockDesc* blocks, int fast_done) {
  __shared__ __attribute__((aligned(16))) uint8_t A[BT_A_BYTES];
  __shared__ __attribute__((aligned(16))) uint8_t B[BT_B_BYTES];
  const BlockDesc b = blocks[blockIdx.x];
  const ChunkDesc& c = chunks[b.chunk];
  if (!(c.mode & CH_BITSHUFFLE) || (c.mode & (CH_MEMCPYED | CH_SKIP))) return;
  if (DIR == 1 && (c.mode & CH_FUSED_BITUNSH)) return;        // the decode kernel has unshuffled this chunk's blocks itself (k_decode.hip)
  if (DIR == 0 && (c.mode & CH_FUSED_SHUF)) return;           // the encode kernel bitshuffles this chunk's blocks itself (enc_shuffle.h)
  const int bsize = b.bsize;
  const int T = c.typesize;
  const size_t boff = (size_t)b.blk * c.blocksize;
  const gu8* src = as_global(DIR == 0 ? c.src : (const uint8_t*)c.filt) + boff;
  gu8* dst = as_global(DIR == 0 ? c.filt : c.dst) + boff;
  const int E = bitshuffle_tile_elems(T);
  if (bsize < T) {  // filter not applied at all (blosc/blosc.c:608-609, :740-741): plain copy
    if (blockIdx.y == 0) for (int k = threadIdx.x; k < bsize; k += FT_THREADS) dst[k] = src[k];
    return;
  }
  const int N = bsize / T;
  if (N & 7) {  // element count not a multiple of 8: whole block copied verbatim (shuffle.c:412-414)
    const int per = E * T;
    const int ntiles = (N + E - 1) / E;           // what the host sized grid.y for
    if ((int)blockIdx.y >= ntiles) return;
    const int lo = blockIdx.y * per;
    const int hi = ((int)blockIdx.y == ntiles - 1) ? bsize : lo + per;   // last tile also takes the tail bytes
    for (int k = lo + threadIdx.x * 16; k < hi; k += FT_THREADS * 16) {
      if (k + 16 <= hi) st16_unaligned(dst + k, ld16_unaligned(src + k));
      else for (int t = k; t < hi; t++) dst[t] = src[t];
    }
    return;
  }
  if (blockIdx.y == 0) {
    for (int k = N * T + threadIdx.x; k < bsize; k += FT_THREADS) dst[k] = src[k];
  }
  const int e0 = blockIdx.y * E;
  if (e0 >= N) return;
  if (fast_done && bit_fast_T(T) && e0 + E <= N) return;     // full tiles of these type sizes: k_bit*_fast has done them
  bit_tile<DIR>(A, B, src, dst, T, N, e0, min(E, N - e0));
}

__global__ __launch_bounds__(FT_THREADS) void k_bitshuffle(const ChunkDesc* __restrict__ chunks,
                                                          const BlockDesc* __restrict__ blocks, int fast_done) {
  bitfilter_block<0>(chunks, blocks, fast_done);
}
__global__ __launch_bounds__(FT_THREADS) void k_bitunshuffle(const ChunkDesc* __restrict__ chunks,
                                                            const BlockDesc* __restrict__ blocks, int fast_done) {
  bitfilter_block<1>(chunks, blocks, fast_done);
}

// ---------------------------------------------------------------------------------------------
// Fast path for typesize 1 / 2 / 4 / 8 and FULL tiles (the BASELINE geometry: config #3 is typesize 4).
// One thread owns 32 consecutive elements: it reads them as 16-byte pieces out of an LDS staging tile (written with
// coalesced 16-byte global loads; chunks of 32 T bytes are 16 bytes apart so that the 128-bit reads of 16 neighbouring
// lanes fall into different banks), transposes 8 x 8 bit matrices in registers (three delta swaps on a 64-bit word per
// byte plane and group of eight elements), and holds 4 consecutive bytes of each of the 8 T bit rows, which go to
// global memory as one dword per row: a wave stores 256 contiguous bytes per row and instruction.  The bit-row side
// never touches LDS (the generic path above moves every byte through LDS twice, one byte per access: 1.9 TB/s).
// The inverse mirrors it: dword loads of the rows, transposes, LDS staging, coalesced 16-byte stores.
// ---------------------------------------------------------------------------------------------
constexpr int BTF_LDS = 32768 + 4096 + 64;   // E T + E / 2 bytes at most (E = 8192 for T <= 4)

template <int T>
__device__ __forceinline__ uint32_t btf_byte(const uint32_t* w, int e, int j) {   // byte j of element e of a chunk held as dwords
  const int idx = e * T + j;
  return (w[idx >> 2] >> (8 * (idx & 3))) & 0xffu;
}

template <int DIR, int T>
__device__ __forceinline__ void bit_tile_fast(uint8_t* S, const gu8* src, gu8* dst, int N, int e0) {
  constexpr int E = (BT_TILE_BYTES / T > 8192 ? 8192 : BT_TILE_BYTES / T) & ~127;
  constexpr int CB = 32 * T, CS = CB + 16, NCH = E / 32, NDW = CB / 4;
  const int tid = threadIdx.x;
  const int rowlen = N >> 3, m0 = e0 >> 3;
  if (DIR == 0) {
    const gu8* in = src + (size_t)e0 * T;
    for (int q = tid; q < E * T / 16; q += FT_THREADS) {
      const int off = 16 * q, c = off / CB;
      *(u128*)(S + c * CS + (off - c * CB)) = ld16_unaligned(in + off);
    }
    __syncthreads();
    for (int t = tid; t < NCH; t += FT_THREADS) {
      uint32_t w[NDW];
#pragma unroll
      for (int k = 0; k < NDW / 4; k++) { const u128 v = *(const u128*)(S + t * CS + 16 * k); w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
#pragma unroll
      for (int j = 0; j < T; j++) {
        uint64_t x[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
          uint64_t v = 0;
#pragma unroll
          for (int k = 0; k < 8; k++) v |= (uint64_t)btf_byte<T>(w, 8 * g + k, j) << (8 * k);
          x[g] = bit_transpose8(v);
        }
#pragma unroll
        for (int b = 0; b < 8; b++) {
          const uint32_t rw = (uint32_t)((x[0] >> (8 * b)) & 0xff) | ((uint32_t)((x[1] >> (8 * b)) & 0xff) << 8) |
                              ((uint32_t)((x[2] >> (8 * b)) & 0xff) << 16) | ((uint32_t)((x[3] >> (8 * b)) & 0xff) << 24);
          g_st4(dst + (size_t)(8 * j + b) * rowlen + m0 + 4 * t, rw);
        }
      }
    }
  } else {
    for (int t = tid; t < NCH; t += FT_THREADS) {
      uint32_t w[NDW];
#pragma unroll
      for (int k = 0; k < NDW; k++) w[k] = 0;
#pragma unroll
      for (int j = 0; j < T; j++) {
        uint32_t rw[8];
#pragma unroll
        for (int b = 0; b < 8; b++) rw[b] = g_ld4(src + (size_t)(8 * j + b) * rowlen + m0 + 4 * t);
#pragma unroll
        for (int g = 0; g < 4; g++) {
          uint64_t v = 0;
#pragma unroll
          for (int b = 0; b < 8; b++) v |= (uint64_t)((rw[b] >> (8 * g)) & 0xffu) << (8 * b);
          v = bit_transpose8(v);
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int idx = (8 * g + k) * T + j;
            w[idx >> 2] |= (uint32_t)((v >> (8 * k)) & 0xff) << (8 * (idx & 3));
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NDW / 4; k++) { u128 v; v.x = w[4 * k]; v.y = w[4 * k + 1]; v.z = w[4 * k + 2]; v.w = w[4 * k + 3]; *(u128*)(S + t * CS + 16 * k) = v; }
    }
    __syncthreads();
    gu8* out = dst + (size_t)e0 * T;
    for (int q = tid; q < E * T / 16; q += FT_THREADS) {
      const int off = 16 * q, c = off / CB;
      st16_unaligned(out + off, *(const u128*)(S + c * CS + (off - c * CB)));
    }
  }
}

template <int DIR>
__global__ __launch_bounds__(FT_THREADS, 2) void k_bitfilter_fast(const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks) {
  __shared__ __attribute__((aligned(16))) uint8_t S[BTF_LDS];
  const BlockDesc b = blocks[blockIdx.x];
  const ChunkDesc& c = chunks[b.chunk];
  if (!(c.mode & CH_BITSHUFFLE) || (c.mode & (CH_MEMCPYED | CH_SKIP))) return;
  if (DIR == 1 && (c.mode & CH_FUSED_BITUNSH)) return;
  if (DIR == 0 && (c.mode & CH_FUSED_SHUF)) return;
  const int T = c.typesize, bsize = b.bsize;
  if (!bit_fast_T(T) || bsize < T) return;
  const int N = bsize / T;
  if (N & 7) return;
  const int E = bitshuffle_tile_elems(T), e0 = blockIdx.y * E;
  if (e0 + E > N) return;                                     // partial tiles stay with the generic kernel
  const size_t boff = (size_t)b.blk * c.blocksize;
  const gu8* src = as_global(DIR == 0 ? c.src : (const uint8_t*)c.filt) + boff;
  gu8* dst = as_global(DIR == 0 ? c.filt : c.dst) + boff;
  switch (T) {
    case 1: bit_tile_fast<DIR, 1>(S, src, dst, N, e0); break;
    case 2: bit_tile_fast<DIR, 2>(S, src, dst, N, e0); break;
    case 4: bit_tile_fast<DIR, 4>(S, src, dst, N, e0); break;
    default: bit_tile_fast<DIR, 8>(S, src, dst, N, e0); break;
  }
}

}  // namespace bamd
