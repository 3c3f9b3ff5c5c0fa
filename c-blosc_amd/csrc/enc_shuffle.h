// enc_shuffle.h — the byte shuffle as work of the encode kernel's own waves (included by k_encode.hip inside namespace bamd):
// shuffle_block_wave_T, the periodic-plane detection, shuffle_block_task, emit_periodic_stream.  DESIGN.md 3.3.
// ---------------------------------------------------------------------------------------------
// Fused byte shuffle of one block by ONE wavefront (typesize 2, 4, 8 or 16): element-major source -> plane-major
// scratch (blosc/shuffle-generic.h:27-58).  The mirror image of unshuffle_block_wave in k_decode.hip: per
// step lane l loads the T*4 contiguous bytes of elements e+4l..e+4l+3 (coalesced 16-byte loads), transposes
// bytes in registers and stores 4 bytes into every plane (each wave store writes 256 contiguous bytes).
// ---------------------------------------------------------------------------------------------
// (Round 3 tried the typesize-8 loads dealt so that a quad reads 64 contiguous bytes per instruction: encode 8.2 -> 10.3 ms,
//  profiles/r03/r03zb_ab_quad_dealt_typesize8_rejected.txt; and non-temporal source loads: no gain.  Both removed in round 4.)
template <int T>
struct ElemRows { uint4 a, b; };

template <int T>
__device__ __forceinline__ ElemRows<T> shuffle_load(const gu8* src, uint32_t e, int lane) {
  ElemRows<T> x;
  const gu8* in = src + (size_t)(e + 4u * (uint32_t)lane) * T;
  x.a = g_ld16(in);
  if (T == 8) x.b = g_ld16(in + 16); else x.b = make_uint4(0, 0, 0, 0);
  return x;
}
template <int T>
__device__ __forceinline__ void shuffle_store(gu8* dst, uint32_t N, uint32_t e, int lane, const ElemRows<T>& x) {
  gu8* o = dst + e + 4u * (uint32_t)lane;
  uint32_t r0, r1, r2, r3;
  if (T == 8) {
    // a = (lo0, hi0, lo1, hi1), b = (lo2, hi2, lo3, hi3): low / high dword of elements 0..3
    transpose4x4(x.a.x, x.a.z, x.b.x, x.b.z, r0, r1, r2, r3);
    g_st4(o, r0); g_st4(o + (size_t)N, r1); g_st4(o + 2 * (size_t)N, r2); g_st4(o + 3 * (size_t)N, r3);
    transpose4x4(x.a.y, x.a.w, x.b.y, x.b.w, r0, r1, r2, r3);
    g_st4(o + 4 * (size_t)N, r0); g_st4(o + 5 * (size_t)N, r1); g_st4(o + 6 * (size_t)N, r2); g_st4(o + 7 * (size_t)N, r3);
  } else {
    transpose4x4(x.a.x, x.a.y, x.a.z, x.a.w, r0, r1, r2, r3);
    g_st4(o, r0); g_st4(o + (size_t)N, r1); g_st4(o + 2 * (size_t)N, r2); g_st4(o + 3 * (size_t)N, r3);
  }
}
// Typesize 2 and 16 (round 3; blosc/shuffle-generic.h:27-58 is the same loop for every typesize): the same step - lane l owns
// elements e + 4l .. e + 4l + 3 and ends up with 4 bytes of every plane - with 8 bytes (T = 2) or 64 bytes (T = 16) per lane.
struct ElemRows2 { uint32_t lo, hi; };                   // (e0b0 e0b1 e1b0 e1b1), (e2b0 e2b1 e3b0 e3b1)
struct ElemRows16 { uint4 v[4]; };                       // element k of the lane: bytes 0-3, 4-7, 8-11, 12-15
__device__ __forceinline__ ElemRows2 shuffle_load2(const gu8* src, uint32_t e, int lane) {
  const uint64_t w = g_ld8(src + (size_t)(e + 4u * (uint32_t)lane) * 2u);
  ElemRows2 x; x.lo = (uint32_t)w; x.hi = (uint32_t)(w >> 32);
  return x;
}
__device__ __forceinline__ void shuffle_rows2(const ElemRows2& x, uint32_t (&r)[16]) {
  r[0] = __builtin_amdgcn_perm(x.hi, x.lo, 0x06040200u);       // byte 0 of elements 0..3
  r[1] = __builtin_amdgcn_perm(x.hi, x.lo, 0x07050301u);       // byte 1
}
// (typesize 16: load k of lane (Q, i) - Q = l >> 2, i = l & 3 - is element e + 16Q + i + 4k, so that the four lanes of a quad read 64
//  contiguous bytes per instruction; quad_byte_transpose (k_decode.hip) then turns every plane dword into the bytes of the lane's
//  own consecutive elements e + 4l .. e + 4l + 3 - the mirror image of unshuffle_store16)
__device__ __forceinline__ ElemRows16 shuffle_load16(const gu8* src, uint32_t e, int lane) {
  const gu8* in = src + (size_t)(e + 16u * ((uint32_t)lane >> 2) + ((uint32_t)lane & 3u)) * 16u;
  ElemRows16 x;
#pragma unroll
  for (int k = 0; k < 4; k++) x.v[k] = g_ld16(in + 64 * k);
  return x;
}
__device__ __forceinline__ void shuffle_rows16(const ElemRows16& x, uint32_t (&r)[16], int lane) {
  transpose4x4(x.v[0].x, x.v[1].x, x.v[2].x, x.v[3].x, r[0], r[1], r[2], r[3]);
  transpose4x4(x.v[0].y, x.v[1].y, x.v[2].y, x.v[3].y, r[4], r[5], r[6], r[7]);
  transpose4x4(x.v[0].z, x.v[1].z, x.v[2].z, x.v[3].z, r[8], r[9], r[10], r[11]);
  transpose4x4(x.v[0].w, x.v[1].w, x.v[2].w, x.v[3].w, r[12], r[13], r[14], r[15]);
#pragma unroll
  for (int j = 0; j < 16; j++) r[j] = quad_byte_transpose(r[j], lane);
}
template <int T>
__device__ __forceinline__ void shuffle_store_rows(gu8* dst, uint32_t N, uint32_t e, int lane, const uint32_t (&r)[16]) {
  gu8* o = dst + e + 4u * (uint32_t)lane;
#pragma unroll
  for (int k = 0; k < T; k++) g_st4(o + (size_t)k * N, r[k]);
}
// whole-row part of a block (N / 256 steps); the tails are the caller's
__device__ __forceinline__ uint32_t shuffle_block_rows2(const gu8* src, gu8* dst, uint32_t N, int lane) {
  uint32_t e = 0, r[16];
  for (; e + 1024u <= N; e += 1024u) {   // 4 steps per iteration: all loads are issued before the first store
    const ElemRows2 a = shuffle_load2(src, e, lane), b = shuffle_load2(src, e + 256u, lane);
    const ElemRows2 c = shuffle_load2(src, e + 512u, lane), d = shuffle_load2(src, e + 768u, lane);
    shuffle_rows2(a, r); shuffle_store_rows<2>(dst, N, e, lane, r);
    shuffle_rows2(b, r); shuffle_store_rows<2>(dst, N, e + 256u, lane, r);
    shuffle_rows2(c, r); shuffle_store_rows<2>(dst, N, e + 512u, lane, r);
    shuffle_rows2(d, r); shuffle_store_rows<2>(dst, N, e + 768u, lane, r);
  }
  for (; e + 256u <= N; e += 256u) { shuffle_rows2(shuffle_load2(src, e, lane), r); shuffle_store_rows<2>(dst, N, e, lane, r); }
  return e;
}
__device__ __forceinline__ uint32_t shuffle_block_rows16(const gu8* src, gu8* dst, uint32_t N, int lane) {
  uint32_t e = 0, r[16];
  for (; e + 512u <= N; e += 512u) {     // 2 steps per iteration (8 KiB of loads in flight per wave, as for typesize 8)
    const ElemRows16 a = shuffle_load16(src, e, lane), b = shuffle_load16(src, e + 256u, lane);
    shuffle_rows16(a, r, lane); shuffle_store_rows<16>(dst, N, e, lane, r);
    shuffle_rows16(b, r, lane); shuffle_store_rows<16>(dst, N, e + 256u, lane, r);
  }
  for (; e + 256u <= N; e += 256u) { shuffle_rows16(shuffle_load16(src, e, lane), r, lane); shuffle_store_rows<16>(dst, N, e, lane, r); }
  return e;
}
template <int T>
__device__ void shuffle_block_wave_T(const gu8* src, gu8* dst, uint32_t bsize, int lane) {
  const uint32_t N = bsize / T;
  uint32_t e = 0;
  if constexpr (T == 2) e = shuffle_block_rows2(src, dst, N, lane);
  else if constexpr (T == 16) e = shuffle_block_rows16(src, dst, N, lane);
  else {
  for (; e + 1024u <= N; e += 1024u) {   // 4 steps per iteration: all loads are issued before the first store
    const ElemRows<T> a = shuffle_load<T>(src, e, lane), b = shuffle_load<T>(src, e + 256u, lane);
    const ElemRows<T> c = shuffle_load<T>(src, e + 512u, lane), d = shuffle_load<T>(src, e + 768u, lane);
    shuffle_store<T>(dst, N, e, lane, a); shuffle_store<T>(dst, N, e + 256u, lane, b);
    shuffle_store<T>(dst, N, e + 512u, lane, c); shuffle_store<T>(dst, N, e + 768u, lane, d);
  }
  for (; e + 256u <= N; e += 256u) shuffle_store<T>(dst, N, e, lane, shuffle_load<T>(src, e, lane));
  }
  // tail: fewer than 256 elements, then the bytes that do not form a whole element (copied as they are)
  for (uint32_t k = e * T + (uint32_t)lane; k < N * T; k += 64u) { const uint32_t el = k / T, j = k - el * T; dst[(size_t)j * N + el] = src[k]; }
  for (uint32_t k = N * T + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}

// ---------------------------------------------------------------------------------------------
// Periodic planes.  A byte plane whose every 256-byte row equals its first row - a constant byte, a counter's low
// byte, the zero top bytes of small integers: the planes shuffling exists to produce - needs no match finder and no
// trip through the scratch: its stream is "first period as literals + one match over the rest".  The shuffle wave
// notices them for free (it holds each row in a register): as long as a plane's rows keep repeating, nothing is
// stored; the first row that differs back-fills the rows skipped so far (copies of row 0) and the plane is an
// ordinary one from there on.  A plane that stays periodic to the end gets its first row stored (the literals' source)
// and its period p (smallest power of two, 1..256) left in its stream's `result` as -p; encode_one_stream turns that
// into the stream (emit_periodic_stream).  The mirror image of the decoder's periodic spans (k_decode.hip).
// Only whole-row blocks (N % 256 == 0, N >= 1024) whose planes are streams of their own (split blocks).
// ---------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void shuffle_rows(const ElemRows<T>& x, uint32_t (&r)[8]) {
  if (T == 8) {
    transpose4x4(x.a.x, x.a.z, x.b.x, x.b.z, r[0], r[1], r[2], r[3]);
    transpose4x4(x.a.y, x.a.w, x.b.y, x.b.w, r[4], r[5], r[6], r[7]);
  } else {
    transpose4x4(x.a.x, x.a.y, x.a.z, x.a.w, r[0], r[1], r[2], r[3]);
    r[4] = r[5] = r[6] = r[7] = 0;
  }
}
// smallest power-of-two period (bytes) of a 256-byte row held one dword per lane; 256 when there is none below
__device__ __forceinline__ uint32_t row_period(uint32_t w, int lane) {
  uint32_t p = 256u;
  for (uint32_t sh = 32u; sh >= 1u; sh >>= 1) {
    const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)lane + sh) & 63u) << 2), (int)w);
    if (__ballot(other != w) != 0ull) return p;
    p = 4u * sh;
  }
  const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);       // all lanes hold the same dword here
  if ((w0 >> 16) != (w0 & 0xffffu)) return 4u;
  return ((w0 >> 8) & 0xffu) == (w0 & 0xffu) ? 1u : 2u;
}
// returns the mask of planes that stayed periodic; their period goes to period[k]
template <int T>
__device__ uint32_t shuffle_block_wave_detect(const gu8* src, gu8* dst, uint32_t bsize, int lane, uint32_t (&period)[8]) {
  const uint32_t N = bsize / T;
  uint32_t row0[8], r[8];
  uint32_t per = (1u << T) - 1u;                       // wave-uniform: planes whose rows all equalled row 0 so far
  shuffle_rows<T>(shuffle_load<T>(src, 0u, lane), row0);
  auto step = [&](const ElemRows<T>& x, uint32_t e) {
    if (per == 0u) { shuffle_store<T>(dst, N, e, lane, x); return; }
    shuffle_rows<T>(x, r);
    gu8* o = dst + e + 4u * (uint32_t)lane;
#pragma unroll
    for (int k = 0; k < T; k++) {
      if (per & (1u << k)) {
        if (__ballot(r[k] != row0[k]) == 0ull) continue;
        per &= ~(1u << k);
        for (uint32_t t = 0; t < e; t += 256u) g_st4(dst + (size_t)k * N + t + 4u * (uint32_t)lane, row0[k]);
      }
      g_st4(o + (size_t)k * N, r[k]);
    }
  };
  uint32_t e = 256u;
  for (; e + 1024u <= N; e += 1024u) {
    const ElemRows<T> a = shuffle_load<T>(src, e, lane), b = shuffle_load<T>(src, e + 256u, lane);
    const ElemRows<T> c = shuffle_load<T>(src, e + 512u, lane), d = shuffle_load<T>(src, e + 768u, lane);
    step(a, e); step(b, e + 256u); step(c, e + 512u); step(d, e + 768u);
  }
  for (; e + 256u <= N; e += 256u) step(shuffle_load<T>(src, e, lane), e);
#pragma unroll
  for (int k = 0; k < T; k++) {
    period[k] = 0u;
    if (per & (1u << k)) {
      g_st4(dst + (size_t)k * N + 4u * (uint32_t)lane, row0[k]);
      period[k] = row_period(row0[k], lane);
    }
  }
  return per;
}

// The same for typesize 2 and 16 (rows through shuffle_rows2 / shuffle_rows16).  Typesize 16 keeps one step in flight
// (64 bytes per lane: 4 KiB per wave; the row registers of 16 planes and their first rows take the rest of the budget).
template <int T>
__device__ uint32_t shuffle_block_wave_detect_x(const gu8* src, gu8* dst, uint32_t bsize, int lane, uint32_t (&period)[16]) {
  static_assert(T == 2 || T == 16, "typesize 4 / 8 use shuffle_block_wave_detect");
  const uint32_t N = bsize / T;
  uint32_t row0[16], r[16];
  uint32_t per = (1u << T) - 1u;                       // wave-uniform: planes whose rows all equalled row 0 so far
  if constexpr (T == 2) shuffle_rows2(shuffle_load2(src, 0u, lane), row0); else shuffle_rows16(shuffle_load16(src, 0u, lane), row0, lane);
  auto step = [&](uint32_t e) {                        // r: the rows of step e
    if (per == 0u) { shuffle_store_rows<T>(dst, N, e, lane, r); return; }
    gu8* o = dst + e + 4u * (uint32_t)lane;
#pragma unroll
    for (int k = 0; k < T; k++) {
      if (per & (1u << k)) {
        if (__ballot(r[k] != row0[k]) == 0ull) continue;
        per &= ~(1u << k);
        for (uint32_t t = 0; t < e; t += 256u) g_st4(dst + (size_t)k * N + t + 4u * (uint32_t)lane, row0[k]);
      }
      g_st4(o + (size_t)k * N, r[k]);
    }
  };
  uint32_t e = 256u;
  if constexpr (T == 2) {
    for (; e + 1024u <= N; e += 1024u) {
      const ElemRows2 a = shuffle_load2(src, e, lane), b = shuffle_load2(src, e + 256u, lane);
      const ElemRows2 c = shuffle_load2(src, e + 512u, lane), d = shuffle_load2(src, e + 768u, lane);
      shuffle_rows2(a, r); step(e); shuffle_rows2(b, r); step(e + 256u); shuffle_rows2(c, r); step(e + 512u); shuffle_rows2(d, r); step(e + 768u);
    }
    for (; e + 256u <= N; e += 256u) { shuffle_rows2(shuffle_load2(src, e, lane), r); step(e); }
  } else {
    for (; e + 256u <= N; e += 256u) { shuffle_rows16(shuffle_load16(src, e, lane), r, lane); step(e); }
  }
#pragma unroll
  for (int k = 0; k < T; k++) {
    period[k] = 0u;
    if (per & (1u << k)) {
      g_st4(dst + (size_t)k * N + 4u * (uint32_t)lane, row0[k]);
      period[k] = row_period(row0[k], lane);
    }
  }
  return per;
}

// queue task "shuffle block gb": afterwards the block's flag tells the encoders of its streams to go ahead.
// Producer and consumers run on the same XCD (per-XCD queues), so the hand-off goes through that XCD's L2:
// drain the stores, then a relaxed agent-scope flag store - no L2 write-back needed.
template <int T>
__device__ __forceinline__ void shuffle_block_detect_T(const gu8* src, gu8* dst, uint32_t bsize, StreamDesc* planes, int lane) {
  uint32_t period[(T == 4 || T == 8) ? 8 : 16];
  uint32_t per;
  if constexpr (T == 4 || T == 8) per = shuffle_block_wave_detect<T>(src, dst, bsize, lane, period);
  else per = shuffle_block_wave_detect_x<T>(src, dst, bsize, lane, period);
#pragma unroll
  for (int k = 0; k < T; k++)
    if ((per & (1u << k)) && lane == 0) planes[k].result = -(int32_t)period[k];
}
__device__ __attribute__((noinline)) void shuffle_block_task_x(const gu8* src_, gu8* dst_, uint32_t bsize_, uint32_t T_, bool det_, StreamDesc* planes_, int lane) {
  // (arguments of a real call count as divergent: back to scalars first, as in unshuffle_block_wave of k_decode.hip)
  const gu8* src = uni_ptr(src_); gu8* dst = uni_ptr(dst_);
  const uint32_t bsize = uni(bsize_), T = uni(T_); const bool det = uni((uint32_t)det_) != 0u;
  const uint64_t pv = (uint64_t)planes_;
  StreamDesc* planes = (StreamDesc*)(((uint64_t)uni((uint32_t)(pv >> 32)) << 32) | uni((uint32_t)pv));
  if (T == 16u) { if (det) shuffle_block_detect_T<16>(src, dst, bsize, planes, lane); else shuffle_block_wave_T<16>(src, dst, bsize, lane); }
  else { if (det) shuffle_block_detect_T<2>(src, dst, bsize, planes, lane); else shuffle_block_wave_T<2>(src, dst, bsize, lane); }
}
// ---------------------------------------------------------------------------------------------
// Fused bitshuffle of one block by ONE wavefront (typesize 1, 2, 4; 8 since round 5), round 4: a "shuffle block" task of a bitshuffle chunk - the
// mirror image of bitunshuffle_block_wave in k_decode.hip, and the end of the k_bitshuffle pass over the batch (3.9 ms per 8 GiB on
// config #3).  blosc_internal_bitshuffle (blosc/shuffle.c:393-443, bitshuffle-generic.c:125-139): element-major source -> 8 T bit rows
// of N / 8 bytes.  Per pass of 2048 elements a lane owns 32 consecutive ones: 32 T bytes as 16-byte loads (its neighbours' lines are the
// same ones: L1), 8 x 8 bit-matrix transposes in registers, and one dword of every bit row out - a wave store writes 256 contiguous bytes
// of a row.  No LDS.  Corner rules as in the filter kernels: not applied when bsize < T, whole block copied when N is not a multiple of 8.
// ---------------------------------------------------------------------------------------------
// EPL = elements a lane owns per pass: 32 (typesize 1 / 2 / 4: a dword of every bit row out) or 16 (typesize 8, round 5: 128 source bytes = 32
// registers per lane, two bytes of each of its 64 rows out - a wave store still writes 128 contiguous bytes of a row)
template <int T, int EPL>
__device__ __forceinline__ void bitshuffle_pass(const gu8* src, gu8* dst, uint32_t rowlen, uint32_t e0, uint32_t nchunks, int lane) {
  constexpr uint32_t CB = (uint32_t)EPL * T, NDW = CB / 4u;
  constexpr int G = EPL / 8;
  static_assert(EPL == 32 || EPL == 16, "a dword or two bytes of every bit row per lane");
  const uint32_t m0 = e0 >> 3, t = (uint32_t)lane;
  if (t >= nchunks) return;
  const gu8* in = src + (size_t)(e0 + (uint32_t)EPL * t) * T;
  uint32_t w[NDW];
#pragma unroll
  for (uint32_t k = 0; k < NDW / 4u; k++) { const uint4 v = g_ld16(in + 16u * k); w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
#pragma unroll
  for (int j = 0; j < T; j++) {
    uint64_t x[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      uint64_t v = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) { const int idx = (8 * g + k) * T + j; v |= (uint64_t)((w[idx >> 2] >> (8 * (idx & 3))) & 0xffu) << (8 * k); }
      x[g] = bit_transpose8(v);
    }
#pragma unroll
    for (int b = 0; b < 8; b++) {
      uint32_t rw = 0;
#pragma unroll
      for (int g = 0; g < G; g++) rw |= (uint32_t)((x[g] >> (8 * b)) & 0xff) << (8 * g);
      gu8* rp = dst + (size_t)(8 * j + b) * rowlen + m0 + (uint32_t)G * t;
      if (G == 4) g_st4(rp, rw); else g_st2(rp, rw);
    }
  }
}
template <int T>
__device__ void bitshuffle_block_wave_T(const gu8* src, gu8* dst, uint32_t bsize, int lane) {
  constexpr int EPL = T == 8 ? 16 : 32;
  const uint32_t N = bsize / T;
  if (bsize < T || (N & 7u)) {                         // not filtered at all (blosc.c:608-609) / copied verbatim by the filter (shuffle.c:412-414)
    for (uint32_t k = 16u * (uint32_t)lane; k < bsize; k += 1024u) {
      if (k + 16u <= bsize) g_st16(dst + k, g_ld16(src + k));
      else for (uint32_t t = k; t < bsize; t++) dst[t] = src[t];
    }
    return;
  }
  const uint32_t rowlen = N >> 3;
  uint32_t e0 = 0;
  for (; e0 + 64u * EPL <= N; e0 += 64u * EPL) bitshuffle_pass<T, EPL>(src, dst, rowlen, e0, 64u, lane);
  if (N - e0 >= (uint32_t)EPL) { const uint32_t nch = (N - e0) / (uint32_t)EPL; bitshuffle_pass<T, EPL>(src, dst, rowlen, e0, nch, lane); e0 += (uint32_t)EPL * nch; }
  if ((uint32_t)lane < ((N - e0) >> 3)) {              // fewer than EPL elements left (a multiple of 8): eight elements per lane, one byte into every bit row
    const uint32_t m = (e0 >> 3) + (uint32_t)lane;
#pragma unroll
    for (int j = 0; j < T; j++) {
      uint64_t v = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) v |= (uint64_t)src[(size_t)(8u * m + (uint32_t)k) * T + (uint32_t)j] << (8 * k);
      v = bit_transpose8(v);
#pragma unroll
      for (int b = 0; b < 8; b++) dst[(size_t)(8 * j + b) * rowlen + m] = (uint8_t)(v >> (8 * b));
    }
  }
  for (uint32_t k = N * T + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}
__device__ __attribute__((noinline)) void bitshuffle_block_task_x(const gu8* src_, gu8* dst_, uint32_t bsize_, uint32_t T_, int lane) {
  const gu8* src = uni_ptr(src_); gu8* dst = uni_ptr(dst_);
  const uint32_t bsize = uni(bsize_), T = uni(T_);
  if (T == 8u) bitshuffle_block_wave_T<8>(src, dst, bsize, lane);
  else if (T == 4u) bitshuffle_block_wave_T<4>(src, dst, bsize, lane);
  else if (T == 2u) bitshuffle_block_wave_T<2>(src, dst, bsize, lane);
  else bitshuffle_block_wave_T<1>(src, dst, bsize, lane);
}

// ---------------------------------------------------------------------------------------------
// Fused byte shuffle of one block by ONE wavefront for every OTHER typesize up to 32 (round 4; the mirror image of unshuffle_block_generic
// in k_decode.hip; blosc/shuffle-generic.h:27-58).  The wave that runs a shuffle task is not encoding: its LDS (the 6 KiB match-finder
// table at least) is free.  Per pass as many elements as that holds: their source bytes in as contiguous 16-byte pieces (1 KiB per load
// instruction), then for every plane a lane gathers the plane's bytes of ITS four elements (of every 256) out of the tile and stores
// one dword - a wave store writes 256 contiguous bytes of the plane.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool shuffle_generic_T(uint32_t T) { return T >= 3u && T <= 32u && T != 4u && T != 8u && T != 16u; }
__device__ __attribute__((noinline)) void shuffle_block_generic(volatile uint32_t* lds_, const gu8* src_, gu8* dst_, uint32_t bsize_, uint32_t T_, int lane) {
  const uint64_t lv = (uint64_t)lds_;
  lu8* S = (lu8*)(BAMD_LAS uint32_t*)(volatile uint32_t*)(((uint64_t)uni((uint32_t)(lv >> 32)) << 32) | uni((uint32_t)lv));
  const gu8* src = uni_ptr(src_); gu8* dst = uni_ptr(dst_);
  const uint32_t bsize = uni(bsize_), T = uni(T_), N = bsize / T;
  // as many 128-element half tiles per pass as the table's LDS holds: typesize 3 moves 6 KiB per pass, typesize 32 one half tile of 4 KiB
  const uint32_t H = ((uint32_t)ENC_TAB_BYTES / 128u) / T;
  static_assert(128u * 32u <= (uint32_t)ENC_TAB_BYTES, "a half tile of the widest element must fit the smallest table a wave owns");
  uint32_t e = 0;
  while (e + 128u <= N) {
    const uint32_t h_here = (N - e) / 128u < H ? (N - e) / 128u : H, K = 128u * h_here;
    const gu8* in = src + (size_t)e * T;
    for (uint32_t q = (uint32_t)lane; q < K * T / 16u; q += 64u) l_st16(S + 16u * q, g_ld16(in + 16u * q));
    LDS_ORDER(); BAMD_LDS_SYNC();
    for (uint32_t el = 4u * (uint32_t)lane; el < K; el += 256u) {      // the lane's four elements el .. el + 3 of every 256
      const lu8* mine = S + el * T;
      if ((T & 3u) == 0u) {                              // whole dwords per element: one dword of each of the four elements -> 4 x 4 byte transpose -> four planes
        const BAMD_LAS uint32_t* w = (const BAMD_LAS uint32_t*)mine;
        for (uint32_t j = 0; j < T; j += 4u) {
          uint32_t r0, r1, r2, r3;
          transpose4x4(w[j / 4u], w[(T + j) / 4u], w[(2u * T + j) / 4u], w[(3u * T + j) / 4u], r0, r1, r2, r3);
          gu8* o = dst + (size_t)j * N + e + el;
          g_st4(o, r0); g_st4(o + (size_t)N, r1); g_st4(o + 2 * (size_t)N, r2); g_st4(o + 3 * (size_t)N, r3);
        }
        continue;
      }
      for (uint32_t j = 0; j < T; j++) {
        const uint32_t v = (uint32_t)mine[j] | ((uint32_t)mine[T + j] << 8) | ((uint32_t)mine[2u * T + j] << 16) | ((uint32_t)mine[3u * T + j] << 24);
        g_st4(dst + (size_t)j * N + e + el, v);
      }
    }
    LDS_ORDER(); BAMD_LDS_SYNC();
    e += K;
  }
  // tail: fewer than K elements, then the bytes that do not form a whole element (copied as they are)
  for (uint32_t k = e * T + (uint32_t)lane; k < N * T; k += 64u) { const uint32_t el = k / T, j = k - el * T; dst[(size_t)j * N + el] = src[k]; }
  for (uint32_t k = N * T + (uint32_t)lane; k < bsize; k += 64u) dst[k] = src[k];
}

__device__ __attribute__((noinline)) void shuffle_block_task(const ChunkDesc* chunks, const BlockDesc* blocks, uint32_t gb,
                                                             uint32_t* blk_ready, StreamDesc* streams, int detect, int lane, volatile uint32_t* lds) {
  const BlockDesc* b = blocks + gb;
  const ChunkDesc* c = chunks + uni((uint32_t)b->chunk);
  const uint32_t blk = uni((uint32_t)b->blk), bsize = uni((uint32_t)b->bsize), bs = uni((uint32_t)c->blocksize);
  const gu8* src = uni_ptr(as_global(c->src)) + (size_t)blk * bs;
  gu8* dst = uni_ptr(as_global(c->filt)) + (size_t)blk * bs;
  const uint32_t T = uni((uint32_t)c->typesize), N = bsize / T;
  // periodic planes (see above): whole rows only, planes that are streams of their own, LZ4 / BloscLZ streams
  const bool det = detect && uni((uint32_t)b->nstreams) == T && N * T == bsize && (N & 255u) == 0u && N >= 1024u &&
                   (uni((uint32_t)c->fmt) == (uint32_t)FMT_LZ4 || uni((uint32_t)c->fmt) == (uint32_t)FMT_BLOSCLZ);
  StreamDesc* planes = streams + uni((uint32_t)b->first_stream);
  if (uni(c->mode) & CH_BITSHUFFLE) bitshuffle_block_task_x(src, dst, bsize, T, lane);      // (typesize 1 / 2 / 4 / 8: the only bitshuffle chunks that carry CH_FUSED_SHUF)
  else if (shuffle_generic_T(T)) shuffle_block_generic(lds, src, dst, bsize, T, lane);
  else if (T == 8u) { if (det) shuffle_block_detect_T<8>(src, dst, bsize, planes, lane); else shuffle_block_wave_T<8>(src, dst, bsize, lane); }
  else if (T == 4u) { if (det) shuffle_block_detect_T<4>(src, dst, bsize, planes, lane); else shuffle_block_wave_T<4>(src, dst, bsize, lane); }
  else shuffle_block_task_x(src, dst, bsize, T, det, planes, lane);       // typesize 2 / 16: out of line, so that the registers of the 16-plane form do not count here
  __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): every store of this wave has reached L2
  if (lane == 0) __hip_atomic_store(&blk_ready[gb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The stream of a plane of n bytes (n % 256 == 0, n >= 1024) that repeats with period p <= 256; `in` holds its first 256
// bytes.  LZ4: p literals, one match at distance p up to n - 5, the last five bytes as literals (lz4.c:245-246 end
// rules).  BloscLZ: the same with the match ending at n - 2 (the encoder's own limit above) and the marker bit of
// blosclz.c:607.  Always smaller than n, so the "store raw" fallback - which would need the plane in memory - cannot hit.
__device__ __forceinline__ uint32_t emit_periodic_stream(const gu8* in, uint32_t n, gu8* out, uint32_t cap, uint32_t p, bool lz4, int lane) {
  uint32_t op;
  if (lz4) {
    op = lz4_emit_seq(out, 0u, cap, in, p, p, n - 5u - p, -1, 0u, lane);
    if (op == 0xffffffffu) return 0u;
    op = lz4_emit_tail(out, op, cap, in + 251u, 5u, lane);
    return op == 0xffffffffu ? 0u : op;
  }
  op = blz_emit_literals(out, 0u, cap, in, p, lane);
  if (op == 0xffffffffu) return 0u;
  op = blz_emit_match(out, op, cap, p, n - 2u - p, lane);
  if (op == 0xffffffffu) return 0u;
  op = blz_emit_literals(out, op, cap, in + 254u, 2u, lane);
  if (op == 0xffffffffu) return 0u;
  if (lane == 0) out[0] |= 0x20u;
  return op;
}

