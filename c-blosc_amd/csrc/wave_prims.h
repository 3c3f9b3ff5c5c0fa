// wave_prims.h — wavefront-level building blocks used by the LZ codecs (gfx950, wave64).
//
// One wavefront owns one LZ stream.  Everything that steers control flow (input position,
// output position, lengths, offsets) is wave-uniform and kept in SGPRs (readfirstlane /
// readlane); the 64 lanes are only used as a 64- to 1024-byte wide copy engine.
//
// Memory-ordering contract relied upon below: vector memory operations of ONE wavefront are
// performed in program order (LLVM AMDGPU memory model, wavefront scope needs no s_waitcnt and
// no cache maintenance: one wave lives on one CU and uses that CU's L1).  A load issued after a
// store of the same wave therefore observes the stored bytes, also when another lane wrote them.
// That is what makes "store literal bytes, then gather match bytes that may overlap them"
// correct without fences; tests/test_gpu_decompress.py (test_handbuilt_lz4_streams, test_handbuilt_blosclz_streams) has adversarial streams for it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mem_prims.h"

namespace bamd {

// Where lanes hand data to each other through LDS between two instructions, relying on the wavefront's lock step: nothing on
// the device; the wavefront emulator of tests/tools/wave_emu (lanes are coroutines there) makes it a rendezvous.
#ifndef BAMD_LDS_SYNC
#define BAMD_LDS_SYNC() ((void)0)
#endif
// The same for global memory: a load that has to see what OTHER lanes of this wave stored in an earlier instruction (the
// memory-ordering contract below).  Nothing on the device.
#ifndef BAMD_MEM_SYNC
#define BAMD_MEM_SYNC() ((void)0)
#endif

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// a pointer every lane agrees on, moved into SGPRs: accesses off it become `global_* v, voffset32, s[base]`
// instead of 64-bit VALU address arithmetic (function arguments and loaded pointers arrive in VGPRs)
__device__ __forceinline__ const gu8* uni_ptr(const gu8* p) {
  const uint64_t v = (uint64_t)p;
  return (const gu8*)(((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v));
}
__device__ __forceinline__ gu8* uni_ptr(gu8* p) {
  const uint64_t v = (uint64_t)p;
  return (gu8*)(((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v));
}
// the same for a pointer of any type (generic address space)
template <typename T>
__device__ __forceinline__ T* uni_gp(T* p) {
  const uint64_t v = (uint64_t)p;
  return (T*)(((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v));
}
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Next index of a global work queue, the same value in every lane.  Lane 0 draws the ticket; the value
// is then read from lane 0 EXPLICITLY (v_readlane ignores the exec mask), so the result is right even if
// the compiler has restructured the surrounding loop with partial exec masks.
__device__ __forceinline__ uint32_t take_ticket(uint32_t* ticket, int lane) {
  uint32_t t = 0;
  if (lane == 0) t = atomicAdd(ticket, 1u);
  return (uint32_t)__builtin_amdgcn_readlane((int)t, 0);
}

__device__ __forceinline__ uint4 ld16u(const gu8* p) { return g_ld16(p); }
__device__ __forceinline__ void st16u(gu8* p, const uint4& v) { g_st16(p, v); }
__device__ __forceinline__ uint32_t ld4u(const gu8* p) { return g_ld4(p); }

// ---- copy of n bytes between disjoint buffers (literals, raw streams): 1 KiB per wave step ----
__device__ __forceinline__ void wave_copy_disjoint(gu8* dst, const gu8* src, uint32_t n, int lane) {
  uint32_t done = 0;
  while (n - done >= 1024) {
    st16u(dst + done + 16 * lane, ld16u(src + done + 16 * lane));
    done += 1024;
  }
  uint32_t rem = n - done;
  if (rem >= 16) {
    uint32_t n16 = rem >> 4;
    if ((uint32_t)lane < n16) st16u(dst + done + 16 * lane, ld16u(src + done + 16 * lane));
    done += n16 << 4;
    rem -= n16 << 4;
  }
  if ((uint32_t)lane < rem) dst[done + lane] = src[done + lane];
}

// ---- LZ match: out[pos + k] = out[pos - off + k], k = 0..len-1, byte-wise forward semantics ----
// (blosc/fastcopy.c:530-639 copy_match, lz4.c:2387-2434).  `off` >= 1, pos - off >= 0 checked by
// the caller.  All arguments wave-uniform.
__device__ __forceinline__ void wave_match_copy(gu8* out, uint32_t pos, uint32_t off, uint32_t len, int lane) {
  uint32_t done = 0;
  uint32_t off_e = off;  // effective distance: a multiple of `off` not exceeding the periodic history
  BAMD_MEM_SYNC();
  if (off < 64 && off < len) {
    // Short period: the match replicates the `off` bytes before pos.  Fetch them once, spread them
    // over the lanes (lane i holds pattern byte i mod off) and store G = off * floor(64/off) bytes
    // per step with no further loads.
    uint32_t pat = ((uint32_t)lane < off) ? out[pos - off + lane] : 0u;
    const uint32_t M = 65536u / off + 1u;                 // floor(i/off) == (i*M)>>16 for i < 64
    const uint32_t reps = (64u * M) >> 16;                 // floor(64/off)
    const uint32_t G = reps * off;
    const uint32_t i_mod = (uint32_t)lane - (((uint32_t)lane * M) >> 16) * off;
    const uint32_t val = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(i_mod << 2), (int)pat);
    const uint32_t head = len < 2048u ? len : 1024u;       // long runs switch to 16-byte copies
    while (done < head) {
      uint32_t chunk = head - done < G ? head - done : G;
      if ((uint32_t)lane < chunk) out[pos + done + lane] = (uint8_t)val;
      done += chunk;
    }
    if (done >= len) return;
    BAMD_MEM_SYNC();
    if ((off & (off - 1u)) == 0u && off <= 16u) {
      // period divides 16: every 16-byte group of the run is the same value -> no more loads at all
      const uint4 v = ld16u(out + pos + done - 16u);
      while (len - done >= 1024u) { st16u(out + pos + done + 16 * lane, v); done += 1024u; }
      if (done >= len) return;
    }
    off_e = G;
    // `done` may not be a multiple of G when head was cut; any multiple of `off` <= off+done is valid
    // only if the copied prefix really is periodic up to `done`, which it is.  Restart from G.
  }
  if (done == 0u && off >= 64u && off <= 1024u && (off & (off - 1u)) == 0u && len >= 2048u) {
    // The period divides 1024: every 1 KiB row of the match is the same, and lane l's 16 bytes of it lie at
    // pos - off + (16 l mod off) - in front of the match, so ONE load serves all rows (round 3: byte planes of a few significant
    // bits decode into dozens of 4 KiB runs of period 128, each of which took five dependent round trips to get going)
    const uint4 row = ld16u(out + pos - off + ((16u * (uint32_t)lane) & (off - 1u)));
    while (len - done >= 1024u) { st16u(out + pos + done + 16 * lane, row); done += 1024u; }
    const uint32_t n16 = (len - done) >> 4;
    if ((uint32_t)lane < n16) st16u(out + pos + done + 16 * lane, row);
    done += n16 << 4;
    // (fewer than 16 bytes are left to the loop below)
  }
  while (done < len) {
    BAMD_MEM_SYNC();
    uint32_t rem = len - done;
    while (off_e < 4096u && 2u * off_e <= off + done) off_e *= 2u;  // history grew: lengthen the stride
    if (off_e == 1024u && (off & (off - 1u)) == 0u && rem >= 2048u) {
      // the period divides 1024: every 1 KiB row from here on equals the previous one -> load one row,
      // then only store (constant and short-period byte planes decode at store-issue speed)
      gu8* d = out + pos + done + 16 * lane;
      const uint4 row = ld16u(d - 1024);
      while (len - done >= 1024u) { st16u(out + pos + done + 16 * lane, row); done += 1024u; }
      continue;
    }
    if (off_e >= 2048u && rem >= 2048u) {
      // up to four independent 1 KiB rows per step: one memory round trip per 2-4 KiB
      uint32_t rows = (rem < off_e ? rem : off_e) >> 10;
      if (rows > 4u) rows = 4u;
      gu8* d = out + pos + done + 16 * lane;
      uint4 a0 = ld16u(d - off_e), a1 = ld16u(d - off_e + 1024), a2 = make_uint4(0, 0, 0, 0), a3 = a2;
      if (rows > 2u) a2 = ld16u(d - off_e + 2048);
      if (rows > 3u) a3 = ld16u(d - off_e + 3072);
      st16u(d, a0); st16u(d + 1024, a1);
      if (rows > 2u) st16u(d + 2048, a2);
      if (rows > 3u) st16u(d + 3072, a3);
      done += rows << 10;
      continue;
    }
    uint32_t chunk = rem < 1024u ? rem : 1024u;
    if (chunk > off_e) chunk = off_e;
    if (chunk >= 16u) {
      uint32_t n16 = chunk >> 4;
      if ((uint32_t)lane < n16) {
        gu8* d = out + pos + done + 16 * lane;
        st16u(d, ld16u(d - off_e));
      }
      done += n16 << 4;
    } else {
      if ((uint32_t)lane < chunk) {
        gu8* d = out + pos + done + lane;
        *d = *(d - off_e);
      }
      done += chunk;
    }
  }
}

// ---- copy of n bytes between DISJOINT buffers by ONE lane (its own literal run / independent match; other lanes copy theirs at the
// same time): 64 bytes per round with all loads ahead of the stores, and overlapping pieces instead of a byte tail - a
// byte-at-a-time tail is up to 15 dependent memory round trips that every lane of the wave waits for ----
__device__ __forceinline__ void lane_copy_disjoint(gu8* d, const gu8* s, uint32_t n) {
  if (n >= 16u) {
    uint32_t k = 0;
    for (; k + 64u <= n; k += 64u) {
      const uint4 a = ld16u(s + k), b = ld16u(s + k + 16), c = ld16u(s + k + 32), e = ld16u(s + k + 48);
      st16u(d + k, a); st16u(d + k + 16, b); st16u(d + k + 32, c); st16u(d + k + 48, e);
    }
    // 0..63 bytes left: up to three whole pieces and one that ends exactly at n (it may overlap the one before)
    const uint32_t r = n - k;
    uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a;
    if (r >= 16u) a = ld16u(s + k);
    if (r >= 32u) b = ld16u(s + k + 16);
    if (r >= 48u) c = ld16u(s + k + 32);
    const uint4 z = ld16u(s + n - 16u);
    if (r >= 16u) st16u(d + k, a);
    if (r >= 32u) st16u(d + k + 16, b);
    if (r >= 48u) st16u(d + k + 32, c);
    st16u(d + n - 16u, z);
  } else if (n >= 8u) {
    const uint64_t a = g_ld8(s), b = g_ld8(s + n - 8u);
    *(BAMD_GAS u64una*)d = a; *(BAMD_GAS u64una*)(d + n - 8u) = b;
  } else if (n >= 4u) {
    const uint32_t a = g_ld4(s), b = g_ld4(s + n - 4u);
    g_st4(d, a); g_st4(d + n - 4u, b);
  } else if (n) {
    const uint8_t a = s[0], b = s[n >> 1], c = s[n - 1u];
    d[0] = a; d[n >> 1] = b; d[n - 1u] = c;
  }
}

// ---- Adler-32 (RFC 1950) of a buffer, all lanes: used by the zlib decoder (k_zlib.hip) and encoder (k_encode.hip) ----
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
  return v;
}
// Adler-32 of p[0..n): a = 1 + sum d_i, b = n + sum (n - i) d_i (mod 65521); every lane takes 16-byte pieces
__device__ __forceinline__ uint32_t wave_adler32(const gu8* p, uint32_t n, int lane) {
  uint64_t s1 = 0, s2 = 0;
  uint32_t i = 16u * (uint32_t)lane;
  for (; i + 16u <= n; i += 1024u) {
    const uint4 v = g_ld16(p + i);
    const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
    uint32_t t1 = 0, t2 = 0;             // t2 = sum k * d_(i+k), k = 0..15
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint32_t d = (wds[k >> 2] >> (8 * (k & 3))) & 0xffu; t1 += d; t2 += (uint32_t)k * d; }
    s1 += t1; s2 += (uint64_t)(n - i) * t1 - t2;
  }
  // the last n % 16 bytes: one byte per lane
  const uint32_t tail0 = n & ~15u;
  if (tail0 + (uint32_t)lane < n) { const uint32_t d = p[tail0 + (uint32_t)lane]; s1 += d; s2 += (uint64_t)(n - tail0 - (uint32_t)lane) * d; }
  const uint32_t a = (1u + wave_sum_u32((uint32_t)(s1 % 65521u)) % 65521u) % 65521u;
  const uint32_t b = (n % 65521u + wave_sum_u32((uint32_t)(s2 % 65521u)) % 65521u) % 65521u;
  return (b << 16) | a;
}

// ---- 512-byte register window over the compressed stream ------------------------------------
// lane l holds stream bytes [base + 4l, +4) in `lo` and [base + 256 + 4l, +4) in `hi`.
// Bytes beyond the stream end read as 0 and are never consumed (every consumer bounds-checks
// against in_size first).
struct Window {
  const gu8* in;
  uint32_t in_size;
  uint32_t base;
  uint32_t lo, hi;
  int lane;

  __device__ __forceinline__ uint32_t fetch(uint32_t pos) const {
    uint32_t p = pos + 4u * (uint32_t)lane;
    uint32_t v = 0;
    if (p + 4u <= in_size) v = ld4u(in + p);
    else if (p < in_size) {
      if (in_size >= 4u) v = ld4u(in + in_size - 4u) >> (8u * (p + 4u - in_size));   // last dword, shifted down
      else for (uint32_t b = 0; p + b < in_size; b++) v |= (uint32_t)in[p + b] << (8u * b);
    }
    return v;
  }
  __device__ __forceinline__ void init(const gu8* in_, uint32_t n, int lane_) {
    in = in_; in_size = n; lane = lane_; base = 0;
    lo = fetch(0); hi = fetch(256);
  }
  // make [p, p+8) readable: base <= p and p + 8 <= base + 512
  __device__ __forceinline__ void seek(uint32_t p) {
    if (p - base < 256u) return;
    if (p - base < 504u) { lo = hi; base += 256u; hi = fetch(base + 256u); return; }
    base = p & ~3u; lo = fetch(base); hi = fetch(base + 256u);
  }
  // four stream bytes starting at p (little endian); needs seek(p) or an earlier seek within 248 bytes
  __device__ __forceinline__ uint32_t peek32(uint32_t p) const {
    uint32_t r = p - base;
    uint32_t i0 = r >> 2, i1 = i0 + 1u;
    // uniform branches (not selects): `hi` may still be in flight right after a slide and must not be
    // waited for unless the bytes really come from it
    uint32_t a, b;
    if (i0 < 64u) a = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)i0);
    else a = (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)(i0 - 64u));
    if (i1 < 64u) b = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)i1);
    else b = (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)((i1 - 64u) & 63u));
    uint64_t w = ((uint64_t)b << 32) | a;
    return (uint32_t)(w >> ((r & 3u) * 8u));
  }
  __device__ __forceinline__ uint32_t byte_at(uint32_t p) { seek(p); return peek32(p) & 0xffu; }

  // lane i < n gets stream byte (p + i) out of the window.  Requires base <= p, p + n <= base + 512, n <= 64.
  __device__ __forceinline__ uint32_t gather_bytes(uint32_t p) const {
    const uint32_t r0 = p - base;                    // uniform
    const uint32_t r = r0 + (uint32_t)lane;
    const uint32_t d = r >> 2;
    uint32_t v;
    if (r0 + 64u <= 256u) {                          // all 64 bytes come from `lo`: do not wait for `hi`
      v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(d << 2), (int)lo);
    } else if (r0 >= 256u) {
      v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((d & 63u) << 2), (int)hi);
    } else {
      const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((d & 63u) << 2), (int)lo);
      const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((d & 63u) << 2), (int)hi);
      v = (d < 64u) ? a : b;
    }
    return (v >> ((r & 3u) * 8u)) & 0xffu;
  }
};

}  // namespace bamd
