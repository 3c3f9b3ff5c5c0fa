// enc_zlib.h — zlib streams on the device (included by k_encode.hip inside namespace bamd): zlib_encode_wave.  The symbol packer
// (dfl_put_symbols / dfl_emit_seq) sits with the match finder that calls it (enc_lz.h); format: deflate_enc.h.  DESIGN.md 3.8.
// ---------------------------------------------------------------------------------------------
// zlib streams (deflate_enc.h has the format).  One stream per blosc stream: header, ONE final block with the fixed
// Huffman codes whose symbols the match finder above emits as it goes (dfl_emit_seq), end-of-block, Adler-32.
// Returns the stream size, or 0 when it would not be smaller than the input (blosc then stores the split raw).
// ---------------------------------------------------------------------------------------------
constexpr int DFL_LDS_BYTES = 65 * 4 + 12;     // the 65-dword strip of dfl_put_symbols, rounded to 16 bytes
template <bool HC = false, int USE = 0>      // USE: 0 = the kernel's own path, 1 = the fallback of zlib_dyn_encode_wave (its own copy: the kernels' code stays as it is)
__device__ uint32_t zlib_encode_wave(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap, int clevel,
                                     enc_entry_t* tab_generic, int lane EPROF_ARG) {
  if (n < 16u || cap < 64u) return 0u;
  DflSink z;
  z.out = dst; z.cap = cap; z.pos = dfl::kHeader;
  z.zb = (volatile BAMD_LAS uint32_t*)((BAMD_LAS uint8_t*)(void*)tab_generic + (HC ? HC_TAB_BYTES : ENC_TAB_BYTES));
  if (lane == 0) { uint8_t h[2]; dfl::write_header(h); dst[0] = h[0]; dst[1] = h[1]; }
  const dfl::Sym bh = dfl::block_header();
  z.acc = bh.bits; z.nb = bh.nbits;
  const uint32_t covered = HC ? hc_encode_wave<EF_ZLIB>(src, n, dst, cap, tab_generic, lane, 0u, nullptr, &z)
                              : lz_encode_wave<EF_ZLIB>(src, n, dst, cap, clevel, tab_generic, lane EPROF_PASS, 0u, nullptr, &z);
  if (covered == 0xffffffffu) return 0u;
  // the literals behind the last match, then the end-of-block symbol.  A literal costs at least 8 bits: when what is left
  // cannot fit below n any more (incompressible planes end here with everything still pending), skip the packing
  if (z.pos + (n - covered) + 8u >= n) return 0u;
  if (dfl_emit_seq(z, src + covered, n - covered, 0u, 0u, -1, 0u, lane) == 0xffffffffu) return 0u;
  const dfl::Sym eob = dfl::end_of_block();
  if (!dfl_put_symbols(z, lane == 0 ? eob.bits : 0u, lane == 0 ? eob.nbits : 0u, lane)) return 0u;
  const uint32_t tailbytes = (z.nb + 7u) >> 3;                   // <= 4, zero padding up to the byte boundary
  if ((uint32_t)lane < tailbytes) dst[z.pos + (uint32_t)lane] = (uint8_t)(z.acc >> (8u * (uint32_t)lane));
  z.pos += tailbytes;
  const uint32_t ad = wave_adler32(src, n, lane);
  if (lane < 4) dst[z.pos + (uint32_t)lane] = (uint8_t)(ad >> (24u - 8u * (uint32_t)lane));
  z.pos += dfl::kTrailer;
  return z.pos < n ? z.pos : 0u;
}


// ---------------------------------------------------------------------------------------------
// Dynamic Huffman codes (deflate_enc.h: "dynamic Huffman blocks"; kernels ENC_ZLIB_DYN / ENC_ZLIB_DYN_HC).  The fixed codes cost
// 8-9 bits per literal and 5 per distance whatever the data looks like; codes made for the stream are worth +30 % (linspace) to
// +60 % (bench19) on shuffled numeric blocks (tests/tools/deflate_enc_cpu.cpp, the same format functions behind a greedy matcher).
// Two passes: the match finder runs with the Zstd path's sink (EF_ZLIB2: literals and (literal length, match length, distance)
// triples go to the wave's sequence scratch), then - histograms of the literal/length and distance symbols (LDS atomics; a match
// longer than 258 bytes counts as its pieces), code lengths by wave_code_lengths (15 bits; 5 symbols per lane / 1 per lane), canonical
// codes by one prefix sum per length, the header (the run-length symbols of the code lengths, their own 7-bit code) written by lane 0 -
// a serial walk over at most 316 lengths, microseconds against the stream's milliseconds - and the symbols, 64 per step as in the
// one-pass form, a match piece as one field of up to 48 bits.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ZD_LCNT = 0u, ZD_DCNT = 320u, ZD_LTAB = 384u, ZD_DTAB = 704u, ZD_LENS = 768u, ZD_CS = 848u, ZD_CCNT = 1008u, ZD_CTAB = 1040u,
                   ZD_STRIP = 1072u, ZD_END = ZD_STRIP + ZV_STRIP;                                  // dword offsets in the wave's scratch
static_assert(ZD_END * 4u <= (uint32_t)ENC_TAB_BYTES, "scratch layout");
constexpr uint32_t ZD_SEQCAP = 32768u, ZD_LITOFF = ZD_SEQCAP * 8u, ZD_LITCAP = 256u * 1024u, ZD_SCRATCH_U64 = (ZD_LITOFF + ZD_LITCAP) / 8u;   // this mode's scratch per wave: 256 KiB of triples, then 256 KiB of literals

struct ZdSink { gu8* out; uint32_t cap, pos, acc, nb; volatile BAMD_LAS uint32_t* strip; };
// one field of up to 48 bits per lane, lane order = stream order
__device__ __forceinline__ bool zd_put(ZdSink& z, uint64_t bits, uint32_t nbits, int lane) {
  const uint32_t incl = wave_incl_scan_u32(nbits, lane);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  z.strip[lane] = lane == 0 ? z.acc : 0u; z.strip[lane + 64] = 0u;
  if (lane < (int)ZV_STRIP - 128) z.strip[lane + 128] = 0u;
  BAMD_LDS_SYNC();
  if (nbits) zv_or_bits(z.strip, z.nb + incl - nbits, bits, nbits);
  const uint32_t fill = z.nb + total, ndw = fill >> 5;                  // <= 97 full dwords
  if (z.pos + 4u * ndw + 16u > z.cap) return false;
  BAMD_LDS_SYNC();
#pragma unroll
  for (uint32_t i = 0; i < 2u; i++) { const uint32_t w = (uint32_t)lane + 64u * i; if (w < ndw) g_st4(z.out + z.pos + 4u * w, z.strip[w]); }
  z.acc = uni(z.strip[ndw]); z.nb = fill & 31u; z.pos += 4u * ndw;
  return true;
}
// canonical Deflate codes (RFC 1951 3.2.2) for lengths l[PER] of symbols PER * lane + j, bit-reversed for the LSB-first stream;
// table[sym] = code | length << 16
template <int PER>
__device__ __forceinline__ void zd_assign_codes(const uint32_t (&l)[PER], volatile BAMD_LAS uint32_t* table, int lane) {
  uint32_t next = 0;                                                  // first code of the current length
  for (uint32_t b = 1u; b <= 15u; b++) {
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) mine += l[j] == b ? 1u : 0u;
    const uint32_t incl = wave_incl_scan_u32(mine, lane);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t r = next + incl - mine;
#pragma unroll
    for (int j = 0; j < PER; j++) if (l[j] == b) { table[(uint32_t)PER * (uint32_t)lane + (uint32_t)j] = dfl::rev(r, b) | (b << 16); r++; }
    next = (next + tot) << 1;
  }
#pragma unroll
  for (int j = 0; j < PER; j++) if (l[j] == 0u) table[(uint32_t)PER * (uint32_t)lane + (uint32_t)j] = 0u;
}

template <bool HC = false>
__device__ uint32_t zlib_dyn_encode_wave(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap, int clevel,
                                         enc_entry_t* tab_generic, BAMD_GAS uint64_t* seqbuf, int lane EPROF_ARG) {
  if (n < 16u || cap < 64u) return 0u;
  if (n > (1u << 18)) return zlib_encode_wave<HC, 1>(src, n, dst, cap, clevel, tab_generic, lane EPROF_PASS);   // the triples hold 18-bit lengths (blosc's zlib blocks are <= 256 KiB unless forced)
  // ---- pass 1: the tokens ----
  ZsSink zs;
  zs.lit = (gu8*)seqbuf + ZD_LITOFF; zs.nlit = 0; zs.litcap = ZD_LITCAP;
  zs.seq = seqbuf; zs.nseq = 0; zs.seqcap = ZD_SEQCAP;
  const uint32_t covered = HC ? hc_encode_wave<EF_ZLIB2>(src, n, dst, cap, tab_generic, lane, 0u, &zs)
                              : lz_encode_wave<EF_ZLIB2>(src, n, dst, cap, clevel, tab_generic, lane EPROF_PASS, 0u, &zs);
  // more tokens than the scratch holds (very many short matches, or mostly literals): the one-pass form with the fixed codes
  if (covered == 0xffffffffu || zs.nlit + (n - covered) > zs.litcap) return zlib_encode_wave<HC, 1>(src, n, dst, cap, clevel, tab_generic, lane EPROF_PASS);
  wave_copy_disjoint(zs.lit + zs.nlit, src + covered, n - covered, lane);
  const uint32_t nlit_bytes = zs.nlit + (n - covered), nseq = zs.nseq;
  const gu8* lit = zs.lit;
  volatile BAMD_LAS uint32_t* scr = (volatile BAMD_LAS uint32_t*)(BAMD_LAS uint8_t*)(void*)tab_generic;
  // ---- histograms ----
  BAMD_LDS_SYNC();                                                    // the scratch overlays the match finder's table; the tokens are in memory
#pragma unroll
  for (int j = 0; j < 6; j++) scr[ZD_LCNT + 64u * (uint32_t)j + (uint32_t)lane] = 0u;
  BAMD_LDS_SYNC();
  BAMD_LAS uint32_t* lcnt = (BAMD_LAS uint32_t*)scr + ZD_LCNT; BAMD_LAS uint32_t* dcnt = (BAMD_LAS uint32_t*)scr + ZD_DCNT;
  for (uint32_t i = 4u * (uint32_t)lane; i < nlit_bytes; i += 256u) {
    if (i + 4u <= nlit_bytes) {
      const uint32_t w = g_ld4(lit + i);
#pragma unroll
      for (int b = 0; b < 4; b++) __hip_atomic_fetch_add(lcnt + ((w >> (8 * b)) & 0xffu), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
      for (uint32_t k = i; k < nlit_bytes; k++) __hip_atomic_fetch_add(lcnt + (uint32_t)lit[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  for (uint32_t base = 0; base < nseq; base += 64u) {
    if (base + (uint32_t)lane < nseq) {
      const uint64_t q = seqbuf[base + (uint32_t)lane];
      const uint32_t ml = zenc::seq_ml(q), dist = zenc::seq_off(q);
      const uint32_t np = dfl::npieces(ml);
      const uint32_t first = np > 2u ? np - 2u : 0u;                  // pieces before these are all 258 bytes long
      if (first) __hip_atomic_fetch_add(lcnt + 285u, first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      for (uint32_t k = first; k < np; k++) __hip_atomic_fetch_add(lcnt + dfl::match_symbols(dfl::piece_len(ml, k, np), dist).lsym, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add(dcnt + dfl::match_symbols(3u, dist).dsym, np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  if (lane == 0) __hip_atomic_fetch_add(lcnt + 256u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // end of block
  BAMD_LDS_SYNC();
  // ---- the two codes ----
  uint32_t lc[5], ll_[5], dc[1], dl_[1];
  uint32_t lsum = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) { lc[j] = scr[ZD_LCNT + 5u * (uint32_t)lane + (uint32_t)j]; lsum += lc[j]; }
  lsum = wave_sum_u32(lsum);
  dc[0] = lane < 30 ? scr[ZD_DCNT + (uint32_t)lane] : 0u;
  const uint32_t dsum = wave_sum_u32(dc[0]);
  if (wave_code_lengths<5, 15>(lc, lsum, ll_, lane) < 2u) return 0u;
  const uint32_t dpresent = dsum ? wave_code_lengths<1, 15>(dc, dsum, dl_, lane) : 0u;
  if (dsum && dpresent == 0u) return 0u;
  if (!dsum) dl_[0] = 0u;
  zd_assign_codes<5>(ll_, scr + ZD_LTAB, lane);
  zd_assign_codes<1>(dl_, scr + ZD_DTAB, lane);
  // how many of each are transmitted, and all their lengths as one byte sequence
  uint32_t hi5 = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) if (ll_[j]) hi5 = 5u * (uint32_t)lane + (uint32_t)j + 1u;
  uint32_t nl = wave_max_u32(hi5); if (nl < 257u) nl = 257u;
  uint32_t nd = wave_max_u32(dl_[0] ? (uint32_t)lane + 1u : 0u); if (nd < 1u) nd = 1u;
  volatile BAMD_LAS uint8_t* lens = (volatile BAMD_LAS uint8_t*)(scr + ZD_LENS);
#pragma unroll
  for (int j = 0; j < 5; j++) if (5u * (uint32_t)lane + (uint32_t)j < nl) lens[5u * (uint32_t)lane + (uint32_t)j] = (uint8_t)ll_[j];
  BAMD_LDS_SYNC();
  if ((uint32_t)lane < nd) lens[nl + (uint32_t)lane] = (uint8_t)dl_[0];
  if (lane < 32) scr[ZD_CCNT + (uint32_t)lane] = 0u;
  BAMD_LDS_SYNC();
  // ---- the code-length symbols (lane 0 walks the lengths) and their code ----
  volatile BAMD_LAS uint16_t* cs = (volatile BAMD_LAS uint16_t*)(scr + ZD_CS);
  uint32_t ncs = 0;
  if (lane == 0) {
    const uint32_t tot = nl + nd;
    for (uint32_t i = 0; i < tot;) {
      const uint32_t v = lens[i];
      uint32_t r = 1u;
      while (i + r < tot && lens[i + r] == v) r++;
      i += r;
      if (v == 0u) {
        while (r >= 11u) { const uint32_t t = r < 138u ? r : 138u; cs[ncs++] = (uint16_t)(18u | ((t - 11u) << 8)); scr[ZD_CCNT + 18u] = scr[ZD_CCNT + 18u] + 1u; r -= t; }
        if (r >= 3u) { cs[ncs++] = (uint16_t)(17u | ((r - 3u) << 8)); scr[ZD_CCNT + 17u] = scr[ZD_CCNT + 17u] + 1u; r = 0u; }
        while (r-- > 0u) { cs[ncs++] = 0u; scr[ZD_CCNT] = scr[ZD_CCNT] + 1u; }
      } else {
        cs[ncs++] = (uint16_t)v; scr[ZD_CCNT + v] = scr[ZD_CCNT + v] + 1u; r--;
        while (r >= 3u) { const uint32_t t = r < 6u ? r : 6u; cs[ncs++] = (uint16_t)(16u | ((t - 3u) << 8)); scr[ZD_CCNT + 16u] = scr[ZD_CCNT + 16u] + 1u; r -= t; }
        while (r-- > 0u) { cs[ncs++] = (uint16_t)v; scr[ZD_CCNT + v] = scr[ZD_CCNT + v] + 1u; }
      }
    }
  }
  ncs = (uint32_t)__builtin_amdgcn_readlane((int)ncs, 0);
  BAMD_LDS_SYNC();
  uint32_t cc[1], cl_[1];
  cc[0] = lane < 19 ? scr[ZD_CCNT + (uint32_t)lane] : 0u;
  const uint32_t cpresent = wave_code_lengths<1, 7>(cc, ncs, cl_, lane);
  if (cpresent == 0u) return 0u;
  if (cpresent == 1u) {                                               // inflate wants this code complete: a second 1-bit code nobody uses
    const uint32_t only = (uint32_t)__builtin_ctzll(__ballot(cl_[0] != 0u));
    if ((uint32_t)lane == (only == 0u ? 1u : 0u)) cl_[0] = 1u;
  }
  zd_assign_codes<1>(cl_, scr + ZD_CTAB, lane);
  BAMD_LDS_SYNC();
  // lengths in the header's order; how many of them are sent (at least 4)
  constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  const uint32_t my_ord = lane < 19 ? (uint32_t)order[lane < 19 ? lane : 0] : 0u;
  const uint32_t my_cl = lane < 19 ? scr[ZD_CTAB + my_ord] >> 16 : 0u;
  const uint64_t used = __ballot(my_cl != 0u);
  uint32_t ncl = used ? 64u - (uint32_t)__builtin_clzll(used) : 0u; if (ncl < 4u) ncl = 4u;
  // ---- header, by lane 0: zlib header, block header, counts, the code-length code, the run-length coded lengths ----
  uint32_t hpos = 0, hacc = 0, hnb = 0;
  if (lane == 0) {
    uint8_t h2[2]; dfl::write_header(h2); dst[0] = h2[0]; dst[1] = h2[1];
    uint64_t acc = 0; uint32_t nb = 0, pos = dfl::kHeader;
    auto put = [&](uint32_t v, uint32_t k) { acc |= (uint64_t)v << nb; nb += k; while (nb >= 8u) { if (pos + 64u < cap) dst[pos] = (uint8_t)acc; pos++; acc >>= 8; nb -= 8u; } };
    put(1u | (2u << 1), 3u);                                          // BFINAL = 1, BTYPE = 10
    put(nl - 257u, 5u); put(nd - 1u, 5u); put(ncl - 4u, 4u);
    for (uint32_t k = 0; k < ncl; k++) put(scr[ZD_CTAB + (uint32_t)order[k]] >> 16, 3u);
    for (uint32_t k = 0; k < ncs; k++) {
      const uint32_t e = cs[k], sy = e & 0xffu, ex = e >> 8;
      const uint32_t t = scr[ZD_CTAB + sy];
      put(t & 0xffffu, t >> 16);
      if (sy == 16u) put(ex, 2u); else if (sy == 17u) put(ex, 3u); else if (sy == 18u) put(ex, 7u);
    }
    hpos = pos; hacc = (uint32_t)acc; hnb = nb;
  }
  ZdSink z;
  z.out = dst; z.cap = cap; z.strip = scr + ZD_STRIP;
  z.pos = (uint32_t)__builtin_amdgcn_readlane((int)hpos, 0); z.acc = (uint32_t)__builtin_amdgcn_readlane((int)hacc, 0); z.nb = (uint32_t)__builtin_amdgcn_readlane((int)hnb, 0);
  if (z.pos + 64u >= cap) return 0u;
  // ---- pass 2: the symbols, sequence by sequence (64 literals / 64 match pieces per step) ----
  uint32_t lp = 0;                                                    // literals consumed
  for (uint32_t base = 0; base <= nseq; base += 64u) {
    const uint32_t cnt = nseq - base < 64u ? nseq - base : 64u;
    const uint64_t q = (uint32_t)lane < cnt ? seqbuf[base + (uint32_t)lane] : 0ull;
    const uint32_t q_ll = zenc::seq_ll(q), q_ml = zenc::seq_ml(q), q_off = zenc::seq_off(q);
    const uint32_t steps = cnt + (base + cnt == nseq ? 1u : 0u);      // behind the last sequence: the closing literals
    for (uint32_t k = 0; k < steps; k++) {
      const bool closing = k == cnt;
      const uint32_t ll = closing ? nlit_bytes - lp : (uint32_t)__builtin_amdgcn_readlane((int)q_ll, (int)(k < cnt ? k : 0u));
      const uint32_t ml = closing ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)q_ml, (int)(k < cnt ? k : 0u));
      const uint32_t dist = closing ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)q_off, (int)(k < cnt ? k : 0u));
      const uint32_t np = ml ? dfl::npieces(ml) : 0u;
      uint32_t ldone = 0, pdone = 0;
      while (ldone < ll || pdone < np) {
        const uint32_t lcnt2 = ll - ldone < 64u ? ll - ldone : 64u;
        uint32_t pcnt = 0;
        if (ldone + lcnt2 == ll) pcnt = np - pdone < 64u - lcnt2 ? np - pdone : 64u - lcnt2;
        uint64_t bits = 0; uint32_t nbits = 0;
        if ((uint32_t)lane < lcnt2) { const uint32_t t = scr[ZD_LTAB + (uint32_t)lit[lp + ldone + (uint32_t)lane]]; bits = t & 0xffffu; nbits = t >> 16; }
        else if ((uint32_t)lane < lcnt2 + pcnt) {
          const dfl::MatchSyms m = dfl::match_symbols(dfl::piece_len(ml, pdone + (uint32_t)lane - lcnt2, np), dist);
          const uint32_t tl = scr[ZD_LTAB + m.lsym], td = scr[ZD_DTAB + m.dsym];
          const uint32_t lb = tl >> 16, db = td >> 16;
          bits = (uint64_t)(tl & 0xffffu) | ((uint64_t)m.lextra << lb) | ((uint64_t)(td & 0xffffu) << (lb + m.lbits)) | ((uint64_t)m.dextra << (lb + m.lbits + db));
          nbits = lb + m.lbits + db + m.dbits;
        }
        if (!zd_put(z, bits, nbits, lane)) return 0u;
        ldone += lcnt2; pdone += pcnt;
      }
      lp += ll;
    }
    if (base + cnt == nseq) break;
  }
  {
    const uint32_t t = scr[ZD_LTAB + 256u];                           // end of block
    if (!zd_put(z, lane == 0 ? (uint64_t)(t & 0xffffu) : 0ull, lane == 0 ? t >> 16 : 0u, lane)) return 0u;
  }
  const uint32_t tailbytes = (z.nb + 7u) >> 3;
  if (z.pos + tailbytes + dfl::kTrailer >= cap) return 0u;
  if ((uint32_t)lane < tailbytes) dst[z.pos + (uint32_t)lane] = (uint8_t)(z.acc >> (8u * (uint32_t)lane));
  z.pos += tailbytes;
  const uint32_t ad = wave_adler32(src, n, lane);
  if (lane < 4) dst[z.pos + (uint32_t)lane] = (uint8_t)(ad >> (24u - 8u * (uint32_t)lane));
  z.pos += dfl::kTrailer;
  return z.pos < n ? z.pos : 0u;
}
