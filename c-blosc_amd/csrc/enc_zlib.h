// enc_zlib.h — zlib streams on the device (included by k_encode.hip inside namespace bamd): zlib_encode_wave.  The symbol packer
// (dfl_put_symbols / dfl_emit_seq) sits with the match finder that calls it (enc_lz.h); format: deflate_enc.h.  DESIGN.md 3.8.
// ---------------------------------------------------------------------------------------------
// zlib streams (deflate_enc.h has the format).  One stream per blosc stream: header, ONE final block with the fixed
// Huffman codes whose symbols the match finder above emits as it goes (dfl_emit_seq), end-of-block, Adler-32.
// Returns the stream size, or 0 when it would not be smaller than the input (blosc then stores the split raw).
// ---------------------------------------------------------------------------------------------
constexpr int DFL_LDS_BYTES = 65 * 4 + 12;     // the 65-dword strip of dfl_put_symbols, rounded to 16 bytes
template <bool HC = false>
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

