// tests/tools/lz_wave_cpu.cpp — the wave-parallel LZ encoders of c-blosc_amd/csrc/k_encode.hip (lz_encode_wave for LZ4 and
// BloscLZ, lz4hc_encode_wave) run on the CPU by the wavefront emulator (tests/tools/wave_emu/wave_emu.h): the SAME source
// the GPU runs, every lane a coroutine.  TEST INFRASTRUCTURE (tests/test_wave_emu_encoders.py): lets the CPU suite check
// that what these functions write is a valid stream that decodes to the input - without a GPU.
//   /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -shared -fPIC -I tests/tools/wave_emu -I c-blosc_amd/csrc -x c++ \
//       tests/tools/lz_wave_cpu.cpp -o tests/tools/liblz_wave_cpu.so
#define WAVE_EMU_IMPLEMENTATION
#include <hip/hip_runtime.h>
#include "dev_types.h"
#include "k_filters.hip"
#include "k_decode.hip"
#include "k_encode.hip"
#include "k_zstd.hip"
#include "k_zstd2.hip"
#include "k_zlib.hip"

namespace {
struct Job { int kind; const uint8_t* src; int n; uint8_t* dst; int cap; int clevel; uint32_t* tab; uint32_t result; uint64_t* seqbuf; };

void lane_body(int lane, void* arg) {
  Job* j = (Job*)arg;
  using namespace bamd;
  uint32_t r;
  if (j->kind == 10) r = zlib_dyn_encode_wave<false>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 11) r = zlib_dyn_encode_wave<true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 8) r = zstd_encode_wave<true, false, true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 9) r = zstd_encode_wave<true, true, true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 6) r = zstd_encode_wave<true, true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 7) r = zlib_encode_wave<true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);
  else if (j->kind == 5) r = zstd_encode_wave<true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 3) r = zstd_encode_wave((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 4) r = zlib_encode_wave((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);
  else if (j->kind == 2) r = lz4hc_encode_wave((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->tab, lane);
  else if (j->kind == 1) r = lz_encode_wave<EF_BLOSCLZ>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);
  else if (j->kind == 12) r = lz_encode_wave<EF_LZ4>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);      // the sequential select / emit loop (BAMD_ENC_PAR=0 builds; still the front end of the other writers)
  else if (j->kind == 13) r = lz4_encode_wave_par<0>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);      // every position probed, whatever the level
  else if (j->kind == 14) r = lz4_encode_wave_par<1>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);      // every other position probed, whatever the level
  else r = BAMD_ENC_PAR ? lz4_encode_wave_auto((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane)
                        : lz_encode_wave<EF_LZ4>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);
  if (lane == 0) j->result = r;
}
}  // namespace

// ---- the fused byte (un)shuffle of one block by one wave (k_encode.hip: shuffle_block_wave_T / shuffle_block_wave_detect with its
// periodic-plane detection and emit_periodic_stream; k_decode.hip: unshuffle_block_wave) ----
namespace {
struct SJob { int T, mode; const uint8_t* src; uint8_t* dst; uint32_t bsize; uint32_t period[16]; uint32_t per; uint32_t* lds; };
void shuf_body(int lane, void* arg) {
  SJob* j = (SJob*)arg;
  using namespace bamd;
  if (unshuffle_generic_T((uint32_t)j->T)) {         // every other typesize up to 32: the LDS-tile forms (round 4)
    if (j->mode == 2) unshuffle_block_generic((volatile uint32_t*)j->lds, j->src, j->dst, j->bsize, (uint32_t)j->T, 0u, lane);
    else shuffle_block_generic((volatile uint32_t*)j->lds, (const gu8*)j->src, (gu8*)j->dst, j->bsize, (uint32_t)j->T, lane);
    if (lane == 0) { j->per = 0; for (int k = 0; k < 16; k++) j->period[k] = 0; }
    return;
  }
  if (j->mode == 2) { unshuffle_block_wave(j->src, j->dst, j->bsize, j->T, lane, nullptr, nullptr, nullptr); return; }
  uint32_t period[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t per = 0;
  if (j->T == 8 || j->T == 4) {
    uint32_t p8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (j->T == 8) { if (j->mode) per = shuffle_block_wave_detect<8>((const gu8*)j->src, (gu8*)j->dst, j->bsize, lane, p8); else shuffle_block_wave_T<8>((const gu8*)j->src, (gu8*)j->dst, j->bsize, lane); }
    else { if (j->mode) per = shuffle_block_wave_detect<4>((const gu8*)j->src, (gu8*)j->dst, j->bsize, lane, p8); else shuffle_block_wave_T<4>((const gu8*)j->src, (gu8*)j->dst, j->bsize, lane); }
    for (int k = 0; k < 8; k++) period[k] = p8[k];
  } else if (j->T == 16) { if (j->mode) per = shuffle_block_wave_detect_x<16>((const gu8*)j->src, (gu8*)j->dst, j->bsize, lane, period); else shuffle_block_wave_T<16>((const gu8*)j->src, (gu8*)j->dst, j->bsize, lane); }
  else { if (j->mode) per = shuffle_block_wave_detect_x<2>((const gu8*)j->src, (gu8*)j->dst, j->bsize, lane, period); else shuffle_block_wave_T<2>((const gu8*)j->src, (gu8*)j->dst, j->bsize, lane); }
  if (lane == 0) { j->per = per; for (int k = 0; k < 16; k++) j->period[k] = period[k]; }
}
struct PJob { const uint8_t* in; uint32_t n; uint8_t* out; uint32_t cap, p; int lz4; uint32_t result; };
void per_body(int lane, void* arg) {
  PJob* j = (PJob*)arg;
  const uint32_t r = bamd::emit_periodic_stream((const bamd::gu8*)j->in, j->n, (bamd::gu8*)j->out, j->cap, j->p, j->lz4 != 0, lane);
  if (lane == 0) j->result = r;
}
}  // namespace
// mode 0: shuffle, 1: shuffle with periodic-plane detection (period_out[k] != 0: plane k repeats with that period and only its first
// 256 bytes were written), 2: unshuffle.  typesize 2, 4, 8 or 16; bsize a multiple of 256 * typesize for modes 0 / 1.  Returns the plane mask.
extern "C" unsigned emu_shuffle_block(int T, int mode, const uint8_t* src, uint8_t* dst, unsigned bsize, unsigned* period_out) {
  SJob j = {T, mode, src, dst, bsize, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, nullptr};
  j.lds = (uint32_t*)aligned_alloc(64, 16384);        // the wave's LDS (the decoder's rings / the encoder's table)
  memset(j.lds, 0xA5, 16384);
  wave_emu::run(shuf_body, &j);
  free(j.lds);
  if (period_out) for (int k = 0; k < 16; k++) period_out[k] = j.period[k];      // room for 16 words
  return j.per;
}
// the whole stream of a plane of n bytes with period p whose first 256 bytes are `in256`
extern "C" unsigned emu_periodic_stream(const uint8_t* in256, unsigned n, uint8_t* out, unsigned cap, unsigned p, int lz4) {
  PJob j = {in256, n, out, cap, p, lz4, 0};
  wave_emu::run(per_body, &j);
  return j.result;
}

// ---- one whole split block through decode_one_stream (k_decode.hip): every stream decoded by its own (emulated) wave, one after the
// other, with the periodic-span bookkeeping of fused chunks; the wave that completes the block's last stream runs the fused unshuffle ----
namespace {
struct BJob { bamd::StreamDesc* sd; int32_t* status; uint32_t* scr; bamd::ChunkDesc* chunks; bamd::BlockDesc* blocks; uint32_t* blk_done; uint32_t sid; uint32_t* spans; uint8_t* pat; };
void blk_body(int lane, void* arg) {
  BJob* j = (BJob*)arg;
  bamd::decode_one_stream(j->sd + j->sid, j->status, (volatile uint32_t*)j->scr, j->chunks, j->blocks, j->blk_done, lane, j->sid, j->spans, j->pat, nullptr);
}
}  // namespace
// streams[k] / csize[k]: the T compressed splits of one block of `bsize` bytes (typesize T = 2, 4, 8 or 16, byte-shuffled), fmt = FMT_LZ4 /
// FMT_BLOSCLZ; dst receives the unshuffled block.  order[k]: the sequence in which the streams are decoded (any permutation).  Returns the
// chunk status word (0 = fine).  spans_out (2 T words) shows which planes took the periodic-span / raw-in-place shortcuts.
extern "C" int emu_decode_block(int T, int fmt, const uint8_t* const* streams, const int* csize, unsigned bsize, uint8_t* dst, const int* order, unsigned* spans_out) {
  using namespace bamd;
  const uint32_t ne = bsize / (uint32_t)T;
  uint8_t* filt = (uint8_t*)malloc(bsize + 4096);          // the padded plane layout of fused chunks (dev_types.h)
  memset(filt, 0xCD, bsize + 4096);
  ChunkDesc c; memset(&c, 0, sizeof c);
  c.src = nullptr; c.dst = dst; c.filt = filt; c.nbytes = (int32_t)bsize; c.blocksize = (int32_t)bsize; c.typesize = T; c.nblocks = 1;
  c.nsplits = T; c.fmt = fmt; c.mode = CH_SHUFFLE | CH_FUSED_UNSHUF; c.first_block = 0; c.first_stream = 0;
  BlockDesc b; memset(&b, 0, sizeof b);
  b.chunk = 0; b.blk = 0; b.first_stream = 0; b.nstreams = T; b.bsize = (int32_t)bsize;
  StreamDesc sd[16];
  for (int k = 0; k < T; k++) { sd[k].in = streams[k]; sd[k].out = filt + (size_t)k * filt_plane_stride(c, bsize, T); sd[k].in_size = csize[k]; sd[k].out_size = (int32_t)ne; sd[k].chunk = 0; sd[k].fmt = fmt; sd[k].aux = 0; sd[k].result = 0; }
  int32_t status = 0; uint32_t blk_done = 0;
  uint32_t spans[32]; memset(spans, 0, sizeof spans);
  uint8_t* pat = (uint8_t*)malloc((size_t)T * SPAN_PAT + 64);
  uint32_t* scr = (uint32_t*)aligned_alloc(64, bamd::DR_LDS_BYTES + 256);
  for (int k = 0; k < T; k++) {
    memset(scr, 0xA5, bamd::DR_LDS_BYTES + 256);
    BJob j = {sd, &status, scr, &c, &b, &blk_done, (uint32_t)order[k], spans, pat};
    wave_emu::run(blk_body, &j);
  }
  if (spans_out) memcpy(spans_out, spans, sizeof(uint32_t) * 2 * (size_t)T);
  free(filt); free(pat); free(scr);
  return status;
}

// ---- the encode side of the same: shuffle_block_task (with the periodic-plane detection) then encode_one_stream<MODE> for every plane
// of one block, as the persistent encode kernel does it task by task ----
namespace {
struct CJob { int mode; bamd::StreamDesc* sd; uint32_t* tab; bamd::ChunkDesc* chunks; bamd::BlockDesc* blocks; uint32_t* blk_ready; uint32_t sid; uint64_t* seqbuf; int shuffle_task; };
void enc_blk_body(int lane, void* arg) {
  CJob* j = (CJob*)arg;
  using namespace bamd;
  if (j->shuffle_task) { shuffle_block_task(j->chunks, j->blocks, 0u, j->blk_ready, j->sd, 1, lane, (volatile uint32_t*)j->tab); return; }
  if (j->mode == ENC_HC) encode_one_stream<ENC_HC>(j->sd + j->sid, j->tab, j->chunks, j->blk_ready, lane, j->blocks, j->sid, nullptr, j->seqbuf);
  else encode_one_stream<ENC_LZ>(j->sd + j->sid, j->tab, j->chunks, j->blk_ready, lane, j->blocks, j->sid, nullptr, j->seqbuf);
}
}  // namespace
// src: one block of bsize bytes (typesize T = 2, 4, 8 or 16); fmt FMT_LZ4 / FMT_BLOSCLZ; mode 0 = plain match finder, 3 = LZ4HC-grade search.
// out: T slots of `slot` bytes each; result[k]: the stream size of plane k (0 = store raw), as the kernel leaves it in StreamDesc::result.
extern "C" void emu_encode_block(int T, int fmt, int mode, int clevel, const uint8_t* src, unsigned bsize, uint8_t* out, unsigned slot, int* result) {
  using namespace bamd;
  const uint32_t ne = bsize / (uint32_t)T;
  uint8_t* filt = (uint8_t*)malloc(bsize + 4096);
  memset(filt, 0xCD, bsize + 4096);
  ChunkDesc c; memset(&c, 0, sizeof c);
  c.src = src; c.dst = nullptr; c.filt = filt; c.nbytes = (int32_t)bsize; c.blocksize = (int32_t)bsize; c.typesize = T; c.nblocks = 1;
  c.nsplits = T; c.fmt = fmt; c.mode = CH_SHUFFLE | CH_FUSED_SHUF; c.clevel = clevel;
  BlockDesc b; memset(&b, 0, sizeof b);
  b.nstreams = T; b.bsize = (int32_t)bsize;
  StreamDesc sd[16];
  for (int k = 0; k < T; k++) { sd[k].in = filt + (size_t)k * ne; sd[k].out = out + (size_t)k * slot; sd[k].in_size = (int32_t)ne; sd[k].out_size = (int32_t)slot; sd[k].chunk = 0; sd[k].fmt = fmt; sd[k].aux = clevel; sd[k].result = 0; }
  uint32_t blk_ready = 0;
  uint32_t* tab = (uint32_t*)aligned_alloc(64, 64 * 1024);
  CJob j = {mode, sd, tab, &c, &b, &blk_ready, 0u, nullptr, 1};
  wave_emu::run(enc_blk_body, &j);                                 // the block's shuffle task
  j.shuffle_task = 0;
  for (int k = 0; k < T; k++) { memset(tab, 0xA5, 64 * 1024); j.sid = (uint32_t)k; wave_emu::run(enc_blk_body, &j); result[k] = sd[k].result; }
  free(filt); free(tab);
}

// ---- the entropy-coded formats' decoders: zlib_decode_wave (k_zlib.hip) and the one-wave-per-frame Zstd decoder (k_zstd.hip) ----
namespace {
struct EJob { int kind; const uint8_t* src; int n; uint8_t* dst; int cap; void* lds; uint8_t* lit; int result; };
void ent_body(int lane, void* arg) {
  EJob* j = (EJob*)arg;
  int r;
  if (j->kind == 4) r = bamd::zlib_decode_wave(j->src, j->n, j->dst, j->cap, *(zi::Tabs*)j->lds, lane);
  else r = bamd::zstd_decode_wave(j->src, j->n, j->dst, j->cap, j->lit, (bamd::ZstdLds*)j->lds, lane);
  if (lane == 0) j->result = r;
}
}  // namespace
// kind: 3 = Zstd frame, 4 = zlib stream.  Returns bytes produced (0 = rejected).
extern "C" int emu_entropy_decode(int kind, const uint8_t* src, int n, uint8_t* dst, int cap) {
  EJob j = {kind, src, n, dst, cap, nullptr, nullptr, 0};
  const size_t lds = kind == 4 ? sizeof(zi::Tabs) : sizeof(bamd::ZstdLds);
  j.lds = aligned_alloc(64, (lds + 63) / 64 * 64);
  memset(j.lds, 0xA5, lds);
  j.lit = (uint8_t*)malloc(128 * 1024 + 64);                    // the Zstd decoder's literals scratch (one block)
  wave_emu::run(ent_body, &j);
  free(j.lds); free(j.lit);
  return j.result;
}

// ---- the decoders of k_decode.hip (lz4_decode_wave, blosclz_decode_wave): one stream, no periodic-span bookkeeping ----
namespace {
struct DJob { int kind; const uint8_t* src; int n; uint8_t* dst; int cap; uint32_t* scr; int result; };
void dec_body(int lane, void* arg) {
  DJob* j = (DJob*)arg;
  using namespace bamd;
  SpanCtx sp = {0u, 0u, 0u, 0u, nullptr};
  int r;
  if (j->kind == 1) r = blosclz_decode_wave((const gu8*)j->src, j->n, (gu8*)j->dst, j->cap, (volatile uint32_t*)j->scr, lane, sp);
  else r = lz4_decode_wave((const gu8*)j->src, j->n, (gu8*)j->dst, j->cap, (volatile uint32_t*)j->scr, lane, sp);
  if (lane == 0) j->result = r;
}
}  // namespace
// kind: 0 = LZ4 block, 1 = BloscLZ stream.  Returns what the device function returns (bytes produced, or its error code).
extern "C" int emu_lz_decode(int kind, const uint8_t* src, int n, uint8_t* dst, int cap) {
  DJob j = {kind, src, n, dst, cap, nullptr, 0};
  j.scr = (uint32_t*)aligned_alloc(64, bamd::DR_LDS_BYTES + 256);      // the wave's LDS: 64 scratch dwords + input ring + history ring (dec_ring.h)
  memset(j.scr, 0xA5, bamd::DR_LDS_BYTES + 256);
  wave_emu::run(dec_body, &j);
  free(j.scr);
  return j.result;
}

// batched steps of the LZ4 ring decoder (dec_ring.h) so far, and matches it served from rows already written to global memory (the tests check
// that reference-written streams really take those paths)
extern "C" unsigned long long emu_ring_steps() { return bamd::g_emu_ring_steps; }
extern "C" unsigned long long emu_ring_far() { return bamd::g_emu_ring_far; }

// kind: 0 = LZ4, 1 = BloscLZ, 2 = LZ4 with the LZ4HC-grade search, 3 = Zstd frame, 4 = zlib stream, 5 = Zstd frame with per-block sequence tables,
// 6 = 5 behind the LZ4HC-grade search, 7 = zlib stream behind the LZ4HC-grade search, 8 / 9 = 5 / 6 with Huffman-coded literals,
// 10 / 11 = zlib stream with dynamic Huffman codes (plain match finder / LZ4HC-grade search).  Returns the stream
// size (0 = "store raw").
extern "C" int emu_lz_encode(int kind, const uint8_t* src, int n, uint8_t* dst, int cap, int clevel, unsigned long long* rendezvous) {
  Job j = {kind, src, n, dst, cap, clevel, nullptr, 0, nullptr};
  // the wave's LDS: big enough for either table, 16-byte aligned, poisoned (the kernels clear what they use)
  j.tab = (uint32_t*)aligned_alloc(64, 64 * 1024);
  memset(j.tab, 0xA5, 64 * 1024);
  if (kind == 10 || kind == 11) j.seqbuf = (uint64_t*)malloc(sizeof(uint64_t) * bamd::ZD_SCRATCH_U64);
  if (kind == 3 || kind == 5 || kind == 6 || kind == 8 || kind == 9) {      // what k_encode_streams_t<ENC_ZSTD> sets up once per persistent wave: the predefined tables behind the hash table, the sequence scratch
    static bamd::zenc::CTabs predefined;
    bamd::zenc::build_predefined(predefined);
    memcpy((uint8_t*)j.tab + ((kind == 6 || kind == 9) ? bamd::HC_TAB_BYTES : bamd::ENC_TAB_BYTES), &predefined, sizeof predefined);
    j.seqbuf = (uint64_t*)malloc(sizeof(uint64_t) * bamd::ZS_SEQCAP);
  }
  wave_emu::run(lane_body, &j);
  if (rendezvous) *rendezvous = wave_emu::last_rendezvous();
  free(j.tab); free(j.seqbuf);
  return (int)j.result;
}
