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
#include "k_decode_blocks.hip"
#include "k_encode.hip"

namespace {
struct Job { int kind; const uint8_t* src; int n; uint8_t* dst; int cap; int clevel; uint32_t* tab; uint32_t result; uint64_t* seqbuf; };

void lane_body(int lane, void* arg) {
  Job* j = (Job*)arg;
  using namespace bamd;
  uint32_t r;
  if (j->kind == 6) r = zstd_encode_wave<true, true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 7) r = zlib_encode_wave<true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);
  else if (j->kind == 5) r = zstd_encode_wave<true>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 3) r = zstd_encode_wave((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, (BAMD_GAS uint64_t*)j->seqbuf, lane);
  else if (j->kind == 4) r = zlib_encode_wave((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);
  else if (j->kind == 2) r = lz4hc_encode_wave((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->tab, lane);
  else if (j->kind == 1) r = lz_encode_wave<EF_BLOSCLZ>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);
  else r = lz_encode_wave<EF_LZ4>((const gu8*)j->src, (uint32_t)j->n, (gu8*)j->dst, (uint32_t)j->cap, j->clevel, j->tab, lane);
  if (lane == 0) j->result = r;
}
}  // namespace

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
  j.scr = (uint32_t*)aligned_alloc(64, 64 * 4 + bamd::LZB_BYTES + 256);      // the wave's LDS: 64 scratch dwords + the step buffer
  memset(j.scr, 0xA5, 64 * 4 + bamd::LZB_BYTES + 256);
  wave_emu::run(dec_body, &j);
  free(j.scr);
  return j.result;
}

// kind: 0 = LZ4, 1 = BloscLZ, 2 = LZ4 with the LZ4HC-grade search, 3 = Zstd frame, 4 = zlib stream, 5 = Zstd frame with per-block sequence tables,
// 6 = 5 behind the LZ4HC-grade search, 7 = zlib stream behind the LZ4HC-grade search.  Returns the stream
// size (0 = "store raw").
extern "C" int emu_lz_encode(int kind, const uint8_t* src, int n, uint8_t* dst, int cap, int clevel, unsigned long long* rendezvous) {
  Job j = {kind, src, n, dst, cap, clevel, nullptr, 0, nullptr};
  // the wave's LDS: big enough for either table, 16-byte aligned, poisoned (the kernels clear what they use)
  j.tab = (uint32_t*)aligned_alloc(64, 64 * 1024);
  memset(j.tab, 0xA5, 64 * 1024);
  if (kind == 3 || kind == 5 || kind == 6) {      // what k_encode_streams_t<ENC_ZSTD> sets up once per persistent wave: the predefined tables behind the hash table, the sequence scratch
    static bamd::zenc::CTabs predefined;
    bamd::zenc::build_predefined(predefined);
    memcpy((uint8_t*)j.tab + (kind == 6 ? bamd::HC_TAB_BYTES : bamd::ENC_TAB_BYTES), &predefined, sizeof predefined);
    j.seqbuf = (uint64_t*)malloc(sizeof(uint64_t) * bamd::ZS_SEQCAP);
  }
  wave_emu::run(lane_body, &j);
  if (rendezvous) *rendezvous = wave_emu::last_rendezvous();
  free(j.tab); free(j.seqbuf);
  return (int)j.result;
}
