// tests/tools/blosc_emu_lib.cpp — the WHOLE library (host engine + every kernel) built for the CPU on top of the wavefront emulator and
// its HIP runtime stand-in (tests/tools/wave_emu/): same sources, same C ABI.  TEST INFRASTRUCTURE ONLY - never shipped, never loaded by
// the product (which has no CPU path); it lets the CPU suite run chunk-level round trips through the real host logic and the real
// kernels, slowly (about 100 KB/s), where no GPU is at hand.
//   /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -shared -fPIC -w -I tests/tools/wave_emu -I c-blosc_amd/csrc -I include -x c++ \
//       tests/tools/blosc_emu_lib.cpp -o tests/tools/libblosc_amd_emu.so -lpthread
#define WAVE_EMU_RUNTIME
#define WAVE_EMU_IMPLEMENTATION
#include <hip/hip_runtime.h>
#include "engine.hip"
#include "blosc_api.hip"

// how often the round-3 Zstd paths ran so far (k_zstd.hip: g_emu_zstd_paths; [4] = blocks unshuffled by their decoding wave): tests/test_emu_library.py
extern "C" void emu_zstd_path_counts(unsigned long long* out) {
  for (int i = 0; i < 4; i++) out[i] = bamd::g_emu_zstd_paths[i];
  out[4] = bamd::g_emu_own_block_unshuffles;
}
