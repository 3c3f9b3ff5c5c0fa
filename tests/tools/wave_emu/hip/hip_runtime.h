// Stand-in for <hip/hip_runtime.h> when device code is compiled for the HOST by the wavefront emulator
// (tests/tools/wave_emu/wave_emu.h).  Test infrastructure only.
#pragma once
#include "../wave_emu.h"
#ifdef WAVE_EMU_RUNTIME
#include "../hip_emu_runtime.h"
#endif
