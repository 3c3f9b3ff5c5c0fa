// engine.h — host engine: turns a batch of chunks into descriptor tables + kernel launches.
//
// This is the GPU analogue of the reference's scheduler (do_job / serial_blosc / t_blosc,
// blosc/blosc.c:803-918, :1706-1887): where the reference hands blocks to a pthread pool, the
// engine flattens every block and split of every chunk of a batch into tables (dev_types.h) and
// launches one grid per pipeline stage over all of them.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace bamd {

struct CompressParams {
  int clevel;
  int doshuffle;          // 0 / 1 / 2  (blosc/blosc.h:54-56)
  size_t typesize;
  int codec;              // blosc/blosc.h:64-69
  int32_t forced_blocksize;
  int splitmode;          // blosc/blosc.h:114-117
};

struct Job {
  const void* src;
  void* dst;
  size_t srcsize;         // compress: nbytes.  decompress: bytes available at src (0 = trust the header)
  size_t dstsize;
};

// All three return 0 when the batch was processed (per-chunk outcomes in results[], with exactly the
// reference's return-value conventions) or a negative number when the device could not be
// used at all (no GPU, out of memory, launch failure) — never a silent CPU fallback.
int engine_compress_batch(const CompressParams& p, int n, const Job* jobs, int* results, bool device_ptrs,
                          hipStream_t stream);
int engine_decompress_batch(int n, const Job* jobs, int* results, bool device_ptrs, hipStream_t stream);
int engine_getitem(const void* src, int start, int nitems, void* dest, bool src_on_device, bool dst_on_device,
                   hipStream_t stream);

// one block through one filter kernel, host buffers (test hook; kind 0..3 = shuffle, unshuffle, bitshuffle, bitunshuffle)
int engine_filter(int kind, size_t typesize, size_t blocksize, const void* src, void* dst);

int engine_set_device(int dev);      // selects the HIP device for this process (default: current)
int engine_thread_device(int dev);   // >= 0: the calling THREAD's calls run on this device (multi-GPU entry points); -1: back to the process-wide one
int engine_device_count();
void engine_release();               // frees workspace memory (blosc_free_resources / blosc_destroy)
bool engine_is_device_pointer(const void* p);

// per-kernel timing (hipEvents on the launch stream), used by bench.py for the roofline numbers
void engine_prof_enable(int on);
void engine_prof_reset();
int engine_prof_get(const char* kernel, double* total_ms, int* launches);

}  // namespace bamd
