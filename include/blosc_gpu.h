/* include/blosc_gpu.h — device-resident, batched entry points of libblosc_amd.
 *
 * These have no counterpart in the reference (blosc/blosc.h): c-blosc's API moves ONE chunk per
 * call between HOST buffers (blosc/blosc.h:221-223, :280, :312), which on a GPU means one PCIe
 * round trip and one tiny launch per chunk.  SURVEY §8b "Required extension": the same operation
 * over MANY chunks whose bytes already live in HBM, so that the work list of the whole batch
 * (every block and split of every chunk) is one set of launches.  Parameter meaning, chunk format
 * and per-chunk return values are exactly those of blosc_compress_ctx / blosc_decompress /
 * blosc_getitem (include/blosc.h); only the transport differs.
 *
 * Pointer arrays (src[], dest[], sizes, results) are HOST arrays; the buffers they point to are
 * DEVICE (or managed) memory on the current device.  Calls are synchronous: they return when the
 * results are in `cbytes_out` / `nbytes_out`.  `stream` is a hipStream_t passed as void* (NULL =
 * default stream); all device work of the call is ordered on it.
 */
#ifndef BLOSC_AMD_BLOSC_GPU_H
#define BLOSC_AMD_BLOSC_GPU_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef BLOSC_EXPORT
#define BLOSC_EXPORT __attribute__((visibility("default")))
#endif

/* Select the HIP device used by this process.  0 on success.
 * Default (never called): the HIP device that is current in the thread that makes the FIRST compute call; it is pinned then, for every
 * thread.  The choice is process-wide: the library's call contexts (up to 8, one per concurrent caller) follow it lazily, each the next
 * time it is used, dropping the arenas it held on the previous device.  Therefore: do NOT call this while other threads are inside (or
 * about to enter) compute calls - a call in flight finishes on the device it started on, but a caller that raced with the switch may run on
 * either device, and pointers of the other device are then invalid for it.  Switch devices only between batches, from one thread.  The
 * _multi calls below bind their own worker threads to their devices and neither read nor change this setting. */
BLOSC_EXPORT int blosc_gpu_set_device(int device);

/* Batched blosc_compress_ctx (blosc/blosc.h:245-248).  Chunk i: nbytes[i] bytes at src[i] ->
 * a chunk of at most destsize[i] bytes at dest[i]; cbytes_out[i] gets what blosc_compress_ctx
 * would return for it.  `compressor` NULL = the global compressor (blosc_set_compressor);
 * `blocksize` 0 = automatic.  Returns 0, or <0 if the device could not be used.  src / dest are DEVICE (or managed) memory. */
BLOSC_EXPORT int blosc_gpu_compress_batch(int clevel, int doshuffle, size_t typesize, const char* compressor,
                                          size_t blocksize, int nchunks, const void* const* src,
                                          const size_t* nbytes, void* const* dest, const size_t* destsize,
                                          int* cbytes_out, void* stream);

/* Batched blosc_decompress (blosc/blosc.h:280).  srcsize may be NULL (trust each header's
 * cbytes, as the reference does) or give the bytes available at src[i] (then a header claiming
 * more is rejected with -1).  nbytes_out[i] gets blosc_decompress's return value. */
BLOSC_EXPORT int blosc_gpu_decompress_batch(int nchunks, const void* const* src, const size_t* srcsize,
                                            void* const* dest, const size_t* destsize, int* nbytes_out,
                                            void* stream);

/* The two batched calls on HOST buffers (staged over PCIe like the stock entry points, one staging pass per batch instead of
 * one per chunk): what a file reader uses - c-blosc_amd/blpk.py feeds whole many-chunk files this way. */
BLOSC_EXPORT int blosc_gpu_compress_batch_host(int clevel, int doshuffle, size_t typesize, const char* compressor,
                                               size_t blocksize, int nchunks, const void* const* src,
                                               const size_t* nbytes, void* const* dest, const size_t* destsize,
                                               int* cbytes_out);
BLOSC_EXPORT int blosc_gpu_decompress_batch_host(int nchunks, const void* const* src, const size_t* srcsize,
                                                 void* const* dest, const size_t* destsize, int* nbytes_out);

/* ---- one call, all GPUs of the node -------------------------------------------------------------------------------------------
 * The reference's one call fans its blocks out over a pool of worker threads (blosc/blosc.c:904-918 do_job, :871-899
 * parallel_blosc); chunks of a batch are independent, so a many-chunk buffer shards over the GPUs with no exchange between them:
 * chunk c belongs to GPU floor(c * ndev / nchunks) - contiguous ranges, the rule blosc_gpu_partition states (and bench.py /
 * c-blosc_amd/multigpu.py use across processes, one process per GPU, with RCCL only for the cbytes table).  The _multi calls run
 * one host thread per GPU inside THIS process, each bound to its device (the process-wide device of blosc_gpu_set_device is not
 * touched).  devices: ndev HIP device ids, NULL = 0 .. ndev-1.  The buffers of range r are host memory (staged) or memory of /
 * visible to devices[r]; results are per chunk exactly as in the single-GPU calls.  Returns 0, or < 0 if a device could not be used. */
BLOSC_EXPORT int blosc_gpu_device_count(void);
BLOSC_EXPORT int blosc_gpu_partition(size_t nchunks, int world, int rank, size_t* lo, size_t* hi);   /* chunk range [lo, hi) of `rank` */
BLOSC_EXPORT int blosc_gpu_compress_batch_multi(int ndev, const int* devices, int clevel, int doshuffle, size_t typesize,
                                                const char* compressor, size_t blocksize, int nchunks, const void* const* src,
                                                const size_t* nbytes, void* const* dest, const size_t* destsize, int* cbytes_out);
BLOSC_EXPORT int blosc_gpu_decompress_batch_multi(int ndev, const int* devices, int nchunks, const void* const* src,
                                                  const size_t* srcsize, void* const* dest, const size_t* destsize, int* nbytes_out);

/* blosc_getitem (blosc/blosc.h:312) on a device-resident chunk into device memory. */
BLOSC_EXPORT int blosc_gpu_getitem(const void* src, int start, int nitems, void* dest, void* stream);

/* Per-kernel timing with hipEvents on the launch stream (bench.py's roofline numbers).
 * Kernel names: k_shuffle k_unshuffle k_bitshuffle k_bitunshuffle k_decode_plan k_decode_streams
 * k_encode_streams k_chunk_scan k_chunk_compact k_copy_chunks.
 * `enable`: bit 0 = collect timings; bit 1 = build the task queues of the calls that follow in plain block order, without the per-plane
 * cost feedback of the previous call (what the first call on new data gets; bench.py's `sched_cold` figure).  0 = both off. */
BLOSC_EXPORT void blosc_gpu_profile(int enable);
BLOSC_EXPORT void blosc_gpu_profile_reset(void);
BLOSC_EXPORT int blosc_gpu_profile_get(const char* kernel, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif
