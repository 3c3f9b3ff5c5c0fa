/* include/blosc.h — C ABI of libblosc_amd: a drop-in for c-blosc 1.x's <blosc.h> on the
 * blocked shuffle->compress / decompress->unshuffle path, executed on an AMD MI355X (gfx950).
 *
 * Every declaration below replaces the reference declaration cited next to it
 * (paths relative to the c-blosc tree, blosc/blosc.h @ 1.21.7.dev).  Signatures, constants,
 * return conventions and the chunk format are the reference's; see INTEGRATION.md for how a
 * maintainer links an existing caller against this library instead of stock libblosc.
 *
 * Differences a caller can observe (all documented in DESIGN.md):
 *  - codecs: BloscLZ, LZ4, "lz4hc" (a higher-effort LZ4 encoder; its chunks carry the shared LZ4
 *    format id, blosc.h:96), Zlib and Zstd are implemented on the GPU in both directions.  Snappy
 *    returns -5 exactly like a stock build configured without it (blosc/blosc.c:525-574, :687-695).
 *  - compressed bytes differ from stock (different match finder, block-order layout) but are
 *    valid chunks: stock blosc_decompress() reads them, and this library reads stock chunks.
 *  - src/dest may be host pointers (stock behaviour; data crosses PCIe) or HIP device / managed
 *    pointers (detected with hipPointerGetAttributes; data stays in HBM).
 *  - nthreads is accepted and remembered but does not change execution (the "thread pool" is the GPU).
 *  - no GPU, no result: calls fail (negative return + message on stderr); there is no CPU fallback.
 */
#ifndef BLOSC_AMD_BLOSC_H
#define BLOSC_AMD_BLOSC_H

#include <limits.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLOSC_EXPORT __attribute__((visibility("default")))

/* blosc/blosc.h:20-29 */
#define BLOSC_VERSION_MAJOR 1
#define BLOSC_VERSION_MINOR 21
#define BLOSC_VERSION_RELEASE 7
#define BLOSC_VERSION_STRING "1.21.7.dev"
#define BLOSC_VERSION_REVISION "$Rev$"                   /* blosc/blosc.h:25 */
#define BLOSC_VERSION_DATE "$Date:: 2024-06-24 #$"       /* blosc/blosc.h:26 */
#define BLOSC_VERSION_FORMAT 2

/* blosc/blosc.h:32-51 */
#define BLOSC_MIN_HEADER_LENGTH 16
#define BLOSC_MAX_OVERHEAD BLOSC_MIN_HEADER_LENGTH
#define BLOSC_MAX_BUFFERSIZE (INT_MAX - BLOSC_MAX_OVERHEAD)
#define BLOSC_MAX_TYPESIZE 255
#define BLOSC_MAX_BLOCKSIZE ((INT_MAX - BLOSC_MAX_TYPESIZE * sizeof(int32_t)) / 3)
#define BLOSC_MAX_THREADS 256

/* blosc/blosc.h:54-61 */
#define BLOSC_NOSHUFFLE 0
#define BLOSC_SHUFFLE 1
#define BLOSC_BITSHUFFLE 2
#define BLOSC_DOSHUFFLE 0x1
#define BLOSC_MEMCPYED 0x2
#define BLOSC_DOBITSHUFFLE 0x4

/* blosc/blosc.h:64-109 */
#define BLOSC_BLOSCLZ 0
#define BLOSC_LZ4 1
#define BLOSC_LZ4HC 2
#define BLOSC_SNAPPY 3
#define BLOSC_ZLIB 4
#define BLOSC_ZSTD 5
#define BLOSC_BLOSCLZ_COMPNAME "blosclz"
#define BLOSC_LZ4_COMPNAME "lz4"
#define BLOSC_LZ4HC_COMPNAME "lz4hc"
#define BLOSC_SNAPPY_COMPNAME "snappy"
#define BLOSC_ZLIB_COMPNAME "zlib"
#define BLOSC_ZSTD_COMPNAME "zstd"
#define BLOSC_BLOSCLZ_LIB 0
#define BLOSC_LZ4_LIB 1
#define BLOSC_SNAPPY_LIB 2
#define BLOSC_ZLIB_LIB 3
#define BLOSC_ZSTD_LIB 4
#define BLOSC_BLOSCLZ_LIBNAME "BloscLZ"
#define BLOSC_LZ4_LIBNAME "LZ4"
#define BLOSC_SNAPPY_LIBNAME "Snappy"
#define BLOSC_ZLIB_LIBNAME "Zlib"
#define BLOSC_ZSTD_LIBNAME "Zstd"
#define BLOSC_BLOSCLZ_FORMAT BLOSC_BLOSCLZ_LIB
#define BLOSC_LZ4_FORMAT BLOSC_LZ4_LIB
#define BLOSC_LZ4HC_FORMAT BLOSC_LZ4_LIB
#define BLOSC_SNAPPY_FORMAT BLOSC_SNAPPY_LIB
#define BLOSC_ZLIB_FORMAT BLOSC_ZLIB_LIB
#define BLOSC_ZSTD_FORMAT BLOSC_ZSTD_LIB
#define BLOSC_BLOSCLZ_VERSION_FORMAT 1
#define BLOSC_LZ4_VERSION_FORMAT 1
#define BLOSC_LZ4HC_VERSION_FORMAT 1
#define BLOSC_SNAPPY_VERSION_FORMAT 1
#define BLOSC_ZLIB_VERSION_FORMAT 1
#define BLOSC_ZSTD_VERSION_FORMAT 1

/* blosc/blosc.h:114-117 */
#define BLOSC_ALWAYS_SPLIT 1
#define BLOSC_NEVER_SPLIT 2
#define BLOSC_AUTO_SPLIT 3
#define BLOSC_FORWARD_COMPAT_SPLIT 4

/* lifecycle — blosc/blosc.h:127, :137 (blosc/blosc.c:2223-2260) */
BLOSC_EXPORT void blosc_init(void);
BLOSC_EXPORT void blosc_destroy(void);

/* blosc/blosc.h:221-223 (blosc/blosc.c:1311-1433).  Honours the BLOSC_CLEVEL, BLOSC_SHUFFLE,
 * BLOSC_TYPESIZE, BLOSC_COMPRESSOR, BLOSC_BLOCKSIZE, BLOSC_NTHREADS, BLOSC_SPLITMODE,
 * BLOSC_NOLOCK and BLOSC_WARN environment variables on every call, like the reference.
 * Returns: >0 compressed bytes; 0 does not fit in destsize / input too large / destsize < 16;
 * -10 bad clevel, shuffle or typesize; -5 codec not available; -1 other error.
 *
 * Which BYTES a compress call writes (the reference pins none: no test of it compares compressed bytes, and with nthreads > 1 its block
 * order depends on thread timing, blosc/blosc.c:1845-1860; single-threaded it is deterministic, :803-867):
 *   - header, blocksize, split decision, bstarts layout (blocks in block order): a function of the arguments alone, equal to stock's;
 *   - "blosclz", "lz4", and "zstd" up to clevel 5 (one match finder, one table probe per probed position - "lz4" at clevel <= 5 probes every
 *     other position, the way the reference's acceleration 10 - clevel skips positions, blosc/blosc.c:577-587): the same input and arguments give the
 *     same chunk, call after call, from any number of threads (tests/test_gpu_threads.py compares the bytes) - but NOT the reference's
 *     bytes: a different, wave-parallel match finder;
 *   - "lz4hc", "zlib" at every clevel and "zstd" from clevel 6: the deeper search inserts into hash buckets from several lanes at
 *     once and the winner of a slot is not fixed, so two calls on the same input may choose different (equally valid) matches: compressed
 *     size and bytes can differ by a fraction of a per cent between calls.  Every such chunk decodes to the input, here and with stock
 *     c-blosc (BLOSC_AMD_LZ4HC=0 / BLOSC_AMD_ZLIB_SEARCH=0 / BLOSC_AMD_ZSTD_SEARCH=0 select the deterministic finder under those names).
 *     Callers that hash or deduplicate COMPRESSED chunks should use one of the deterministic settings above. */
BLOSC_EXPORT int blosc_compress(int clevel, int doshuffle, size_t typesize, size_t nbytes, const void* src,
                                void* dest, size_t destsize);

/* blosc/blosc.h:245-248 (blosc/blosc.c:1282-1308) */
BLOSC_EXPORT int blosc_compress_ctx(int clevel, int doshuffle, size_t typesize, size_t nbytes, const void* src,
                                    void* dest, size_t destsize, const char* compressor, size_t blocksize,
                                    int numinternalthreads);

/* blosc/blosc.h:280 (blosc/blosc.c:1537-1572).  Returns bytes written, 0 for an empty chunk,
 * <0 on error (-1 malformed chunk or dest too small, -5 / -9 unknown codec / codec format). */
BLOSC_EXPORT int blosc_decompress(const void* src, void* dest, size_t destsize);

/* blosc/blosc.h:301-302 (blosc/blosc.c:1520-1535) */
BLOSC_EXPORT int blosc_decompress_ctx(const void* src, void* dest, size_t destsize, int numinternalthreads);

/* blosc/blosc.h:312 (blosc/blosc.c:1574-1703) */
BLOSC_EXPORT int blosc_getitem(const void* src, int start, int nitems, void* dest);

/* blosc/blosc.h:318, :329 (blosc/blosc.c:1951-1973) */
BLOSC_EXPORT int blosc_get_nthreads(void);
BLOSC_EXPORT int blosc_set_nthreads(int nthreads);

/* blosc/blosc.h:335, :347 (blosc/blosc.c:2002-2020) */
BLOSC_EXPORT const char* blosc_get_compressor(void);
BLOSC_EXPORT int blosc_set_compressor(const char* compname);

/* blosc/blosc.h:357, :366 (blosc/blosc.c:330-409) */
BLOSC_EXPORT int blosc_compcode_to_compname(int compcode, const char** compname);
BLOSC_EXPORT int blosc_compname_to_compcode(const char* compname);

/* blosc/blosc.h:380 (blosc/blosc.c:2022-2045) */
BLOSC_EXPORT const char* blosc_list_compressors(void);

/* blosc/blosc.h:387 (blosc/blosc.c:2047-2050) */
BLOSC_EXPORT const char* blosc_get_version_string(void);

/* blosc/blosc.h:403 (blosc/blosc.c:2052-2109); strings are malloc'ed, caller frees */
BLOSC_EXPORT int blosc_get_complib_info(const char* compname, char** complib, char** version);

/* blosc/blosc.h:412 (blosc/blosc.c:2311-2317) */
BLOSC_EXPORT int blosc_free_resources(void);

/* blosc/blosc.h:427-428, :441-442, :464-465, :475-476, :484 (blosc/blosc.c:2112-2180).
 * These read only the 16 header bytes; `cbuffer` must be a HOST pointer. */
BLOSC_EXPORT void blosc_cbuffer_sizes(const void* cbuffer, size_t* nbytes, size_t* cbytes, size_t* blocksize);
BLOSC_EXPORT int blosc_cbuffer_validate(const void* cbuffer, size_t cbytes, size_t* nbytes);
BLOSC_EXPORT void blosc_cbuffer_metainfo(const void* cbuffer, size_t* typesize, int* flags);
BLOSC_EXPORT void blosc_cbuffer_versions(const void* cbuffer, int* version, int* versionlz);
BLOSC_EXPORT const char* blosc_cbuffer_complib(const void* cbuffer);

/* blosc/blosc.h:498, :507, :527 (blosc/blosc.c:2184-2200) */
BLOSC_EXPORT int blosc_get_blocksize(void);
BLOSC_EXPORT void blosc_set_blocksize(size_t blocksize);
BLOSC_EXPORT void blosc_set_splitmode(int splitmode);

#ifdef __cplusplus
}
#endif
#endif /* BLOSC_AMD_BLOSC_H */
