/* include/blosc_gpu_rccl.h — the exchange steps of a many-chunk buffer sharded over the GPUs of one node, in C, on RCCL.
 *
 * chunks are independent (the reference's only intra-chunk couplings are the bstarts table and the contiguous
 * packing, blosc/blosc.c:816, :1845-1856), so compress / decompress need NO data-path collective: rank r owns the contiguous
 * chunk range blosc_gpu_partition() gives it and runs blosc_gpu_{compress,decompress}_batch (include/blosc_gpu.h) on it.  What a
 * caller that wants ONE container (a file, a message) still needs are the two exchanges below - the same two c-blosc_amd/multigpu.py
 * does through torch.distributed for bench.py, here without Python:
 *   (1) every rank learns the cbytes of every chunk: one ncclAllGather of 4 bytes per chunk;
 *   (2) the compressed chunks go back to back, in chunk order, onto one rank (or all): an all-gather-v written as ONE group of
 *       ncclSend / ncclRecv with the counts (1) gives - each pair of GPUs uses its own xGMI link, nothing is padded;
 *   and the inverse of (2) in front of a sharded decompress.
 * Chunks are moved as opaque runs of cbytes bytes: their bstarts are offsets from the chunk's own start (blosc/blosc.c:816).
 *
 * This is a library of its own (c-blosc_amd/libblosc_amd_rccl.so, links librccl): the drop-in libblosc.so.1 does not depend on RCCL.
 * Loading it: its DT_NEEDED names the drop-in by its SONAME (libblosc.so.1) and its run path is $ORIGIN, so the drop-in must sit NEXT TO
 * this library under that name - `make -C c-blosc_amd rccl` puts the symlink libblosc.so.1 -> libblosc_amd.so there.  (Without it the
 * loader would pick a stock c-blosc from the system's library path, which has none of the blosc_gpu_* symbols.)
 *
 * A communicator rank belongs to the thread that uses it; the calls are collective: every rank of the communicator makes the same
 * call with the same nchunks / table / root - those are equal on all ranks BY CONTRACT and decide alike everywhere.  What only one rank
 * can see (a missing container or too small a buffer on a receiver, a NULL pointer for its own range, a batch call that failed on its
 * device) is AGREED ON before any payload moves: every exchange first takes the minimum of all ranks' verdicts (one 4-byte all-reduce),
 * so either every rank enters the group of sends and receives or none does - a mis-call on one rank cannot leave the others waiting.
 * All return 0, or < 0: -1 bad arguments on this rank, -2 a HIP or RCCL call (or the batched call) failed on this rank (the message goes
 * to stderr), -3 this rank was fine but a peer reported -1 / -2, and nothing was exchanged.  The one thing that cannot be agreed on is a
 * call without a communicator (-1 at once).  Device pointers are memory of the communicator rank's device.
 */
#ifndef BLOSC_AMD_BLOSC_GPU_RCCL_H
#define BLOSC_AMD_BLOSC_GPU_RCCL_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef BLOSC_EXPORT
#define BLOSC_EXPORT __attribute__((visibility("default")))
#endif

typedef struct blosc_gpu_comm blosc_gpu_comm;       /* one rank of an RCCL communicator + its stream and scratch */
#define BLOSC_GPU_COMM_ID_BYTES 128                 /* = NCCL_UNIQUE_ID_BYTES */

/* One process per GPU: rank 0 makes the id, the caller carries it to the other ranks (file, socket, MPI ...), every rank
 * creates its communicator rank on `device`. */
BLOSC_EXPORT int blosc_gpu_comm_unique_id(void* id /* BLOSC_GPU_COMM_ID_BYTES */);
BLOSC_EXPORT int blosc_gpu_comm_create(blosc_gpu_comm** out, int world, int rank, const void* id, int device);
/* One process, ndev GPUs (the layout of blosc_gpu_*_batch_multi): comms[r] lives on devices[r] (NULL: 0 .. ndev-1); use each
 * from a thread of its own (the calls below block until the rank's part is done). */
BLOSC_EXPORT int blosc_gpu_comm_create_all(blosc_gpu_comm** comms, int ndev, const int* devices);
BLOSC_EXPORT void blosc_gpu_comm_destroy(blosc_gpu_comm* comm);
BLOSC_EXPORT int blosc_gpu_comm_rank(const blosc_gpu_comm* comm, int* world, int* rank, int* device);

/* (1) local_cbytes: the results of this rank's blosc_gpu_compress_batch, one per chunk of its range [lo, hi) =
 *     blosc_gpu_partition(nchunks, world, rank); table: nchunks ints, the same on every rank afterwards.  Host arrays. */
BLOSC_EXPORT int blosc_gpu_allgather_cbytes(blosc_gpu_comm* comm, size_t nchunks, const int* local_cbytes, int* table);

/* (2) local_chunks[i]: DEVICE pointer of chunk lo + i (table[lo + i] bytes are taken; entries <= 0 contribute nothing).
 *     container: DEVICE buffer of sum(max(table[c], 0)) bytes on `root` (root = -1: on every rank); ignored elsewhere.
 *     offsets (host, nchunks entries, may be NULL): where chunk c starts inside the container. */
BLOSC_EXPORT int blosc_gpu_gather_chunks(blosc_gpu_comm* comm, size_t nchunks, const int* table, const void* const* local_chunks,
                                         void* container, int root, size_t* offsets);

/* The inverse: `root` holds the container; every rank receives the bytes of its own range into local_packed (DEVICE,
 * sum over its range of max(table[c], 0) bytes); local_offsets (host, hi - lo entries, may be NULL): where chunk lo + i starts in it. */
BLOSC_EXPORT int blosc_gpu_scatter_chunks(blosc_gpu_comm* comm, size_t nchunks, const int* table, const void* container, int root,
                                          void* local_packed, size_t* local_offsets);

/* The whole sharded compress of SURVEY 8e in one collective call per rank: this rank's range [lo, hi) = blosc_gpu_partition(nchunks, world, rank)
 * goes through blosc_gpu_compress_batch (src / nbytes / dest / destsize are indexed by GLOBAL chunk number; only the entries of the own range are
 * touched, and they name memory of this rank's device), the table is all-gathered, the chunks are gathered into `container` on `root` (-1: every
 * rank).  table (host, nchunks) and offsets (host, nchunks, may be NULL) are filled on every rank.  container must hold the worst case on the
 * receivers: sum(destsize) of all chunks, or any bound the caller knows; *container_bytes (may be NULL) gets what was used.
 * Parameters clevel ... blocksize as in blosc_gpu_compress_batch.  Needs libblosc_amd.so (the drop-in) next to this library. */
BLOSC_EXPORT int blosc_gpu_compress_sharded(blosc_gpu_comm* comm, int clevel, int doshuffle, size_t typesize, const char* compressor, size_t blocksize,
                                            size_t nchunks, const void* const* src, const size_t* nbytes, void* const* dest, const size_t* destsize,
                                            int* table, void* container, size_t container_capacity, int root, size_t* offsets, size_t* container_bytes);
/* ... and its inverse: `root` holds the container and the table; every rank receives its range's chunks (into `packed`, device memory of
 * packed_capacity bytes), decompresses them into dest[lo .. hi) (device memory of this rank, destsize as in blosc_gpu_decompress_batch) and
 * reports blosc_decompress's return value per chunk in nbytes_out[lo .. hi); the other entries of dest / nbytes_out are not touched. */
BLOSC_EXPORT int blosc_gpu_decompress_sharded(blosc_gpu_comm* comm, size_t nchunks, const int* table, const void* container, int root,
                                              void* packed, size_t packed_capacity, void* const* dest, const size_t* destsize, int* nbytes_out);

#ifdef __cplusplus
}
#endif
#endif
