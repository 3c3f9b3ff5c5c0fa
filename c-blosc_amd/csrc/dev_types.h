// dev_types.h — descriptor tables shared by the host engine and the HIP kernels.
//
// The reference walks chunk -> block -> split with nested loops on one thread
// (blosc/blosc.c:803-867 serial_blosc, :591-800 blosc_c/blosc_d).  On the GPU every level is
// flattened into a table so that one launch covers every block / stream of every chunk of a
// batch:  ChunkDesc[nchunks]  ->  BlockDesc[nblocks_total]  ->  StreamDesc[nstreams_total]
// ("stream" = one split of a block, or the whole block when it is not split: the unit a codec sees).
#pragma once
#include <stdint.h>

namespace bamd {

enum : int32_t {
  FMT_BLOSCLZ = 0,  // header flag bits 5-7 (blosc/blosc.h:93-99)
  FMT_LZ4 = 1,
  FMT_ZLIB = 3,     // k_zlib.hip (decode), deflate_enc.h (encode)
  FMT_ZSTD = 4,     // k_zstd.hip / k_zstd2.hip (decode), zstd_enc.h (encode)
};

enum : uint32_t {
  CH_SHUFFLE = 1u,     // byte shuffle active for this chunk's blocks (flag bit0 && typesize > 1)
  CH_BITSHUFFLE = 2u,  // bit shuffle active (flag bit2; applied per block when bsize >= typesize)
  CH_MEMCPYED = 4u,    // payload is the raw input after the 16-byte header (flag bit1)
  CH_SKIP = 8u,        // chunk needs no device work (error found on host, or empty)
  CH_FUSED_UNSHUF = 16u,  // decompress: the decode kernel itself unshuffles each block when its last stream is done
  CH_FUSED_SHUF = 32u,    // compress: the encode kernel itself shuffles each block (queue task) before its streams are encoded
  CH_FUSED_BITUNSH = 64u, // decompress, bitshuffle chunks: the wave that completes a block's last stream bit-unshuffles the block (k_decode.hip)
};

struct ChunkDesc {
  const uint8_t* src;   // decompress: compressed chunk.  compress: uncompressed input
  uint8_t* dst;         // decompress: output buffer.     compress: chunk being written
  uint8_t* filt;        // plane-major scratch for this chunk's filtered bytes (nullptr: no filter)
  uint8_t* stage;       // compress only: staging area, one slot of `neblock` bytes per stream
  int32_t nbytes;       // uncompressed size
  int32_t cbytes;       // decompress: header cbytes.  compress: maxbytes (destsize clamped)
  int32_t blocksize;
  int32_t typesize;
  int32_t nblocks;
  int32_t leftover;     // nbytes % blocksize
  int32_t nsplits;      // streams per full block (1 or typesize); the leftover block always has 1
  int32_t fmt;          // FMT_*
  uint32_t mode;        // CH_* bits
  int32_t first_block;  // index of this chunk's block 0 in BlockDesc[]
  int32_t first_stream; // index of its first stream in StreamDesc[]
  int32_t clevel;       // compress only
  int32_t hdr_flags;    // compress only: header flag byte to write
};

// Scratch layout of a filtered chunk: plane-major per block, planes bsize / typesize bytes apart.  (Round 3 padded the planes of fused chunks
// apart to keep them off one HBM channel: no effect, profiles/r03/r03p_dec_ab_plane_padding_no_effect.txt - removed in round 4.)
#if defined(__HIPCC__) || defined(BAMD_WAVE_EMU)
#define BAMD_HD __host__ __device__
#else
#define BAMD_HD
#endif
BAMD_HD inline size_t filt_block_stride(const ChunkDesc& c) {
  return (size_t)c.blocksize;
}
// plane stride inside one block: split blocks of fused chunks are padded, everything else is the plain plane-major image
BAMD_HD inline uint32_t filt_plane_stride(const ChunkDesc& c, uint32_t bsize, int nstreams) {
  (void)nstreams;
  return bsize / (uint32_t)(c.typesize > 0 ? c.typesize : 1);      // bytes per plane (an unsplit block is ONE stream, but still typesize planes)
}

struct BlockDesc {
  int32_t chunk;        // owning chunk
  int32_t blk;          // block index inside the chunk
  int32_t first_stream; // global stream index of split 0
  int32_t nstreams;     // 1 or typesize
  int32_t bsize;        // bytes in this block: blocksize, or `leftover` for a short last block.  Computed
                        // on the host: selecting between two ChunkDesc fields on the device tripped an
                        // AMDGPU backend miscompile (ROCm 7.2, the `leftover > 0` test was dropped).
  int32_t flags;        // BLK_* bits
};
enum : int32_t {
  BLK_Z = 2,            // decompress: a block of a Zstd / zlib chunk: every stream of it belongs to k_zstd_* / k_zlib_streams, k_decode_streams' queues leave it out
  BLK_ZLIB = 4,         // ... of a zlib chunk (set together with BLK_Z): k_zlib_streams has per-XCD queues of its own (queue_order.h)
};

struct StreamDesc {
  const uint8_t* in;    // decode: compressed bytes.  encode: (filtered) plain bytes
  uint8_t* out;         // decode: where the plain bytes go.  encode: staging slot
  int32_t in_size;      // decode: csize.  encode: neblock
  int32_t out_size;     // decode: neblock (must be produced exactly).  encode: slot capacity
  int32_t chunk;
  int32_t fmt;          // FMT_*
  int32_t aux;          // encode: clevel | (global block index << 4).  decode: global index of the owning block
  int32_t result;       // encode: compressed size (0 = store raw).  decode: bytes produced or <0
};

// Negative per-chunk status codes written by kernels (host maps them to the reference's
// return values, blosc/blosc.c:762-782): -1 bad csize chain / bounds, -2 codec produced wrong size.
enum : int32_t { ST_OK = 0, ST_BADCHAIN = -1, ST_BADCODEC = -2 };

}  // namespace bamd
