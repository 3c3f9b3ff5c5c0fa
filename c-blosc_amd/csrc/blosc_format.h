// blosc_format.h — host-side chunk-format constants and the blocksize / split policy.
//
// These are the parts of the reference that MUST be reproduced bit-for-bit on the host so that
// headers written here equal stock headers for the same inputs:
//   header layout            blosc/blosc.c:1148-1247, README_CHUNK_FORMAT.rst:15-76
//   compute_blocksize        blosc/blosc.c:962-1060
//   split_block              blosc/blosc.c:929-959
// tests/test_host_abi.py::test_policy_equals_oracle checks them against the oracle (itself pinned to the reference).
#pragma once
#include <limits.h>
#include <stdint.h>

namespace bamd {

constexpr int kVersionFormat = 2;        // blosc/blosc.h:29
constexpr int kMaxOverhead = 16;         // blosc/blosc.h:32-37
constexpr int kMaxBufferSize = INT_MAX - kMaxOverhead;              // blosc/blosc.h:40
constexpr int kMaxTypeSize = 255;        // blosc/blosc.h:43
constexpr int kMaxBlockSize = (INT_MAX - kMaxTypeSize * 4) / 3;     // blosc/blosc.h:47-48
constexpr int kMaxThreads = 256;         // blosc/blosc.h:51
constexpr int kMinBufferSize = 128;      // blosc/blosc.c MIN_BUFFERSIZE
constexpr int kMaxSplits = 16;           // blosc/blosc.c MAX_SPLITS
constexpr int kL1 = 32 * 1024;           // blosc/blosc.c L1

enum { kBloscLZ = 0, kLZ4 = 1, kLZ4HC = 2, kSnappy = 3, kZlib = 4, kZstd = 5 };   // blosc.h:64-69
enum { kAlwaysSplit = 1, kNeverSplit = 2, kAutoSplit = 3, kForwardCompatSplit = 4 };  // blosc.h:114-117
enum { kFlagShuffle = 0x1, kFlagMemcpyed = 0x2, kFlagBitShuffle = 0x4, kFlagReserved = 0x8, kFlagDontSplit = 0x10 };

inline bool codec_is_hcr(int codec) { return codec == kLZ4HC || codec == kZlib || codec == kZstd; }

// -1 for an unknown split mode (the reference prints a message and returns -1 as a "boolean")
inline int split_block(int codec, int typesize, int blocksize, int splitmode) {
  switch (splitmode) {
    case kAlwaysSplit: return 1;
    case kNeverSplit: return 0;
    case kAutoSplit:
      return (codec == kBloscLZ || codec == kSnappy) && typesize <= kMaxSplits && blocksize / typesize >= kMinBufferSize;
    case kForwardCompatSplit:
      return codec != kZstd && typesize <= kMaxSplits && blocksize / typesize >= kMinBufferSize;
  }
  return -1;
}

inline int32_t compute_blocksize(int clevel, int32_t typesize, int32_t nbytes, int32_t forced, int codec, int splitmode) {
  if (nbytes < typesize) return 1;
  int32_t bs = nbytes;
  if (forced) {
    bs = forced;
    if (bs < kMinBufferSize) bs = kMinBufferSize;
    if (bs > kMaxBlockSize) bs = kMaxBlockSize;
  } else if (nbytes >= kL1) {
    bs = kL1;
    if (codec_is_hcr(codec)) bs *= 2;
    switch (clevel) {
      case 0: bs /= 4; break;
      case 1: bs /= 2; break;
      case 2: break;
      case 3: bs *= 2; break;
      case 4: case 5: bs *= 4; break;
      default: bs *= 8; if (clevel == 9 && codec_is_hcr(codec)) bs *= 2; break;
    }
  }
  if (clevel > 0 && split_block(codec, typesize, bs, splitmode)) {
    if (bs > (1 << 18)) bs = 1 << 18;
    bs *= typesize;
    if (bs < (1 << 16)) bs = 1 << 16;
    if (bs > 1024 * 1024) bs = 1024 * 1024;
  }
  if (bs > nbytes) bs = nbytes;
  if (bs > typesize) bs = bs / typesize * typesize;
  return bs;
}

inline int codec_to_format(int codec) {  // blosc.h:93-99
  switch (codec) {
    case kBloscLZ: return 0;
    case kLZ4: case kLZ4HC: return 1;
    case kSnappy: return 2;
    case kZlib: return 3;
    case kZstd: return 4;
  }
  return -1;
}

inline int32_t rd_i32(const uint8_t* p) {
  return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

struct Header {  // parsed view of the 16 header bytes
  int version, versionlz, flags, typesize;
  int32_t nbytes, blocksize, cbytes;
};
inline Header parse_header(const uint8_t* h) {
  Header r;
  r.version = h[0]; r.versionlz = h[1]; r.flags = h[2]; r.typesize = h[3];
  r.nbytes = rd_i32(h + 4); r.blocksize = rd_i32(h + 8); r.cbytes = rd_i32(h + 12);
  return r;
}

}  // namespace bamd
