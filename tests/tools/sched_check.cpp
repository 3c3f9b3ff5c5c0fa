// tests/tools/sched_check.cpp — CPU check of the host-side queue builders (c-blosc_amd/csrc/queue_order.h).
// Invariants: every stream is queued exactly once, on the XCD that owns its block (g & 7); a fused block's
// shuffle task is queued exactly once, on the same XCD, BEFORE all of that block's streams (the encoder's
// no-deadlock argument); blocks without streams and non-fused blocks have no shuffle task; offsets are a
// proper prefix sum.  With and without cost feedback, for mixed geometries.
//   g++ -O1 -std=c++17 -I c-blosc_amd/csrc tests/tools/sched_check.cpp -o /tmp/sched_check && /tmp/sched_check
#include <stdio.h>
#include <string.h>
#include <map>
#include <random>
#include "queue_order.h"

using namespace bamd;

static int fail(const char* what, int trial) { printf("FAIL trial %d: %s\n", trial, what); return 1; }

int main() {
  std::mt19937 rng(12345);
  for (int trial = 0; trial < 400; trial++) {
    std::vector<ChunkDesc> chunks; std::vector<BlockDesc> blocks;
    const int nchunks = 1 + (int)(rng() % 6);
    int nstr = 0;
    for (int c = 0; c < nchunks; c++) {
      ChunkDesc cd; memset(&cd, 0, sizeof cd);
      const int T = (int[]){1, 2, 4, 8, 8, 8, 16}[rng() % 7];
      const bool split = rng() % 4 != 0, memcpyed = rng() % 9 == 0;
      cd.typesize = T; cd.mode = 0;
      if (!memcpyed && (T == 4 || T == 8) && rng() % 3 != 0) cd.mode |= CH_SHUFFLE | CH_FUSED_SHUF;
      const int nb = trial < 20 ? (int)(rng() % 4) : (int)(rng() % 700);
      for (int j = 0; j < nb; j++) {
        BlockDesc b; memset(&b, 0, sizeof b);
        b.chunk = c; b.blk = j; b.first_stream = nstr;
        const bool last = j == nb - 1 && rng() % 2;
        b.nstreams = memcpyed ? 0 : ((split && !last) ? T : 1);
        nstr += b.nstreams; blocks.push_back(b);
      }
      chunks.push_back(cd);
    }
    uint32_t cost[256];
    for (int k = 0; k < 256; k++) cost[k] = (rng() % 3 == 0) ? 0u : (uint32_t)(rng() % 100000);
    const bool valid = trial % 2;
    const int nq = (trial % 5 == 4) ? 1 : 8;     // 1: the single-queue fallback
    // ---- encode queues ----
    std::vector<int32_t> q;
    build_encode_queues(blocks, chunks, cost, valid, q, nq);
    if (q.size() < 10 || q[0] != 0) return fail("enc: header", trial);
    for (int x = 0; x < 8; x++) if (q[x + 1] < q[x]) return fail("enc: offsets not monotone", trial);
    size_t nfused = 0;
    for (auto& b : blocks) if (b.nstreams > 0 && (chunks[b.chunk].mode & CH_FUSED_SHUF)) nfused++;
    if ((size_t)q[8] != (size_t)nstr + nfused) return fail("enc: task count", trial);
    // behind the entries: shoff[9] | the shuffle list (the same blocks per XCD, in the order of the negative entries)
    const size_t sh_at = (size_t)9 + ((size_t)q[8] ? (size_t)q[8] : 1);
    if (q.size() != sh_at + 9 + nfused || q[sh_at] != 0 || (size_t)q[sh_at + 8] != nfused) return fail("enc: vector size / shuffle list header", trial);
    for (int x = 0; x < 8; x++) {
      size_t k = (size_t)q[sh_at + (size_t)x];
      for (int i = q[x]; i < q[x + 1]; i++) if (q[9 + i] < 0) {
        if (k >= (size_t)q[sh_at + (size_t)x + 1] || q[sh_at + 9 + k] != -(q[9 + i] + 1)) return fail("enc: shuffle list differs from the negative entries", trial);
        k++;
      }
      if (k != (size_t)q[sh_at + (size_t)x + 1]) return fail("enc: shuffle list length", trial);
    }
    std::vector<int> owner((size_t)nstr, -1);
    for (size_t g = 0; g < blocks.size(); g++) for (int k = 0; k < blocks[g].nstreams; k++) owner[(size_t)blocks[g].first_stream + k] = (int)g;
    std::vector<int> seen((size_t)nstr, 0); std::vector<int> shuffled(blocks.size(), 0);
    for (int x = 0; x < 8; x++)
      for (int i = q[x]; i < q[x + 1]; i++) {
        const int32_t t = q[9 + i];
        if (t < 0) {
          const size_t g = (size_t)(-(t + 1));
          if (g >= blocks.size() || (int)(g % (size_t)nq) != x) return fail("enc: shuffle task on the wrong XCD", trial);
          if (!(chunks[blocks[g].chunk].mode & CH_FUSED_SHUF) || blocks[g].nstreams == 0) return fail("enc: shuffle task for a non-fused block", trial);
          if (shuffled[g]++) return fail("enc: shuffle task twice", trial);
        } else {
          if (t >= nstr) return fail("enc: stream index out of range", trial);
          const int g = owner[(size_t)t];
          if ((g % nq) != x) return fail("enc: stream on the wrong XCD", trial);
          if (seen[(size_t)t]++) return fail("enc: stream twice", trial);
          if ((chunks[blocks[(size_t)g].chunk].mode & CH_FUSED_SHUF) && !shuffled[(size_t)g]) return fail("enc: stream queued before its block's shuffle task", trial);
        }
      }
    for (int s = 0; s < nstr; s++) if (seen[(size_t)s] != 1) return fail("enc: stream missing", trial);
    for (size_t g = 0; g < blocks.size(); g++)
      if ((blocks[g].nstreams > 0 && (chunks[blocks[g].chunk].mode & CH_FUSED_SHUF)) != (shuffled[g] == 1)) return fail("enc: shuffle task missing", trial);
    // ---- decode queues ----
    std::vector<int32_t> d;
    build_xcd_queues(blocks, (size_t)nstr, cost, valid, d, nq);
    if (d.size() != (size_t)9 + (nstr ? (size_t)nstr : 1) || d[0] != 0 || d[8] != nstr) return fail("dec: header", trial);
    std::fill(seen.begin(), seen.end(), 0);
    for (int x = 0; x < 8; x++) {
      if (d[x + 1] < d[x]) return fail("dec: offsets not monotone", trial);
      for (int i = d[x]; i < d[x + 1]; i++) {
        const int32_t t = d[9 + i];
        if (t < 0 || t >= nstr) return fail("dec: stream index out of range", trial);
        if ((owner[(size_t)t] % nq) != x) return fail("dec: stream on the wrong XCD", trial);
        if (seen[(size_t)t]++) return fail("dec: stream twice", trial);
      }
    }
    for (int s = 0; s < nstr; s++) if (seen[(size_t)s] != 1) return fail("dec: stream missing", trial);
  }
  // ---- round 3: blocks of Zstd chunks (BLK_Z) are in nobody's queue - their kernels take tickets over all streams -, blocks of zlib chunks
  //      (BLK_Z | BLK_ZLIB) are in the zlib kernel's queues and only there, everything else is k_decode_streams': a partition, every
  //      stream on its block's XCD, in mixed batches of any geometry ----
  for (int trial = 0; trial < 200; trial++) {
    const int nblk = 1 + (int)(rng() % 300), nq = (trial % 5 == 4) ? 1 : 8;
    std::vector<BlockDesc> blocks; std::vector<int> owner; std::vector<int> kind;      // kind per stream: 0 LZ, 1 Zstd, 2 zlib
    int nstr = 0;
    for (int j = 0; j < nblk; j++) {
      BlockDesc b; memset(&b, 0, sizeof b);
      const int k = (int)(rng() % 3), ns = (rng() % 4 == 0) ? 1 : (int)(1 + rng() % 16);
      b.blk = j; b.first_stream = nstr; b.nstreams = ns; b.flags = k == 1 ? BLK_Z : (k == 2 ? (BLK_Z | BLK_ZLIB) : 0);
      for (int t = 0; t < ns; t++) { owner.push_back(j); kind.push_back(k); }
      nstr += ns; blocks.push_back(b);
    }
    std::vector<int32_t> qa, qz;
    build_xcd_queues(blocks, (size_t)nstr, nullptr, false, qa, nq);
    build_xcd_queues(blocks, (size_t)nstr, nullptr, false, qz, nq, BLK_ZLIB);
    std::vector<int> seen((size_t)nstr, 0);
    for (int pass = 0; pass < 2; pass++) {
      const std::vector<int32_t>& d = pass ? qz : qa;
      if (d[0] != 0) return fail("z: header", trial);
      for (int x = 0; x < 8; x++) {
        if (d[x + 1] < d[x]) return fail("z: offsets not monotone", trial);
        for (int i = d[x]; i < d[x + 1]; i++) {
          const int32_t t = d[9 + i];
          if (t < 0 || t >= nstr) return fail("z: stream index out of range", trial);
          if (kind[(size_t)t] != (pass ? 2 : 0)) return fail(pass ? "z: a stream that is not zlib's in the zlib queues" : "z: a Zstd / zlib stream in k_decode_streams' queues", trial);
          if ((owner[(size_t)t] % nq) != x) return fail("z: stream on the wrong XCD", trial);
          if (seen[(size_t)t]++) return fail("z: stream twice", trial);
        }
      }
    }
    for (int t = 0; t < nstr; t++) if (seen[(size_t)t] != ((kind[(size_t)t] == 0 || kind[(size_t)t] == 2) ? 1 : 0)) return fail("z: partition", trial);
  }
  // the tail property the feedback is for: with one plane 10x as expensive, no XCD's encode queue ends with it
  {
    std::vector<ChunkDesc> chunks(1); memset(&chunks[0], 0, sizeof(ChunkDesc)); chunks[0].typesize = 8; chunks[0].mode = CH_SHUFFLE | CH_FUSED_SHUF;
    std::vector<BlockDesc> blocks;
    for (int j = 0; j < 4096; j++) { BlockDesc b; memset(&b, 0, sizeof b); b.blk = j; b.first_stream = 8 * j; b.nstreams = 8; blocks.push_back(b); }
    uint32_t cost[256] = {0}; for (int k = 0; k < 8; k++) cost[k] = 100; cost[5] = 1000;
    std::vector<int32_t> q; build_encode_queues(blocks, chunks, cost, true, q);
    for (int x = 0; x < 8; x++) for (int i = q[x + 1] - 512; i < q[x + 1]; i++) if (q[9 + i] >= 0 && q[9 + i] % 8 == 5) return fail("enc: expensive plane in the tail", -1);
    std::vector<int32_t> d; build_xcd_queues(blocks, 8 * 4096, cost, true, d);
    // The decode queues' invariant (the old check only looked at the last 7 * kDecLead entries): with blocks numbered per XCD in
    // queue order, the expensive plane of block i stands in front of every cheap plane of the blocks >= i - kDecLead and behind every cheap
    // plane of the blocks < i - kDecLead - i.e. it leads its own block by exactly kDecLead blocks: never behind its block's cheap planes (the
    // kernel's tail is cheap streams), never further ahead (its bytes would leave the caches before the block completes).
    for (int x = 0; x < 8; x++) {
      std::vector<long> heavy_pos(4096 / 8, -1), cheap_lo(4096 / 8, 1L << 40), cheap_hi(4096 / 8, -1);
      for (int i = d[x]; i < d[x + 1]; i++) {
        const int sid = d[9 + i], g = sid / 8, bi = g / 8;                   // block g lives on XCD g % 8 as its bi-th block
        if (g % 8 != x) return fail("dec: stream on the wrong XCD", x);
        if (sid % 8 == 5) { if (heavy_pos[(size_t)bi] >= 0) return fail("dec: expensive plane twice", x); heavy_pos[(size_t)bi] = i; }
        else { if (i < cheap_lo[(size_t)bi]) cheap_lo[(size_t)bi] = i; if (i > cheap_hi[(size_t)bi]) cheap_hi[(size_t)bi] = i; }
      }
      const long L = (long)kDecLead, nb = 4096 / 8;
      for (long bi = 0; bi < nb; bi++) {
        if (heavy_pos[(size_t)bi] < 0 || cheap_hi[(size_t)bi] < 0) return fail("dec: a block without its planes", x);
        if (heavy_pos[(size_t)bi] > cheap_lo[(size_t)bi]) return fail("dec: expensive plane behind its own block's cheap planes", (int)bi);
        if (bi - L >= 0 && heavy_pos[(size_t)bi] > cheap_lo[(size_t)(bi - L)]) return fail("dec: expensive plane leads by less than kDecLead blocks", (int)bi);
        if (bi - L - 1 >= 0 && heavy_pos[(size_t)bi] < cheap_hi[(size_t)(bi - L - 1)]) return fail("dec: expensive plane leads by more than kDecLead blocks", (int)bi);
      }
      // the tail: behind the last expensive plane, the cheap planes of kDecLead blocks at least
      if (d[x + 1] - 1 - heavy_pos[(size_t)(nb - 1)] < 7 * L) return fail("dec: tail shorter than kDecLead blocks of cheap planes", x);
    }
  }
  printf("sched_check OK\n");
  return 0;
}
