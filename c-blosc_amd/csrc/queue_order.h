// queue_order.h — host-side queue builders of the two persistent kernels (plain C++, no HIP): which XCD serves
// which block, where a block's shuffle task sits relative to its streams, and the order that keeps the
// expensive byte planes out of the kernels' tails.  Replaces the reference's static block -> thread
// partition (blosc/blosc.c:1706-1887 t_blosc: contiguous block ranges per pthread).
// Compiled into engine.hip; tests/tools/sched_check.cpp checks the invariants with g++ on the CPU.
#pragma once
#include <atomic>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "dev_types.h"

namespace bamd {

// Per-XCD task queues of the encode kernel: out = off[9] | entries | shoff[9] | shuffle list.  Block g belongs to queue
// g % nq (nq = 8: one queue per XCD; nq = 1: the single-queue fallback of engine.hip, which also switches the fusion off); an
// entry >= 0 is a stream index, an entry < 0 "one shuffle task" (written as -(block+1) of the block it stands for).  A block's
// shuffle task is queued kEncLookahead blocks ahead of its streams: by the time a wave draws one of the streams the transpose is
// normally finished.  The shuffle list names the same blocks per XCD in the same order; shuffle tasks are CLAIMED from it through
// a counter of their own, by the waves that draw a negative entry and by waves that would otherwise wait for a block that is not
// ready (k_encode.hip): when the streams are cheap (incompressible data) the transposes then run on every waiting wave instead of
// on the few that drew the negative entries.  *sh_at = index of shoff[0] in `out`.
constexpr size_t kEncLookaheadDefault = 32;
inline size_t enc_lookahead() { return kEncLookaheadDefault; }     // (swept 1 ... 64 in round 4: 8.4 ... 8.0 ms, profiles/r04/r04q_enc_lookahead_sweep.txt - the distance hardly matters)
// BLOSC_AMD_SCHED=0: plain block order (no cost feedback)
// (read once: getenv is not safe against a setenv from another host thread.  blosc_gpu_profile(2) switches the feedback off for the calls
//  that follow - bench.py's `sched_cold` leg: one step in the order a first call on new data gets)
inline std::atomic<bool>& sched_override_off() { static std::atomic<bool> off{false}; return off; }
inline bool sched_enabled() {
  static const bool env_on = [] { const char* e = getenv("BLOSC_AMD_SCHED"); return !(e && atoi(e) == 0); }();
  return env_on && !sched_override_off().load(std::memory_order_relaxed);
}

// plane indices 0..T-1 in descending cost; *nheavy = how many of them count as expensive (> max/2)
inline void plane_order(const uint32_t* cost, bool valid, int T, std::vector<int>& order, int* nheavy) {
  order.resize((size_t)T);
  for (int j = 0; j < T; j++) order[(size_t)j] = j;
  *nheavy = T;
  if (!valid || T <= 1 || T > 256) return;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  const uint32_t mx = cost[order[0]];
  if (mx == 0) return;
  int h = 0;
  while (h < T && cost[order[(size_t)h]] > mx / 2) h++;
  *nheavy = h;
}

// With cost feedback the queue has two passes: the first, in block order, carries the shuffle tasks and the
// streams of the expensive planes; the second carries the cheap planes, plane by plane in descending cost.
// The kernel's tail (waves finishing their last stream while the queue is already empty) then consists of
// cheap streams instead of 3 ms ones.  (One pass in the decode queues' manner - expensive planes of block i + lead with the cheap
// planes of block i, lead 16 / 64 / 256 - is 4 ... 7 % slower: profiles/r04/r04zg_enc_ab_one_pass_lead_order_rejected.txt.)
inline void build_encode_queues(const std::vector<BlockDesc>& blocks, const std::vector<ChunkDesc>& chunks,
                                const uint32_t* cost, bool cost_valid, std::vector<int32_t>& out, int nq = 8, size_t* sh_at = nullptr) {
  std::vector<int32_t> q[8], sh[8];
  std::vector<uint32_t> mine[8];
  for (int x = 0; x < nq; x++) mine[x].reserve(blocks.size() / (size_t)nq + 1);
  size_t nstr_all = 0;
  for (size_t g = 0; g < blocks.size(); g++) if (blocks[g].nstreams > 0) { mine[g % (size_t)nq].push_back((uint32_t)g); nstr_all += (size_t)blocks[g].nstreams; }
  std::vector<int> order; int nheavy = 0, lastT = -1;
  for (int x = 0; x < 8; x++) {
    const std::vector<uint32_t>& B = mine[x];
    if (!B.empty()) { q[x].reserve(2 * (nstr_all / (size_t)nq) + B.size() + 64); sh[x].reserve(B.size()); }
    auto push_shuffle = [&](size_t i) {
      if (chunks[(size_t)blocks[B[i]].chunk].mode & CH_FUSED_SHUF) { q[x].push_back(-(int32_t)B[i] - 1); sh[x].push_back((int32_t)B[i]); }
    };
    int maxT = 1;
    const size_t kEncLookahead = enc_lookahead();
    for (size_t i = 0; i < B.size() && i < kEncLookahead; i++) push_shuffle(i);
    for (size_t i = 0; i < B.size(); i++) {
      if (i + kEncLookahead < B.size()) push_shuffle(i + kEncLookahead);
      const BlockDesc& b = blocks[B[i]];
      if (b.nstreams != lastT) { plane_order(cost, cost_valid && sched_enabled(), b.nstreams, order, &nheavy); lastT = b.nstreams; }
      if (b.nstreams > maxT) maxT = b.nstreams;
      for (int k = 0; k < nheavy; k++) q[x].push_back(b.first_stream + order[(size_t)k]);
    }
    // second pass: the cheap planes, most expensive first (rank k of each block's own order)
    for (int k = 1; k < maxT; k++)
      for (size_t i = 0; i < B.size(); i++) {
        const BlockDesc& b = blocks[B[i]];
        if (b.nstreams != lastT) { plane_order(cost, cost_valid && sched_enabled(), b.nstreams, order, &nheavy); lastT = b.nstreams; }
        if (k >= nheavy && k < b.nstreams) q[x].push_back(b.first_stream + order[(size_t)k]);
      }
  }
  out.clear();
  out.reserve(32 + nstr_all + 2 * blocks.size());
  out.assign(9, 0);
  for (int x = 0; x < 8; x++) { out[(size_t)x + 1] = out[(size_t)x] + (int32_t)q[x].size(); }
  for (int x = 0; x < 8; x++) out.insert(out.end(), q[x].begin(), q[x].end());
  if (out.size() == 9) out.push_back(0);
  const size_t at = out.size();
  if (sh_at) *sh_at = at;
  out.resize(at + 9, 0);
  for (int x = 0; x < 8; x++) out[at + (size_t)x + 1] = out[at + (size_t)x] + (int32_t)sh[x].size();
  for (int x = 0; x < 8; x++) out.insert(out.end(), sh[x].begin(), sh[x].end());
}

// Per-XCD stream queues of the decode kernel: out = off[9] | stream indices.  Block g belongs to XCD g & 7
// (all streams of a block on one XCD: the fused unshuffle hands over through that XCD's L2).
// With cost feedback the expensive planes of block i + kDecLead are queued together with the cheap planes of
// block i: blocks still complete in order (the unshuffles stay spread over the whole kernel), but the
// streams drawn last - the kernel's tail - are cheap ones.
// (The lead was 256 blocks until round 4: the expensive planes then sat in the scratch for 2048 blocks' worth of output before their block
//  completed and came back from HBM instead of the L2 / Infinity Cache.  64 / 16 / 2: - 9 % on reference-written config-2 chunks,
//  profiles/r04/r04zc_dec_ab_heavy_plane_lead_256_64_16_2.txt; the tail the lead is there for is as short at 16.)
#ifndef BAMD_DEC_LEAD
#define BAMD_DEC_LEAD 16
#endif
constexpr size_t kDecLead = BAMD_DEC_LEAD;
// pick = 0: the blocks of k_decode_streams (neither k_decode_blocks' nor the entropy-coded formats'); pick = BLK_ZLIB: the zlib kernel's.
inline void build_xcd_queues(const std::vector<BlockDesc>& blocks, size_t nstr, const uint32_t* cost, bool cost_valid,
                             std::vector<int32_t>& out, int nq = 8, uint32_t pick = 0u) {
  std::vector<int32_t> q[8];
  std::vector<uint32_t> mine[8];
  for (int x = 0; x < nq; x++) { mine[x].reserve(blocks.size() / (size_t)nq + 1); q[x].reserve(nstr / (size_t)nq + 64); }
  for (size_t g = 0; g < blocks.size(); g++) {
    const uint32_t f = (uint32_t)blocks[g].flags;
    if (blocks[g].nstreams > 0 && (pick ? (f & pick) != 0u : !(f & BLK_Z))) mine[g % (size_t)nq].push_back((uint32_t)g);
  }
  std::vector<int> order; int nheavy = 0, lastT = -1;
  auto prep = [&](const BlockDesc& b) {
    if (b.nstreams != lastT) { plane_order(cost, cost_valid && sched_enabled(), b.nstreams, order, &nheavy); lastT = b.nstreams; }
  };
  for (int x = 0; x < 8; x++) {
    const std::vector<uint32_t>& B = mine[x];
    const size_t lead = B.size() > 2 * kDecLead ? kDecLead : B.size() / 4;
    auto push_heavy = [&](size_t i) { const BlockDesc& b = blocks[B[i]]; prep(b); if (nheavy < b.nstreams) for (int k = 0; k < nheavy; k++) q[x].push_back(b.first_stream + order[(size_t)k]); };
    for (size_t i = 0; i < B.size() && i < lead; i++) push_heavy(i);
    for (size_t i = 0; i < B.size(); i++) {
      if (i + lead < B.size()) push_heavy(i + lead);
      const BlockDesc& b = blocks[B[i]];
      prep(b);
      if (nheavy < b.nstreams) { for (int k = nheavy; k < b.nstreams; k++) q[x].push_back(b.first_stream + order[(size_t)k]); }
      else for (int k = 0; k < b.nstreams; k++) q[x].push_back(b.first_stream + k);   // no feedback: plain order
    }
  }
  out.clear();
  out.reserve(10 + nstr);
  out.assign(9, 0);
  for (int x = 0; x < 8; x++) out[(size_t)x + 1] = out[(size_t)x] + (int32_t)q[x].size();
  for (int x = 0; x < 8; x++) out.insert(out.end(), q[x].begin(), q[x].end());
  out.resize(9 + (nstr ? nstr : 1), 0);
}

}  // namespace bamd
