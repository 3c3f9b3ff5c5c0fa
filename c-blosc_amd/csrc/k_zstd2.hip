// k_zstd2.hip — two-phase Zstd decode for the common frame shape (one compressed block per frame: what blosc writes
// for blocks of at most 128 KiB).
//
// Why: with one wavefront per frame (k_zstd.hip) the entropy decoding - 4000 dependent FSE steps per frame on bench19 -
// is a scalar program, and ALL waves of a CU share its one scalar unit (or, as vector code, its four SIMDs): ~120
// instructions per sequence x 12 waves = 1500 cycles per sequence and wave (profiles/r02/r02_e_zstd_decode_phases.txt).
// The instruction count per sequence is what it is; the way to use the machine is to let the LANES of a wave work on
// DIFFERENT frames.
//
//   phase A  k_zstd_entropy: 16 frames per wavefront, 4 lanes per frame.  Lane 0 of a group runs the serial pieces of
//            zstd_serial.h (headers, table builds, the FSE sequence stream) with the frame's tables in the group's own
//            LDS slice; the four Huffman streams of the literals run on the group's four lanes.  Output: the literals
//            (scratch), the sequences as packed (literal length, match length, offset) triples, and a small record.
//   phase B  k_zstd_exec: one wavefront per frame, no tables, no LDS: 64 triples per load, executed 16 at a time with
//            the wave-cooperative copies of k_zstd.hip (zstd_exec16).
// Frames of any other shape (several blocks, raw / RLE blocks, treeless literals, more sequences than the triple
// scratch holds) are left to k_zstd_streams, which skips the frames phase A has taken.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_types.h"
#include "wave_prims.h"
#include "zstd_serial.h"

#ifndef BAMD_ZSTD_LDS_FSE
#define BAMD_ZSTD_LDS_FSE 0      // the three sequence tables of a frame compact in LDS (40 KiB per wave: 4 waves per CU).  Measured (8 GiB of reference-written frames, k_zstd_entropy): bench19 27.1 ms with, 28.8 without; linspace 14.5 with, 6.1 without - the lookups are not what the kernel waits for, the occupancy is what it needs.  Off.
#endif
#ifndef BAMD_ZSTD_SEQ_KERNEL
#define BAMD_ZSTD_SEQ_KERNEL 1    // the FSE sequence streams of the global two-phase path in a kernel of their own, one lane per frame (k_zstd_seq); 0: inside k_zstd_entropy, on one lane in four
#endif
namespace bamd {

enum : uint32_t { ZM_FALLBACK = 0, ZM_READY = 1, ZM_ERROR = 2, ZM_SEQ = 3 };      // ZM_SEQ: everything but the sequence stream is done (k_zstd_seq turns it into READY or ERROR)
struct ZMeta {
  uint32_t state;
  uint32_t nseq, regen;
  uint32_t lit_mode;     // 0: decoded into the literal scratch, 1: raw at in + lit_src, 2: RLE byte lit_src
  uint32_t lit_src;
  uint32_t pad_[3];      // ZM_SEQ: offset of the sequence bit stream in the frame, its length, the three accuracy logs
};

// LDS of one frame group: the Huffman table and the three FSE tables are never needed at the same time
struct ZgLds {
  union { uint16_t huf[2048]; uint32_t fse[3][512]; } t;
  uint32_t ftab[64];
  uint16_t next[3][256];     // one table-build scratch per table: the three FSE tables are built by three lanes at once
  int16_t norm[3][64];
  uint8_t w[256];
};
constexpr int ZG_FRAMES = 16;        // frames per wavefront (4 lanes each)
// The three sequence tables of a frame as 16-bit cells (zd::fse_build's `compact`), one dense record per frame: what k_zstd_seq reads.
// 2.5 KiB per frame = 160 MiB for the 65 536 frames of the benchmark batch, all of them live at once: inside the 256 MiB Infinity
// Cache, where the 32-bit tables inside the 8.4 KiB ZgLds records (550 MiB) were not - the sequence loop is bound by random
// 64-byte sector reads (41 GB per launch for 0.33 GB of input, profiles/r03/r03_traffic_cfg4.json), not by its instructions
// (one lane per frame instead of one in four: 18 -> 16.3 ms, profiles/r03/r03v_zent_split.txt).
struct ZcTab { uint16_t ll[512], of[256], ml[512]; };

__device__ __forceinline__ uint64_t zpack(uint32_t ll, uint32_t ml, uint32_t off) { return (uint64_t)ll | ((uint64_t)ml << 18) | ((uint64_t)off << 36); }

// where the triples of a stream live: same offset as its literal scratch, in the arena `zseq_delta` bytes further on
__device__ __forceinline__ uint64_t* zseq_ptr(const uint8_t* lit, ptrdiff_t zseq_delta) {
  return (uint64_t*)(((uintptr_t)lit + (uintptr_t)zseq_delta) & ~(uintptr_t)7);
}

// The FSE sequence stream of one frame on ONE LANE: zd::seq_next on plain locals (same order of reads, same checks - zstd_serial.h
// is the specification and what the CPU tests pin to the reference); symbol codes above the format's limits cannot occur: a
// 6-bit symbol of a table whose description was accepted is at most 35 / 31 / 52 by fse_read_ncount's max_sym.
// Everything the loop carries is a plain local: SeqState's repeat offsets are an array indexed by a decoded value, which put the
// WHOLE state - bit reader included - on the stack (every field access a scratch load or store, every refill a flat load with a
// vmcnt(0) + lgkmcnt(0) behind it: the ISA of round 2's loop).  The bit stream is read through a global pointer, four bytes per
// refill, the NEXT refill's word already on its way.  gl / go / gm: the frame's three tables (32-bit cells, global scratch) or,
// with BAMD_ZSTD_LDS_FSE, tl / to / tm: their compact LDS copies.
template <int GC = 0>            // GC 1: tl / to / tm are 16-bit cells in GLOBAL memory (ZcTab); GC 2: the same in LDS (k_zstd_seq_lds)
__device__ __forceinline__ bool zseq_run(zd::SeqState& st, int nseq, const uint32_t* gl_, const uint32_t* go_, const uint32_t* gm_,
                                         const uint16_t* tl, const uint16_t* to, const uint16_t* tm, int al_l, int al_o, int al_m, uint64_t* sq_,
                                         volatile uint64_t* sqbuf_ = nullptr, int lane = 0, const volatile uint32_t* codes_ = nullptr) {
  // codes (LDS, 36 + 53 words: base | extra bits << 24 of the literal-length and match-length codes, zseq_fill_codes): one read instead of
  // the comparison chains of zd::ll_base / ll_bits / ml_base / ml_bits - the loop is a chain of dependent instructions, every one counts
  const volatile __attribute__((address_space(3))) uint32_t* codes = (const volatile __attribute__((address_space(3))) uint32_t*)codes_;
  // sqbuf (LDS, 8 x 64 words, entry k of lane l at [k * 64 + l]): the triples leave in runs of eight.  gfx950 counts loads and stores
  // with ONE in-order counter: a store per sequence meant that every wait for a loaded value - the next stream word, a table cell -
  // was also a wait for the previous sequence's store to be acknowledged by L2 (~0.9 us per sequence even with the tables in LDS,
  // profiles/r03/r03x_zent_split.txt); with the run buffer that wait comes once per eight sequences, and a lane writes 64 contiguous bytes.
  volatile __attribute__((address_space(3))) uint64_t* sqbuf = (volatile __attribute__((address_space(3))) uint64_t*)sqbuf_;
  bool fine = true;
  const BAMD_GAS uint32_t* gl = (const BAMD_GAS uint32_t*)gl_; const BAMD_GAS uint32_t* go = (const BAMD_GAS uint32_t*)go_;      // (global_load, not flat_load)
  const BAMD_GAS uint32_t* gm = (const BAMD_GAS uint32_t*)gm_; BAMD_GAS uint64_t* sq = (BAMD_GAS uint64_t*)sq_;
  const gu8* bp = as_global(st.b.p);
  int bytepos = st.b.bytepos, nacc = st.b.nacc, off = st.b.off;
  uint64_t acc = st.b.acc;          // the low nacc bits are unread stream bits (what lies above them is stale: every extraction masks)
  uint32_t sl = st.sl, so = st.so, sm = st.sm, r0 = 1u, r1 = 4u, r2 = 8u;
  uint32_t nxt = bytepos >= 4 ? g_ld4(bp + bytepos - 4) : 0u;          // the word the next refill will take (valid while bytepos >= 4)
  auto rd = [&](int nb_) -> uint32_t {                                   // zd::back_read on the locals: any state of the stream, its last bytes included
    if (nb_ == 0) return 0u;
    if (nacc < nb_ && bytepos >= 4) {
      bytepos -= 4; acc = (acc << 32) | nxt; nacc += 32;
      if (bytepos >= 4) nxt = g_ld4(bp + bytepos - 4);
    }
    while (nacc < nb_ && bytepos > 0) { bytepos--; acc = (acc << 8) | bp[bytepos]; nacc += 8; }
    if (nacc < nb_) { acc <<= (nb_ - nacc); nacc = nb_; }
    const uint32_t v = (uint32_t)(acc >> (nacc - nb_)) & (nb_ >= 32 ? 0xffffffffu : ((1u << nb_) - 1u));
    nacc -= nb_;
    acc &= (nacc ? ((1ull << nacc) - 1ull) : 0ull);
    off -= nb_;
    return v;
  };
  // The same reads while at least 12 bytes of the stream are left - i.e. for all but the last two or three sequences of a frame: three
  // refill points per sequence instead of a test in front of each of the six fields.  A refill leaves >= 32 unread bits; the fields
  // between two refill points need at most 31 (offset bits), 16 + 16 (match and literal length bits) and 9 + 9 + 8 (the three
  // state updates).  rd() above was ~130 instructions per field with its byte loop and its end-of-stream cases, six times per
  // sequence, and the loop as a whole ~800 (profiles/r03/r03u_zent_split.txt: 18 of the entropy kernel's 27 ms).
  auto refill = [&]() {
    if (nacc <= 32) { bytepos -= 4; acc = (acc << 32) | nxt; nacc += 32; if (bytepos >= 4) nxt = g_ld4(bp + bytepos - 4); }
  };
  auto take = [&](int nb_) -> uint32_t {                                 // nb_ <= 31 and nacc >= nb_ (nacc - nb_ = 64 only with nb_ = 0: masked to nothing)
    const uint32_t v = (uint32_t)(acc >> ((nacc - nb_) & 63)) & ((1u << nb_) - 1u);
    nacc -= nb_; off -= nb_;
    return v;
  };
  for (int i = 0; fine && i < nseq; i++) {
    uint32_t cl, co, cm;
    if (GC == 2) {
      typedef const volatile __attribute__((address_space(3))) uint16_t* lp16;
      cl = ((lp16)tl)[sl]; co = ((lp16)to)[so]; cm = ((lp16)tm)[sm];
    } else if (GC == 1) { cl = ((const BAMD_GAS uint16_t*)tl)[sl]; co = ((const BAMD_GAS uint16_t*)to)[so]; cm = ((const BAMD_GAS uint16_t*)tm)[sm]; }
    else if (BAMD_ZSTD_LDS_FSE) { cl = tl[sl]; co = to[so]; cm = tm[sm]; }
    else {                                                                // the same compact form out of the 32-bit cells
      const uint32_t el = gl[sl], eo = go[so], em = gm[sm];
      cl = (el & 63u) | ((((el >> 16) + (1u << al_l)) >> ((el >> 8) & 0xffu)) << 6);
      co = (eo & 63u) | ((((eo >> 16) + (1u << al_o)) >> ((eo >> 8) & 0xffu)) << 6);
      cm = (em & 63u) | ((((em >> 16) + (1u << al_m)) >> ((em >> 8) & 0xffu)) << 6);
    }
    const int lc = (int)(cl & 63u), oc = (int)(co & 63u), mc = (int)(cm & 63u);
    if (oc > 31 || mc > 52 || lc > 35) { fine = false; break; }
    const uint32_t xl = cl >> 6, xm = cm >> 6, xo = co >> 6;
    const int nl = al_l - zd::hb32(xl), nm = al_m - zd::hb32(xm), no = al_o - zd::hb32(xo);
    uint32_t ov, q_ml, q_ll;
    if (bytepos >= 12) {
      refill(); ov = (1u << oc) + take(oc);
      refill();
      if (codes_) {
        const uint32_t pm = codes[36 + mc], pl = codes[lc];
        q_ml = (pm & 0xffffffu) + take((int)(pm >> 24)); q_ll = (pl & 0xffffffu) + take((int)(pl >> 24));
      } else { q_ml = zd::ml_base(mc) + take(zd::ml_bits(mc)); q_ll = zd::ll_base(lc) + take(zd::ll_bits(lc)); }
      if (i + 1 != nseq) {
        refill();
        sl = ((xl << nl) - (1u << al_l)) + take(nl);
        sm = ((xm << nm) - (1u << al_m)) + take(nm);
        so = ((xo << no) - (1u << al_o)) + take(no);
      }
    } else {
      ov = (1u << oc) + rd(oc);
      q_ml = zd::ml_base(mc) + rd(zd::ml_bits(mc));
      q_ll = zd::ll_base(lc) + rd(zd::ll_bits(lc));
      if (i + 1 != nseq) {
        sl = ((xl << nl) - (1u << al_l)) + rd(nl);
        sm = ((xm << nm) - (1u << al_m)) + rd(nm);
        so = ((xo << no) - (1u << al_o)) + rd(no);
      }
    }
    if (off < 0) { fine = false; break; }
    uint32_t q_off;
    if (ov > 3) { q_off = ov - 3u; r2 = r1; r1 = r0; r0 = q_off; }
    else {
      uint32_t idx = ov - 1u;
      if (q_ll == 0) idx++;
      if (idx == 0) q_off = r0;
      else {
        q_off = idx == 1u ? r1 : (idx == 2u ? r2 : r0 - 1u);
        if (q_off == 0) { fine = false; break; }
        if (idx > 1) r2 = r1;
        r1 = r0; r0 = q_off;
      }
    }
    if (!sqbuf_) sq[i] = zpack(q_ll, q_ml, q_off);
    else {
      sqbuf[(uint32_t)(i & 7) * 64u + (uint32_t)lane] = zpack(q_ll, q_ml, q_off);
      if ((i & 7) == 7) {
        uint64_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = sqbuf[(uint32_t)k * 64u + (uint32_t)lane];
        BAMD_GAS uint8_t* o = (BAMD_GAS uint8_t*)(sq + (i - 7));
#pragma unroll
        for (int k = 0; k < 4; k++) g_st16(o + 16 * k, make_uint4((uint32_t)v[2 * k], (uint32_t)(v[2 * k] >> 32), (uint32_t)v[2 * k + 1], (uint32_t)(v[2 * k + 1] >> 32)));
      }
    }
  }
  if (sqbuf_ && fine) {                                    // the last, incomplete run
    const int done = nseq;
    for (int i = done & ~7; i < done; i++) sq[i] = sqbuf[(uint32_t)(i & 7) * 64u + (uint32_t)lane];
  }
  st.b.off = off;
  return fine;
}

// One Huffman literal stream on one lane: zd::huf_decode_stream (zstd_serial.h, the CPU-checked statement of the rules: exactly n
// symbols, the stream consumed exactly) in the form the device needs - the bit reader of zseq_run above (one refill test per
// symbol while four stream bytes are left, the general reader for the last ones), the table (GLOBAL = true: 4 KiB of global
// scratch per frame) read through a global pointer, and the output leaving EIGHT bytes at a time: one byte store per symbol put
// L2's acknowledgement of the previous store in front of every table read (the one in-order counter again; 7 us per symbol,
// profiles/r03/r03u_zent_split.txt: 4.5 of the kernel's 27 ms for 2 400 literals per frame).
template <bool GLOBAL>
__device__ __forceinline__ bool zhuf_run(const uint16_t* te_, int mb, const uint8_t* src, int len, uint8_t* out_, int n) {
  zd::Back b0;
  if (!zd::back_init(b0, src, len)) return false;
  const gu8* bp = as_global(src); gu8* out = as_global(out_);
  const BAMD_GAS uint16_t* tg = (const BAMD_GAS uint16_t*)te_;
  int bytepos = b0.bytepos, nacc = b0.nacc, off = b0.off;
  uint64_t acc = b0.acc;
  uint32_t nxt = bytepos >= 4 ? g_ld4(bp + bytepos - 4) : 0u;
  auto rd = [&](int nb_) -> uint32_t {                                   // zd::back_read on the locals (see zseq_run)
    if (nb_ == 0) return 0u;
    if (nacc < nb_ && bytepos >= 4) {
      bytepos -= 4; acc = (acc << 32) | nxt; nacc += 32;
      if (bytepos >= 4) nxt = g_ld4(bp + bytepos - 4);
    }
    while (nacc < nb_ && bytepos > 0) { bytepos--; acc = (acc << 8) | bp[bytepos]; nacc += 8; }
    if (nacc < nb_) { acc <<= (nb_ - nacc); nacc = nb_; }
    const uint32_t v = (uint32_t)(acc >> (nacc - nb_)) & (nb_ >= 32 ? 0xffffffffu : ((1u << nb_) - 1u));
    nacc -= nb_;
    acc &= (nacc ? ((1ull << nacc) - 1ull) : 0ull);
    off -= nb_;
    return v;
  };
  const uint32_t mask = (1u << mb) - 1u;
  uint32_t state = rd(mb);
  uint64_t pend = 0;
  int i = 0;
  for (; i < n && off > -mb; i++) {
    const uint32_t e = GLOBAL ? (uint32_t)tg[state] : (uint32_t)te_[state];
    pend |= (uint64_t)(e & 0xffu) << (8 * (i & 7));
    if ((i & 7) == 7) { g_st8(out + (i - 7), pend); pend = 0; }
    const int nb = (int)(e >> 8);
    uint32_t v;
    if (bytepos >= 4) {                                                  // nb <= 11 <= the 32 bits a refill leaves
      if (nacc <= 32) { bytepos -= 4; acc = (acc << 32) | nxt; nacc += 32; if (bytepos >= 4) nxt = g_ld4(bp + bytepos - 4); }
      v = (uint32_t)(acc >> ((nacc - nb) & 63)) & ((1u << nb) - 1u);
      nacc -= nb; off -= nb;
    } else v = rd(nb);
    state = ((state << nb) & mask) | v;
  }
  for (int k = i & ~7; k < i; k++) out[k] = (uint8_t)(pend >> (8 * (k & 7)));      // the last, incomplete run
  return i == n && off == -mb;
}

// GLOBAL = false: the tables of the 16 frames in LDS (128 KiB: ONE wave per CU).  GLOBAL = true: the same structure per frame
// in a global scratch (gscr[sid]): every table access becomes a trip to L2 / HBM, but nothing limits the number of waves
// per CU any more - 65 536 frames are 4096 waves, all of them resident at once (BLOSC_AMD_ZSTD2=2).
template <bool GLOBAL>
__global__ __launch_bounds__(64) void k_zstd_entropy_t(const StreamDesc* __restrict__ streams, int nstreams, const ChunkDesc* __restrict__ chunks,
                                                       const BlockDesc* __restrict__ blocks, ZMeta* __restrict__ meta, ptrdiff_t zseq_delta, ZgLds* __restrict__ gscr,
                                                       ZcTab* __restrict__ ctab) {
  __shared__ uint64_t lds_raw[GLOBAL ? 1 : (sizeof(ZgLds) * ZG_FRAMES + 7) / 8];
  ZgLds* lds = (ZgLds*)lds_raw;
  // Round 3 (GLOBAL only): the three SEQUENCE tables of every frame of the wave, compact, in LDS.  With all tables in the global
  // scratch every FSE step was three dependent 4-byte reads out of an 8.4 KiB per-frame structure: 31 GB fetched to decode 0.3 GB
  // (profiles/r02/r02g_traffic_cfg4.json), ~7 us per sequence.  A cell is kept as symbol | x << 6 (x = the cell's "next state"
  // counter, < 1024): 16 bits, from which nbBits = al - floor(log2 x) and baseline = (x << nbBits) - 2^al follow - so the
  // largest tables the format allows (LL 512 + OF 256 + ML 512 cells) take 2.5 KiB per frame, 40 KiB per wave, 4 waves per CU.
  // The Huffman table (4 KiB, used once per literal) and the build scratch stay global.
  __shared__ uint16_t lfse[(GLOBAL && BAMD_ZSTD_LDS_FSE) ? ZG_FRAMES : 1][(GLOBAL && BAMD_ZSTD_LDS_FSE) ? 1280 : 4];
  const int lane = threadIdx.x & 63, g = lane >> 2, sub = lane & 3;
  const int sid = (int)blockIdx.x * ZG_FRAMES + g;
  ZgLds* L = GLOBAL ? gscr + (sid < nstreams ? sid : 0) : &lds[GLOBAL ? 0 : g];
  // ---- lane 0 of the group: everything up to the literal streams ----
  uint32_t state = ZM_FALLBACK;
  bool take = false;
  const uint8_t* in = nullptr; int n = 0, want = 0;
  uint8_t* lit = nullptr;
  const uint8_t* b = nullptr; int size = 0, p = 0;
  zd::LitHdr lh = {0, 0, 0, 1, 0};
  zd::Huf huf = {L->t.huf, 0};
  uint32_t lit_mode = 0, lit_src = 0;
  int hs_off = 0, hlen = 0;                 // Huffman payload (behind the table description)
  if (sid < nstreams) {
    const StreamDesc& sd = streams[sid];
    in = (const uint8_t*)as_global(sd.in); n = sd.in_size; want = sd.out_size;      // (through the global address space: what is inlined below - zstd_serial.h's header, table and Huffman-table code - then uses global_load, not flat_load: 58 flat instructions -> 0)
    take = sd.fmt == FMT_ZSTD && n >= 0 && n != want;
    if (take) {
      const ChunkDesc& c = chunks[sd.chunk];
      const BlockDesc& bk = blocks[sd.aux];
      lit = c.stage + (size_t)bk.blk * (size_t)c.blocksize + (size_t)(sid - bk.first_stream) * (size_t)want;
    }
  }
  if (take && sub == 0) {
    long long fcs = -1; bool checksum = false;
    const int ip = zd::frame_header(in, n, &fcs, &checksum);
    if (ip < 0 || (fcs >= 0 && fcs != (long long)want)) state = ZM_ERROR;
    else if (ip + 3 <= n) {
      const uint32_t bh = (uint32_t)in[ip] | ((uint32_t)in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16);
      const int last = bh & 1u, type = (bh >> 1) & 3u, bsize = (int)(bh >> 3);
      // exactly one compressed block that fills the frame: everything else goes the general way
      if (last && type == 2 && bsize <= (1 << 17) && !checksum && ip + 3 + bsize == n) {      // (frames with a content checksum: general path)
        b = in + ip + 3; size = bsize;
        if (!zd::lit_header(b, size, lh) || lh.regen > want) state = ZM_ERROR;
        else {
          p = lh.hdr;
          state = ZM_READY;
          if (lh.type == 0) {
            if (p + lh.regen > size) state = ZM_ERROR;
            lit_mode = 1; lit_src = (uint32_t)(ip + 3 + p); p += lh.regen;
          } else if (lh.type == 1) {
            if (p + 1 > size) state = ZM_ERROR; else { lit_mode = 2; lit_src = b[p]; p += 1; }
          } else if (lh.type == 2) {
            if (p + lh.csize > size) state = ZM_ERROR;
            else {
              const int used = zd::huf_read_table(huf, b + p, lh.csize, L->w, L->ftab, L->next[0], L->norm[0]);
              if (used < 0) state = ZM_ERROR; else { hs_off = p + used; hlen = lh.csize - used; }
            }
          } else state = ZM_FALLBACK;          // treeless literals need the previous block's table
        }
      }
    }
  }
  // ---- the group's four lanes: Huffman streams ----
  const int src0 = (lane & ~3) << 2;          // ds_bpermute address of the group's lane 0
  const uint32_t gstate = (uint32_t)__builtin_amdgcn_ds_bpermute(src0, (int)state);
  const int gtype = __builtin_amdgcn_ds_bpermute(src0, lh.type);
  uint32_t lit_ok = 1;
  if (take && gstate == ZM_READY && gtype == 2) {
    const int g_regen = __builtin_amdgcn_ds_bpermute(src0, lh.regen), g_nstreams = __builtin_amdgcn_ds_bpermute(src0, lh.nstreams);
    const int g_hs = __builtin_amdgcn_ds_bpermute(src0, hs_off), g_hlen = __builtin_amdgcn_ds_bpermute(src0, hlen);
    const int g_mb = __builtin_amdgcn_ds_bpermute(src0, huf.maxbits);
    const int g_boff = __builtin_amdgcn_ds_bpermute(src0, (int)(b - in));
    const uint8_t* hs = in + g_boff + g_hs;
    zd::Huf h2 = {L->t.huf, g_mb};
    if (g_nstreams == 1) {
      if (sub == 0) lit_ok = zhuf_run<GLOBAL>(h2.e, h2.maxbits, hs, g_hlen, lit, g_regen) ? 1u : 0u;
    } else if (g_hlen < 6) lit_ok = 0;
    else {
      const int s1 = hs[0] | (hs[1] << 8), s2 = hs[2] | (hs[3] << 8), s3 = hs[4] | (hs[5] << 8), s4 = g_hlen - 6 - s1 - s2 - s3;
      const int q = (g_regen + 3) / 4;
      if (s4 < 1 || 3 * q > g_regen) lit_ok = 0;
      else {
        const int so = sub == 0 ? 0 : (sub == 1 ? s1 : (sub == 2 ? s1 + s2 : s1 + s2 + s3));
        const int sl = sub == 0 ? s1 : (sub == 1 ? s2 : (sub == 2 ? s3 : s4));
        const int cnt = sub < 3 ? q : g_regen - 3 * q;
        lit_ok = zhuf_run<GLOBAL>(h2.e, h2.maxbits, hs + 6 + so, sl, lit + sub * q, cnt) ? 1u : 0u;
      }
    }
  }
  // all four verdicts to lane 0 of the group
  const uint32_t v1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src0 + 4, (int)lit_ok), v2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src0 + 8, (int)lit_ok);
  const uint32_t v3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src0 + 12, (int)lit_ok);
  // ---- lane 0 again: sequence count and the three table descriptions ----
  int nseq = 0;
  int t_nsym[3] = {0, 0, 0}, t_al[3] = {0, 0, 0}, t_kind[3] = {-1, -1, -1};    // kind: 0 / 2 build from norm, 1 RLE (done), -1 nothing to build
  if (take && sub == 0 && state == ZM_READY) {
    if (!(lit_ok & v1 & v2 & v3)) state = ZM_ERROR;
    if (lh.type == 2) p += lh.csize;
    if (state == ZM_READY) {
      const int u0 = zd::seq_count(b + p, size - p, &nseq);
      if (u0 < 0) state = ZM_ERROR; else p += u0;
    }
    if (state == ZM_READY && nseq > want / 8) state = ZM_FALLBACK;       // the triple scratch of this stream is `want` bytes
    if (state == ZM_READY && nseq > 0) {
      bool fine = p < size;
      int modes = 0;
      if (fine) { modes = b[p++]; fine = (modes & 3) == 0; }
      for (int k = 0; fine && k < 3; k++) {                               // table order in the stream: LL, OF, ML
        const int mode = (modes >> (6 - 2 * k)) & 3;
        const int max_sym = k == 0 ? 35 : (k == 1 ? 31 : 52), max_al = k == 1 ? 8 : 9;
        if (mode == 0) {
          t_nsym[k] = k == 0 ? 36 : (k == 1 ? 29 : 53); t_al[k] = k == 1 ? 5 : 6; t_kind[k] = 0;
          for (int s_ = 0; s_ < t_nsym[k]; s_++) L->norm[k][s_] = k == 0 ? zd::ll_default(s_) : (k == 1 ? zd::of_default(s_) : zd::ml_default(s_));
        } else if (mode == 1) {
          if (size - p < 1 || b[p] > max_sym) fine = false;
          else {
            L->t.fse[k][0] = (uint32_t)b[p]; t_al[k] = 0; t_kind[k] = 1;
            if (ctab) (k == 0 ? ctab[sid].ll : (k == 1 ? ctab[sid].of : ctab[sid].ml))[0] = (uint16_t)(b[p] | (1u << 6));      // one cell: x = 1
            p += 1;
          }
        } else if (mode == 2) {
          const int h = zd::fse_read_ncount(b + p, size - p, max_al, max_sym, L->norm[k], &t_nsym[k], &t_al[k]);
          if (h < 0) fine = false; else { t_kind[k] = 2; p += h; }
        } else fine = false;                                               // repeat mode: there is no previous block in these frames
      }
      if (!fine) state = ZM_ERROR;
    } else if (state == ZM_READY && p != size) state = ZM_ERROR;            // no sequences: nothing may follow the count byte
  }
  // ---- lanes 0 / 1 / 2 of the group build the LL / OF / ML tables side by side ----
  uint32_t built = 1;
  {
    const uint32_t gs = (uint32_t)__builtin_amdgcn_ds_bpermute(src0, (int)state);
    const int gn = __builtin_amdgcn_ds_bpermute(src0, nseq);
    int kd = -1, ns = 0, al = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int a = __builtin_amdgcn_ds_bpermute(src0, t_kind[k]), bb = __builtin_amdgcn_ds_bpermute(src0, t_nsym[k]), c2 = __builtin_amdgcn_ds_bpermute(src0, t_al[k]);
      if (sub == k) { kd = a; ns = bb; al = c2; }
    }
    if (take && gs == ZM_READY && gn > 0 && sub < 3 && (kd == 0 || kd == 2)) {
      zd::Fse ft = {L->t.fse[sub], 0};
      uint16_t* cc = ctab ? (sub == 0 ? ctab[sid].ll : (sub == 1 ? ctab[sid].of : ctab[sid].ml)) : nullptr;
      built = zd::fse_build(ft, L->norm[sub], ns, al, L->next[sub], cc) ? 1u : 0u;
    }
    if (GLOBAL && BAMD_ZSTD_LDS_FSE && take && gs == ZM_READY && gn > 0 && sub < 3 && kd >= 0 && built) {      // the lane that built a table packs it into LDS (RLE: one cell)
      const int size = 1 << al, o = sub == 0 ? 0 : (sub == 1 ? 512 : 768);                    // LL | OF | ML
      for (int i = 0; i < size; i++) {
        const uint32_t e = L->t.fse[sub][i];
        lfse[g][o + i] = (uint16_t)((e & 63u) | ((((e >> 16) + (uint32_t)size) >> ((e >> 8) & 0xffu)) << 6));
      }
    }
  }
  const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src0 + 4, (int)built), b2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src0 + 8, (int)built);
  // (lane 0 of a group reads below what lanes 1 and 2 packed into LDS above: lock step on the device; on the wavefront emulator the
  //  two cross-lane reads of `built` just above are the rendezvous - lane 0 cannot get past them before lanes 1 and 2 arrive there)
  // ---- lane 0: the FSE sequence stream ----
  uint32_t pend_off = 0, pend_len = 0, pend_al = 0;
  if (take && sub == 0) {
    if (state == ZM_READY && nseq > 0) {
      bool fine = (built & b1 & b2) != 0u;
      uint64_t* sq = zseq_ptr(lit, zseq_delta);
      zd::SeqTabs tb = {{L->t.fse[0], t_al[0]}, {L->t.fse[1], t_al[1]}, {L->t.fse[2], t_al[2]}, true, true, true};
      zd::SeqState st;
      st.rep[0] = 1u; st.rep[1] = 4u; st.rep[2] = 8u;
      bool seq_later = false; uint32_t seq_off = 0, seq_len = 0, seq_al = 0;
      if (fine) fine = size - p >= 1;
      if (fine && !(GLOBAL && !BAMD_ZSTD_LDS_FSE && BAMD_ZSTD_SEQ_KERNEL)) fine = zd::seq_begin(st, tb, b + p, size - p);
      if (GLOBAL && !BAMD_ZSTD_LDS_FSE && BAMD_ZSTD_SEQ_KERNEL) {
        // the sequence stream is k_zstd_seq's (below: one LANE per frame instead of one lane in four); what it needs travels in the record
        if (fine) { seq_off = (uint32_t)((b + p) - in); seq_len = (uint32_t)(size - p); seq_al = (uint32_t)t_al[0] | ((uint32_t)t_al[1] << 8) | ((uint32_t)t_al[2] << 16); seq_later = true; }
      } else if (GLOBAL) {
        const uint16_t* tl = &lfse[BAMD_ZSTD_LDS_FSE ? g : 0][0]; const uint16_t* to = &lfse[BAMD_ZSTD_LDS_FSE ? g : 0][BAMD_ZSTD_LDS_FSE ? 512 : 0]; const uint16_t* tm = &lfse[BAMD_ZSTD_LDS_FSE ? g : 0][BAMD_ZSTD_LDS_FSE ? 768 : 0];
        if (fine) fine = zseq_run(st, nseq, L->t.fse[0], L->t.fse[1], L->t.fse[2], tl, to, tm, t_al[0], t_al[1], t_al[2], sq);
      } else
      for (int i = 0; fine && i < nseq; i++) {
        zd::Seq q;
        if (!zd::seq_next(st, tb, i + 1 == nseq, q)) { fine = false; break; }
        sq[i] = zpack(q.ll, q.ml, q.off);
      }
      if (fine && !seq_later && st.b.off != 0) fine = false;                 // the bit stream must be consumed exactly
      if (!fine) state = ZM_ERROR;
      else if (seq_later) { pend_off = seq_off; pend_len = seq_len; pend_al = seq_al; state = ZM_SEQ; }
    }
    ZMeta m;
    m.state = state; m.nseq = (uint32_t)nseq; m.regen = (uint32_t)lh.regen; m.lit_mode = lit_mode; m.lit_src = lit_src;
    m.pad_[0] = pend_off; m.pad_[1] = pend_len; m.pad_[2] = pend_al;
    meta[sid] = m;
  } else if (sid < nstreams && sub == 0) {
    ZMeta m;
    m.state = ZM_FALLBACK; m.nseq = 0; m.regen = 0; m.lit_mode = 0; m.lit_src = 0; m.pad_[0] = m.pad_[1] = m.pad_[2] = 0;
    meta[sid] = m;
  }
}

// phase A2 (round 3): the sequence streams, ONE LANE PER FRAME.  Inside k_zstd_entropy the stream of a frame ran on lane 0 of the frame's four
// lanes (the other three are there for the four Huffman streams): 16 of 64 lanes busy for two thirds of that kernel's time
// (profiles/r03/r03u_zent_split.txt: 27.3 ms, 9.2 without the sequence loop, 4.2 without the Huffman streams as well).  Here a wave
// carries 64 frames through the same loop (zseq_run), the tables where phase A built them (global scratch).
__device__ __forceinline__ void zseq_fill_codes(volatile uint32_t* codes, int lane) {      // before any lane leaves: the whole wave fills, then a rendezvous
  if (lane < 36) codes[lane] = zd::ll_base(lane) | ((uint32_t)zd::ll_bits(lane) << 24);
  if (lane < 53) codes[36 + lane] = zd::ml_base(lane) | ((uint32_t)zd::ml_bits(lane) << 24);
  BAMD_LDS_SYNC();
}
#ifndef BAMD_ZSEQ_FRAMES
#define BAMD_ZSEQ_FRAMES 64       // frames per wavefront of k_zstd_seq (lanes above that idle: fewer frames per wave = more waves to hide the table reads behind)
#endif
constexpr int ZSEQ_FRAMES = BAMD_ZSEQ_FRAMES;
__global__ __launch_bounds__(64) void k_zstd_seq(const StreamDesc* __restrict__ streams, int nstreams, const ChunkDesc* __restrict__ chunks,
                                                 const BlockDesc* __restrict__ blocks, ZMeta* __restrict__ meta, ptrdiff_t zseq_delta, ZgLds* __restrict__ gscr,
                                                 const ZcTab* __restrict__ ctab) {
  __shared__ uint64_t sqb[8 * 64];           // the triples' run buffer (zseq_run)
  __shared__ uint32_t codes[36 + 53];
  zseq_fill_codes(codes, (int)(threadIdx.x & 63));
  const int sid = (int)blockIdx.x * ZSEQ_FRAMES + (int)(threadIdx.x & 63);
  if ((int)(threadIdx.x & 63) >= ZSEQ_FRAMES || sid >= nstreams) return;
  if (meta[sid].state != ZM_SEQ) return;
  const StreamDesc& sd = streams[sid];
  const ChunkDesc& c = chunks[sd.chunk];
  const BlockDesc& bk = blocks[sd.aux];
  uint8_t* lit = c.stage + (size_t)bk.blk * (size_t)c.blocksize + (size_t)(sid - bk.first_stream) * (size_t)sd.out_size;
  uint64_t* sq = zseq_ptr(lit, zseq_delta);
  ZgLds* L = gscr + sid;
  const int nseq = (int)meta[sid].nseq;
  const uint32_t al = meta[sid].pad_[2];
  const int al_l = (int)(al & 0xffu), al_o = (int)((al >> 8) & 0xffu), al_m = (int)((al >> 16) & 0xffu);
  zd::SeqTabs tb = {{L->t.fse[0], al_l}, {L->t.fse[1], al_o}, {L->t.fse[2], al_m}, true, true, true};
  zd::SeqState st;
  st.rep[0] = 1u; st.rep[1] = 4u; st.rep[2] = 8u;
  bool fine = zd::seq_begin(st, tb, sd.in + meta[sid].pad_[0], (int)meta[sid].pad_[1]);
  if (fine) {
    volatile uint64_t* rb = sqb;
    if (ctab) fine = zseq_run<1>(st, nseq, nullptr, nullptr, nullptr, ctab[sid].ll, ctab[sid].of, ctab[sid].ml, al_l, al_o, al_m, sq, rb, (int)(threadIdx.x & 63), codes);
    else fine = zseq_run(st, nseq, L->t.fse[0], L->t.fse[1], L->t.fse[2], nullptr, nullptr, nullptr, al_l, al_o, al_m, sq, rb, (int)(threadIdx.x & 63));
  }
  if (fine && st.b.off != 0) fine = false;                                  // the bit stream must be consumed exactly
  meta[sid].state = fine ? ZM_READY : ZM_ERROR;
#ifdef BAMD_WAVE_EMU
  g_emu_zstd_paths[3]++;
#endif
}

// (Round 3 also built this kernel with the frames' tables in LDS - 30 frames per wave, two waves per CU: 16.2 - 16.8 ms against 11.7 ms, the
//  dependent-issue latency of the loop at one wave per SIMD outweighs the table reads; removed in round 4, record: profiles/r03/r03y_zent_split.txt.)

// phase B: one wavefront per frame, persistent + ticket
#ifndef BAMD_ZEXEC_MINWAVES
#define BAMD_ZEXEC_MINWAVES 8     // waves per SIMD the register allocator plans for (64 VGPRs at 8)
#endif
constexpr int ZEXEC_WAVES_PER_CU = 4 * BAMD_ZEXEC_MINWAVES;
__global__ __launch_bounds__(64, BAMD_ZEXEC_MINWAVES) void k_zstd_exec(StreamDesc* __restrict__ streams, int nstreams, int32_t* __restrict__ status, uint32_t* __restrict__ ticket,
                                                     const ChunkDesc* __restrict__ chunks, const BlockDesc* __restrict__ blocks,
                                                     const ZMeta* __restrict__ meta, ptrdiff_t zseq_delta) {
  __shared__ uint32_t xbuf[ZXB_WORDS];      // the LDS-assembled groups of zstd_exec16 (k_zstd.hip): 3 KiB per wave, 32 waves per CU
  const int lane = threadIdx.x & 63;
  uint32_t sid = take_ticket(ticket, lane);
  while (sid < (uint32_t)nstreams) {
    const uint32_t state = uni(meta[sid].state);
    StreamDesc* sd = streams + sid;
    if (state == ZM_ERROR) {
      if (lane == 0) { sd->result = 0; atomicMin(&status[sd->chunk], (int32_t)ST_BADCODEC); }
    } else if (state == ZM_READY) {
      const uint32_t want = uni((uint32_t)sd->out_size);
      const ChunkDesc* c = chunks + uni((uint32_t)sd->chunk);
      const BlockDesc* bk = blocks + uni((uint32_t)sd->aux);
      uint8_t* litbuf = c->stage + (size_t)uni((uint32_t)bk->blk) * (size_t)uni((uint32_t)c->blocksize) +
                        (size_t)(sid - uni((uint32_t)bk->first_stream)) * (size_t)want;
      const uint64_t* sq = zseq_ptr(litbuf, zseq_delta);
      const uint32_t nseq = uni(meta[sid].nseq), regen = uni(meta[sid].regen), lmode = uni(meta[sid].lit_mode), lsrc = uni(meta[sid].lit_src);
      const uint8_t* lit = litbuf;
      if (lmode == 1u) lit = sd->in + lsrc;
      else if (lmode == 2u) wave_fill(as_global(litbuf), lsrc, regen, lane);
      uint32_t op = 0, lp = 0;
      ZxState zx = {0u, 0u, 0u};
      bool ok = true;
      for (uint32_t done = 0; ok && done < nseq; done += 64u) {
        const uint32_t m = nseq - done < 64u ? nseq - done : 64u;
        const uint64_t q = (uint32_t)lane < m ? sq[done + (uint32_t)lane] : 0ull;
        const uint32_t ll_b = (uint32_t)q & 0x3ffffu, ml_b = (uint32_t)(q >> 18) & 0x3ffffu, off_b = (uint32_t)(q >> 36);
        for (uint32_t g = 0; ok && g < m; g += 16u)
          ok = zstd_exec16(ll_b, ml_b, off_b, (int)g, (int)(m - g < 16u ? m - g : 16u), sd->out, want, op, lit, lp, regen, lane, xbuf, &zx);
      }
      if (ok) {
        const uint32_t rest = regen - lp;
        if ((uint64_t)op + rest > (uint64_t)want) ok = false;
        else { if (rest) wave_copy_disjoint(as_global(sd->out) + op, as_global(lit) + lp, rest, lane); op += rest; }
      }
      if (lane == 0) {
        sd->result = ok ? (int32_t)op : 0;
        if (!ok || op != want) atomicMin(&status[sd->chunk], (int32_t)ST_BADCODEC);   // blosc.c:780-782
      }
      if (ok && op == want) fused_unshuffle_own_block(c, bk, lane);
    }
    sid = take_ticket(ticket, lane);
  }
}

}  // namespace bamd
