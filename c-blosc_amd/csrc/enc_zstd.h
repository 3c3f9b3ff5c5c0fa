// enc_zstd.h — Zstd frames on the device (included by k_encode.hip inside namespace bamd): the sequence section writers
// (scalar and vector form), the per-block sequence tables, zstd_encode_wave.  Format: zstd_enc.h.  DESIGN.md 3.6.
// ---------------------------------------------------------------------------------------------
// Zstd frames (zstd_enc.h has the format; this is its wave-parallel use).  One frame per stream, blocks of at most
// 128 KiB; per block the match finder above fills a ZsSink, then the sequence section is coded: code numbers and extra
// bits of 64 sequences at a time in the lanes, the three FSE state chains and the bit writer as a wave-uniform
// (scalar) loop over them, tables in the wave's LDS (the hash table is rebuilt per stream anyway).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zs_emit_seq(ZsSink& z, const gu8* lit, uint32_t ll, uint32_t off, uint32_t mlen, int lit_lane0, uint32_t ownbyte, int lane) {
  if (z.nlit + ll > z.litcap || z.nseq >= z.seqcap) return 0xffffffffu;
  emit_literals(z.lit + z.nlit, lit, ll, lit_lane0, ownbyte, lane);
  if (lane == 0) z.seq[z.nseq] = zenc::pack_seq(ll, mlen, off);
  z.nlit += ll; z.nseq++;
  return 0u;
}

// distances -> Offset_Values (repeat codes, zstd_enc.h: rep_value), in stream order: 64 sequences per load, the history
// as wave-uniform state
__device__ __forceinline__ void zs_assign_offset_values(BAMD_GAS uint64_t* seqs, uint32_t nseq, zenc::RepState& rep, int lane) {
  for (uint32_t base = 0; base < nseq; base += 64u) {
    const uint32_t cnt = nseq - base < 64u ? nseq - base : 64u;
    const uint64_t q = (uint32_t)lane < cnt ? seqs[base + (uint32_t)lane] : 0ull;
    const uint32_t off_l = zenc::seq_off(q), ll_l = zenc::seq_ll(q);
    uint32_t val_l = 0;
    for (uint32_t k = 0; k < cnt; k++) {
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)off_l, (int)k), ll = (uint32_t)__builtin_amdgcn_readlane((int)ll_l, (int)k);
      const uint32_t v = zenc::rep_value(rep, off, ll);
      val_l = (uint32_t)lane == k ? v : val_l;
    }
    if ((uint32_t)lane < cnt) seqs[base + (uint32_t)lane] = zenc::pack_seq(ll_l, zenc::seq_ml(q), val_l);
  }
}

typedef BAMD_LAS const zenc::CTab LdsCTab;
__device__ __forceinline__ uint32_t zs_tab_u32(const BAMD_LAS uint32_t* p, uint32_t i) { return uni(p[i]); }

// out: start of the Sequences_Section, room: bytes available.  Returns the section size or 0xffffffff.
__device__ __forceinline__ uint32_t zs_write_sequences(gu8* out, uint32_t room, const BAMD_GAS uint64_t* seqs, uint32_t nseq,
                                                       const BAMD_LAS zenc::CTabs* T, int lane) {
  if (room < 8u) return 0xffffffffu;
  uint32_t pos = 0;
  if (nseq == 0u) { if (lane == 0) out[0] = 0; return 1u; }
  if (nseq < 128u) { if (lane == 0) out[0] = (uint8_t)nseq; pos = 1; }
  else if (nseq < 0x7f00u) { if (lane == 0) { out[0] = (uint8_t)((nseq >> 8) + 128u); out[1] = (uint8_t)nseq; } pos = 2; }
  else { if (lane == 0) { out[0] = 255u; out[1] = (uint8_t)(nseq - 0x7f00u); out[2] = (uint8_t)((nseq - 0x7f00u) >> 8); } pos = 3; }
  if (lane == 0) out[pos] = 0;                               // three predefined tables
  pos += 1;
  const BAMD_LAS uint32_t* ll_dnb = (const BAMD_LAS uint32_t*)T->ll.dnb; const BAMD_LAS int32_t* ll_dfs = (const BAMD_LAS int32_t*)T->ll.dfs; const BAMD_LAS uint16_t* ll_st = (const BAMD_LAS uint16_t*)T->ll.st;
  const BAMD_LAS uint32_t* ml_dnb = (const BAMD_LAS uint32_t*)T->ml.dnb; const BAMD_LAS int32_t* ml_dfs = (const BAMD_LAS int32_t*)T->ml.dfs; const BAMD_LAS uint16_t* ml_st = (const BAMD_LAS uint16_t*)T->ml.st;
  const BAMD_LAS uint32_t* of_dnb = (const BAMD_LAS uint32_t*)T->of.dnb; const BAMD_LAS int32_t* of_dfs = (const BAMD_LAS int32_t*)T->of.dfs; const BAMD_LAS uint16_t* of_st = (const BAMD_LAS uint16_t*)T->of.st;
  uint64_t acc = 0; uint32_t nb = 0; bool ovf = false;
  auto add = [&](uint32_t v, uint32_t n) {
    acc |= (uint64_t)v << nb; nb += n;
    if (nb >= 32u) {
      if (pos + 4u > room) ovf = true; else if (lane == 0) g_st4(out + pos, (uint32_t)acc);
      pos += 4u; acc >>= 32; nb -= 32u;
    }
  };
  uint32_t sll = 0, sml = 0, sof = 0;
  bool first = true;
  for (uint32_t base = ((nseq - 1u) >> 6) << 6;; base -= 64u) {
    const uint32_t cnt = nseq - base < 64u ? nseq - base : 64u;
    const uint64_t q = (uint32_t)lane < cnt ? seqs[base + (uint32_t)lane] : zenc::pack_seq(0, 3, 4);
    const zenc::Code l = zenc::ll_code(zenc::seq_ll(q)), m = zenc::ml_code(zenc::seq_ml(q)), o = zenc::of_code_value(zenc::seq_off(q));
    // everything that does not depend on the FSE states is prepared per lane, 64 sequences at once: the table rows of the
    // sequence's three codes and its extra bits as ONE field (literal-length | match-length | offset bits, <= 49 bits).
    // The serial loop below - a scalar program, and all waves of a CU share one scalar unit - only walks the states.
    const uint32_t dl_v = ll_dnb[l.code], dm_v = ml_dnb[m.code], do_v = of_dnb[o.code];
    const uint32_t nbx_v = l.bits + m.bits + o.bits;
    const uint32_t fpk_v = ((uint32_t)ll_dfs[l.code] & 0xffu) | (((uint32_t)ml_dfs[m.code] & 0xffu) << 8) | (((uint32_t)of_dfs[o.code] & 0xffu) << 16) | (nbx_v << 24);
    const uint64_t ext_v = (uint64_t)l.extra | ((uint64_t)m.extra << l.bits) | ((uint64_t)o.extra << (l.bits + m.bits));
    const uint32_t exl_v = (uint32_t)ext_v, exh_v = (uint32_t)(ext_v >> 32);
    for (int k = (int)cnt - 1; k >= 0; k--) {
      const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)dl_v, k), dm = (uint32_t)__builtin_amdgcn_readlane((int)dm_v, k);
      const uint32_t dO = (uint32_t)__builtin_amdgcn_readlane((int)do_v, k), fpk = (uint32_t)__builtin_amdgcn_readlane((int)fpk_v, k);
      const uint32_t exl = (uint32_t)__builtin_amdgcn_readlane((int)exl_v, k), exh = (uint32_t)__builtin_amdgcn_readlane((int)exh_v, k);
      const int32_t fl = (int32_t)(int8_t)(fpk & 0xffu), fm = (int32_t)(int8_t)((fpk >> 8) & 0xffu), fo = (int32_t)(int8_t)((fpk >> 16) & 0xffu);
      const uint32_t nbx = fpk >> 24;
      if (first) {
        first = false;
        const uint32_t nm = (dm + (1u << 15)) >> 16, nO = (dO + (1u << 15)) >> 16, nl = (dl + (1u << 15)) >> 16;
        sml = uni((uint32_t)ml_st[(int32_t)(((nm << 16) - dm) >> nm) + fm]);
        sof = uni((uint32_t)of_st[(int32_t)(((nO << 16) - dO) >> nO) + fo]);
        sll = uni((uint32_t)ll_st[(int32_t)(((nl << 16) - dl) >> nl) + fl]);
      } else {
        // the three state transitions: their bits (<= 5 + 6 + 6) go out as one field
        const uint32_t nO = (sof + dO) >> 16, nm = (sml + dm) >> 16, nl = (sll + dl) >> 16;
        const uint32_t bitsv = (sof & ((1u << nO) - 1u)) | ((sml & ((1u << nm) - 1u)) << nO) | ((sll & ((1u << nl) - 1u)) << (nO + nm));
        const uint32_t nxo = uni((uint32_t)of_st[(int32_t)(sof >> nO) + fo]), nxm = uni((uint32_t)ml_st[(int32_t)(sml >> nm) + fm]);
        const uint32_t nxl = uni((uint32_t)ll_st[(int32_t)(sll >> nl) + fl]);
        add(bitsv, nO + nm + nl);
        sof = nxo; sml = nxm; sll = nxl;
      }
      if (nbx > 24u) { add(exl & 0xffffffu, 24u); add((exl >> 24) | (exh << 8), nbx - 24u); }      // <= 49 bits: two pieces of <= 25
      else add(exl, nbx);
    }
    if (base == 0u) break;
  }
  add(sml & 63u, (uint32_t)zenc::kMLLog); add(sof & 31u, (uint32_t)zenc::kOFLog); add(sll & 63u, (uint32_t)zenc::kLLLog);
  add(1u, 1u);
  while (nb > 0u) {
    if (pos >= room) { ovf = true; break; }
    if (lane == 0) out[pos] = (uint8_t)acc;
    pos++; acc >>= 8; nb = nb > 8u ? nb - 8u : 0u;
  }
  return ovf ? 0xffffffffu : pos;
}

// ---------------------------------------------------------------------------------------------
// The same section, written without the scalar unit (the form in use).  The scalar version above costs ~60 SALU
// instructions per sequence, and the 20 waves of a CU share ONE scalar unit: 262 M sequences per 8 GiB of bench19 were
// ~30 of the kernel's 42 ms.  Here, per batch of 64 sequences:
//   * every lane prepares its sequence (codes, table rows, extra bits) as before;
//   * the three FSE state chains run on THREE LANES (0: literal lengths, 1: match lengths, 2: offsets), vector code: per
//     step one dependent LDS read (the next state); the bits each transition emits go to an LDS array;
//   * all 64 lanes then place their sequence's bits - state bits, then extra bits, <= 66 per sequence - with a prefix sum
//     over the bit counts (in stream order: last sequence first), OR them into an LDS strip and store the full dwords.
// Bit for bit the output of zenc::write_sequences (tests: every frame is read by ZSTD_decompress and by the oracle).
// Scratch: 64 x 3 transition words + a 136-dword strip, taken from the START of this wave's hash table - the section is
// written after the block's match finding; a later block of the same stream finds some stale entries there, which the
// candidate check (tag, then bytes) rejects like any other stale entry.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ZV_STRIP = 136u;

// OR the low n (<= 49) bits of v into the strip at bit position bitpos
__device__ __forceinline__ void zv_or_bits(volatile BAMD_LAS uint32_t* strip, uint32_t bitpos, uint64_t v, uint32_t n) {
  if (n == 0u) return;
  const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
  const uint64_t lo = v << sh;
  __hip_atomic_fetch_or((BAMD_LAS uint32_t*)strip + w, (uint32_t)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  if (sh + n > 32u) __hip_atomic_fetch_or((BAMD_LAS uint32_t*)strip + w + 1u, (uint32_t)(lo >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  if (sh + n > 64u) __hip_atomic_fetch_or((BAMD_LAS uint32_t*)strip + w + 2u, (uint32_t)(v >> (64u - sh)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// ---------------------------------------------------------------------------------------------
// Per-block sequence tables (zstd_enc.h: "per-block tables"; kernel k_encode_streams_t<ENC_ZSTD_T>).  The predefined
// distributions are made for text-like sequences; byte planes of numeric data have a handful of literal-length and match-length
// codes and two or three offset codes, and coding them with tables made for the block is worth +29 % ratio on the SURVEY 8d
// planes (+67 % on linspace; tests/tools/zstd_enc_cpu.cpp measures it on the CPU with the same format functions).
// Per block, after the match finder: histogram of the three code alphabets (LDS atomics), then per alphabet with one lane
// per symbol: RLE when a single code occurs; else probabilities normalised to 64 (zenc::fse_normalize's rule), the table
// description written with a prefix sum over the field widths, its cost compared with the predefined table's, and the
// encoder table built with one lane per CELL - with Accuracy_Log 6 and no "less than 1" probabilities the k-th cell
// handed out lies at (43 k) & 63, so the spread, its inverse (k = 3 u & 63) and every cell's rank inside its symbol need no
// serial pass.  Everything lives in the scratch at the start of this wave's hash table, like the writer's strips.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ZT_HIST = 328u, ZT_CTABS = 520u, ZT_DESC = 934u, ZT_END = 970u;     // dword offsets in the wave's scratch
static_assert(sizeof(zenc::CTabs) == 1656 && ZT_CTABS + sizeof(zenc::CTabs) / 4 == ZT_DESC && ZT_END * 4u <= (uint32_t)ENC_TAB_BYTES, "scratch layout");
static_assert(zenc::kCustomLog == 6, "one lane per cell");
struct ZsTabs {                  // wave-uniform; index 0 literal lengths, 1 offsets, 2 match lengths (the order of the modes byte)
  uint32_t mode[3], rle[3], log[3], desc_len[3];
};
// the predefined distributions (RFC 8878 3.1.1.3.2.2.1), one row per alphabet in ZsTabs order, -1 = "less than 1"
__device__ const int8_t kZtPredef[3][64] = {
  {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1},
  {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1},
  {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1}};
// -log2(n / 2^log) in 1/256 bits, as zenc::fse_cost256 counts it
__device__ __forceinline__ uint32_t zt_bits256(uint32_t n, uint32_t log) {
  const uint32_t hbit = 31u - (uint32_t)__builtin_clz(n);
  const uint32_t frac = ((n << 8) >> hbit) - 256u;
  return ((log - hbit) << 8) - frac;
}
// One alphabet.  `c`: this lane's symbol count (0 for lanes beyond the alphabet).  Fills zt.* for index a and, in FSE mode, the
// encoder table `ct` and the description at desc[0 .. 12).  Returns nothing; all decisions are wave-uniform.
// FORCE: no predefined alternative (the weights of a Huffman tree description): FSE mode whenever the counts can be normalised
template <bool FORCE = false>
__device__ __forceinline__ void zt_make_table(uint32_t c, uint32_t total, int a, uint32_t predef_log, ZsTabs& zt, BAMD_LAS zenc::CTab* ct,
                                              volatile BAMD_LAS uint32_t* desc, int lane) {
  zt.mode[a] = zenc::kModePredefined; zt.log[a] = predef_log; zt.rle[a] = 0u; zt.desc_len[a] = 0u;
  const uint64_t present = __ballot(c != 0u);
  if (present == 0ull) return;
  if ((present & (present - 1ull)) == 0ull) { zt.mode[a] = zenc::kModeRLE; zt.rle[a] = (uint32_t)__builtin_ctzll(present); zt.log[a] = 0u; return; }
  // ---- probabilities: floor of the proportional share, at least 1; the rest to the most frequent symbol ----
  uint32_t v = c ? (c << 6) / total : 0u;
  if (c && v == 0u) v = 1u;
  const uint32_t sum = wave_sum_u32(v);
  const int big = 63 - (int)(wave_max_u32((c << 6) | (63u - (uint32_t)lane)) & 63u);       // first lane among the largest counts
  if (sum <= 64u) { if (lane == big) v += 64u - sum; }
  else {
    for (uint32_t over = sum - 64u; over > 0u; over--) {
      const uint32_t best = wave_max_u32((v << 6) | (63u - (uint32_t)lane));
      if ((best >> 6) < 2u) return;                                  // cannot be normalised: predefined
      if (lane == 63 - (int)(best & 63u)) v--;
    }
  }
  const uint64_t nz = __ballot(v != 0u);
  const int last = 63 - __builtin_clzll(nz);
  const uint32_t incl = wave_incl_scan_u32(v, lane);
  const uint32_t excl = incl - v;                                    // cells handed out before this symbol
  // ---- cost with this table against the predefined one ----
  uint32_t cost_pre = 0xffffffffu, cost_new = 0u;
  if (!FORCE) {
    const int32_t pn = (int32_t)kZtPredef[a][lane];
    cost_pre = wave_sum_u32(c ? c * zt_bits256(pn < 0 ? 1u : (uint32_t)pn, predef_log) : 0u);
    cost_new = wave_sum_u32(c ? c * zt_bits256(v, 6u) : 0u);
  }
  // ---- the description: 4 bits log - 5, then per symbol up to `last` a field whose width follows from the points still left ----
  uint64_t field = 0; uint32_t width = 0;
  if (lane <= last) {
    const uint32_t remaining = 64u - excl;
    const uint32_t value = v + 1u;
    const uint32_t bits = (31u - (uint32_t)__builtin_clz(remaining + 1u)) + 1u;
    const uint32_t low = (1u << bits) - 1u - (remaining + 1u);
    if (value < low) { field = value; width = bits - 1u; }
    else if (value < (1u << (bits - 1u))) { field = value; width = bits; }
    else { field = value + low; width = bits; }
  }
  const uint32_t vprev = (uint32_t)__shfl_up((int)v, 1, 64);
  if (lane <= last && v == 0u) {
    if (lane != 0 && vprev == 0u) { field = 0; width = 0; }          // inside a run of absent symbols: counted by its first one
    else {
      const uint32_t next = (uint32_t)lane + 1u + (uint32_t)__builtin_ctzll(nz >> ((uint32_t)lane + 1u));   // lane < last here
      const uint32_t z = next - (uint32_t)lane - 1u;                 // further zeros behind this one
      const uint32_t threes = z / 3u;
      const uint64_t flags = ((1ull << (2u * threes)) - 1ull) | ((uint64_t)(z - 3u * threes) << (2u * threes));
      field |= flags << width; width += 2u * (threes + 1u);
    }
  }
  if (lane == 0) { field = (field << 4) | (uint64_t)(6u - 5u); width += 4u; }
  const uint32_t wincl = wave_incl_scan_u32(width, lane);
  const uint32_t dbits = (uint32_t)__builtin_amdgcn_readlane((int)wincl, 63);
  const uint32_t dlen = (dbits + 7u) >> 3;
  cost_new += dlen << 11;
  if (!FORCE && cost_new >= cost_pre) return;                        // the predefined table is cheaper (short blocks)
  if (lane < 12) desc[lane] = 0u;
  BAMD_LDS_SYNC();
  zv_or_bits(desc, wincl - width, field, width);
  // ---- encoder table (zenc::build_ctab): per symbol deltaNbBits / deltaFindState ... ----
  BAMD_LAS uint32_t* dnb = (BAMD_LAS uint32_t*)ct->dnb; BAMD_LAS int32_t* dfs = (BAMD_LAS int32_t*)ct->dfs; BAMD_LAS uint16_t* st = (BAMD_LAS uint16_t*)ct->st;
  if (lane < 53) {
    if (v == 0u) { dnb[lane] = (7u << 16) - 64u; dfs[lane] = 0; }
    else if (v == 1u) { dnb[lane] = (6u << 16) - 64u; dfs[lane] = (int32_t)excl - 1; }
    else {
      const uint32_t maxbits = 6u - (31u - (uint32_t)__builtin_clz(v - 1u));
      dnb[lane] = (maxbits << 16) - (v << maxbits);
      dfs[lane] = (int32_t)excl - (int32_t)v;
    }
  }
  // ---- ... and per cell the coded state: lane u is cell u; it was the k-th cell handed out, k = 3 u & 63 (43 * 3 = 129) ----
  // symbol of hand-out index k: the last symbol whose first index is <= k
  uint32_t mark = 0;
  {
    // lane k learns the symbol that starts at index k (if any): scatter through the cross-lane network, one symbol at a time
    // would be serial; instead every lane k counts the symbols whose range starts at or before k
    uint32_t sy = 0;
    for (int sidx = 0; sidx <= last; sidx++) {
      const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)excl, sidx), n = (uint32_t)__builtin_amdgcn_readlane((int)v, sidx);
      if (n != 0u && e <= (uint32_t)lane) sy = (uint32_t)sidx;
    }
    mark = sy;                                                       // symbol of hand-out index `lane`
  }
  const uint32_t k = (3u * (uint32_t)lane) & 63u;
  const uint32_t S = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(k << 2), (int)mark);                 // symbol in cell `lane`
  const uint32_t cS = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(S << 2), (int)excl);
  const uint32_t nS = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(S << 2), (int)v);
  uint32_t rank = 0;
  for (uint32_t j = 0; j < nS; j++) rank += ((((cS + j) * 43u) & 63u) < (uint32_t)lane) ? 1u : 0u;
  st[cS + rank] = (uint16_t)(64u + (uint32_t)lane);
  BAMD_LDS_SYNC();
  zt.mode[a] = zenc::kModeFSE; zt.log[a] = 6u; zt.desc_len[a] = dlen;
}

// histogram + the three tables of one block; `pre`: the predefined tables, `scr`: the wave's scratch
__device__ __forceinline__ void zt_make_tables(const BAMD_GAS uint64_t* seqs, uint32_t nseq, volatile BAMD_LAS uint32_t* scr, ZsTabs& zt, int lane) {
  volatile BAMD_LAS uint32_t* hist = scr + ZT_HIST;
  BAMD_LDS_SYNC();                                     // the scratch overlays the match finder's table: every lane is done with that
  hist[lane] = 0u; hist[lane + 64] = 0u; hist[lane + 128] = 0u;
  BAMD_LDS_SYNC();
  for (uint32_t base = 0; base < nseq; base += 64u) {
    if (base + (uint32_t)lane < nseq) {
      const uint64_t q = seqs[base + (uint32_t)lane];
      __hip_atomic_fetch_add((BAMD_LAS uint32_t*)hist + zenc::ll_code(zenc::seq_ll(q)).code, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add((BAMD_LAS uint32_t*)hist + 64u + zenc::of_code_value(zenc::seq_off(q)).code, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add((BAMD_LAS uint32_t*)hist + 128u + zenc::ml_code(zenc::seq_ml(q)).code, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  BAMD_LDS_SYNC();
  BAMD_LAS zenc::CTabs* C = (BAMD_LAS zenc::CTabs*)((BAMD_LAS uint32_t*)scr + ZT_CTABS);
  zt_make_table(lane < zenc::kLLSyms ? hist[lane] : 0u, nseq, 0, (uint32_t)zenc::kLLLog, zt, &C->ll, scr + ZT_DESC, lane);
  zt_make_table(lane < 32 ? hist[64 + lane] : 0u, nseq, 1, (uint32_t)zenc::kOFLog, zt, &C->of, scr + ZT_DESC + 12, lane);
  zt_make_table(lane < zenc::kMLSyms ? hist[128 + lane] : 0u, nseq, 2, (uint32_t)zenc::kMLLog, zt, &C->ml, scr + ZT_DESC + 24, lane);
}
// Code lengths of a complete prefix code, at most LIMIT bits, for an alphabet of 64 * PER symbols (this lane: symbols PER * lane ..
// PER * lane + PER - 1, counts c[], `total` their sum): ceil(log2(total / count)) cut to LIMIT, then the code space is brought to exactly
// full - over-subscription is paid by the longest codes below LIMIT (the rarest of them first), slack goes to where it saves the most
// bits (count << length) and still fits - one wave-wide maximum per move.  Returns the number of symbols present; 0: the space could
// not be filled (never seen; the caller falls back); 1: a lone symbol, length 1 (an incomplete code: what Deflate wants there).
template <int PER, int LIMIT>
__device__ __forceinline__ uint32_t wave_code_lengths(const uint32_t (&c)[PER], uint32_t total, uint32_t (&l)[PER], int lane) {
  constexpr uint32_t FULL = 1u << LIMIT;
  uint32_t kraft = 0, npresent = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    l[j] = 0u;
    if (c[j]) {
      uint32_t b = 1u;
      while (b < (uint32_t)LIMIT && ((uint64_t)c[j] << b) < (uint64_t)total) b++;
      l[j] = b; kraft += FULL >> b; npresent++;
    }
  }
  npresent = wave_sum_u32(npresent);
  if (npresent < 2u) return npresent;                               // (the lone symbol already has length 1)
  kraft = wave_sum_u32(kraft);
  while (kraft > FULL) {
    uint32_t key = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) if (l[j] && l[j] < (uint32_t)LIMIT) { const uint32_t k2 = (l[j] << 20) | (0xfffffu - (c[j] < 0xfffffu ? c[j] : 0xfffffu)); key = k2 > key ? k2 : key; }
    const uint32_t best = wave_max_u32(key);
    if (best == 0u) return 0u;
    const uint64_t who = __ballot(key == best);
    if (lane == __builtin_ctzll(who)) {
      bool done = false;
#pragma unroll
      for (int j = 0; j < PER; j++) if (!done && l[j] && l[j] < (uint32_t)LIMIT && ((l[j] << 20) | (0xfffffu - (c[j] < 0xfffffu ? c[j] : 0xfffffu))) == best) { l[j]++; done = true; }
    }
    kraft -= FULL >> ((best >> 20) + 1u);
  }
  uint32_t slack = FULL - kraft;
  while (slack) {
    uint32_t key = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) if (l[j] > 1u && (FULL >> l[j]) <= slack) { const uint32_t k2 = (c[j] < 0xffffu ? c[j] : 0xffffu) << l[j]; key = k2 > key ? k2 : key; }
    const uint32_t best = wave_max_u32(key);
    if (best == 0u) return 0u;
    const uint64_t who = __ballot(key == best);
    const int wl = __builtin_ctzll(who);
    uint32_t step = 0;
    if (lane == wl) {
      bool done = false;
#pragma unroll
      for (int j = 0; j < PER; j++) if (!done && l[j] > 1u && (FULL >> l[j]) <= slack && ((c[j] < 0xffffu ? c[j] : 0xffffu) << l[j]) == best) { step = FULL >> l[j]; l[j]--; done = true; }
    }
    slack -= (uint32_t)__builtin_amdgcn_readlane((int)step, wl);
  }
  return npresent;
}

// ---------------------------------------------------------------------------------------------
// Huffman-coded literals (zstd_enc.h: "Huffman-coded literals"; flag bit 0 of the tables kernels).  Noisy planes keep most of
// their bytes as literals, and those are what the reference's ratio on such data comes from (small integers: 2.0 -> 2.35).
// Per block, after the match finder and before the sequence tables (same scratch): byte histogram (LDS atomics); code lengths with
// one lane per four byte values - ceil(log2(total / count)) cut to 11 bits, then the code space is brought to exactly full:
// over-subscription is paid by the longest codes below 11 bits, slack goes to where it saves the most bits (count << length),
// one wave-wide maximum per move (tests/test_zstd_enc_cpu.py's true Huffman tree is within 3 % of this); canonical codes in the
// format's order by one prefix sum per length; tree description direct (up to 128 weights) or FSE-compressed (weights' table by
// zt_make_table<FORCE>, the two interleaved states walked by lane 0); then one or four streams, 64 literals per step placed with a
// prefix sum over their code lengths.  The section is assembled in the sequence scratch behind the block's triples and copied
// in front of the sequences section if it is smaller than the raw form.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t ZH_CNT = 0u, ZH_CODE = 256u, ZH_STRIP = 512u, ZH_CTAB = 544u, ZH_DESC = 688u, ZH_W = 704u, ZH_H12 = 768u, ZH_END = 784u;   // dwords
static_assert(ZH_CTAB + (sizeof(zenc::CTab) + 3) / 4 <= ZH_DESC && ZH_END * 4u <= (uint32_t)ENC_TAB_BYTES, "scratch layout");
struct ZhOut { gu8* p; uint32_t pos, cap, acc, nb; };            // forward bit writer, wave-uniform state
// the codes of up to 64 symbols, lane order = stream order (lane 0 lands in the lowest bits); <= 11 bits each
__device__ __forceinline__ bool zh_put(ZhOut& o, volatile BAMD_LAS uint32_t* strip, uint32_t bits, uint32_t nbits, int lane) {
  const uint32_t incl = wave_incl_scan_u32(nbits, lane);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  if (lane < 24) strip[lane] = lane == 0 ? o.acc : 0u;
  BAMD_LDS_SYNC();
  if (nbits) zv_or_bits(strip, o.nb + incl - nbits, (uint64_t)bits, nbits);
  const uint32_t fill = o.nb + total, ndw = fill >> 5;            // <= 22 full dwords
  if (o.pos + 4u * ndw + 8u > o.cap) return false;
  BAMD_LDS_SYNC();
  if ((uint32_t)lane < ndw) g_st4(o.p + o.pos + 4u * (uint32_t)lane, strip[lane]);
  o.acc = uni(strip[ndw]); o.nb = fill & 31u; o.pos += 4u * ndw;
  return true;
}
// one stream: literals lit[0..len) coded last byte first, the mark bit, zero padding to the byte
__device__ __forceinline__ bool zh_stream(ZhOut& o, volatile BAMD_LAS uint32_t* scr, const gu8* lit, uint32_t len, int lane) {
  o.acc = 0u; o.nb = 0u;
  for (uint32_t base = 0; base < len; base += 64u) {
    uint32_t e = 0;
    if (base + (uint32_t)lane < len) e = scr[ZH_CODE + (uint32_t)lit[len - 1u - base - (uint32_t)lane]];
    if (!zh_put(o, scr + ZH_STRIP, e & 0xfffu, e >> 12, lane)) return false;
  }
  if (!zh_put(o, scr + ZH_STRIP, lane == 0 ? 1u : 0u, lane == 0 ? 1u : 0u, lane)) return false;
  const uint32_t tail = (o.nb + 7u) >> 3;
  if (o.pos + tail > o.cap) return false;
  if ((uint32_t)lane < tail) o.p[o.pos + (uint32_t)lane] = (uint8_t)(o.acc >> (8u * (uint32_t)lane));
  o.pos += tail;
  return true;
}
// the whole Literals_Section of lit[0..n) into stage[0..cap); returns its size, 0 = keep the raw form
__device__ __forceinline__ uint32_t zh_literals(const gu8* lit, uint32_t n, gu8* stage, uint32_t cap, volatile BAMD_LAS uint32_t* scr, int lane) {
  if (n < 64u || n >= (1u << 18) || cap < 256u) return 0u;
  volatile BAMD_LAS uint32_t* cnt = scr + ZH_CNT;
  BAMD_LDS_SYNC();                                                  // the scratch overlays the match finder's table
#pragma unroll
  for (int j = 0; j < 4; j++) cnt[64 * j + lane] = 0u;
  BAMD_LDS_SYNC();
  for (uint32_t i = 4u * (uint32_t)lane; i < n; i += 256u) {
    if (i + 4u <= n) {
      const uint32_t w = g_ld4(lit + i);
#pragma unroll
      for (int b = 0; b < 4; b++) __hip_atomic_fetch_add((BAMD_LAS uint32_t*)cnt + ((w >> (8 * b)) & 0xffu), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
      for (uint32_t k = i; k < n; k++) __hip_atomic_fetch_add((BAMD_LAS uint32_t*)cnt + (uint32_t)lit[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  BAMD_LDS_SYNC();
  // ---- code lengths: this lane's byte values are 4 lane .. 4 lane + 3 ----
  uint32_t c[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) c[j] = cnt[4 * lane + j];
  if (wave_code_lengths<4, 11>(c, n, l, lane) < 2u) return 0u;
  // ---- canonical codes: longer codes first, within a length by byte value ----
  uint32_t lmax = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) lmax = l[j] > lmax ? l[j] : lmax;
  const uint32_t maxbits = wave_max_u32(lmax);
  uint32_t code[4] = {0, 0, 0, 0};
  uint32_t at = 0;                                                  // first table index of the current length
  for (uint32_t b = maxbits; b >= 1u; b--) {
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) mine += l[j] == b ? 1u : 0u;
    const uint32_t incl = wave_incl_scan_u32(mine, lane);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t r = incl - mine;
#pragma unroll
    for (int j = 0; j < 4; j++) if (l[j] == b) { code[j] = (at >> (maxbits - b)) + r; r++; }
    at += tot << (maxbits - b);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) scr[ZH_CODE + 4u * (uint32_t)lane + (uint32_t)j] = code[j] | (l[j] << 12);
  // ---- tree description: weights of byte values 0 .. last - 1 ----
  const uint64_t anyl = __ballot((l[0] | l[1] | l[2] | l[3]) != 0u);
  const int hl = 63 - __builtin_clzll(anyl);
  const uint32_t lj = l[3] ? 3u : (l[2] ? 2u : (l[1] ? 1u : 0u));
  const uint32_t last = 4u * (uint32_t)hl + (uint32_t)__builtin_amdgcn_readlane((int)lj, hl);
  const uint32_t nw = last;
  const uint32_t hdr = n < 1024u ? 3u : (n < 16384u ? 4u : 5u);
  const bool single = n < 256u;
  uint32_t wv[4];
#pragma unroll
  for (int j = 0; j < 4; j++) wv[j] = l[j] ? maxbits + 1u - l[j] : 0u;
  volatile BAMD_LAS uint32_t* h12 = scr + ZH_H12;
  if (lane < 16) h12[lane] = 0u;
  scr[ZH_W + (uint32_t)lane] = wv[0] | (wv[1] << 8) | (wv[2] << 16) | (wv[3] << 24);       // weights as bytes, in byte-value order
  BAMD_LDS_SYNC();
#pragma unroll
  for (int j = 0; j < 4; j++) if (4u * (uint32_t)lane + (uint32_t)j < nw) __hip_atomic_fetch_add((BAMD_LAS uint32_t*)h12 + wv[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  BAMD_LDS_SYNC();
  gu8* tree = stage + hdr;
  uint32_t tree_len = 0;
  {
    ZsTabs zt;
    BAMD_LAS zenc::CTab* ct = (BAMD_LAS zenc::CTab*)((BAMD_LAS uint32_t*)scr + ZH_CTAB);
    zt_make_table<true>(lane < 12 ? h12[lane] : 0u, nw, 0, 6u, zt, ct, scr + ZH_DESC, lane);
    if (zt.mode[0] == zenc::kModeFSE) {
      // description bytes, then the two interleaved state chains (lane 0: a serial walk over <= 255 weights)
      const uint32_t dl = zt.desc_len[0];
      const uint32_t dwv = scr[ZH_DESC + ((uint32_t)lane >> 2)];
      if ((uint32_t)lane < dl) tree[1u + (uint32_t)lane] = (uint8_t)(dwv >> (8u * ((uint32_t)lane & 3u)));
      uint32_t total_len = 0;
      if (lane == 0) {
        const BAMD_LAS uint32_t* dnb = (const BAMD_LAS uint32_t*)ct->dnb; const BAMD_LAS int32_t* dfs = (const BAMD_LAS int32_t*)ct->dfs;
        const BAMD_LAS uint16_t* stt = (const BAMD_LAS uint16_t*)ct->st;
        const volatile BAMD_LAS uint8_t* wb = (const volatile BAMD_LAS uint8_t*)(scr + ZH_W);
        uint64_t acc = 0; uint32_t nb = 0, pos = 1u + dl;
        uint32_t s1 = 0, s2 = 0; bool i1 = false, i2 = false;
        for (int k = (int)nw - 1; k >= 0; k--) {
          const uint32_t sym = wb[k];
          const uint32_t d = dnb[sym]; const int32_t f = dfs[sym];
          uint32_t& st = (k & 1) ? s2 : s1; bool& inited = (k & 1) ? i2 : i1;
          if (!inited) { const uint32_t nbb = (d + (1u << 15)) >> 16; st = stt[(int32_t)(((nbb << 16) - d) >> nbb) + f]; inited = true; }
          else {
            const uint32_t nbb = (st + d) >> 16;
            acc |= (uint64_t)(st & ((1u << nbb) - 1u)) << nb; nb += nbb;
            st = stt[(int32_t)(st >> nbb) + f];
            while (nb >= 8u) { if (pos < 140u) tree[pos] = (uint8_t)acc; pos++; acc >>= 8; nb -= 8u; }
          }
        }
        acc |= (uint64_t)(s2 & 63u) << nb; nb += 6u;
        acc |= (uint64_t)(s1 & 63u) << nb; nb += 6u;
        acc |= 1ull << nb; nb += 1u;
        while (nb > 0u) { if (pos < 140u) tree[pos] = (uint8_t)acc; pos++; acc >>= 8; nb = nb > 8u ? nb - 8u : 0u; }
        total_len = pos - 1u;
      }
      total_len = (uint32_t)__builtin_amdgcn_readlane((int)total_len, 0);
      if (total_len < 128u && (nw > 128u || total_len < (nw + 1u) / 2u)) { if (lane == 0) tree[0] = (uint8_t)total_len; tree_len = 1u + total_len; }
    }
  }
  if (tree_len == 0u) {                                             // direct: 4 bits per weight
    if (nw > 128u) return 0u;
    BAMD_MEM_SYNC();
    const uint32_t nbytes = (nw + 1u) / 2u;
    const volatile BAMD_LAS uint8_t* wb = (const volatile BAMD_LAS uint8_t*)(scr + ZH_W);
    if ((uint32_t)lane < nbytes) {
      const uint32_t a = wb[2 * lane], b2 = 2u * (uint32_t)lane + 1u < nw ? (uint32_t)wb[2 * lane + 1] : 0u;
      tree[1u + (uint32_t)lane] = (uint8_t)((a << 4) | b2);
    }
    if (lane == 0) tree[0] = (uint8_t)(127u + nw);
    tree_len = 1u + nbytes;
  }
  // ---- the streams ----
  ZhOut o = {stage, hdr + tree_len, cap, 0u, 0u};
  if (single) { if (!zh_stream(o, scr, lit, n, lane)) return 0u; }
  else {
    const uint32_t jump = o.pos; o.pos += 6u;
    const uint32_t q = (n + 3u) / 4u;
#pragma unroll 1
    for (uint32_t k = 0; k < 4u; k++) {
      const uint32_t from = q * k, len = k < 3u ? q : n - 3u * q, p0 = o.pos;
      if (!zh_stream(o, scr, lit + from, len, lane)) return 0u;
      const uint32_t sz = o.pos - p0;
      if (k < 3u) { if (sz > 0xffffu) return 0u; if (lane == 0) { stage[jump + 2u * k] = (uint8_t)sz; stage[jump + 2u * k + 1u] = (uint8_t)(sz >> 8); } }
    }
  }
  const uint32_t csize = o.pos - hdr;
  if (o.pos >= n + zenc::kLitHeaderRaw) return 0u;
  if (hdr == 3u) {
    if (csize >= 1024u) return 0u;
    const uint32_t v = 2u | ((single ? 0u : 1u) << 2) | (n << 4) | (csize << 14);
    if (lane < 3) stage[lane] = (uint8_t)(v >> (8 * lane));
  } else if (hdr == 4u) {
    if (csize >= 16384u) return 0u;
    const uint32_t v = 2u | (2u << 2) | (n << 4) | (csize << 18);
    if (lane < 4) stage[lane] = (uint8_t)(v >> (8 * lane));
  } else {
    const uint64_t v = 2u | (3u << 2) | ((uint64_t)n << 4) | ((uint64_t)csize << 22);
    if (lane < 5) stage[lane] = (uint8_t)(v >> (8 * lane));
  }
  return o.pos;
}

template <bool TABLES>
__device__ __forceinline__ uint32_t zs_write_sequences_v(gu8* out, uint32_t room, const BAMD_GAS uint64_t* seqs, uint32_t nseq,
                                                         const BAMD_LAS zenc::CTabs* T, volatile BAMD_LAS uint32_t* scr, int lane, const ZsTabs* ztp = nullptr) {
  if (room < 8u) return 0xffffffffu;
  uint32_t pos = 0;
  if (nseq == 0u) { if (lane == 0) out[0] = 0; return 1u; }
  if (nseq < 128u) { if (lane == 0) out[0] = (uint8_t)nseq; pos = 1; }
  else if (nseq < 0x7f00u) { if (lane == 0) { out[0] = (uint8_t)((nseq >> 8) + 128u); out[1] = (uint8_t)nseq; } pos = 2; }
  else { if (lane == 0) { out[0] = 255u; out[1] = (uint8_t)(nseq - 0x7f00u); out[2] = (uint8_t)((nseq - 0x7f00u) >> 8); } pos = 3; }
  // the block's three tables: the predefined ones, or (TABLES) what zt_make_tables chose - index 0 literal lengths, 1 offsets, 2 match lengths
  const BAMD_LAS zenc::CTab* tl = &T->ll; const BAMD_LAS zenc::CTab* tm = &T->ml; const BAMD_LAS zenc::CTab* to = &T->of;
  uint32_t log_l = (uint32_t)zenc::kLLLog, log_m = (uint32_t)zenc::kMLLog, log_o = (uint32_t)zenc::kOFLog;
  bool my_rle = false;                                       // this chain lane's alphabet has one symbol: no state, no bits
  if (TABLES) {
    const ZsTabs& zt = *ztp;
    if (lane == 0) out[pos] = (uint8_t)((zt.mode[0] << 6) | (zt.mode[1] << 4) | (zt.mode[2] << 2));      // Symbol_Compression_Modes
    pos += 1;
    const BAMD_LAS zenc::CTabs* C = (const BAMD_LAS zenc::CTabs*)((BAMD_LAS uint32_t*)scr + ZT_CTABS);
    uint32_t extra = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) extra += zt.mode[a] == zenc::kModeRLE ? 1u : (zt.mode[a] == zenc::kModeFSE ? zt.desc_len[a] : 0u);
    if (pos + extra + 8u > room) return 0xffffffffu;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      if (zt.mode[a] == zenc::kModeRLE) { if (lane == 0) out[pos] = (uint8_t)zt.rle[a]; pos += 1u; }
      else if (zt.mode[a] == zenc::kModeFSE) {
        const uint32_t w = scr[ZT_DESC + 12u * (uint32_t)a + ((uint32_t)lane >> 2)];
        if ((uint32_t)lane < zt.desc_len[a]) out[pos + (uint32_t)lane] = (uint8_t)(w >> (8u * ((uint32_t)lane & 3u)));
        pos += zt.desc_len[a];
      }
    }
    if (zt.mode[0] == zenc::kModeFSE) tl = &C->ll;
    if (zt.mode[1] == zenc::kModeFSE) to = &C->of;
    if (zt.mode[2] == zenc::kModeFSE) tm = &C->ml;
    log_l = zt.log[0]; log_o = zt.log[1]; log_m = zt.log[2];
    my_rle = (lane == 0 ? zt.mode[0] : (lane == 1 ? zt.mode[2] : zt.mode[1])) == zenc::kModeRLE;
  } else {
    if (lane == 0) out[pos] = 0;                             // three predefined tables
    pos += 1;
  }
  const BAMD_LAS uint32_t* ll_dnb = (const BAMD_LAS uint32_t*)tl->dnb; const BAMD_LAS int32_t* ll_dfs = (const BAMD_LAS int32_t*)tl->dfs;
  const BAMD_LAS uint32_t* ml_dnb = (const BAMD_LAS uint32_t*)tm->dnb; const BAMD_LAS int32_t* ml_dfs = (const BAMD_LAS int32_t*)tm->dfs;
  const BAMD_LAS uint32_t* of_dnb = (const BAMD_LAS uint32_t*)to->dnb; const BAMD_LAS int32_t* of_dfs = (const BAMD_LAS int32_t*)to->dfs;
  // this lane's chain (lanes 0 / 1 / 2): its state table
  const BAMD_LAS uint16_t* my_st = lane == 0 ? (const BAMD_LAS uint16_t*)tl->st : (lane == 1 ? (const BAMD_LAS uint16_t*)tm->st : (const BAMD_LAS uint16_t*)to->st);
  volatile BAMD_LAS uint32_t* trans = scr;                   // [64][3]: bits | count << 16 of every state transition
  volatile BAMD_LAS uint32_t* strip = scr + 192;             // [ZV_STRIP]
  uint32_t state = 0;                                        // lanes 0..2
  uint32_t pend = 0, npend = 0;                              // bits not yet stored (< 32), wave-uniform
  bool first = true, ovf = false;
  for (uint32_t base = ((nseq - 1u) >> 6) << 6;; base -= 64u) {
    const uint32_t cnt = nseq - base < 64u ? nseq - base : 64u;
    const uint64_t q = (uint32_t)lane < cnt ? seqs[base + (uint32_t)lane] : zenc::pack_seq(0, 3, 4);
    const zenc::Code l = zenc::ll_code(zenc::seq_ll(q)), m = zenc::ml_code(zenc::seq_ml(q)), o = zenc::of_code_value(zenc::seq_off(q));
    const uint32_t dl_v = ll_dnb[l.code], dm_v = ml_dnb[m.code], do_v = of_dnb[o.code];
    const uint32_t nbx = l.bits + m.bits + o.bits;
    const uint32_t fpk_v = ((uint32_t)ll_dfs[l.code] & 0xffu) | (((uint32_t)ml_dfs[m.code] & 0xffu) << 8) | (((uint32_t)of_dfs[o.code] & 0xffu) << 16);
    const uint64_t ext = (uint64_t)l.extra | ((uint64_t)m.extra << l.bits) | ((uint64_t)o.extra << (l.bits + m.bits));
    // ---- the three chains, last sequence of the batch first ----
    for (int k = (int)cnt - 1; k >= 0; k--) {
      const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)dl_v, k), dm = (uint32_t)__builtin_amdgcn_readlane((int)dm_v, k);
      const uint32_t dO = (uint32_t)__builtin_amdgcn_readlane((int)do_v, k), fpk = (uint32_t)__builtin_amdgcn_readlane((int)fpk_v, k);
      if (TABLES && my_rle) { if (lane < 3) trans[3 * k + lane] = 0u; }
      else if (lane < 3) {
        const uint32_t d = lane == 0 ? dl : (lane == 1 ? dm : dO);
        const int32_t f = (int32_t)(int8_t)((fpk >> (8u * (uint32_t)lane)) & 0xffu);
        if (first) {
          const uint32_t nb = (d + (1u << 15)) >> 16;
          state = my_st[(int32_t)(((nb << 16) - d) >> nb) + f];
          trans[3 * k + lane] = 0u;
        } else {
          const uint32_t nb = (state + d) >> 16;
          trans[3 * k + lane] = (state & ((1u << nb) - 1u)) | (nb << 16);
          state = my_st[(int32_t)(state >> nb) + f];
        }
      }
      first = false;
    }
    // ---- placement: per sequence [offset-state bits | match-length-state bits | literal-length-state bits | extra bits] ----
    BAMD_LDS_SYNC();
    uint32_t a_bits = 0, a_n = 0;
    if ((uint32_t)lane < cnt) {
      const uint32_t tl = trans[3 * lane], tm = trans[3 * lane + 1], to = trans[3 * lane + 2];
      const uint32_t nO = to >> 16, nm = tm >> 16, nl = tl >> 16;
      a_bits = (to & 0xffffu) | ((tm & 0xffffu) << nO) | ((tl & 0xffffu) << (nO + nm));
      a_n = nO + nm + nl;
    }
    const uint32_t mylen = (uint32_t)lane < cnt ? a_n + nbx : 0u;
    const uint32_t incl = wave_incl_scan_u32(mylen, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    strip[lane] = lane == 0 ? pend : 0u; strip[lane + 64] = 0u;
    if (lane < (int)ZV_STRIP - 128) strip[lane + 128] = 0u;
    BAMD_LDS_SYNC();
    if (mylen) {
      const uint32_t at = npend + (total - incl);            // stream order: the batch's last sequence first
      zv_or_bits(strip, at, (uint64_t)a_bits, a_n);
      zv_or_bits(strip, at + a_n, ext, nbx);
    }
    const uint32_t fill = npend + total, ndw = fill >> 5;     // <= 133 full dwords
    if (pos + 4u * ndw + 8u > room) { ovf = true; break; }
    BAMD_LDS_SYNC();
#pragma unroll
    for (uint32_t i = 0; i < 3u; i++) { const uint32_t w = (uint32_t)lane + 64u * i; if (w < ndw) g_st4(out + pos + 4u * w, strip[w]); }
    pend = uni(strip[ndw]); npend = fill & 31u; pos += 4u * ndw;
    if (base == 0u) break;
  }
  if (ovf) return 0xffffffffu;
  // final states (match length, offset, literal length), the mark bit, the pending bytes
  const uint32_t sll = (uint32_t)__builtin_amdgcn_readlane((int)state, 0), sml = (uint32_t)__builtin_amdgcn_readlane((int)state, 1), sof = (uint32_t)__builtin_amdgcn_readlane((int)state, 2);
  uint64_t acc = (uint64_t)pend; uint32_t nb = npend;
  if (TABLES) {
    acc |= (uint64_t)(sml & ((1u << log_m) - 1u)) << nb; nb += log_m;
    acc |= (uint64_t)(sof & ((1u << log_o) - 1u)) << nb; nb += log_o;
    acc |= (uint64_t)(sll & ((1u << log_l) - 1u)) << nb; nb += log_l;
  } else {
  acc |= (uint64_t)(sml & 63u) << nb; nb += (uint32_t)zenc::kMLLog;
  acc |= (uint64_t)(sof & 31u) << nb; nb += (uint32_t)zenc::kOFLog;
  acc |= (uint64_t)(sll & 63u) << nb; nb += (uint32_t)zenc::kLLLog;
  }
  acc |= 1ull << nb; nb += 1u;
  const uint32_t nbytes = (nb + 7u) >> 3;                    // <= 7
  if (pos + nbytes > room) return 0xffffffffu;
  if ((uint32_t)lane < nbytes) out[pos + (uint32_t)lane] = (uint8_t)(acc >> (8u * (uint32_t)lane));
  return pos + nbytes;
}

// One stream -> one frame.  Returns the frame size, or 0 when it would not be smaller than the input (the split is
// then stored raw by blosc's own rule, blosc.c:703-717).  `seqbuf`: zenc::kBlockMax / 4 entries of this wave.
constexpr uint32_t ZS_SEQCAP = zenc::kBlockMax / 4u;
constexpr int ZS_LDS_BYTES = (int)((sizeof(zenc::CTabs) + 15) / 16 * 16);
// HC: the LZ4HC-grade search (hc_encode_wave) as the match finder, its 24 KiB table in front of the FSE tables; HUF (with TABLES):
// Huffman-coded literals where they are smaller than the raw form
template <bool TABLES = false, bool HC = false, bool HUF = false>
__device__ uint32_t zstd_encode_wave(const gu8* __restrict__ src, uint32_t n, gu8* __restrict__ dst, uint32_t cap, int clevel,
                                     enc_entry_t* tab_generic, BAMD_GAS uint64_t* seqbuf, int lane EPROF_ARG) {
  if (n < 32u || cap < 64u) return 0u;
  if (lane == 0) {
    uint8_t h[16];
    zenc::write_frame_header(h, n);
    for (int i = 0; i < 9; i++) dst[i] = h[i];
  }
  uint32_t op = zenc::kFrameHeader;
  zenc::RepState rep;
  zenc::rep_init(rep);
  for (uint32_t s0 = 0; s0 < n; s0 += zenc::kBlockMax) {
    const uint32_t s1 = s0 + zenc::kBlockMax < n ? s0 + zenc::kBlockMax : n;
    const bool last = s1 == n;
    const uint32_t seg = s1 - s0;
    if (op + zenc::kBlockHeader + zenc::kLitHeader + 16u >= cap) return 0u;
    gu8* bh = dst + op;
    ZsSink z;
    z.lit = bh + zenc::kBlockHeader + zenc::kLitHeader; z.nlit = 0;
    z.litcap = cap - (op + zenc::kBlockHeader + zenc::kLitHeader);
    z.seq = seqbuf; z.nseq = 0; z.seqcap = ZS_SEQCAP;
    const uint32_t covered = HC ? hc_encode_wave<EF_ZSTD>(src, s1, dst, cap, tab_generic, lane, s0, &z)
                                : lz_encode_wave<EF_ZSTD>(src, s1, dst, cap, clevel, tab_generic, lane EPROF_PASS, s0, &z);
    uint32_t bsize = 0xffffffffu;
    const zenc::RepState rep_before = rep;
    if (covered != 0xffffffffu && z.nlit + (s1 - covered) <= z.litcap) {
      wave_copy_disjoint(z.lit + z.nlit, src + covered, s1 - covered, lane);
      z.nlit += s1 - covered;
      if (lane == 0) { uint8_t h[4]; zenc::write_raw_literals_header(h, z.nlit); bh[3] = h[0]; bh[4] = h[1]; bh[5] = h[2]; }
      // the FSE tables sit behind this wave's hash table in LDS (k_encode_streams_t<true> puts them there once)
      const BAMD_LAS zenc::CTabs* T = (const BAMD_LAS zenc::CTabs*)((BAMD_LAS uint8_t*)(void*)tab_generic + (HC ? HC_TAB_BYTES : ENC_TAB_BYTES));
      __builtin_amdgcn_s_waitcnt(0);      // this wave's sequence triples are in memory before other lanes load them
      zs_assign_offset_values(seqbuf, z.nseq, rep, lane);
      __builtin_amdgcn_s_waitcnt(0);
      PROF_LAP(4);                        // Zstd: slot 4 = tail literals + offset values, slot 5 = sequences section
      uint32_t ss;
      uint32_t lit_bytes = zenc::kLitHeader + z.nlit;              // size of the literals section (raw form so far)
      if (TABLES) {
        volatile BAMD_LAS uint32_t* scr = (volatile BAMD_LAS uint32_t*)(BAMD_LAS uint8_t*)(void*)tab_generic;
        gu8* seq_out = z.lit + z.nlit;
        if (HUF) {
          // the section is put together behind the block's triples in the sequence scratch, then moved in front of the sequences
          gu8* stage = (gu8*)(seqbuf + z.nseq);
          const uint32_t hs = zh_literals(z.lit, z.nlit, stage, (ZS_SEQCAP - z.nseq) * 8u, scr, lane);
          if (hs) {
            BAMD_MEM_SYNC();
            wave_copy_disjoint(bh + zenc::kBlockHeader, stage, hs, lane);
            seq_out = bh + zenc::kBlockHeader + hs; lit_bytes = hs;
          }
        }
        // per-block tables (zt_make_tables): worth their description from a few dozen sequences on
        ZsTabs zt;
        zt_make_tables(seqbuf, z.nseq, scr, zt, lane);
        ss = zs_write_sequences_v<true>(seq_out, cap - (uint32_t)(seq_out - dst), seqbuf, z.nseq, T, scr, lane, &zt);
        if (ss != 0xffffffffu) bsize = lit_bytes + ss;
      } else {
      ss = zs_write_sequences_v<false>(z.lit + z.nlit, z.litcap - z.nlit, seqbuf, z.nseq, T, (volatile BAMD_LAS uint32_t*)(BAMD_LAS uint8_t*)(void*)tab_generic, lane);
      if (ss != 0xffffffffu) bsize = zenc::kLitHeader + z.nlit + ss;
      }
      PROF_LAP(5);
    }
    if (bsize >= seg) {                   // no gain: Raw_Block
      if (op + zenc::kBlockHeader + seg >= cap) return 0u;
      BAMD_MEM_SYNC();                    // the copy overwrites what other lanes have just written of the compressed form
      wave_copy_disjoint(bh + zenc::kBlockHeader, src + s0, seg, lane);
      bsize = seg;
      rep = rep_before;                   // a raw block leaves the decoder's repeat offsets alone
      if (lane == 0) { uint8_t h[4]; zenc::write_block_header(h, last, 0u, bsize); bh[0] = h[0]; bh[1] = h[1]; bh[2] = h[2]; }
    } else if (lane == 0) { uint8_t h[4]; zenc::write_block_header(h, last, 2u, bsize); bh[0] = h[0]; bh[1] = h[1]; bh[2] = h[2]; }
    op += zenc::kBlockHeader + bsize;
  }
  return op < n ? op : 0u;
}

