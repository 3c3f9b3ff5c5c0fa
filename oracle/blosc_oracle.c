/* oracle/blosc_oracle.c — CPU oracle for the c-blosc hot path.  TEST INFRASTRUCTURE ONLY.
 * See blosc_oracle.h for scope and parity status.  Every function cites the reference code it
 * restates (paths relative to /root/reference).  Nothing here is used by the product library.
 *
 * Style note: this is written index-based (positions into byte arrays) rather than with the
 * pointer-walking idiom of the reference; the arithmetic is the same, the code is not.
 */
#include "blosc_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * little helpers
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline int32_t  ldi32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; } /* LE host */
static inline void     sti32(uint8_t* p, int32_t v) { memcpy(p, &v, 4); }

/* ------------------------------------------------------------------------------------------
 * Filters
 * ---------------------------------------------------------------------------------------- */

/* blosc/shuffle-generic.h:32-52 : dest[j*N + i] = src[i*T + j]; tail (bs mod T) copied. */
void orc_shuffle(size_t T, size_t bs, const uint8_t* src, uint8_t* dst) {
  size_t N = bs / T, rem = bs % T;
  for (size_t j = 0; j < T; j++)
    for (size_t i = 0; i < N; i++) dst[j * N + i] = src[i * T + j];
  memcpy(dst + (bs - rem), src + (bs - rem), rem);
}

/* blosc/shuffle-generic.h:61-81 */
void orc_unshuffle(size_t T, size_t bs, const uint8_t* src, uint8_t* dst) {
  size_t N = bs / T, rem = bs % T;
  for (size_t i = 0; i < N; i++)
    for (size_t j = 0; j < T; j++) dst[i * T + j] = src[j * N + i];
  memcpy(dst + (bs - rem), src + (bs - rem), rem);
}

/* blosc/shuffle.c:393-416 + bitshuffle-generic.c:125-139.
 * With N = bs/T elements and N % 8 == 0, output row r = 8*j + b (N/8 bytes long) holds bit b of
 * byte j of every element; element 8*m + k sits in bit k (LSB first) of byte m of the row.
 * When N % 8 != 0 the whole block is copied verbatim.  Returns what the reference returns:
 * N*T on the transposing path (count of bytes processed), N on the memcpy path. */
int orc_bitshuffle(size_t T, size_t bs, const uint8_t* src, uint8_t* dst) {
  size_t N = bs / T;
  if (N % 8) { memcpy(dst, src, bs); return (int)N; }
  size_t rowlen = N / 8;
  memset(dst, 0, N * T);
  for (size_t e = 0; e < N; e++) {
    size_t m = e >> 3; unsigned k = (unsigned)(e & 7);
    for (size_t j = 0; j < T; j++) {
      unsigned v = src[e * T + j];
      for (unsigned b = 0; b < 8; b++)
        dst[(j * 8 + b) * rowlen + m] |= (uint8_t)(((v >> b) & 1u) << k);
    }
  }
  memcpy(dst + N * T, src + N * T, bs - N * T);
  return (int)(N * T);
}

/* blosc/shuffle.c:420-443 + bitshuffle-generic.c:208-220 : exact inverse. */
int orc_bitunshuffle(size_t T, size_t bs, const uint8_t* src, uint8_t* dst) {
  size_t N = bs / T;
  if (N % 8) { memcpy(dst, src, bs); return (int)N; }
  size_t rowlen = N / 8;
  for (size_t e = 0; e < N; e++) {
    size_t m = e >> 3; unsigned k = (unsigned)(e & 7);
    for (size_t j = 0; j < T; j++) {
      unsigned v = 0;
      for (unsigned b = 0; b < 8; b++)
        v |= (unsigned)((src[(j * 8 + b) * rowlen + m] >> k) & 1u) << b;
      dst[e * T + j] = (uint8_t)v;
    }
  }
  memcpy(dst + N * T, src + N * T, bs - N * T);
  return (int)(N * T);
}

/* ------------------------------------------------------------------------------------------
 * LZ4 block codec
 * ---------------------------------------------------------------------------------------- */
#define LZ4_MINMATCH      4
#define LZ4_LASTLITERALS  5     /* lz4.c:245 */
#define LZ4_MFLIMIT       12    /* lz4.c:246 */
#define LZ4_MAXDIST       65535 /* lz4.h LZ4_DISTANCE_MAX */
#define LZ4_64KLIMIT      (65536 + LZ4_MFLIMIT - 1) /* lz4.c:710 */
#define LZ4_SKIPTRIGGER   6     /* lz4.c:711 */
#define LZ4_MAXINPUT      0x7E000000

/* number of equal bytes at a[0..], b[0..] with a limited to `alimit` (lz4.c:LZ4_count) */
static int lz4_common(const uint8_t* s, int a, int b, int alimit) {
  int n = 0;
  while (a + n < alimit && s[a + n] == s[b + n]) n++;
  return n;
}

/* hashes: lz4.c:777-806.  Streams >= 65547 bytes use the 5-byte hash over a 64-bit little
 * endian load into a 4096-entry u32 table, shorter ones the 4-byte hash into 8192 u16 slots. */
static inline uint32_t lz4_hash_small(const uint8_t* p) { return (ld32(p) * 2654435761u) >> (32 - 13); }
static inline uint32_t lz4_hash_big(const uint8_t* p) {
  return (uint32_t)(((ld64(p) << 24) * 889523592379ULL) >> (64 - 12));
}

/* write a 255-run length extension (value v already reduced by 15) */
static int lz4_put_ext(uint8_t* dst, int op, int v) {
  for (; v >= 255; v -= 255) dst[op++] = 255;
  dst[op++] = (uint8_t)v;
  return op;
}

/* LZ4_compress_fast -> LZ4_compress_fast_extState -> LZ4_compress_generic_validated
 * (lz4.c:1453-1469, 1382-1403, 930-1338) for the only mode blosc uses: fresh state, no
 * dictionary, `limitedOutput` when dstcap < LZ4_compressBound(srclen) else `notLimited`. */
int orc_lz4_compress(const uint8_t* src, int n, uint8_t* dst, int cap, int accel) {
  if (accel < 1) accel = 1;
  if (accel > 65537) accel = 65537;
  if ((uint32_t)n > (uint32_t)LZ4_MAXINPUT) return 0;
  const int bound = n + n / 255 + 16;
  const int limited = cap < bound;
  if (n == 0) {                        /* lz4.c:1361-1371 */
    if (limited && cap <= 0) return 0;
    dst[0] = 0;
    return 1;
  }
  const int small = n < LZ4_64KLIMIT;
  uint32_t* tab = (uint32_t*)calloc(4096, sizeof(uint32_t)); /* also holds 8192 u16 */
  uint16_t* tab16 = (uint16_t*)tab;
  if (!tab) return 0;
#define H(p)       (small ? lz4_hash_small(src + (p)) : lz4_hash_big(src + (p)))
#define TGET(h)    (small ? (uint32_t)tab16[h] : tab[h])
#define TPUT(h, v) do { if (small) tab16[h] = (uint16_t)(v); else tab[h] = (uint32_t)(v); } while (0)
#define FAIL()     do { free(tab); return 0; } while (0)

  const int iend = n;
  const int mfl1 = iend - LZ4_MFLIMIT + 1;     /* first position where no match may start */
  const int mlimit = iend - LZ4_LASTLITERALS;  /* matches may not extend past this */
  int ip = 0, anchor = 0, op = 0;
  int match = 0, token = 0;
  uint32_t fwdH;

  if (n < LZ4_MFLIMIT + 1) goto tail;           /* lz4.c:1002 */
  TPUT(H(0), 0);
  ip = 1;
  fwdH = H(1);

  for (;;) {
    /* ---- search: single-probe hash lookup with accelerating stride (lz4.c:1040-1102) ---- */
    {
      int fwd = ip, step = 1, tries = accel << LZ4_SKIPTRIGGER;
      for (;;) {
        uint32_t h = fwdH;
        uint32_t cur = (uint32_t)fwd;
        uint32_t cand = TGET(h);
        ip = fwd;
        fwd += step;
        step = tries++ >> LZ4_SKIPTRIGGER;
        if (fwd > mfl1) goto tail;
        fwdH = H(fwd);
        TPUT(h, cur);
        if (!small && cand + LZ4_MAXDIST < cur) continue;   /* too far (byU32 only) */
        if (ld32(src + cand) == ld32(src + ip)) { match = (int)cand; break; }
      }
    }
    /* ---- catch up backwards (lz4.c:1105-1109) ---- */
    while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { ip--; match--; }

    /* ---- literals (lz4.c:1111-1136) ---- */
    {
      int lit = ip - anchor;
      token = op++;
      if (limited && op + lit + (2 + 1 + LZ4_LASTLITERALS) + lit / 255 > cap) FAIL();
      if (lit >= 15) { dst[token] = 15 << 4; op = lz4_put_ext(dst, op, lit - 15); }
      else dst[token] = (uint8_t)(lit << 4);
      memcpy(dst + op, src + anchor, (size_t)lit);
      op += lit;
    }
  next_match:
    /* ---- offset + match length (lz4.c:1154-1226) ---- */
    {
      int off = ip - match;
      dst[op++] = (uint8_t)off; dst[op++] = (uint8_t)(off >> 8);
      int code = lz4_common(src, ip + LZ4_MINMATCH, match + LZ4_MINMATCH, mlimit);
      ip += code + LZ4_MINMATCH;
      if (limited && op + (1 + LZ4_LASTLITERALS) + (code + 240) / 255 > cap) FAIL();
      if (code >= 15) { dst[token] += 15; op = lz4_put_ext(dst, op, code - 15); }
      else dst[token] += (uint8_t)code;
    }
    anchor = ip;
    if (ip >= mfl1) break;                       /* lz4.c:1233 */
    TPUT(H(ip - 2), (uint32_t)(ip - 2));         /* lz4.c:1236-1242 */
    /* ---- immediate re-test at the new position (lz4.c:1253-1295) ---- */
    {
      uint32_t h = H(ip), cur = (uint32_t)ip, cand = TGET(h);
      TPUT(h, cur);
      if ((small || cand + LZ4_MAXDIST >= cur) && ld32(src + cand) == ld32(src + ip)) {
        token = op++;
        dst[token] = 0;
        match = (int)cand;
        goto next_match;
      }
    }
    fwdH = H(++ip);
  }

tail:
  /* ---- last literals (lz4.c:1302-1329) ---- */
  {
    int run = iend - anchor;
    if (limited && op + run + 1 + (run + 255 - 15) / 255 > cap) FAIL();
    if (run >= 15) { dst[op++] = 15 << 4; op = lz4_put_ext(dst, op, run - 15); }
    else dst[op++] = (uint8_t)(run << 4);
    memcpy(dst + op, src + anchor, (size_t)run);
    op += run;
  }
  free(tab);
  return op;
#undef H
#undef TGET
#undef TPUT
#undef FAIL
}

/* LZ4_decompress_safe, restated from the "safe" loop of LZ4_decompress_generic
 * (lz4.c:2215-2445; decode_full_block, noDict).  Returns bytes written or a negative number.
 * Rules: literal-length extension may not read past iend-15 (lz4.c:2265, 1979-2014); a literal
 * run that gets within 12 bytes of the output end or 8 of the input end must be the final one and
 * end exactly at iend (lz4.c:2279-2318); offset may not reach before the output start
 * (lz4.c:2356); match-length extension may not read past iend-4 (lz4.c:2346); a match may not end
 * within the last 5 output bytes (lz4.c:2423).  Offset 0 is not rejected by the reference (output
 * then unspecified); here it copies from the current position like a byte-wise forward copy. */
static long g_lz4_offset0_seen = 0;   /* test aid: how many offset-0 matches the decoder has executed */
long orc_lz4_offset0_seen(int reset) { long v = g_lz4_offset0_seen; if (reset) g_lz4_offset0_seen = 0; return v; }

int orc_lz4_decompress(const uint8_t* src, int n, uint8_t* dst, int cap) {
  if (src == NULL || cap < 0) return -1;
  if (cap == 0) return (n == 1 && src[0] == 0) ? 0 : -1;
  if (n == 0) return -1;
  int ip = 0, op = 0;
  for (;;) {
    unsigned token = src[ip++];
    long len = token >> 4;
    if (len == 15) {
      int lim = n - 15;
      if (ip >= lim) return -ip - 1;
      unsigned s;
      do { s = src[ip++]; len += s; if (ip > lim) return -ip - 1; } while (s == 255);
    }
    if (op + len > cap - LZ4_MFLIMIT || ip + len > n - (2 + 1 + LZ4_LASTLITERALS)) {
      if (ip + len != n || op + len > cap) return -ip - 1;
      memmove(dst + op, src + ip, (size_t)len);
      op += (int)len;
      break;
    }
    memcpy(dst + op, src + ip, (size_t)len);
    ip += (int)len; op += (int)len;
    int off = src[ip] | (src[ip + 1] << 8);
    ip += 2;
    len = token & 15;
    if (len == 15) {
      int lim = n - LZ4_LASTLITERALS + 1;
      unsigned s;
      do { s = src[ip++]; len += s; if (ip > lim) return -ip - 1; } while (s == 255);
    }
    len += LZ4_MINMATCH;
    if (off > op) return -ip - 1;
    if (op + len > cap - LZ4_LASTLITERALS) return -ip - 1;
    if (off == 0) g_lz4_offset0_seen++;
    for (long k = 0; k < len; k++) dst[op + k] = dst[op - off + k]; /* forward, overlap-safe */
    op += (int)len;
  }
  return op;
}

/* ------------------------------------------------------------------------------------------
 * BloscLZ codec
 * ---------------------------------------------------------------------------------------- */
#define BLZ_MAX_COPY     32
#define BLZ_MAX_DIST     8191
#define BLZ_MAX_FARDIST  (65535 + BLZ_MAX_DIST - 1)   /* blosclz.c:42-44 */
#define BLZ_HASHLOG      14
#define BLZ_HASHLOG2     12

static inline uint32_t blz_hash(uint32_t seq, unsigned bits) { return (seq * 2654435761u) >> (32 - bits); }

/* How far a candidate match runs (blosclz.c:117-243).  `ip` is the first unverified position,
 * `ref` the position it is compared with, `bound` = last index of the buffer.  The reference has
 * two scanners whose results differ by one: the run scanner (distance 1) stops ON the first
 * differing byte, the match scanner stops ONE PAST it (but never past `bound`). */
static int blz_extent(const uint8_t* b, int ip, int bound, int ref, int run) {
  if (run) {
    uint8_t x = b[ip - 1];
    while (ip < bound && b[ref] == x) { ip++; ref++; }
    return ip;
  }
  while (ip < bound) {
    int same = b[ref] == b[ip];
    ref++; ip++;
    if (!same) break;
  }
  return ip;
}

/* blosclz.c:318-418 : cheap estimate of the compression ratio of (at most) the first 4096
 * bytes of b[0..maxlen) with a 4096-entry u16 table, counting output bytes instead of
 * writing them. */
static double blz_probe(const uint8_t* b, int maxlen, int minlen, int ipshift) {
  uint16_t tab[1u << BLZ_HASHLOG2];
  memset(tab, 0, sizeof tab);
  const int limit = (uint16_t)((maxlen > (1 << BLZ_HASHLOG2)) ? (1 << BLZ_HASHLOG2) : maxlen);
  const int bound = limit - 1, iplimit = limit - 12;
  int ip = 0, oc = 5;
  uint8_t copy = 4;
#define PROBE_LITERAL() do { oc++; anchor++; ip = anchor; copy++; \
                             if (copy == BLZ_MAX_COPY) { copy = 0; oc++; } } while (0)
  while (ip < iplimit) {
    int anchor = ip;
    uint32_t h = blz_hash(ld32(b + ip), BLZ_HASHLOG2);
    int ref = tab[h];
    unsigned dist = (unsigned)(anchor - ref);
    tab[h] = (uint16_t)anchor;
    if (dist == 0 || dist >= BLZ_MAX_FARDIST) { PROBE_LITERAL(); continue; }
    if (ld32(b + ref) != ld32(b + ip)) { PROBE_LITERAL(); continue; }
    dist--;
    ip = blz_extent(b, anchor + 4, bound, ref + 4, dist == 0);
    ip -= ipshift;
    int len = ip - anchor;
    if (len < minlen) { PROBE_LITERAL(); continue; }
    if (!copy) oc--;
    copy = 0;
    if (len >= 7) oc += (len - 7) / 255 + 1;
    oc += (dist < BLZ_MAX_DIST) ? 2 : 4;
    tab[blz_hash(ld32(b + ip), BLZ_HASHLOG2)] = (uint16_t)ip;
    ip += 2;
    oc++;
  }
#undef PROBE_LITERAL
  return (double)ip / (double)oc;
}

/* blosclz.c:421-613.  Stream layout (blosclz.c:246-314): ctrl < 32 -> literal run of ctrl+1
 * bytes; otherwise match of length (ctrl>>5)+2 (+ 255-extension bytes when the field is 7) at
 * distance ((ctrl&31)<<8) + next byte + 1, or with the 31/255 escape a 16-bit big-endian
 * distance biased by 8191+1.  The first control byte carries marker bit 5. */
int orc_blosclz_compress(int clevel, const uint8_t* in, int length, uint8_t* out, int maxout,
                         int split_block) {
  static const double min_ratio[10] = {0, 2, 1.5, 1.2, 1.2, 1.2, 1.2, 1.15, 1.1, 1.0};
  static const uint8_t hashlog_of[10] = {0, BLZ_HASHLOG - 2, BLZ_HASHLOG - 1, BLZ_HASHLOG,
      BLZ_HASHLOG, BLZ_HASHLOG, BLZ_HASHLOG, BLZ_HASHLOG, BLZ_HASHLOG, BLZ_HASHLOG};
  /* entropy probe on the last quarter (blosclz.c:425-435) */
  const int maxlen = length / 4;
  const double ratio = blz_probe(in + (length - maxlen), maxlen, 3, 3);
  if (ratio < min_ratio[clevel]) return 0;

  unsigned ipshift = 4, minlen = 4;              /* blosclz.c:445-457 */
  if (!split_block || ratio < 4) { ipshift = 3; minlen = 3; }
  const unsigned hashlog = hashlog_of[clevel];

  if (length < 16 || maxout < 66) return 0;      /* blosclz.c:473-475 */
  uint32_t* tab = (uint32_t*)calloc((size_t)1 << BLZ_HASHLOG, sizeof(uint32_t));
  if (!tab) return 0;

  const int bound = length - 1, iplimit = length - 12;
  int ip = 0, op = 0;
  unsigned copy = 4;
#define BAIL() do { free(tab); return 0; } while (0)
#define EMIT_LITERAL() do { if (op + 2 > maxout) BAIL(); out[op++] = in[anchor++]; ip = anchor; \
    copy++; if (copy == BLZ_MAX_COPY) { copy = 0; out[op++] = BLZ_MAX_COPY - 1; } } while (0)

  out[op++] = BLZ_MAX_COPY - 1;                  /* blosclz.c:481-487 */
  for (int k = 0; k < 4; k++) out[op++] = in[ip++];

  while (ip < iplimit) {
    int anchor = ip;
    uint32_t h = blz_hash(ld32(in + ip), hashlog);
    int ref = (int)tab[h];
    unsigned dist = (unsigned)(anchor - ref);
    tab[h] = (uint32_t)anchor;
    if (dist == 0 || dist >= BLZ_MAX_FARDIST) { EMIT_LITERAL(); continue; }
    if (ld32(in + ref) != ld32(in + ip)) { EMIT_LITERAL(); continue; }
    dist--;
    ip = blz_extent(in, anchor + 4, bound, ref + 4, dist == 0);
    ip -= (int)ipshift;
    unsigned len = (unsigned)(ip - anchor);
    if (len < minlen || (len <= 5 && dist >= BLZ_MAX_DIST)) { EMIT_LITERAL(); continue; }

    if (copy) out[op - (int)copy - 1] = (uint8_t)(copy - 1);   /* close the literal run */
    else op--;
    copy = 0;

    int far = dist >= BLZ_MAX_DIST;
    if (far) dist -= BLZ_MAX_DIST;
    unsigned hi = far ? 31u : (dist >> 8);
    if (len < 7) {
      if (op + (far ? 4 : 2) > maxout) BAIL();
      out[op++] = (uint8_t)((len << 5) + hi);
    } else {
      if (op + 1 > maxout) BAIL();
      out[op++] = (uint8_t)((7u << 5) + hi);
      for (len -= 7; len >= 255; len -= 255) { if (op + 1 > maxout) BAIL(); out[op++] = 255; }
      if (op + (far ? 4 : 2) > maxout) BAIL();
      out[op++] = (uint8_t)len;
    }
    if (far) { out[op++] = 255; out[op++] = (uint8_t)(dist >> 8); out[op++] = (uint8_t)dist; }
    else out[op++] = (uint8_t)dist;

    /* re-seed the table at the match boundary (blosclz.c:567-580) */
    uint32_t seq = ld32(in + ip);
    tab[blz_hash(seq, hashlog)] = (uint32_t)ip++;
    if (clevel == 9) tab[blz_hash(seq >> 8, hashlog)] = (uint32_t)ip++;
    else ip++;

    if (op + 1 > maxout) BAIL();
    out[op++] = BLZ_MAX_COPY - 1;
  }
  while (ip <= bound) {                          /* blosclz.c:589-598 */
    if (op + 2 > maxout) BAIL();
    out[op++] = in[ip++];
    copy++;
    if (copy == BLZ_MAX_COPY) { copy = 0; out[op++] = BLZ_MAX_COPY - 1; }
  }
  if (copy) out[op - (int)copy - 1] = (uint8_t)(copy - 1);
  else op--;
  out[0] |= (1u << 5);                           /* marker, blosclz.c:607 */
  free(tab);
  return op;
#undef BAIL
#undef EMIT_LITERAL
}

/* blosclz.c:679-789.  Returns bytes written, 0 on any violation.  Two quirks are kept: a match
 * is only executed if at least one more input byte follows it (blosclz.c:736) — otherwise the
 * loop ends WITHOUT copying it; and the ">= ip_limit" look-ahead checks of blosclz.c:700-721. */
int orc_blosclz_decompress(const uint8_t* in, int length, uint8_t* out, int maxout) {
  if (length == 0) return 0;
  int ip = 0, op = 0;
  unsigned ctrl = in[ip++] & 31u;
  for (;;) {
    if (ctrl >= 32) {
      int len = (int)(ctrl >> 5) - 1;
      int ofs = (int)(ctrl & 31u) << 8;
      unsigned code;
      if (len == 6) {
        do {
          if (ip + 1 >= length) return 0;
          code = in[ip++];
          len += (int)code;
        } while (code == 255);
      } else if (ip + 1 >= length) return 0;
      code = in[ip++];
      len += 3;
      long dist = (long)ofs + code;                 /* distance - 1 */
      if (code == 255 && ofs == (31 << 8)) {
        if (ip + 1 >= length) return 0;
        ofs = (in[ip] << 8) + in[ip + 1];
        ip += 2;
        dist = (long)ofs + BLZ_MAX_DIST;
      }
      if (op + len > maxout) return 0;
      if ((long)op - dist - 1 < 0) return 0;
      if (ip >= length) break;
      ctrl = in[ip++];
      long from = (long)op - dist - 1;
      for (int k = 0; k < len; k++) out[op + k] = out[from + k];
      op += len;
    } else {
      int run = (int)ctrl + 1;
      if (op + run > maxout) return 0;
      if (ip + run > length) return 0;
      memcpy(out + op, in + ip, (size_t)run);
      op += run; ip += run;
      if (ip >= length) break;
      ctrl = in[ip++];
    }
  }
  return op;
}

/* ------------------------------------------------------------------------------------------
 * Policy: blocksize and split decisions (blosc/blosc.c:922-1060)
 * ---------------------------------------------------------------------------------------- */
#define ORC_L1             (32 * 1024)  /* blosc.c L1 */
#define ORC_MIN_BUFFERSIZE 128
#define ORC_MAX_SPLITS     16
#define ORC_MAX_OVERHEAD   16
#define ORC_MAX_BUFFERSIZE (0x7fffffff - ORC_MAX_OVERHEAD)
#define ORC_MAX_TYPESIZE   255
#define ORC_MAX_BLOCKSIZE  ((0x7fffffff - ORC_MAX_TYPESIZE * 4) / 3)

static int orc_hcr(int codec) { return codec == ORC_LZ4HC || codec == ORC_ZLIB || codec == ORC_ZSTD; }

int orc_split_block(int codec, int typesize, int blocksize, int splitmode) {
  switch (splitmode) {
    case ORC_SPLIT_ALWAYS: return 1;
    case ORC_SPLIT_NEVER:  return 0;
    case ORC_SPLIT_AUTO:
      return (codec == ORC_BLOSCLZ || codec == ORC_SNAPPY) && typesize <= ORC_MAX_SPLITS &&
             blocksize / typesize >= ORC_MIN_BUFFERSIZE;
    case ORC_SPLIT_FWD_COMPAT:
      return codec != ORC_ZSTD && typesize <= ORC_MAX_SPLITS &&
             blocksize / typesize >= ORC_MIN_BUFFERSIZE;
    default: return -1;
  }
}

int orc_compute_blocksize(int clevel, int typesize, int nbytes, int forced, int codec, int splitmode) {
  static const int num[10] = {1, 1, 1, 2, 4, 4, 8, 8, 8, 8}, den[10] = {4, 2, 1, 1, 1, 1, 1, 1, 1, 1};
  if (nbytes < typesize) return 1;
  int bs = nbytes;
  if (forced) {
    bs = forced;
    if (bs < ORC_MIN_BUFFERSIZE) bs = ORC_MIN_BUFFERSIZE;
    if (bs > ORC_MAX_BLOCKSIZE) bs = ORC_MAX_BLOCKSIZE;
  } else if (nbytes >= ORC_L1) {
    bs = ORC_L1;
    if (orc_hcr(codec)) bs *= 2;
    bs = bs * num[clevel] / den[clevel];
    if (clevel == 9 && orc_hcr(codec)) bs *= 2;
  }
  if (clevel > 0 && orc_split_block(codec, typesize, bs, splitmode)) {
    if (bs > (1 << 18)) bs = 1 << 18;
    bs *= typesize;
    if (bs < (1 << 16)) bs = 1 << 16;
    if (bs > 1024 * 1024) bs = 1024 * 1024;
  }
  if (bs > nbytes) bs = nbytes;
  if (bs > typesize) bs = bs / typesize * typesize;
  return bs;
}

/* ------------------------------------------------------------------------------------------
 * Chunk level
 * ---------------------------------------------------------------------------------------- */
static int codec_format(int codec) { /* blosc.h:93-99 */
  switch (codec) {
    case ORC_BLOSCLZ: return 0;
    case ORC_LZ4: case ORC_LZ4HC: return 1;
    case ORC_SNAPPY: return 2;
    case ORC_ZLIB: return 3;
    case ORC_ZSTD: return 4;
  }
  return -1;
}

/* blosc_c, blosc/blosc.c:591-722 */
static int orc_block_c(int codec, int clevel, int flags, int typesize, int bs, int leftover,
                       int ntbytes, int maxbytes, const uint8_t* src, uint8_t* dest, uint8_t* tmp) {
  const int dont_split = (flags >> 4) & 1;
  const uint8_t* data = src;
  if ((flags & 1) && typesize > 1) { orc_shuffle((size_t)typesize, (size_t)bs, src, tmp); data = tmp; }
  else if ((flags & 4) && bs >= typesize) { orc_bitshuffle((size_t)typesize, (size_t)bs, src, tmp); data = tmp; }
  const int nsplits = (!dont_split && !leftover) ? typesize : 1;
  const int neblock = bs / nsplits;
  int ctbytes = 0;
  for (int j = 0; j < nsplits; j++) {
    dest += 4; ntbytes += 4; ctbytes += 4;
    int maxout = neblock;
    if (ntbytes + maxout > maxbytes) {
      maxout = maxbytes - ntbytes;
      if (maxout <= 0) return 0;
    }
    int cb;
    if (codec == ORC_BLOSCLZ) cb = orc_blosclz_compress(clevel, data + j * neblock, neblock, dest, maxout, !dont_split);
    else if (codec == ORC_LZ4) cb = orc_lz4_compress(data + j * neblock, neblock, dest, maxout, 10 - clevel);
    else return -5;
    if (cb > maxout) return -1;
    if (cb < 0) return -2;
    if (cb == 0 || cb == neblock) {
      if (ntbytes + neblock > maxbytes) return 0;
      memcpy(dest, data + j * neblock, (size_t)neblock);
      cb = neblock;
    }
    sti32(dest - 4, cb);
    dest += cb; ntbytes += cb; ctbytes += cb;
  }
  return ctbytes;
}

/* blosc_compress_ctx with one thread: initialize_context_compression (blosc.c:1062-1145),
 * write_compression_header (1148-1247), blosc_compress_context (1250-1279), serial_blosc
 * (803-867). */
int orc_compress(int clevel, int doshuffle, size_t typesize, size_t nbytes, const void* src_,
                 void* dest_, size_t destsize, int codec, size_t forced_bs, int splitmode) {
  const uint8_t* src = (const uint8_t*)src_;
  uint8_t* dest = (uint8_t*)dest_;
  if (nbytes > (size_t)ORC_MAX_BUFFERSIZE) return 0;
  if (destsize < ORC_MAX_OVERHEAD) return 0;
  if (destsize - ORC_MAX_OVERHEAD > nbytes) destsize = nbytes + ORC_MAX_OVERHEAD;
  if (clevel < 0 || clevel > 9) return -10;
  if (doshuffle != 0 && doshuffle != 1 && doshuffle != 2) return -10;
  if (typesize == 0) return -10;
  if (typesize > ORC_MAX_TYPESIZE) typesize = 1;
  const int T = (int)typesize, n = (int)nbytes, maxbytes = (int)destsize;
  const int bs = orc_compute_blocksize(clevel, T, n, (int)forced_bs, codec, splitmode);
  int nblocks = n / bs; const int leftover = n % bs;
  if (leftover > 0) nblocks++;

  const int fmt = codec_format(codec);
  if (fmt < 0 || (codec != ORC_BLOSCLZ && codec != ORC_LZ4)) return -5;
  dest[0] = 2; dest[1] = 1;
  int flags = 0;
  dest[3] = (uint8_t)T;
  sti32(dest + 4, n); sti32(dest + 8, bs);
  int ntbytes = 16 + 4 * nblocks;
  if (clevel == 0) { flags |= 2; ntbytes = 16; }
  if (n < ORC_MIN_BUFFERSIZE) { flags |= 2; ntbytes = 16; }
  if (doshuffle == 1) flags |= 1;
  if (doshuffle == 2) flags |= 4;
  flags |= (!orc_split_block(codec, T, bs, splitmode)) << 4;
  flags |= fmt << 5;
  dest[2] = (uint8_t)flags;

  if ((flags & 2) && n + ORC_MAX_OVERHEAD > maxbytes) return 0;

  uint8_t* tmp = (uint8_t*)malloc((size_t)bs + 16);
  if (!(flags & 2)) {
    int total = ntbytes;
    for (int j = 0; j < nblocks; j++) {
      sti32(dest + 16 + 4 * j, total);
      int bsz = bs, lo = 0;
      if (j == nblocks - 1 && leftover > 0) { bsz = leftover; lo = 1; }
      int cb = orc_block_c(codec, clevel, flags, T, bsz, lo, total, maxbytes, src + (size_t)j * bs,
                           dest + total, tmp);
      if (cb == 0) { total = 0; break; }
      if (cb < 0) { free(tmp); return -1; }
      total += cb;
    }
    ntbytes = total;
    if (ntbytes == 0 && n + ORC_MAX_OVERHEAD <= maxbytes) { flags |= 2; dest[2] = (uint8_t)flags; }
  }
  free(tmp);
  if (flags & 2) {
    if (n + ORC_MAX_OVERHEAD > maxbytes) return 0;
    memcpy(dest + 16, src, (size_t)n);
    ntbytes = 16 + n;
  }
  sti32(dest + 12, ntbytes);
  return ntbytes;
}

/* blosc_d, blosc/blosc.c:725-800.  `tmp` holds one block. */
static int orc_block_d(int flags, int versionlz_fmt, int typesize, int compressedsize, int bs,
                       int leftover, const uint8_t* base, int32_t src_offset, uint8_t* dest,
                       uint8_t* tmp) {
  const int dont_split = (flags >> 4) & 1;
  const int doshuffle = (flags & 1) && typesize > 1;
  const int dobitshuffle = (flags & 4) && bs >= typesize;
  uint8_t* out = (doshuffle || dobitshuffle) ? tmp : dest;
  int nsplits = 1;
  if (!dont_split && typesize <= ORC_MAX_SPLITS && bs / typesize >= ORC_MIN_BUFFERSIZE && !leftover)
    nsplits = typesize;
  const int neblock = bs / nsplits;
  int ntbytes = 0;
  for (int j = 0; j < nsplits; j++) {
    if (src_offset < 0 || src_offset > compressedsize - 4) return -1;
    int cb = ldi32(base + src_offset);
    src_offset += 4;
    if (cb < 0 || cb > compressedsize - src_offset) return -1;
    const uint8_t* s = base + src_offset;
    int nb;
    if (cb == neblock) { memcpy(out, s, (size_t)neblock); nb = neblock; }
    else {
      if (versionlz_fmt == 0) nb = orc_blosclz_decompress(s, cb, out, neblock);
      else if (versionlz_fmt == 4) nb = orc_zstd_decompress(s, cb, out, neblock);   /* zstd_wrap_decompress, blosc.c:515-522 */
      else if (versionlz_fmt == 3) nb = orc_zlib_decompress(s, cb, out, neblock);   /* zlib_wrap_decompress, blosc.c:484-495 */
      else nb = orc_lz4_decompress(s, cb, out, neblock);
      if (nb != neblock) return -2;
    }
    src_offset += cb; out += nb; ntbytes += nb;
  }
  if (doshuffle) orc_unshuffle((size_t)typesize, (size_t)bs, tmp, dest);
  else if (dobitshuffle) orc_bitunshuffle((size_t)typesize, (size_t)bs, tmp, dest);
  return ntbytes;
}

/* initialize_decompress_func, blosc/blosc.c:525-574: BloscLZ, LZ4(+HC), Zlib and Zstd are restated here; Snappy
 * answers -5 like a build configured without it.  All known formats carry version 1. */
static int orc_pick_format(int flags, int versionlz) {
  int fmt = (flags & 0xe0) >> 5;
  if (fmt == 0 || fmt == 1 || fmt == 3 || fmt == 4) return versionlz == 1 ? fmt : -9;
  return -5;
}

/* blosc_run_decompression_with_context + serial_blosc, blosc.c:1435-1518 / 803-867 */
int orc_decompress(const void* src_, void* dest_, size_t destsize) {
  const uint8_t* src = (const uint8_t*)src_;
  uint8_t* dest = (uint8_t*)dest_;
  const int version = src[0], versionlz = src[1], flags = src[2], T = src[3];
  const int n = ldi32(src + 4), bs = ldi32(src + 8), cbytes = ldi32(src + 12);
  if (n == 0) return 0;
  if (bs <= 0 || (size_t)bs > destsize || bs > ORC_MAX_BLOCKSIZE || T <= 0) return -1;
  if (version != 2) return -1;
  if (flags & 0x08) return -1;
  int nblocks = n / bs; const int leftover = n % bs;
  if (leftover > 0) nblocks++;
  if (n > (int32_t)destsize) return -1;
  int fmt = 0;
  if (flags & 2) {
    if (n + ORC_MAX_OVERHEAD != cbytes) return -1;
    if (n < 0) return 0;                   /* a negative size counts nblocks <= 0: serial_blosc's loop (blosc.c:814) runs no block and returns 0 */
    memcpy(dest, src + 16, (size_t)n);     /* serial_blosc:843-848, block by block == one copy */
    return n;
  }
  fmt = orc_pick_format(flags, versionlz);
  if (fmt < 0) return fmt;
  if (nblocks > (cbytes - 16) / 4) return -1;
  uint8_t* tmp = (uint8_t*)malloc((size_t)bs);
  int ntbytes = 0;
  for (int j = 0; j < nblocks; j++) {
    int bsz = bs, lo = 0;
    if (j == nblocks - 1 && leftover > 0) { bsz = leftover; lo = 1; }
    int cb = orc_block_d(flags, fmt, T, cbytes, bsz, lo, src, ldi32(src + 16 + 4 * j),
                         dest + (size_t)j * bs, tmp);
    if (cb < 0) { free(tmp); return -1; }
    ntbytes += cb;
  }
  free(tmp);
  return ntbytes;
}

/* blosc_getitem, blosc/blosc.c:1574-1703 */
int orc_getitem(const void* src_, int start, int nitems, void* dest_) {
  const uint8_t* src = (const uint8_t*)src_;
  uint8_t* dest = (uint8_t*)dest_;
  const int version = src[0], versionlz = src[1], flags = src[2], T = src[3];
  const int n = ldi32(src + 4), bs = ldi32(src + 8), cbytes = ldi32(src + 12);
  const int stop = start + nitems;
  if (version != 2) return -9;
  if (bs <= 0 || bs > n || bs > ORC_MAX_BLOCKSIZE || T <= 0) return -1;
  int nblocks = n / bs; const int leftover = n % bs;
  if (leftover > 0) nblocks++;
  int fmt = 0;
  if (flags & 2) { if (n + ORC_MAX_OVERHEAD != cbytes) return -1; }
  else {
    fmt = orc_pick_format(flags, versionlz);
    if (fmt < 0) return fmt;
    if (nblocks >= (cbytes - 16) / 4) return -1;
  }
  if (start < 0 || start * T > n) return -1;
  if (stop < 0 || stop * T > n) return -1;
  uint8_t* tmp = (uint8_t*)malloc(2 * (size_t)bs);
  uint8_t* blk = tmp + bs;
  int ntbytes = 0;
  for (int j = 0; j < nblocks; j++) {
    int bsz = bs, lo = 0;
    if (j == nblocks - 1 && leftover > 0) { bsz = leftover; lo = 1; }
    int startb = start * T - j * bs, stopb = stop * T - j * bs;
    if (startb >= bs || stopb <= 0) continue;
    if (startb < 0) startb = 0;
    if (stopb > bs) stopb = bs;
    int take = stopb - startb;
    if (flags & 2) memcpy(dest + ntbytes, src + 16 + (size_t)j * bs + startb, (size_t)take);
    else {
      int cb = orc_block_d(flags, fmt, T, cbytes, bsz, lo, src, ldi32(src + 16 + 4 * j), blk, tmp);
      if (cb < 0) { ntbytes = cb; break; }
      memcpy(dest + ntbytes, blk + startb, (size_t)take);
    }
    ntbytes += take;
  }
  free(tmp);
  return ntbytes;
}
