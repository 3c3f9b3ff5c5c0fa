/* tests/tools/enc_model.c — CPU model of the wave-parallel LZ4 match finder of
 * c-blosc_amd/csrc/k_encode.hip (64 positions probed per step against a table that is updated once per
 * step).  Used to explore ratio vs. design knobs without a GPU; output is a real LZ4 block, so the
 * model is checked by decoding it with the oracle.  NOT part of the product.
 *   gcc -O2 -shared -fPIC -o /tmp/enc_model.so tests/tools/enc_model.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

typedef struct {
  int hash_bits;      /* table entries = 1 << hash_bits */
  int entry_bits;     /* 16: positions mod 65536, 32: full */
  int near_mask;      /* bit d set: try distance d+1 .. (bit0 = distance 1) ; uses power-of-two list below */
  int select;         /* 0 first hit, 1 best (longest, capped probe) among first K hits */
  int probe_cap;      /* bytes compared when ranking hits */
  int insert_in_match;/* 0 none, 1 every 'stride'-th position inside matches */
  int stride;
  int accel_div;      /* fail skip: adv = 1 + nfail*accel/accel_div */
  int accel;
  int ways;           /* 1 or 2 table ways (second = previous occupant) */
} Knobs;

static int count_fwd(const uint8_t* s, int a, int b, int lim) { int n = 0; while (a + n < lim && s[a + n] == s[b + n]) n++; return n; }

static int put_ext(uint8_t* d, int op, int v) { for (; v >= 255; v -= 255) d[op++] = 255; d[op++] = (uint8_t)v; return op; }

int enc_model(const uint8_t* src, int n, uint8_t* dst, int cap, const Knobs* k, int* nseq_out) {
  if (n < 13) return 0;
  const int last_start = n - 12, mlimit = n - 5;
  const int tabn = 1 << k->hash_bits;
  uint32_t* tab = (uint32_t*)calloc((size_t)tabn * 2, 4);
  uint32_t* tab2 = tab + tabn;
  int ip = 0, anchor = 0, op = 0, nfail = 0, nseq = 0;
#define HASH(v) (((v) * 2654435761u) >> (32 - k->hash_bits))
  while (ip <= last_start) {
    int cand[64], hit[64], len[64];
    /* phase 1: all lanes look up the table state left by previous steps */
    for (int l = 0; l < 64; l++) {
      hit[l] = 0; cand[l] = 0; len[l] = 0;
      int p = ip + l;
      if (p > last_start) continue;
      uint32_t seq = ld32(src + p), h = HASH(seq);
      for (int w = 0; w < k->ways && !hit[l]; w++) {
        uint32_t e = (w ? tab2 : tab)[h];
        int c;
        if (k->entry_bits == 16) { uint32_t d = ((uint32_t)p - e) & 0xffffu; if (d == 0 || d > (uint32_t)p) continue; c = p - (int)d; }
        else { c = (int)e - 1; if (c < 0 || p - c > 65535 || c >= p) continue; }
        if (ld32(src + c) == seq) { hit[l] = 1; cand[l] = c; }
      }
      if (!hit[l]) for (int d = 1; d <= 32; d <<= 1) if ((k->near_mask & d) && p >= d && ld32(src + p - d) == seq) { hit[l] = 1; cand[l] = p - d; break; }
    }
    int f = -1;
    if (k->select == 0) { for (int l = 0; l < 64; l++) if (hit[l]) { f = l; break; } }
    else if (k->select == 2) {
      /* rank by in-batch evidence only: consecutive hit lanes with the same distance form a match of
       * length >= run + 4 that starts at the first of them; a run that reaches the end of the batch is
       * "long" (bonus = probe_cap).  gain = estimated length - literals left before it. */
      int best = -1000000;
      for (int l = 0; l < 64; l++) if (hit[l]) {
        int d = ip + l - cand[l], r = 0;
        while (l + r + 1 < 64 && hit[l + r + 1] && (ip + l + r + 1 - cand[l + r + 1]) == d) r++;
        int est = r + 4;
        if (l + r + 1 >= 64 || ip + l + r + 1 > last_start) est += k->probe_cap;
        int gain = est - l;
        if (gain > best) { best = gain; f = l; }
      }
    }
    else {
      /* rank: gain = matched bytes (capped) - lane index (literals before it) */
      int best = -1000000;
      for (int l = 0; l < 64; l++) if (hit[l]) {
        int p = ip + l;
        int m = 4 + count_fwd(src, p + 4, cand[l] + 4, (p + k->probe_cap < mlimit) ? p + k->probe_cap : mlimit);
        int gain = m - l;           /* bytes covered by match minus literals it leaves uncovered before it */
        if (gain > best) { best = gain; f = l; }
      }
    }
    /* phase 2: insert.  stride/insert_in_match == 2: only lanes up to the chosen match start (LZ4 inserts
     * nothing inside a match, which is what lets it later find the START of a repeated run) */
    for (int l = 0; l < 64; l++) {
      int p = ip + l;
      if (p > last_start) continue;
      if (k->insert_in_match == 2 && f >= 0 && l > f) continue;
      uint32_t h = HASH(ld32(src + p));
      if (k->ways == 2) tab2[h] = tab[h];
      tab[h] = (k->entry_bits == 16) ? (uint32_t)(p & 0xffff) : (uint32_t)(p + 1);
    }
    if (f < 0) { nfail++; int adv = 1 + nfail * k->accel / k->accel_div; if (adv > 16) adv = 16; ip += 64 * adv; continue; }
    nfail = 0;
    int pm = ip + f, cm = cand[f];
    while (pm > anchor && cm > 0 && src[pm - 1] == src[cm - 1]) { pm--; cm--; }
    int mlen = 4 + count_fwd(src, pm + 4, cm + 4, mlimit);
    int ll = pm - anchor, mc = mlen - 4;
    if (op + 1 + ll + 8 + ll / 255 > cap) { free(tab); return 0; }
    int tok = op++;
    dst[tok] = (uint8_t)(((ll < 15 ? ll : 15) << 4) | (mc < 15 ? mc : 15));
    if (ll >= 15) op = put_ext(dst, op, ll - 15);
    memcpy(dst + op, src + anchor, (size_t)ll); op += ll;
    int off = pm - cm; dst[op++] = (uint8_t)off; dst[op++] = (uint8_t)(off >> 8);
    if (op + 6 + (mc + 240) / 255 > cap) { free(tab); return 0; }
    if (mc >= 15) op = put_ext(dst, op, mc - 15);
    nseq++;
    if (k->insert_in_match == 2 && pm + mlen - 2 <= last_start && pm + mlen - 2 > pm) { int q = pm + mlen - 2; uint32_t h = HASH(ld32(src + q)); if (k->ways == 2) tab2[h] = tab[h]; tab[h] = (k->entry_bits == 16) ? (uint32_t)(q & 0xffff) : (uint32_t)(q + 1); }
    if (k->insert_in_match == 1) for (int q = pm + 1; q < pm + mlen && q <= last_start; q += k->stride) { uint32_t h = HASH(ld32(src + q)); if (k->ways == 2) tab2[h] = tab[h]; tab[h] = (k->entry_bits == 16) ? (uint32_t)(q & 0xffff) : (uint32_t)(q + 1); }
    anchor = pm + mlen; ip = anchor;
  }
  int run = n - anchor;
  if (op + run + 1 + (run + 240) / 255 > cap) { free(tab); return 0; }
  dst[op++] = (uint8_t)((run < 15 ? run : 15) << 4);
  if (run >= 15) op = put_ext(dst, op, run - 15);
  memcpy(dst + op, src + anchor, (size_t)run); op += run;
  free(tab);
  if (nseq_out) *nseq_out = nseq;
  return op < n ? op : 0;
}
