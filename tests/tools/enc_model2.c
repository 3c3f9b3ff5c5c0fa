/* tests/tools/enc_model2.c — CPU model of lz_encode_wave<EF_LZ4> (c-blosc_amd/csrc/k_encode.hip): 64 positions per step,
 * tagged single-entry table, candidates ranked by their first RANK_CAP bytes, the winner maximises (len - lane),
 * re-selection behind a match that ends inside the step, distance-1 runs, backward extension.  On top of it the knobs an
 * LZ4HC-grade search adds (buckets, chains, exact ranking, inserts inside matches, a better parse of the step), so that
 * their worth in ratio can be judged on the CPU before any device code is written.  The output is a real LZ4 block (the
 * callers decode it with the reference's LZ4_decompress_safe).  NOT part of the product.
 *   gcc -O2 -shared -fPIC -o /tmp/enc_model2.so tests/tools/enc_model2.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int hash_bits;     /* head table entries = 1 << hash_bits (device: 11) */
  int tag_bits;      /* 0..16 tag bits checked before a candidate is read (device: 8) */
  int rank_cap;      /* bytes compared when ranking (device: 20); <= 0: exact lengths */
  int minlen;        /* shortest match taken (device: 4 at clevel 9) */
  int accel;         /* 10 - clevel */
  int ways;          /* entries per bucket (FIFO), all tried (device: 1) */
  int depth;         /* 0: no chains; >= 1: walk up to `depth` candidates of the hash chain behind the head */
  int insert;        /* 0: as the device (lanes up to the winner, anchor-2, everything on a miss); 1: every position, also inside
                        matches; 2: every position of the steps visited */
  int instep;        /* 1: positions of the same step are candidates for each other (exact chain order); 0: pre-step state only */
  int near;          /* also try distances 1..near by direct comparison (device: 1 = runs) */
  int parse;         /* 0: device; 1: + earlier matches kept (truncated) in front of the winner; 2: optimal parse of the step */
  int max_dist;      /* 65535 */
  int replace;       /* buckets: 0 = FIFO shift, 1 = the way is the step counter mod ways (no read-modify-write) */
  int groups;        /* > 0 with rank_cap > 0: at most this many capped candidate groups (runs of lanes with one distance) get
                        their exact length per step, the others rank with rank_cap */
  int conflict;      /* 1: of the lanes of one insert call that share a slot only the highest one gets in (what parallel
                        LDS stores do); 0: all of them, in order */
} Knobs2;

static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static int count_eq(const uint8_t* s, int a, int b, int maxlen) { int k = 0; while (k < maxlen && s[a + k] == s[b + k]) k++; return k; }
static int put_ext(uint8_t* d, int op, int v) { for (; v >= 255; v -= 255) d[op++] = 255; d[op++] = (uint8_t)v; return op; }

typedef struct { uint8_t* dst; int op, cap, nseq; } Out;

/* optional capture of the sequences (literal length, match length, distance) for analyses outside the LZ4 format */
static uint32_t* g_cap_buf = 0; static int g_cap_max = 0, g_cap_n = 0;
void enc_model2_capture(uint32_t* buf, int max_triples) { g_cap_buf = buf; g_cap_max = max_triples; g_cap_n = 0; }
int enc_model2_captured(void) { return g_cap_n; }

static int emit_seq(Out* o, const uint8_t* src, int anchor, int pm, int dist, int mlen) {
  int ll = pm - anchor, mc = mlen - 4;
  if (g_cap_buf && g_cap_n < g_cap_max) { g_cap_buf[3 * g_cap_n] = (uint32_t)ll; g_cap_buf[3 * g_cap_n + 1] = (uint32_t)mlen; g_cap_buf[3 * g_cap_n + 2] = (uint32_t)dist; g_cap_n++; }
  if (o->op + 1 + ll + ll / 255 + 1 + 2 + (mc + 240) / 255 + 1 > o->cap) return 0;
  int tok = o->op++;
  o->dst[tok] = (uint8_t)(((ll < 15 ? ll : 15) << 4) | (mc < 15 ? mc : 15));
  if (ll >= 15) o->op = put_ext(o->dst, o->op, ll - 15);
  memcpy(o->dst + o->op, src + anchor, (size_t)ll); o->op += ll;
  o->dst[o->op++] = (uint8_t)dist; o->dst[o->op++] = (uint8_t)(dist >> 8);
  if (mc >= 15) o->op = put_ext(o->dst, o->op, mc - 15);
  o->nseq++;
  return 1;
}

typedef struct {
  const Knobs2* k; const uint8_t* src; int n;
  int32_t* head;     /* [slots][ways] position or -1; way 0 = newest */
  uint16_t* tag;     /* [slots][ways] */
  uint16_t* chain;   /* delta to the previous position of the same slot, 0 = none */
  int step;
} Tab;

static uint32_t mix_of(const Tab* t, int p) { return ld32(t->src + p) * 2654435761u; }
static uint32_t slot_of(const Tab* t, uint32_t m) { return m >> (32 - t->k->hash_bits); }
static uint16_t tag_of(const Tab* t, uint32_t m) { return t->k->tag_bits ? (uint16_t)((m << t->k->hash_bits) >> (32 - t->k->tag_bits)) : 0; }

static void tab_insert(Tab* t, int p) {
  const uint32_t m = mix_of(t, p), h = slot_of(t, m);
  const int W = t->k->ways;
  int32_t* hd = t->head + (size_t)h * W; uint16_t* tg = t->tag + (size_t)h * W;
  if (t->k->replace) { const int w = t->step & (W - 1); hd[w] = p; tg[w] = tag_of(t, m); return; }
  if (hd[0] == p) return;
  if (t->k->depth > 0) { int d = hd[0] >= 0 ? p - hd[0] : 0; t->chain[p] = (uint16_t)((d > 0 && d <= t->k->max_dist) ? d : 0); }
  for (int w = W - 1; w > 0; w--) { hd[w] = hd[w - 1]; tg[w] = tg[w - 1]; }
  hd[0] = p; tg[0] = tag_of(t, m);
}

static void tab_insert_lanes(Tab* t, int ip, int l0, int l1, const int* live) {
  for (int l = l0; l <= l1 && l < 64; l++) {
    if (!live[l]) continue;
    if (t->k->conflict) {
      const uint32_t h = slot_of(t, mix_of(t, ip + l));
      int later = 0;
      for (int j = l + 1; j <= l1 && j < 64; j++) if (live[j] && slot_of(t, mix_of(t, ip + j)) == h) { later = 1; break; }
      if (later) continue;
    }
    tab_insert(t, ip + l);
  }
}

int enc_model2(const uint8_t* src, int n, uint8_t* dst, int cap, const Knobs2* k, int* nseq_out, long* work_out) {
  if (n < 13) return 0;
  const int last_start = n - 12, mlimit = n - 5;
  const int tabn = 1 << k->hash_bits, W = k->ways;
  Tab t = {k, src, n, (int32_t*)malloc((size_t)tabn * W * 4), (uint16_t*)calloc((size_t)tabn * W, 2), (uint16_t*)calloc((size_t)n + 64, 2), 0};
  for (int i = 0; i < tabn * W; i++) t.head[i] = -1;
  Out o = {dst, 0, cap, 0};
  long work = 0;                                            /* candidate reads (each = one 20-byte gather on the device) */
  int ip = 0, anchor = 0, nfail = 0, ins_from = 0;
  while (ip <= last_start) {
    int cand[64], len[64], live[64];
    int gcount = 0, gdist[64], glane[64], glen[64];   /* capped groups resolved in this step */
    t.step++;
    if (k->insert == 1) { for (int q = ins_from; q < ip && q <= last_start; q++) tab_insert(&t, q); if (ins_from < ip) ins_from = ip; }
    /* ---- candidates: every lane against the state left by the previous steps (+ the earlier lanes of this step) ---- */
    for (int l = 0; l < 64; l++) {
      const int p = ip + l;
      live[l] = p <= last_start; cand[l] = 0; len[l] = 0;
      if (!live[l]) continue;
      const int limit = mlimit - p;
      const int rank_lim = (k->rank_cap <= 0 || limit < k->rank_cap) ? limit : k->rank_cap;
      const uint32_t m = mix_of(&t, p), h = slot_of(&t, m); const uint16_t tg = tag_of(&t, m);
      int best = 0, bestc = 0;
      int tries = k->depth > 0 ? k->depth : 1 << 30;
      if (k->instep) for (int j = l - 1; j >= 0 && tries > 0; j--) if (slot_of(&t, mix_of(&t, ip + j)) == h) {
        tries--;
        if (tag_of(&t, mix_of(&t, ip + j)) != tg) continue;
        work++;
        int mm = count_eq(src, p, ip + j, rank_lim);
        if (mm > best) { best = mm; bestc = ip + j; }
      }
      for (int w = 0; w < W && tries > 0; w++) {
        int c = t.head[(size_t)h * W + w];
        if (c < 0 || c >= p || p - c > k->max_dist) continue;
        const int tag_ok = t.tag[(size_t)h * W + w] == tg;
        int first = 1;
        while (tries-- > 0) {
          if (!first || tag_ok) {
            work++;
            int mm = count_eq(src, p, c, rank_lim);
            if (k->groups > 0 && k->rank_cap > 0 && mm == k->rank_cap && mm < limit) {
              /* same distance as a resolved group that still covers this lane: exact length for free; else resolve a new group */
              int g, hit = 0;
              for (g = 0; g < gcount; g++) if (gdist[g] == p - c && glane[g] <= l && glane[g] + glen[g] - l >= mm) { mm = glane[g] + glen[g] - l; hit = 1; break; }
              if (!hit && gcount < k->groups) { gdist[gcount] = p - c; glane[gcount] = l; glen[gcount] = count_eq(src, p, c, limit); mm = glen[gcount]; gcount++; }
            }
            if (mm > best) { best = mm; bestc = c; }
          }
          first = 0;
          if (k->depth == 0 || w != 0) break;
          const int dd = t.chain[c];
          if (dd == 0 || p - (c - dd) > k->max_dist) break;
          c -= dd;
        }
      }
      if (best < k->minlen) best = 0;
      for (int d = 1; d <= k->near && d <= p; d++) {
        int mm = count_eq(src, p, p - d, rank_lim);
        if (k->groups > 0 && k->rank_cap > 0 && mm == k->rank_cap && mm < limit) {
          int g, hit = 0;
          for (g = 0; g < gcount; g++) if (gdist[g] == d && glane[g] <= l && glane[g] + glen[g] - l >= mm) { mm = glane[g] + glen[g] - l; hit = 1; break; }
          if (!hit && gcount < k->groups) { gdist[gcount] = d; glane[gcount] = l; glen[gcount] = count_eq(src, p, p - d, limit); mm = glen[gcount]; gcount++; }
        }
        if (mm >= k->minlen && mm > best) { best = mm; bestc = p - d; }
      }
      len[l] = best; cand[l] = bestc;
    }
    const int step_end = ip + 64;
    if (k->insert >= 1) { for (int l = 0; l < 64; l++) if (live[l]) tab_insert(&t, ip + l); ins_from = step_end; }
    /* ---- parse ---- */
    int lane_lo = 0, any = 0;
    int plan_lane[64], plan_len[64], nplan = 0, pi = 0;
    if (k->parse == 2) {
      /* cost[i]: bytes written for the step's positions i..63 (and what a match reaching past the step saves there) */
      double cost[64 + 1]; int choice[64];
      cost[64] = 0;
      for (int i = 63; i >= 0; i--) {
        cost[i] = 1.0 + 1.0 / 255 + cost[i + 1]; choice[i] = 0;
        if (!live[i] || !len[i]) continue;
        for (int m2 = 4; m2 <= len[i]; m2++) {
          if (i + m2 > 64 && m2 != len[i]) continue;               /* beyond the step only the full length */
          double c = 3.0 + (m2 >= 19 ? 1 + (m2 - 19) / 255 : 0) + (i + m2 >= 64 ? -0.95 * (i + m2 - 64) : cost[i + m2]);
          if (c < cost[i]) { cost[i] = c; choice[i] = m2; }
        }
      }
      for (int i = 0; i < 64;) { if (choice[i]) { plan_lane[nplan] = i; plan_len[nplan++] = choice[i]; i += choice[i]; } else i++; }
    }
    for (;;) {
      int f = -1, flen = 0;
      if (k->parse == 2) {
        if (pi >= nplan) break;
        f = plan_lane[pi]; flen = plan_len[pi]; pi++;
        if (f < lane_lo) { /* a backward extension ate into it */ int cut = lane_lo - f; if (flen - cut < 4) continue; f += cut; flen -= cut; cand[f] = cand[f - cut] + cut; }
      } else {
        int bestkey = 0;
        for (int l = lane_lo; l < 64; l++) if (len[l]) { int key = ((len[l] + 64 - l) << 6) | (63 - l); if (key > bestkey) { bestkey = key; f = l; } }
        if (f < 0) break;
        flen = len[f];
        if (k->parse == 1) {
          /* an earlier match that fits (possibly truncated) in front of the winner goes first */
          for (int g = lane_lo; g < f; g++) if (len[g]) {
            int tl = len[g] < f - g ? len[g] : f - g;
            if (tl >= 4) { f = g; flen = tl; break; }
          }
        }
      }
      any = 1;
      if (k->insert == 0) tab_insert_lanes(&t, ip, lane_lo, f, live);
      int pm = ip + f, cm = cand[f];
      int mlen = flen;
      const int full = (flen == len[f]);
      if (full && k->rank_cap > 0 && len[f] >= k->rank_cap && pm + len[f] < mlimit)
        mlen += count_eq(src, pm + len[f], cm + len[f], mlimit - (pm + len[f]));
      int maxb = pm - anchor; if (cm < maxb) maxb = cm; if (maxb > 64) maxb = 64;
      int back = 0; while (back < maxb && src[pm - 1 - back] == src[cm - 1 - back]) back++;
      pm -= back; cm -= back; mlen += back;
      if (!emit_seq(&o, src, anchor, pm, pm - cm, mlen)) goto fail;
      anchor = pm + mlen;
      if (anchor >= step_end) break;
      lane_lo = anchor - ip;
      if (k->insert == 0 && lane_lo >= 2 && live[lane_lo - 2]) tab_insert(&t, ip + lane_lo - 2);
    }
    if (!any) {
      if (k->insert == 0) tab_insert_lanes(&t, ip, 0, 63, live);
      nfail++;
      int adv = 1 + nfail * k->accel / 16; if (adv > 16) adv = 16;
      ip += 64 * adv;
      continue;
    }
    nfail = 0;
    if (anchor >= step_end) {
      ip = anchor;
      if (k->insert == 0 && ip - 2 <= last_start) tab_insert(&t, ip - 2);
    } else {
      if (k->insert == 0) tab_insert_lanes(&t, ip, lane_lo, 63, live);
      ip = step_end;
    }
  }
  {
    int run = n - anchor;
    if (o.op + run + 1 + (run + 240) / 255 > cap) goto fail;
    dst[o.op++] = (uint8_t)((run < 15 ? run : 15) << 4);
    if (run >= 15) o.op = put_ext(dst, o.op, run - 15);
    memcpy(dst + o.op, src + anchor, (size_t)run); o.op += run;
  }
  free(t.head); free(t.tag); free(t.chain);
  if (nseq_out) *nseq_out = o.nseq;
  if (work_out) *work_out = work;
  return o.op < n ? o.op : 0;
fail:
  free(t.head); free(t.tag); free(t.chain);
  return 0;
}
