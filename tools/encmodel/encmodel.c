/*
 * encmodel.c — CPU model of the MI355X tile encoder (design tool + kernel debugging aid).
 *
 * NOT the oracle and NOT part of the product: it models OUR OWN wave-parallel algorithm
 * (64-position steps, look-ups before inserts, scalar greedy selection) so that design
 * parameters (tile size, table bits, far-match tables) can be evaluated for compression ratio
 * without a GPU, and so that the HIP kernel's output can be compared byte-for-byte against a
 * sequential statement of the same algorithm.  The token emitters follow the MinLZ format
 * (SPEC.md:68-266).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int tile_log;     /* tile size = 1<<tile_log */
    int hash_bits;    /* near table entries */
    int hash_bytes;   /* 4,5,6 */
    int use_rep;      /* check repeat offset */
    int back_ext;     /* backward extension */
    int far;          /* far matches via first-occurrence epoch tables */
    int far_bits;     /* entries per epoch table */
    int far_stride;   /* insertion stride */
    int far_min;      /* min far match length */
    int epoch_log;    /* epoch size log2 */
    int skip;         /* incompressible skipping */
    int wave;         /* lanes per step (64) */
    int nlevels;      /* leveled tiles: far sources must be in lower-level tiles (0 = unconstrained) */
    int lpat;         /* level pattern id */
    int pw_tiles;     /* >0: far table = per-tile table over the last pw_tiles lower-level tiles (most recent wins) */
    int pw_bits;
    int ways;         /* near table ways (1 or 2: most recent + previous) */
    int lazy;         /* selection: prefer a match at p+1 that is longer by >= lazy (0 = off) */
    int far_all;      /* probe both epochs for every position */
    int preseed;      /* near table pre-seeded with up to this many preceding tiles while they have a lower level */
    int nohole;       /* near matches may read far-copied bytes (the current decoder allows it) */
} params;

static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

#define NOPOS INT32_MIN
static inline uint32_t hashN(uint64_t v, int bytes, int bits) {
    if (bytes == 4) return ((uint32_t)v * 2654435761u) >> (32 - bits);
    if (bytes == 5) return (uint32_t)(((v << 24) * 889523592379ull) >> (64 - bits));
    if (bytes == 6) return (uint32_t)(((v << 16) * 227718039650203ull) >> (64 - bits));
    return (uint32_t)((v * 0xcf1bbcdcb7a56463ull) >> (64 - bits));
}

/* token size models */
static size_t lit_size(size_t n) { if (!n) return 0; return n + (n <= 29 ? 1 : n < 29 + 256 + 1 ? 2 : n < 29 + 65536 + 1 ? 3 : 4); }
static size_t rep_size(size_t len) { return len < 30 ? 1 : len - 30 < 256 ? 2 : len - 30 < 65536 ? 3 : 4; }
static size_t copy_size(size_t off, size_t len) {
    if (off > 65599) { size_t l = len - 4; return l <= 60 ? 4 : l - 60 < 256 ? 5 : l - 60 < 65536 ? 6 : 7; }
    if (off <= 1024) { if (len < 19) return 2; if (len < 274) return 3; return 2 + rep_size(len - 18); }
    size_t l = len - 4; return l <= 60 ? 3 : l - 60 < 256 ? 4 : l - 60 < 65536 ? 5 : 6;
}
static size_t match_cost(size_t lits, size_t off, size_t len, int is_rep) {
    if (is_rep) return lit_size(lits) + rep_size(len);
    if (lits > 0 && off >= 64) {
        if (off <= 65599 && lits <= 4) return 3 + lits + (len > 11 ? rep_size(len - 11) : 0);
        if (off > 65599 && lits <= 3) return copy_size(off, len) + lits;
    }
    return lit_size(lits) + copy_size(off, len);
}

static size_t mlen(const uint8_t* a, const uint8_t* b, size_t max) {
    size_t n = 0;
    while (n + 8 <= max) { uint64_t d = ld64(a + n) ^ ld64(b + n); if (d) return n + (__builtin_ctzll(d) >> 3); n += 8; }
    while (n < max && a[n] == b[n]) n++;
    return n;
}

static const int LPAT[16][16] = {
 {0,1,2,3,0,1,2,3,0,1,2,3,0,1,2,3},{0,2,1,2,0,2,1,2,0,2,1,2,0,2,1,2},{0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1},{0,1,2,3,4,5,6,7,0,1,2,3,4,5,6,7},
 {0,1,2,3,1,2,3,2,0,1,2,3,1,2,3,2},{0,1,2,3,1,2,3,3,0,1,2,3,1,2,3,3},{0,1,2,3,2,1,2,3,0,1,2,3,2,1,2,3},{0,1,2,1,2,3,2,3,0,1,2,1,2,3,2,3},
 {0,1,2,3,1,2,3,2,1,2,3,2,1,2,3,3},{0,1,2,3,1,2,3,1,2,3,1,2,3,1,2,3},{0,1,2,3,2,3,1,2,3,2,3,1,2,3,2,3},{0,1,2,3,3,1,2,3,3,1,2,3,3,1,2,3},
 {0,1,2,2,3,3,1,2,2,3,3,1,2,2,3,3},{0,1,1,2,2,3,3,1,1,2,2,3,3,2,3,3},{0,1,2,3,1,2,3,2,3,1,2,3,2,3,2,3},{0,1,2,3,1,2,3,1,2,3,2,3,1,2,3,3}};
static inline int tile_level(size_t t, const void* Pv);
typedef struct { size_t out; size_t n_near, n_rep, n_far, far_bytes, near_bytes, lit_bytes; size_t depth_hist[16]; } stats;

static inline int tile_level(size_t t, const void* Pv) { const params* P = (const params*)Pv; if (!P->nlevels) return 0; return LPAT[P->lpat][t & 15]; }
/* encode one block of n bytes; returns token-stream size (no header) */
size_t model_block(const uint8_t* src, size_t n, const params* P, stats* st) {
    size_t T = (size_t)1 << P->tile_log;
    size_t ntiles = (n + T - 1) / T;
    size_t total = 0;
    uint32_t* far_tab = NULL;
    size_t E = (size_t)1 << P->epoch_log, nepoch = (n + E - 1) / E;
    if (P->far) {
        int NL = P->nlevels ? P->nlevels : 1;
        far_tab = (uint32_t*)malloc(sizeof(uint32_t) * NL * nepoch << P->far_bits);
        memset(far_tab, 0xff, sizeof(uint32_t) * NL * nepoch << P->far_bits);
        for (size_t q = 0; q + 8 <= n; q += P->far_stride) {
            uint32_t h = hashN(ld64(src + q), 8, P->far_bits);
            int lv = tile_level(q >> P->tile_log, P);
            for (int L2 = lv; L2 < NL; L2++) {
                uint32_t* e = &far_tab[(((size_t)L2 * nepoch + (q >> P->epoch_log)) << P->far_bits) + h];
                if (*e > q) *e = (uint32_t)q;
            }
        }
    }
    uint8_t* depth = (uint8_t*)calloc(n + 8, 1);
    int32_t* table = (int32_t*)malloc(sizeof(int32_t) << P->hash_bits);
    int32_t* table2 = (int32_t*)malloc(sizeof(int32_t) << P->hash_bits);
    uint32_t* pwtab = P->pw_tiles ? (uint32_t*)malloc(sizeof(uint32_t) << P->pw_bits) : NULL;
    uint8_t* hole = (uint8_t*)malloc(T + 8);
    for (size_t t = 0; t < ntiles; t++) {
        size_t base = t * T, tl = n - base < T ? n - base : T;
        const uint8_t* s = src + base;
        for (size_t q = 0; q < ((size_t)1 << P->hash_bits); q++) table[q] = table2[q] = NOPOS;
        long seed_lo = 0;
        if (P->preseed && P->nlevels) {
            int mylv0 = tile_level(t, P); long tt = (long)t - 1; int got = 0;
            while (tt >= 0 && got < P->preseed && tile_level(tt, P) < mylv0) { tt--; got++; }
            seed_lo = -(long)got * (long)T;
            for (long q = seed_lo; q < 0; q++) { uint32_t h = hashN(ld64(s + q), P->hash_bytes, P->hash_bits); if (P->ways > 1) table2[P->ways >= 3 ? h >> (P->ways - 2) : h] = table[h]; table[h] = (int32_t)q; }
        }
        memset(hole, 0, T + 8);
        if (pwtab) {
            memset(pwtab, 0xff, sizeof(uint32_t) << P->pw_bits);
            int mylv0 = tile_level(t, P), got = 0;
            /* oldest first so that the most recent source tile wins */
            long srcs[64]; 
            for (long tt = (long)t - 1; tt >= 0 && got < P->pw_tiles && (t - tt) * T <= 2162687 - T; tt--) if (tile_level(tt, P) < mylv0) srcs[got++] = tt;
            for (int gi = got - 1; gi >= 0; gi--) {
                size_t b0 = (size_t)srcs[gi] * T;
                for (size_t q = b0; q + 8 <= b0 + T && q + 8 <= n; q += P->far_stride) pwtab[hashN(ld64(src + q), 8, P->pw_bits)] = (uint32_t)q;
            }
        }
        size_t cur = 0, next_emit = 0, out = 0, miss = 0;
        size_t rep = 0;
        const int W = P->wave;
        while (cur + 8 <= tl) {
            size_t s0 = cur;
            /* phase 1: all lanes look up */
            int32_t cand[64]; int32_t cand2[64]; uint32_t hh[64]; int valid[64];
            size_t len_[64], off_[64]; int isrep[64], isfar[64];
            for (int i = 0; i < W; i++) {
                size_t p = s0 + i;
                valid[i] = p + 8 <= tl;
                len_[i] = 0; isrep[i] = 0; isfar[i] = 0; off_[i] = 0;
                if (!valid[i]) continue;
                uint64_t v = ld64(s + p);
                hh[i] = hashN(v, P->hash_bytes, P->hash_bits);
                cand[i] = table[hh[i]]; cand2[i] = table2[P->ways >= 3 ? hh[i] >> (P->ways - 2) : hh[i]];
            }
            for (int i = 0; i < W; i++) if (valid[i]) { if (P->ways > 1) { int32_t old = table[hh[i]]; if (old < (long)s0) table2[P->ways >= 3 ? hh[i] >> (P->ways - 2) : hh[i]] = old; } table[hh[i]] = (int32_t)(s0 + i); } /* highest lane wins */
            for (int i = 0; i < W; i++) {
                if (!valid[i]) continue;
                size_t p = s0 + i, maxl = tl - p;
                size_t best = 0, boff = 0; int brep = 0, bfar = 0;
                if (P->use_rep && rep && p >= rep) {
                    size_t l = mlen(s + p, s + p - rep, maxl);
                    if (l >= 4) { best = l; boff = rep; brep = 1; }
                }
                if (cand[i] != NOPOS && cand[i] < (long)p) {
                    size_t l = mlen(s + p, s + cand[i], maxl);
                    if (l >= 4 && (!brep || l > best + 1)) { if (!brep || l > best + 1) { best = l; boff = p - cand[i]; brep = 0; } }
                }
                if (P->ways > 1 && cand2[i] != NOPOS && cand2[i] < (long)p && cand2[i] != cand[i]) {
                    size_t l = mlen(s + p, s + cand2[i], maxl);
                    if (l >= 4 && l > best + 1) { best = l; boff = p - cand2[i]; brep = 0; }
                }
                int mylv = tile_level(t, P);
                if (pwtab && mylv > 0) {
                    uint64_t v = ld64(s + p);
                    uint32_t q = pwtab[hashN(v, 8, P->pw_bits)];
                    if (q != 0xffffffffu && q < base) {
                        size_t off = base + p - q;
                        size_t srcleft = T - (q & (T - 1));
                        size_t ml = maxl < srcleft ? maxl : srcleft;
                        size_t l = mlen(s + p, src + q, ml);
                        if (off <= 2162687 && l >= (size_t)P->far_min && l > best + 2) { best = l; boff = off; brep = 0; bfar = 1; }
                    }
                } else if (P->far && (!P->nlevels || mylv > 0)) {
                    uint64_t v = ld64(s + p);
                    uint32_t h = hashN(v, 8, P->far_bits);
                    size_t ep = (base + p) >> P->epoch_log;
                    for (int k = 0; k < 2; k++) {
                        if ((size_t)k > ep) break;
                        if (k == 1 && !P->far_all && ((base + p) & (E - 1)) >= E / 2) break;
                        size_t LS = P->nlevels ? (size_t)(mylv - 1) : 0;
                        uint32_t q = far_tab[((LS * nepoch + (ep - k)) << P->far_bits) + h];
                        if (q == 0xffffffffu || q >= base) continue; /* strictly before this tile */
                        size_t off = base + p - q;
                        if (off > 2162687) continue;
                        size_t l = mlen(s + p, src + q, maxl);
                        if (l >= (size_t)P->far_min && l > best + 2) { best = l; boff = off; brep = 0; bfar = 1; }
                    }
                }
                len_[i] = best; off_[i] = boff; isrep[i] = brep; isfar[i] = bfar;
            }
            /* phase 2: greedy selection */
            size_t pos = cur; int any = 0;
            for (int i = 0; i < W; i++) {
                size_t p = s0 + i;
                if (p < pos || !valid[i] || len_[i] < 4) continue;
                if (P->lazy && i + 1 < W && valid[i + 1] && len_[i + 1] >= len_[i] + (size_t)P->lazy) continue;
                size_t L = len_[i], off = off_[i]; int r = isrep[i], f = isfar[i];
                size_t pp = p;
                if (P->back_ext) {
                    const uint8_t* a = s + pp; const uint8_t* b = f ? src + (base + pp - off) : s + pp - off;
                    while (pp > next_emit && (f ? (base + pp - off) > 0 : (long)pp - (long)off > seed_lo) && a[-1] == b[-1]) { a--; b--; pp--; L++; }
                }
                /* near matches must not read holes (C1): trim at first hole byte in source */
                if (!f && !P->nohole && off <= pp) {
                    size_t q = pp - off, k = 0;
                    while (k < L && !hole[q + k]) k++;
                    /* overlapping copies read dest bytes which are non-hole by construction */
                    if (k < L) { if (off < L && q + k >= pp) {} else L = k; }
                    if (L < 4) continue;
                }
                size_t lits = pp - next_emit;
                size_t c = match_cost(lits, off, L, r && off == rep);
                if (c >= lits + L + (lits ? (lits <= 29 ? 1 : 2) : 0) && !r) { if (L < 5) continue; }
                out += c;
                if (st) { st->lit_bytes += lits; if (f) { st->n_far++; st->far_bytes += L; } else if (r) { st->n_rep++; st->near_bytes += L; } else { st->n_near++; st->near_bytes += L; } }
                if (f) { memset(hole + pp, 1, L); size_t q = base + pp - off; int dm = 0; for (size_t k = 0; k < L; k++) if (depth[q + k] > dm) dm = depth[q + k];
                    dm++; if (dm > 15) dm = 15; memset(depth + base + pp, dm, L); if (st) st->depth_hist[dm]++; }
                rep = off; pos = pp + L; next_emit = pos; any = 1;
            }
            size_t nxt = s0 + W;
            if (!any) miss++; else miss = 0;
            if (P->skip && miss > 4) nxt += (size_t)W * ((miss - 4) / 2 > 15 ? 15 : (miss - 4) / 2);
            cur = pos > nxt ? pos : nxt;
        }
        if (next_emit < tl) { out += lit_size(tl - next_emit); if (st) st->lit_bytes += tl - next_emit; }
        if (out > tl + 4) out = tl + 4;
        total += out;
    }
    free(table); free(table2); free(hole); free(far_tab); free(depth); free(pwtab);
    if (st) st->out += total;
    return total;
}
