// encmodel2.cpp — sequential CPU statement of the round-2 tile encoder (match kernel + serialize kernel).
//
// NOT the oracle and NOT part of the product: a design tool (ratio of a parameter set without a GPU) and a
// debugging aid (the HIP kernels' token records can be compared with these one for one).  It models OUR OWN
// algorithm: 32 KiB tiles staged in LDS, `sub` independent pieces per tile (one wavefront each, near table
// pre-seeded with the tile's earlier positions), fixed 64-position windows (look-ups before inserts), a greedy
// walk over the candidate mask, token records {match start, length, offset}, and a second pass that extends
// matches backwards, chooses the token forms and writes the bytes (the emitters are the product's mlz_format.h).
//
// build: g++ -O2 -shared -fPIC -o libencmodel2.so encmodel2.cpp -I../../minlz_amd/csrc
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <vector>

#include "mlz_format.h"

using namespace mlz;

namespace {
#ifndef MLZ_TILE_LOG
#define MLZ_TILE_LOG 15
#endif
constexpr uint32_t kTileLog = MLZ_TILE_LOG, kTile = 1u << kTileLog;   // (-DMLZ_TILE_LOG=14: the 16 KiB tile experiment)
constexpr int kEpochLog = 21, kFarTagBits = 9, kLevels = 4;
constexpr uint32_t kFarTagMask = (1u << kFarTagBits) - 1;
constexpr uint32_t kPatternFast = 0xE4E4E4E4u, kPatternDense = 0xEEE7B9E4u;
inline int tile_level_p(uint32_t t, uint32_t pat) { return int((pat >> (2 * (t & 15))) & 3); }

inline uint64_t ld64z(const uint8_t* s, size_t p, size_t n) {  // bytes past n read as zero (the LDS pad)
    uint64_t v = 0;
    if (p + 8 <= n) memcpy(&v, s + p, 8);
    else if (p < n) memcpy(&v, s + p, n - p);
    return v;
}
inline uint32_t hash4(uint64_t v, int bits) { return (uint32_t(v) * 2654435761u) >> (32 - bits); }
struct FarHash { uint32_t idx, tag; };
static int g_hash24 = 0;
inline uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline FarHash far_hash(uint64_t v, int bits) {
    if (g_hash24 == 2) {   // two 32-bit multiplies (mul_lo issues at the same 4 cycles as the 24-bit forms on gfx950)
        const uint32_t lo = uint32_t(v), hi = uint32_t(v >> 32);
        const uint32_t a = (lo * 0x9E3779B1u + hi) * 0x85EBCA77u;
        return {a >> (32 - bits), (a >> (32 - bits - kFarTagBits)) & kFarTagMask};
    }
    if (g_hash24) {
        const uint32_t lo = uint32_t(v), hi = uint32_t(v >> 32);
        uint32_t a = mul24(lo, 0x9E3779u);
        a += mul24(lo >> 8, 0x85EBCBu);
        a += mul24(hi, 0xC2B2AFu);
        a += mul24(hi >> 8, 0x27D4EBu);
        return {a >> (32 - bits), (a >> (32 - bits - kFarTagBits)) & kFarTagMask};
    }
    const uint32_t h = (uint32_t(v) * 0x9E3779B1u) ^ (uint32_t(v >> 32) * 0x85EBCA77u);
    const uint32_t g = h * 0xC2B2AE3Du;
    return {g >> (32 - bits), (g >> (32 - bits - kFarTagBits)) & kFarTagMask};
}
inline uint32_t common8(uint64_t a, uint64_t b) { const uint64_t d = a ^ b; return d ? uint32_t(__builtin_ctzll(d)) >> 3 : 8; }
}  // namespace

struct Params {
    int sub;        // pieces per tile (1, 2, 4, 8)
    int nw;         // windows per iteration (1 or 2): the repeat offset is the one in force before the iteration
    int near_bits;  // near table entries (log2)
    int lazy;       // look-ahead positions (0..3)
    int use_rep;    // repeat-offset probe (in-tile sources only)
    int far;        // far tables
    int lane_cap;   // per-lane extension cap (32)
    int back;       // backward extension limit in the serialize pass (0..8)
    int dense;      // level pattern: 0 fast, 1 dense
    int skip_shift; // growing skip on misses: extra windows = min(7, run >> skip_shift); 0 = off
    int far_gate;   // 1: far candidate only looked at when the near/repeat winner is shorter than 8
    int seed;       // 1: pre-seed the near table with the tile's earlier positions
    int lazy_cost;  // 1: lazy compares length minus token size
    int min_far;    // minimum far match (8)
    int probe_stride; // 1: every position is probed; 2: only even positions are (all positions are still inserted); experiment
    int far_stride2;  // far probes only at positions that are multiples of this (0 / 1: every position); coprime with far_stride so that every (p - q) is found within far_stride * far_stride2 bytes
    int lazy_local;   // 1: the look-ahead does not cross a 64-position window
    int far_hash24;   // 1: far hash from 24-bit multiply-adds
    int far_prev;     // 1: the previous epoch's table is probed as well (LevelBalanced)
    unsigned pattern; // != 0: level pattern override (2 bits per tile, period 16)
    int graded;       // 1: piece k of a tile has (k+1)/sub of the near-table entries (same positions-per-entry for every piece)
    int far_bits;     // far table entries per (epoch, level set), log2 (LevelFastest 17, LevelBalanced 18)
    int far_stride;   // every far_stride-th 8-byte window is entered into the far tables (4 / 2)
    int seed_stride;  // every seed_stride-th earlier position of the tile seeds a piece's near table (2 / 1)
    int pm;           // round-5 experiments (built as kernel variants, measured, NOT kept: DESIGN.md section 3 "Own-phase compares"; the kernels are 0).
                      // 2: own-phase compares — a lane compares 8 raw dwords = 32 - (p & 3) bytes, and far candidates below position 4 are not used (their
                      // bytes are loaded at q - (p & 3)).  1: in addition the four windows of an iteration are the four PHASES — window k = positions
                      // cur + 4 l + k —, all look-ups of the iteration come before its inserts (inserts window by window, lanes in order).
    int min_tiles;    // no tile levels: a far source lies at least this many tiles back (MLZ_OPT_L2_GAP; 0 / 1 = anywhere)
    int near_unit;    // graded near tables: entries per 8 KiB of indexed positions (0: 2^near_bits / sub; the kernels' 12-bit class in phase-major form: 992)
};

struct Rec { uint32_t mp, len, off; };

extern "C" {

// Encodes one block; writes the token stream (no block header) to out (cap >= n + n/8 + 64); returns its size.
// stats[0] = tokens, [1] = literal bytes, [2] = far tokens, [3] = repeat tokens, [4] = positions probed,
// [8] = near candidates passing the tag bit, [9] = near matches >= 4, [10] = near matches of 8 (extended), [11] = far tag hits, [12] = far candidates equal in 8 bytes  (stats: 16 words)
size_t model2_block(const uint8_t* src, size_t n, const Params* P, uint8_t* out, uint64_t* stats) {
    g_hash24 = P->far_hash24;
    // no tile levels (LevelBalanced's default since round 4, kPatternFree of the kernels): pattern 0xffffffff here, or the environment switch
    const bool x_nolevel = P->pattern == 0xffffffffu || getenv("MODEL_NOLEVEL") != nullptr;
    const uint32_t pat = P->pattern ? P->pattern : P->dense ? kPatternDense : kPatternFast;
    const int kFarBits = P->far_bits;
    const size_t ntiles = (n + kTile - 1) >> kTileLog;
    const size_t nepoch = (n + (size_t(1) << kEpochLog) - 1) >> kEpochLog;
    std::vector<uint32_t> far_tab;
    if (P->far && n > kTile) {
        far_tab.assign(size_t(kLevels - 1) * nepoch << kFarBits, 0xffffffffu);
        for (size_t q = 0; q + 8 <= n; q += size_t(P->far_stride)) {
            uint64_t v; memcpy(&v, src + q, 8);
            const FarHash fh = far_hash(v, kFarBits);
            const int lv = x_nolevel ? 0 : tile_level_p(uint32_t(q >> kTileLog), pat);   // MODEL_NOLEVEL: one table of ALL positions (a copy may read any earlier tile: the block becomes a general one for the decoder)
            for (int ls = lv; ls < kLevels - 1; ls++) {
                uint32_t& e = far_tab[((size_t(ls) * nepoch + (q >> kEpochLog)) << kFarBits) + fh.idx];
                const uint32_t val = (uint32_t(q) << kFarTagBits) | fh.tag;
                if (val < e) e = val;
            }
        }
    }
    // design experiments (not part of the kernels): MODEL_WAYS=2 two entries per near bucket (half the buckets), MODEL_LONG=b a second
    // near table of 2^b entries indexed by a hash of 8 bytes (MODEL_LONGBYTES: 5..8), candidates of all of them compared, longest taken
    const int x_heads = getenv("MODEL_FARHEADS") ? atoi(getenv("MODEL_FARHEADS")) : 0;   // a far hit whose offset equals that of the hit 4 lanes before (same 16-lane row) is dropped
    const int x_ways = getenv("MODEL_WAYS") ? atoi(getenv("MODEL_WAYS")) : 1;
    const int x_long = getenv("MODEL_LONG") ? atoi(getenv("MODEL_LONG")) : 0;
    const int x_lbytes = getenv("MODEL_LONGBYTES") ? atoi(getenv("MODEL_LONGBYTES")) : 8;
    // MODEL_MINOFF="a,b,c": tiles of level 0 / 1 / 2 take no near match closer than a / b / c bytes (the decoder's ordered copies of a round of 64
    // tokens then depend on each other less often: fewer passes on the latency-bound low levels)
    // MODEL_PM: the windows of an iteration are PHASE classes — window w holds the positions cur + 4 l + w (round 5: lane l owns one aligned dword group,
    // so the four windows share the lane's raw dwords and a position's phase p & 3 is a compile-time constant); 1: look-ups and inserts window by window,
    // 2: all look-ups of the iteration before all its inserts.  MODEL_PMCAP: the per-lane compare covers 8 raw dwords = 32 - (p & 3) bytes.
    const int x_pm = getenv("MODEL_PM") ? atoi(getenv("MODEL_PM")) : P->pm == 1 ? 2 : 0;
    const int x_pmcap = getenv("MODEL_PMCAP") ? atoi(getenv("MODEL_PMCAP")) : P->pm != 0;
    // MODEL_MINTILES=K (round 5, with no tile levels): a far source lies at least K tiles back (or in the own tile: near matches) — K consecutive tiles of a
    // block then never read each other and a decoder could settle them side by side
    const uint32_t x_mintiles = getenv("MODEL_MINTILES") ? uint32_t(atoi(getenv("MODEL_MINTILES"))) : (x_nolevel && P->min_tiles > 1 ? uint32_t(P->min_tiles) : 1u);
    const int x_l0pieces = getenv("MODEL_L0PIECES") ? atoi(getenv("MODEL_L0PIECES")) : 0;   // 1: the four 8 KiB pieces of a level-0 tile do not see each other (no seeding, no source before the piece)
    uint32_t x_minoff[4] = {0, 0, 0, 0};
    if (getenv("MODEL_MINOFF")) sscanf(getenv("MODEL_MINOFF"), "%u,%u,%u", &x_minoff[0], &x_minoff[1], &x_minoff[2]);
    std::vector<uint16_t> table2(size_t(1) << P->near_bits), ltable(x_long ? size_t(1) << x_long : 1);
    auto lhash = [&](uint64_t v) -> uint32_t { v <<= 8 * (8 - x_lbytes); return uint32_t((v * 0xcf1bbcdcb7a56463ull) >> (64 - x_long)); };
    const uint32_t piece_len = kTile / uint32_t(P->sub);
    const int W = 64, NW = P->nw;
    std::vector<uint16_t> table(size_t(1) << P->near_bits);
    std::vector<Rec> recs;
    size_t o = 0;
    for (size_t t = 0; t < ntiles; t++) {
        const size_t base = t << kTileLog;
        const uint32_t tl = uint32_t(n - base < kTile ? n - base : kTile);
        const uint8_t* s = src + base;
        const int mylv = x_nolevel ? (t > 0 ? 1 : 0) : tile_level_p(uint32_t(t), pat);
        const bool far_tile = !far_tab.empty() && mylv > 0;
        const uint32_t* ftab = far_tile ? &far_tab[(size_t(mylv - 1) * nepoch) << kFarBits] : nullptr;
        for (uint32_t ps = 0; ps < tl; ps += piece_len) {
            const uint32_t pe = ps + piece_len < tl ? ps + piece_len : tl;
            recs.clear();
            std::fill(table.begin(), table.end(), uint16_t(0));
            std::fill(table2.begin(), table2.end(), uint16_t(0)); std::fill(ltable.begin(), ltable.end(), uint16_t(0));
            const uint32_t tsize = P->graded ? uint32_t(((ps / piece_len) + 1) * (P->near_unit ? size_t(P->near_unit) : table.size() / size_t(P->sub))) : uint32_t(table.size());
            const bool mulhi = P->far_hash24 == 2;   // round-3 hashes: near bucket = mul_hi(hash32, table size), far hash from two 32-bit multiplies
            auto tix = [&](uint32_t h) -> uint32_t { return uint32_t((uint64_t(h) * tsize) >> P->near_bits); };
            auto nidx = [&](uint64_t v8, uint32_t& tag) -> uint32_t {
                if (mulhi) {
                    const uint32_t h32 = uint32_t(v8) * 2654435761u;
                    tag = (h32 >> (16 - P->near_bits)) & 0x8000u;
                    return uint32_t((uint64_t(h32) * tsize) >> 32);
                }
                const uint32_t h1 = hash4(v8, P->near_bits + 1);
                tag = (h1 & 1) << 15;
                return tix(h1 >> 1);
            };
            if (P->seed && !(x_l0pieces && mylv == 0))
                for (uint32_t p = 0; p < ps; p += uint32_t(P->seed_stride)) {   // (mlz_encode2.hip.inc: kSeedStride)
                    uint32_t stag;
                    const uint32_t sidx = nidx(ld64z(s, p, tl), stag);
                    if (x_ways == 2) table2[sidx >> 1] = table[sidx >> 1];
                    table[x_ways == 2 ? sidx >> 1 : sidx] = uint16_t(p | stag);
                    if (x_long) ltable[lhash(ld64z(s, p, tl))] = uint16_t(p);
                }
            uint32_t cur = ps, pos = ps, rep = 0;
            // the kernel's far pipeline: candidates exist for an iteration only if its windows were the expected ones
            // two iterations earlier (table entries) and one iteration earlier (candidate bytes); the two start-up trips
            // fill it for the piece's first windows
            const uint32_t kNone = 0xffffffffu;
            uint32_t pf_cur = kNone, cand_cur = kNone;
            if (far_tile) { pf_cur = ps; cand_cur = ps; pf_cur = ps + uint32_t(W * NW); }
            while (cur + 4 <= pe) {
                const bool use_far = far_tile && cand_cur == cur;
                if (far_tile) { const uint32_t nx = cur + uint32_t(W * NW); if (pf_cur == nx) cand_cur = nx; pf_cur = nx + uint32_t(W * NW); }
                uint32_t best[256], boff[256];
                bool valid[256];
                const int NP = W * NW;
                uint32_t eA[4][64], hhA[4][64], tgA[4][64];
                auto PI = [&](int ww, int ll) -> int { return x_pm ? 4 * ll + ww : ww * W + ll; };
                if (x_pm == 2) {   // all look-ups of the iteration, then all its inserts (window by window = phase by phase, lanes in order: what four LDS store instructions do)
                    for (int w = 0; w < NW; w++) for (int l = 0; l < W; l++) {
                        const uint32_t p = cur + PI(w, l);
                        hhA[w][l] = nidx(ld64z(s, p, tl), tgA[w][l]);
                        eA[w][l] = table[hhA[w][l]];
                    }
                    for (int w = 0; w < NW; w++) for (int l = 0; l < W; l++) { const int i = 4 * l + w; if (cur + i + 4 <= pe) table[hhA[w][l]] = uint16_t((cur + i) | tgA[w][l]); }
                }
                for (int w = 0; w < NW; w++) {
                    uint32_t *e = eA[w], *hh = hhA[w], *tg = tgA[w]; uint32_t e2[64], el[64], hl[64];
                    for (int l = 0; l < W; l++) {
                        const uint32_t p = cur + PI(w, l);
                        valid[PI(w, l)] = p + 4 <= pe;
                        if (x_pm == 2) continue;
                        const uint32_t bidx = nidx(ld64z(s, p, tl), tg[l]);
                        hh[l] = x_ways == 2 ? bidx >> 1 : bidx;
                        e[l] = table[hh[l]];
                        e2[l] = x_ways == 2 ? table2[hh[l]] : 0;
                        hl[l] = x_long ? lhash(ld64z(s, p, tl)) : 0; el[l] = x_long ? ltable[hl[l]] : 0;
                    }
                    if (x_pm != 2)
                    for (int l = 0; l < W; l++) if (valid[PI(w, l)]) {
                        if (x_ways == 2) table2[hh[l]] = table[hh[l]];
                        table[hh[l]] = uint16_t((cur + PI(w, l)) | tg[l]);
                        if (x_long) ltable[hl[l]] = uint16_t(cur + PI(w, l));
                    }
                    uint32_t hitoff[64];
                    bool w_far16 = false, w_far12 = false, w_near8 = false; if (stats) stats[13]++;
                    for (int l = 0; l < W; l++) {
                        const int i = PI(w, l);
                        const uint32_t p = cur + i;
                        best[i] = 0; boff[i] = 0; hitoff[l] = 0;
                        if (!valid[i]) continue;
                        if (P->probe_stride > 1 && (p & uint32_t(P->probe_stride - 1))) continue;
                        if (stats) stats[4]++;
                        const uint64_t v = ld64z(s, p, tl);
                        const uint32_t maxl = pe - p;
                        const uint32_t cap_p = uint32_t(P->lane_cap) - (x_pmcap ? (p & 3u) : 0u);
                        const uint32_t lim = maxl < cap_p ? maxl : cap_p;
                        const uint32_t cand = e[l] & 0x7fffu;
                        const bool near_ok = cand < p && (e[l] & 0x8000u) == tg[l] && p - cand >= x_minoff[mylv];
                        const bool rep_ok = P->use_rep && rep != 0 && rep <= p;
                        const uint32_t l_near = near_ok ? common8(v, ld64z(s, cand, tl)) : 0;
                        if (stats) { stats[8] += near_ok; stats[9] += l_near >= 4; stats[10] += l_near == 8; }
                        const uint32_t l_rep = rep_ok ? common8(v, ld64z(s, p - rep, tl)) : 0;
                        const bool brep = l_rep >= 4;
                        uint32_t b = brep ? l_rep : 0, bo = brep ? rep : 0;
                        if (l_near >= 4 && (!brep || l_near > b + 1)) { b = l_near; bo = p - cand; }
                        if (b == 8 && lim > 8) {
                            uint32_t k = 8;
                            while (k < lim && s[p + k] == s[p - bo + k]) k++;   // p + k < pe <= tl
                            b = k;
                        }
                        for (int xc = 0; xc < 2; xc++) {   // experiments: further near candidates, the longest wins
                            uint32_t c2 = 0xffffffffu;
                            if (xc == 0 && x_ways == 2 && (e2[l] & 0x8000u) == tg[l]) c2 = e2[l] & 0x7fffu;
                            if (xc == 1 && x_long) c2 = el[l];
                            if (c2 >= p) continue;
                            uint32_t k = 0;
                            while (k < lim && s[p + k] == s[c2 + k]) k++;
                            if (k >= 4 && k > b) { b = k; bo = p - c2; }
                        }
                        if (b > maxl) b = maxl;
                        if (use_far && !(P->far_gate && b >= 8) && !(P->far_stride2 > 1 && (p % uint32_t(P->far_stride2)))) {
                            const FarHash fh = far_hash(v, kFarBits);
                            const size_t ep0 = (base + p) >> kEpochLog;
                            for (int kk = 0; kk <= (P->far_prev && ep0 > 0 ? 1 : 0); kk++) {
                            const size_t ep = ep0 - size_t(kk);
                            const uint32_t en = ftab[(ep << kFarBits) + fh.idx];
                            if ((en & kFarTagMask) == fh.tag) {
                                if (stats) stats[11]++;
                                const uint32_t fq = en >> kFarTagBits;
                                const uint32_t off = uint32_t(base) + p - fq;
                                const uint32_t left = kTile - (fq & (kTile - 1));
                                if (kk == 0 && fq < base) hitoff[l] = off;
                                if (x_heads && kk == 0 && fq < base && (l & 15) >= x_heads && hitoff[l - x_heads] == off) continue;
                                if (fq < base && off <= kMaxCopy3Offset && left >= 8 && !(x_pmcap && fq < 4) && (uint32_t(t) - (fq >> kTileLog)) >= x_mintiles) {
                                    uint64_t fv; memcpy(&fv, src + fq, 8);
                                    const bool deep = base + p + 40 <= n;  // the kernel looks at a far candidate only when 32 bytes are readable on both sides
                                    if (stats && fv == v) stats[12]++;
                                    if (fv == v && maxl >= 8 && deep) {
                                        const uint32_t lm = left < lim ? left : lim;
                                        uint32_t k = 8;
                                        while (k < lm && s[p + k] == src[fq + k]) k++;
                                        if (k >= 16) w_far16 = true;
                                        if (k >= 12) w_far12 = true;
                                        if (k >= uint32_t(P->min_far) && k > b + 2) { b = k; bo = off; }
                                    }
                                }
                            }
                            }
                        }
                        if (b < 4) b = 0;
                        best[i] = b; boff[i] = bo;
                    }
                    if (stats) { stats[14] += w_far16; stats[15] += w_far12; }
                }
                // lazy
                bool take[256];
                for (int i = 0; i < NP; i++) {
                    take[i] = best[i] >= 4;
                    if (!take[i]) continue;
                    auto gain = [&](int j) -> uint32_t {
                        if (!P->lazy_cost || best[j] < 4) return best[j];
                        return best[j] - (boff[j] > kMaxCopy2Offset ? 4u : boff[j] > kMaxCopy1Offset ? 3u : 2u);
                    };
                    for (int k = 1; k <= P->lazy && i + k < NP && !(P->lazy_local && (i & 63) + k >= 64); k++)
                        if (gain(i + k) > gain(i) + uint32_t(k - 1)) { take[i] = false; break; }
                }
                // greedy walk
                bool any = false;
                for (int i = 0; i < NP; i++) {
                    const uint32_t p = cur + i;
                    if (p < pos || !take[i]) continue;
                    uint32_t L = best[i];
                    const uint32_t off = boff[i];
                    if (L >= uint32_t(P->lane_cap) - (x_pmcap ? 3u : 0u)) {  // cooperative extension
                        uint32_t end = pe;
                        if (off > p) {
                            const uint32_t q = uint32_t(base) + p - off;
                            const uint32_t left = kTile - (q & (kTile - 1));
                            if (p + left < end) end = p + left;
                            while (p + L < end && s[p + L] == src[q + L]) L++;
                        } else {
                            while (p + L < end && s[p + L] == s[p - off + L]) L++;
                        }
                    }
                    if (stats && off > p) { stats[5]++; if (L >= 13) stats[6]++; if (L >= 29) stats[7]++; }
                    recs.push_back({p, L, off});
                    pos = p + L; rep = off; any = true;
                }
                uint32_t next = cur + uint32_t(NP);
                if (pos > next) next = pos & ~63u;
                else if (!any && P->skip_shift) {
                    uint32_t extra = (next - pos) >> P->skip_shift;
                    if (extra > 7) extra = 7;
                    next += 64u * uint32_t(NW) * extra;
                }
                cur = next;
            }
            // ---- serialize pass ----
            const size_t o0 = o;
            uint32_t prev_end = ps, prev_off = 0;
            for (const Rec& r : recs) {
                uint32_t bk = 0;
                {
                    uint32_t room = r.mp - prev_end;
                    if (room > uint32_t(P->back)) room = uint32_t(P->back);
                    uint32_t sroom;
                    const uint8_t* a = s + r.mp;
                    const uint8_t* b;
                    if (r.off <= r.mp) { sroom = r.mp - r.off; if (x_l0pieces && mylv == 0) sroom -= ps; b = s + r.mp - r.off; }
                    else { const uint32_t q = uint32_t(base) + r.mp - r.off; sroom = q & (kTile - 1); b = src + q; }
                    if (room > sroom) room = sroom;
                    if (uint32_t(base) + r.mp - r.off < 8) room = 0;  // the kernel compares the 8 bytes before both positions
                    while (bk < room && a[-1 - int(bk)] == b[-1 - int(bk)]) bk++;
                }
                const uint32_t ms = r.mp - bk, len = r.len + bk;
                const uint32_t lits = ms - prev_end;
                const bool is_rep = r.off == prev_off;
                const Emit e = plan_emit(lits, r.off, len, is_rep);
                for (uint32_t k = 0; k < e.pre.n; k++) out[o++] = uint8_t(e.pre.bits >> (8 * k));
                memcpy(out + o, s + prev_end, lits); o += lits;
                for (uint32_t k = 0; k < e.post.n; k++) out[o++] = uint8_t(e.post.bits >> (8 * k));
                if (stats) { stats[0]++; stats[1] += lits; if (r.off > r.mp) stats[2]++; if (is_rep) stats[3]++; }
                prev_end = ms + len; prev_off = r.off;
            }
            if (prev_end < pe) {
                const uint32_t lits = pe - prev_end;
                const Hdr h = lit_header(lits);
                for (uint32_t k = 0; k < h.n; k++) out[o++] = uint8_t(h.bits >> (8 * k));
                memcpy(out + o, s + prev_end, lits); o += lits;
                if (stats) stats[1] += lits;
            }
            const uint32_t plen = pe - ps;
            const uint32_t raw = lit_header(plen).n + plen;
            if (o - o0 >= raw) {  // piece stored as one literal run
                o = o0;
                const Hdr h = lit_header(plen);
                for (uint32_t k = 0; k < h.n; k++) out[o++] = uint8_t(h.bits >> (8 * k));
                memcpy(out + o, s + ps, plen); o += plen;
            }
        }
    }
    return o;
}

}  // extern "C"
