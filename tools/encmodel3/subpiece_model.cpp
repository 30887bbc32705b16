// subpiece_model.cpp — CPU cost model for the "lane-sequential greedy parsers over sub-pieces sharing the wave's LDS table" division of the
// match kernel's work (round-5 review, item 3b).  Test infrastructure, not product.  It prices, on the same input and with the same size
// model, two tile-local match finders (no far tables, no lazy step — both would sit on top of either):
//   A  "all positions": what match_tiles_kernel does — every position of an 8 KiB piece probes a 2^HB-entry table of most-recent positions
//      (seeded with every 2nd position of the tile's earlier pieces), candidates verified, greedy first-hit parse over the candidates;
//   B  "lane-sequential": 64 lanes, lane i parses sub-piece i of SUB bytes with the reference's greedy loop (probe at p, on a hit emit and
//      jump past it, else p += 1), all lanes in lockstep sharing ONE table (look-ups of a step before its inserts), matches clipped at the
//      sub-piece's end (B1) or allowed to run on to the piece's end, the next lanes' tokens inside them dropped by the serializer (B2).
// Output: bytes of the token stream under the MinLZ size model (copy1 2 B for offsets <= 1024 and lengths 4..18, copy2 3 B up to 64 KiB,
// literal runs 1 + n, + 1 / 2 extra bytes for long lengths) and, for B, the number of lockstep steps per piece (the divergence price: a
// wave runs as long as its slowest lane).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static const uint32_t kTile = 32768, kPiece = 8192;

static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t hash4(uint32_t v, int hb) { return (v * 2654435761u) >> (32 - hb); }
static uint32_t mlen(const uint8_t* a, const uint8_t* b, uint32_t max) { uint32_t n = 0; while (n < max && a[n] == b[n]) n++; return n; }

static uint64_t lit_cost(uint32_t n) { return n == 0 ? 0 : n + (n < 30 ? 1 : n < 286 ? 2 : 3); }
static uint64_t copy_cost(uint32_t off, uint32_t len) {
    if (off <= 1024 && len <= 18) return 2;
    uint64_t c = off <= 65536 + 63 ? 3 : 4;
    if (len > 64) c += len > 64 + 255 ? 2 : 1;   // (approximation of the extended-length forms)
    return c;
}

struct Tok { uint32_t pos, len, off; };

static uint64_t size_of(const std::vector<Tok>& t, uint32_t ps, uint32_t pe) {
    uint64_t sz = 0; uint32_t at = ps;
    for (const Tok& k : t) {
        if (k.pos < at) continue;                       // (B2: a token inside an earlier match is dropped)
        sz += lit_cost(k.pos - at) + copy_cost(k.off, k.len);
        at = k.pos + k.len;
    }
    return sz + lit_cost(pe - at);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: subpiece_model file [hb=12]\n"); return 2; }
    const int hb = argc > 2 ? atoi(argv[2]) : 12;
    FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
    std::vector<uint8_t> d; { uint8_t buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n); } fclose(f);
    d.resize(d.size() + 64, 0);
    const size_t N = d.size() - 64;
    uint64_t szA = 0, szB1[3] = {0, 0, 0}, szB2[3] = {0, 0, 0}, steps[3] = {0, 0, 0}, probesB[3] = {0, 0, 0}, pieces = 0;
    const uint32_t subs[3] = {128, 256, 512};
    std::vector<uint16_t> tab(1u << hb);
    for (size_t t0 = 0; t0 + kTile <= N; t0 += kTile) {
        const uint8_t* s = d.data() + t0;
        for (uint32_t ps = 0; ps < kTile; ps += kPiece) {
            const uint32_t pe = ps + kPiece;
            pieces++;
            auto seed = [&]() { std::fill(tab.begin(), tab.end(), 0xffff); for (uint32_t q = 0; q < ps; q += 2) tab[hash4(ld32(s + q), hb)] = uint16_t(q); };
            // ---- A ----
            {
                seed();
                std::vector<Tok> tk; uint32_t next = ps;
                for (uint32_t p = ps; p + 4 <= pe; p++) {
                    const uint32_t h = hash4(ld32(s + p), hb); const uint32_t c = tab[h]; tab[h] = uint16_t(p);
                    if (p < next || c == 0xffff) continue;
                    const uint32_t l = mlen(s + p, s + c, pe - p);
                    if (l >= 4) { tk.push_back({p, l, p - c}); next = p + l; }
                }
                szA += size_of(tk, ps, pe);
            }
            // ---- B: 64 / (SUB / 128) lanes in lockstep ----
            for (int v = 0; v < 3; v++) {
                const uint32_t SUB = subs[v], L = kPiece / SUB;
                for (int clip = 0; clip < 2; clip++) {
                    seed();
                    std::vector<uint32_t> p(L), end(L);
                    std::vector<std::vector<Tok>> tk(L);
                    for (uint32_t i = 0; i < L; i++) { p[i] = ps + i * SUB; end[i] = p[i] + SUB; }
                    uint64_t st = 0, pr = 0;
                    for (;;) {
                        bool any = false;
                        std::vector<uint32_t> cand(L, 0xffff);
                        for (uint32_t i = 0; i < L; i++) if (p[i] + 4 <= end[i]) { any = true; cand[i] = tab[hash4(ld32(s + p[i]), hb)]; pr++; }
                        if (!any) break;
                        st++;
                        for (uint32_t i = 0; i < L; i++) if (p[i] + 4 <= end[i]) tab[hash4(ld32(s + p[i]), hb)] = uint16_t(p[i]);
                        for (uint32_t i = 0; i < L; i++) {
                            if (p[i] + 4 > end[i]) continue;
                            const uint32_t c = cand[i];
                            uint32_t l = 0;
                            if (c != 0xffff && c < p[i]) l = mlen(s + p[i], s + c, (clip ? end[i] : pe) - p[i]);
                            if (l >= 4) {
                                tk[i].push_back({p[i], l, p[i] - c});
                                // the reference indexes two positions inside a match (encode_l1.go:214-230): its second byte and its last-but-one
                                if (p[i] + 1 + 4 <= pe) tab[hash4(ld32(s + p[i] + 1), hb)] = uint16_t(p[i] + 1);
                                if (l >= 3 && p[i] + l - 2 + 4 <= pe) tab[hash4(ld32(s + p[i] + l - 2), hb)] = uint16_t(p[i] + l - 2);
                                p[i] += l;
                            } else p[i]++;
                        }
                    }
                    std::vector<Tok> all;
                    for (uint32_t i = 0; i < L; i++) all.insert(all.end(), tk[i].begin(), tk[i].end());
                    (clip ? szB1[v] : szB2[v]) += size_of(all, ps, pe);
                    if (clip) { steps[v] += st; probesB[v] += pr; }
                }
            }
        }
    }
    const double in = double(pieces) * kPiece;
    printf("input %.0f bytes in %llu pieces of 8 KiB, near tables of 2^%d entries (tile-local matching only, greedy)\n", in, (unsigned long long)pieces, hb);
    printf("A  all positions (the kernel's division):           ratio %.4f   probes per piece %u\n", szA / in, kPiece);
    for (int v = 0; v < 3; v++)
        printf("B  lane-sequential, sub-pieces of %3u B (%2u lanes): ratio %.4f clipped (%+.1f %%), %.4f running on (%+.1f %%); lockstep steps per piece %.0f "
               "(x %u lanes = %.0f lane-slots, %.0f useful probes: %.0f %% of the slots)\n",
               subs[v], kPiece / subs[v], szB1[v] / in, 100.0 * (double(szB1[v]) / szA - 1), szB2[v] / in, 100.0 * (double(szB2[v]) / szA - 1),
               double(steps[v]) / pieces, kPiece / subs[v], double(steps[v]) / pieces * (kPiece / subs[v]), double(probesB[v]) / pieces,
               100.0 * probesB[v] / (double(steps[v]) * (kPiece / subs[v])));
    return 0;
}
