// CPU simulation of the general-block decoder's phases (E: source words, L: in-tile chains, J: rounds of hops):
// how many words are still positions after each step?  usage: genstats <block.mzb-body> <dlen>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <fstream>
static const uint32_t kMinCopy2Offset = 64, kMinCopy3Offset = 65536;
struct Tok { uint32_t hdr, lit, cp, off; };
static Tok decode_tok(uint64_t w) {
    const uint32_t lo = uint32_t(w); const uint32_t b = lo & 0xff; const uint32_t tag = b & 3;
    const uint32_t x = b >> 3; const uint32_t e0 = x >= 29 ? x - 28 : 0;
    const uint32_t v0 = (lo >> 8) & (0xffffffu >> (8 * (3 - (e0 ? e0 : 1))));
    const uint32_t len0 = e0 ? 30 + v0 : x + 1; const bool rep0 = (b & 4) != 0;
    const uint32_t l1 = (b >> 2) & 15; const uint32_t cp1 = l1 == 15 ? ((lo >> 16) & 0xff) + 18 : l1 + 4;
    const uint32_t l2 = b >> 2; const uint32_t e2 = l2 > 60 ? l2 - 60 : 0;
    const uint32_t v2 = uint32_t(w >> 24) & (0xffffffu >> (8 * (3 - (e2 ? e2 : 1)))); const uint32_t cp2 = e2 ? 64 + v2 : l2 + 4;
    const bool c3 = (lo & 4) != 0; const uint32_t lits = (lo >> 3) & 3; const uint32_t l3 = (lo >> 5) & 63; const uint32_t e3 = l3 > 60 ? l3 - 60 : 0;
    const uint32_t v3 = uint32_t(w >> 32) & (0xffffffu >> (8 * (3 - (e3 ? e3 : 1)))); const uint32_t cp3 = e3 ? 64 + v3 : l3 + 4;
    const uint32_t off16 = ((lo >> 8) & 0xffff) + kMinCopy2Offset;
    Tok t;
    t.hdr = tag == 0 ? 1 + e0 : tag == 1 ? 2 + (l1 == 15) : tag == 2 ? 3 + e2 : (c3 ? 4 + e3 : 3);
    t.lit = tag == 0 ? (rep0 ? 0 : len0) : tag == 3 ? (c3 ? lits : lits + 1) : 0;
    t.cp = tag == 0 ? (rep0 ? len0 : 0) : tag == 1 ? cp1 : tag == 2 ? cp2 : (c3 ? cp3 : 4 + ((lo >> 5) & 7));
    t.off = tag == 0 ? 0 : tag == 1 ? ((lo & 0xffff) >> 6) + 1 : tag == 2 ? off16 : (c3 ? (lo >> 11) + kMinCopy3Offset : off16);
    return t;
}
int main(int argc, char** argv) {
    std::ifstream f(argv[1], std::ios::binary); std::vector<uint8_t> s((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const uint32_t dlen = atoi(argv[2]); const uint32_t hops = argc > 3 ? atoi(argv[3]) : 4; const uint32_t tile = argc > 4 ? atoi(argv[4]) : 32768;
    s.resize(s.size() + 16);
    const uint32_t LIT = 0x80000000u;
    std::vector<uint32_t> idx(dlen);
    uint32_t sp = 0, d = 0, rep = 0; size_t ntok = 0, ncopy = 0;
    while (d < dlen) {
        uint64_t w; memcpy(&w, &s[sp], 8);
        Tok t = decode_tok(w);
        uint32_t off = t.off ? t.off : rep; if (t.off) rep = t.off;
        for (uint32_t i = 0; i < t.lit; i++) idx[d + i] = LIT | (sp + t.hdr + i);
        d += t.lit;
        for (uint32_t i = 0; i < t.cp; i++) idx[d + i] = d + i - off;
        d += t.cp; sp += t.hdr + t.lit; ntok++; ncopy += t.cp != 0;
    }
    auto count = [&](const char* what) { size_t n = 0; for (uint32_t j = 0; j < dlen; j++) n += !(idx[j] & LIT); printf("%-28s positions left: %9zu (%.2f %%)\n", what, n, 100.0 * n / dlen); };
    printf("tokens %zu copies %zu bytes/token %.2f\n", ntok, ncopy, double(dlen) / ntok);
    count("after E");
    // L: in-tile chains to their end (sequential = fully resolved inside the tile)
    for (uint32_t j = 0; j < dlen; j++) { uint32_t v = idx[j]; const uint32_t ts = j / tile * tile; if (!(v & LIT) && v >= ts) idx[j] = idx[v]; }
    count("after L (in-tile)");
    // distribution of distinct 256-word groups that are entirely literal
    for (int round = 0; round < 8; round++) {
        std::vector<uint32_t> nx(idx);
        for (uint32_t j = 0; j < dlen; j++) { uint32_t v = idx[j]; for (uint32_t h = 0; h < hops && !(v & LIT); h++) v = idx[v]; nx[j] = v; }
        idx.swap(nx);
        char nm[64]; snprintf(nm, 64, "after J round %d (%u hops)", round + 1, hops);
        count(nm);
        size_t n = 0; for (uint32_t j = 0; j < dlen; j++) n += !(idx[j] & LIT); if (!n) break;
    }
    return 0;
}
