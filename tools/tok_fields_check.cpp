// Checks the table form of the token decode (minlz_amd/csrc/mlz_toktab.h: tok_entry / tok_adv / tok_olen / tok_lit / tok_off) against
// decode_tok: every value of a token's first two bytes x 2^16 random values + the all-zero / all-one patterns of the six bytes behind them.
// build + run: g++ -O2 -o /tmp/tfc tools/tok_fields_check.cpp && /tmp/tfc [stride]   (all values: a minute; tests/test_toktab.py runs every 41st)
#include <cstdio>
#include <cstdlib>
#include "../minlz_amd/csrc/mlz_toktab.h"
using namespace mlz;
int main(int argc, char** argv) {
    const uint32_t stride = argc > 1 ? uint32_t(atoi(argv[1])) : 1;   // every stride-th value of the first two bytes (tests: a quick pass)
    uint32_t tab[256];
    for (uint32_t b = 0; b < 256; b++) tab[b] = tok_entry(b);
    uint64_t rng = 0x9e3779b97f4a7c15ull, bad = 0, n = 0;
    for (uint32_t b01 = 0; b01 < 65536; b01 += stride ? stride : 1) {
        for (uint32_t r = 0; r < 65536 + 4; r++) {
            uint64_t hi;
            if (r == 65536) hi = 0; else if (r == 65537) hi = ~0ull; else if (r == 65538) hi = 0x0000ffffffull; else if (r == 65539) hi = 0xffffff000000ull;
            else { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; hi = rng; }
            const uint64_t w = uint64_t(b01) | (hi << 16);
            const Tok t = decode_tok(w);
            const uint32_t e = tab[w & 0xff], lo = uint32_t(w);
            const uint32_t adv = tok_adv(lo, e), olen = tok_olen(w, e), lit = tok_lit(olen, e), off = tok_off(lo, e);
            n++;
            if (adv != t.hdr + t.lit || olen != t.lit + t.cp || lit != t.lit || olen - lit != t.cp || off != t.off) {
                if (bad++ < 10) printf("MISMATCH w=%016llx: adv %u/%u olen %u/%u lit %u/%u off %u/%u\n", (unsigned long long)w, adv, t.hdr + t.lit, olen, t.lit + t.cp, lit, t.lit, off, t.off);
            }
        }
    }
    printf("%llu tokens checked, %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad);
    return bad ? 1 : 0;
}
