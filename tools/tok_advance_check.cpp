// Checks the exit pass's table form of the token advance (mlz_decode.hip.inc: tok_advance_entry / tok_advance_tab) against the arithmetic
// form (tok_advance) for all 2^32 values of a token's first four bytes.  build + run: g++ -O2 -o /tmp/tac tools/tok_advance_check.cpp && /tmp/tac  (18 s)
#include <cstdint>
#include <cstdio>
static uint32_t ubfe(uint32_t v, uint32_t off, uint32_t w) { off &= 31; w &= 31; if (!w) return 0; return (v >> off) & ((1u << w) - 1); }   // v_bfe_u32 semantics (width 5 bits)
static uint32_t tok_advance(uint32_t lo) {
    const uint32_t tag = lo & 3;
    const uint32_t x = (lo >> 3) & 31;
    const uint32_t e0 = x > 28 ? x - 28 : 0;
    const uint32_t v0 = ubfe(lo, 8, 8 * e0);
    const uint32_t lit0 = (lo & 4) ? 0 : x + 1 + v0 - (e0 > 1 ? e0 - 1 : 0);
    const uint32_t a0 = 1 + e0 + lit0;
    const uint32_t a1 = 2 + (((lo >> 2) & 15) == 15 ? 1u : 0u);
    const uint32_t l2 = (lo >> 2) & 63;
    const uint32_t a2 = 3 + (l2 > 60 ? l2 - 60 : 0);
    const uint32_t lits = (lo >> 3) & 3, l3 = (lo >> 5) & 63;
    const uint32_t a3 = 4 + lits + ((lo & 4) && l3 > 60 ? l3 - 60 : 0);
    return tag == 0 ? a0 : tag == 1 ? a1 : tag == 2 ? a2 : a3;
}
static uint32_t entry(uint32_t b) {
    const uint32_t tag = b & 3;
    uint32_t adv = 0, e8 = 0, c = 0;
    if (tag == 0) {
        const uint32_t x = b >> 3; const bool rep = b & 4;
        if (x <= 28) adv = 1 + (rep ? 0 : x + 1);
        else { const uint32_t e0 = x - 28; adv = 1 + e0 + (rep ? 0 : 30); e8 = rep ? 0 : 8 * e0; }
    } else if (tag == 1) adv = 2 + (((b >> 2) & 15) == 15);
    else if (tag == 2) { const uint32_t l2 = b >> 2; adv = 3 + (l2 > 60 ? l2 - 60 : 0); }
    else { const uint32_t lits = (b >> 3) & 3; adv = 4 + lits; if ((b & 4) && (b >> 5) >= 5) c = (b >> 5) - 4; }
    return adv | (e8 << 8) | (c << 16);
}
static uint32_t tab[256];
static uint32_t tok_advance_tab(uint32_t lo) {
    const uint32_t t = tab[lo & 0xff];
    uint32_t adv = t & 63;
    adv += ubfe(lo, 8, ubfe(t, 8, 5));
    adv += ((lo >> 8) & 7) == 7 ? ubfe(t, 16, 2) : 0;
    return adv;
}
int main() {
    for (uint32_t b = 0; b < 256; b++) tab[b] = entry(b);
    uint64_t bad = 0;
    for (uint64_t lo = 0; lo < (1ull << 32); lo += 1) {
        if (tok_advance(uint32_t(lo)) != tok_advance_tab(uint32_t(lo))) { if (bad < 5) printf("lo %08x: %u vs %u\n", uint32_t(lo), tok_advance(uint32_t(lo)), tok_advance_tab(uint32_t(lo))); bad++; }
    }
    printf("mismatches %llu\n", (unsigned long long)bad);
}
