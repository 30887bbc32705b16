// Host check: the product token builders (minlz_amd/csrc/mlz_format.h, used by the HIP encoder) produce the
// same bytes as the oracle restatement of emitLiteral/emitRepeat/emitCopy/emitCopyLits2/3 (asm_none.go:84-323).
#include "mlz_format.h"
#include "minlz_oracle.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
using namespace mlz;
static size_t put(uint8_t* d, Hdr h) { for (uint32_t i = 0; i < h.n; i++) d[i] = uint8_t(h.bits >> (8 * i)); return h.n; }
int main() {
    uint8_t a[64], b[64]; uint8_t lits[8] = {1,2,3,4,5,6,7,8};
    long bad = 0, n = 0;
    uint32_t lens[] = {4,5,11,12,18,19,63,64,65,68,273,274,300,319,320,1000,65535,65536,65600,70000,1<<20, 8<<20};
    uint32_t offs[] = {1,2,63,64,65,1024,1025,65535,65536,65599,65600,70000,2162687};
    for (uint32_t off : offs) for (uint32_t len : lens) {
        size_t na = put(a, copy_header(off, len)); size_t nb = mlzo_emit_copy(b, off, len); n++;
        if (na != nb || memcmp(a, b, na)) { bad++; if (bad < 5) printf("copy off=%u len=%u mismatch\n", off, len); }
        for (uint32_t nl = 0; nl <= 6; nl++) for (int rep = 0; rep < 2; rep++) {
            Emit e = plan_emit(nl, off, len, rep);
            size_t na2 = put(a, e.pre); memcpy(a + na2, lits, nl); na2 += nl; na2 += put(a + na2, e.post);
            size_t nb2;
            if (rep) { nb2 = mlzo_emit_literal(b, lits, nl); nb2 += mlzo_emit_repeat(b + nb2, len); }
            else if (nl > 0 && off >= 64 && off <= 65599 && nl <= 4) nb2 = mlzo_emit_copy_lits2(b, lits, nl, off, len);
            else if (nl > 0 && off > 65599 && nl <= 3) nb2 = mlzo_emit_copy_lits3(b, lits, nl, off, len);
            else { nb2 = mlzo_emit_literal(b, lits, nl); nb2 += mlzo_emit_copy(b + nb2, off, len); }
            n++;
            if (na2 != nb2 || memcmp(a, b, na2)) { bad++; if (bad < 10) printf("emit nl=%u off=%u len=%u rep=%d mismatch (%zu vs %zu)\n", nl, off, len, rep, na2, nb2); }
        }
    }
    // the short-token fast path must agree with the general builder on its whole domain
    {
        uint32_t offs2[] = {1, 2, 63, 64, 65, 1024, 1025, 65535, 65536, 65599, 65600, 70000, 2162687};
        for (uint32_t off : offs2) for (uint32_t len = 1; len <= 64; len++) for (uint32_t nl = 0; nl <= 29; nl++) for (int rep = 0; rep < 2; rep++) {
            if (!rep && len < 4) continue;
            if (!plan_emit_is_short(nl, len, rep)) continue;
            const Emit x = plan_emit(nl, off, len, rep), y = plan_emit_short(nl, off, len, rep);
            const uint64_t mx = x.pre.n ? (x.pre.n == 8 ? ~0ull : (1ull << (8 * x.pre.n)) - 1) : 0, my = x.post.n ? (x.post.n == 8 ? ~0ull : (1ull << (8 * x.post.n)) - 1) : 0;
            n++;
            if (x.pre.n != y.pre.n || x.post.n != y.post.n || ((x.pre.bits ^ y.pre.bits) & mx) || ((x.post.bits ^ y.post.bits) & my)) {
                bad++; if (bad < 10) printf("short nl=%u off=%u len=%u rep=%d mismatch\n", nl, off, len, rep);
            }
        }
    }
    uint32_t runs[] = {1,2,29,30,31,285,286,65565,65566,1000000};
    for (uint32_t r : runs) { size_t na = put(a, lit_header(r)); uint8_t* big = (uint8_t*)malloc(r + 8); uint8_t* src = (uint8_t*)calloc(r, 1); size_t nb = mlzo_emit_literal(big, src, r) - r; if (na != nb || memcmp(a, big, na)) { bad++; printf("lit %u mismatch\n", r); } free(big); free(src);
        na = put(a, repeat_header(r)); nb = mlzo_emit_repeat(b, r); if (na != nb || memcmp(a, b, na)) { bad++; printf("rep %u mismatch\n", r); } n += 2; }
    printf("%ld cases, %ld bad\n", n, bad);
    return bad != 0;
}
