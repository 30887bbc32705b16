// mlz_format.h — MinLZ block-format constants and token builders shared by host and device code.
//
// Normative source: the reference's SPEC.md:16-266 (block format) and its emitters
// asm_none.go:84-323 / encode.go:247-282.  The builders here produce the same bytes as
// emitLiteral / emitRepeat / emitCopy / emitCopyLits2 / emitCopyLits3 but in a "header word"
// form (up to 8 bytes packed little-endian in a uint64 + a length) so that a wavefront can
// scatter the bytes with one predicated store per lane instead of a byte-serial emitter.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MLZ_HD __host__ __device__ __forceinline__
#else
#define MLZ_HD inline
#endif

namespace mlz {

constexpr uint32_t kMaxBlockSize = 8u << 20;        // minlz.go:84
constexpr uint32_t kMinNonLiteralBlock = 16;        // encode.go:220
constexpr uint32_t kMaxCopy1Offset = 1024;          // encode.go:48
constexpr uint32_t kMinCopy2Offset = 64;            // encode.go:50
constexpr uint32_t kMaxCopy2Offset = 64 + 65535;    // encode.go:51
constexpr uint32_t kMinCopy3Offset = 65536;         // encode.go:57
constexpr uint32_t kMaxCopy3Offset = (2u << 20) + 65535;  // encode.go:58
constexpr uint32_t kCopy2LitMaxLen = 11;            // encode.go:52

struct Hdr {
    uint64_t bits;  // bytes, little-endian
    uint32_t n;     // number of valid bytes (0..8)
};

// The builders below are written with selects instead of branches: on the GPU every lane of a
// wavefront builds a different token, and divergent branches cost more than the arithmetic.

// Shared shape of emitLiteral / emitRepeat headers (asm_none.go:84-156): n >= 1 items,
// tag bit 2 set for repeats.  1 byte for n <= 29, else code 29/30/31 + 1/2/3 bytes of (n - 30).
MLZ_HD Hdr run_header(uint32_t n, uint32_t tagbit) {
    const uint32_t sz = 1u + (n > 29u) + (n > 29u + 256u) + (n > 29u + 65536u);
    const uint64_t small = uint64_t(((n - 1u) << 3) | tagbit);
    const uint64_t big = uint64_t(((27u + sz) << 3) | tagbit) | (uint64_t(n - 30u) << 8);
    return {sz == 1 ? small : big, sz};
}
// emitLiteral header (asm_none.go:84-122); len >= 1.
MLZ_HD Hdr lit_header(uint32_t len) { return run_header(len, 0); }
// emitRepeat (asm_none.go:125-156); len >= 1.
MLZ_HD Hdr repeat_header(uint32_t len) { return run_header(len, 4); }

MLZ_HD Hdr hdr_cat(Hdr a, Hdr b) { return {a.bits | (b.bits << (8 * a.n)), a.n + b.n}; }
MLZ_HD Hdr hdr_sel(bool c, Hdr a, Hdr b) { return {c ? a.bits : b.bits, c ? a.n : b.n}; }

// Length field shared by copy2 / copy3 (encode.go:247-282, asm_none.go:160-200): code 0..60 =
// len - 4, else 61/62/63 followed by 1/2/3 bytes of (len - 64).
struct LenExt { uint32_t code, extra_n; uint64_t extra; };
MLZ_HD LenExt len_ext(uint32_t len) {
    const uint32_t l = len - 4;
    const uint32_t e = len - 64;  // only meaningful when l > 60
    const uint32_t xn = (l > 60u) + (l > 60u + 255u) + (l > 60u + 65535u);
    return {xn ? 60u + xn : l, xn, xn ? uint64_t(e) : 0};
}

// encodeCopy3 (asm_none.go:160-200): 4..7 bytes; lits = fused literal count 0..3.
MLZ_HD Hdr copy3_header(uint32_t off, uint32_t len, uint32_t lits) {
    const LenExt x = len_ext(len);
    const uint32_t enc = ((off - 65536) << 11) | 7u | (lits << 3) | (x.code << 5);
    return {uint64_t(enc) | (x.extra << 32), 4 + x.extra_n};
}

// encodeCopy2 (encode.go:247-282): 3..6 bytes.
MLZ_HD Hdr copy2_header(uint32_t off, uint32_t len) {
    const LenExt x = len_ext(len);
    return {uint64_t((x.code << 2) | 2) | (uint64_t(off - kMinCopy2Offset) << 8) | (x.extra << 24), 3 + x.extra_n};
}

// copy1 (asm_none.go:252-270): 2 bytes for len < 19, 3 for len < 274, else copy1(18) + repeat.
MLZ_HD Hdr copy1_header(uint32_t off, uint32_t len, Hdr rep_tail /* repeat_header(len - 18) */) {
    const uint32_t o = ((off - 1) << 6) & 0xffff;
    const Hdr a{uint64_t(o | ((len - 4) << 2) | 1), 2};
    const Hdr b{uint64_t(o | (15 << 2) | 1) | uint64_t(len - 18) << 16, 3};
    const Hdr c = hdr_cat({uint64_t(o | (14 << 2) | 1), 2}, rep_tail);
    return hdr_sel(len < 19, a, hdr_sel(len < 274, b, c));
}

// emitCopy (asm_none.go:207-278): picks copy1 / copy2 / copy3 by offset.
MLZ_HD Hdr copy_header(uint32_t off, uint32_t len) {
    const Hdr c1 = copy1_header(off, len, repeat_header(len > 18 ? len - 18 : 1));
    return hdr_sel(off > kMaxCopy2Offset, copy3_header(off, len, 0), hdr_sel(off <= kMaxCopy1Offset, c1, copy2_header(off, len)));
}

// emitCopyLits2 header (asm_none.go:284-308): 3 bytes; the 1..4 literals follow, then (for
// len > 11) a repeat for the remainder, returned through *tail.
MLZ_HD Hdr fused2_header(uint32_t off, uint32_t len, uint32_t lits, Hdr* tail) {
    const uint32_t l = len - 4;
    const uint64_t o = uint64_t(off - kMinCopy2Offset) << 8;
    *tail = hdr_sel(l > 7, repeat_header(l > 7 ? l - 7 : 1), Hdr{0, 0});
    return {uint64_t(3u | ((l > 7 ? 7u : l) << 5) | ((lits - 1) << 3)) | o, 3};
}

// A complete match emission: [pre][literals][post].  Mirrors the choice made by the reference's
// L1 encoder (encode_l1.go:190-206, :432-446): repeat if the offset equals the previous one,
// fused forms for 1..4 (copy2) / 1..3 (copy3) pending literals with offset >= 64, else
// literal + copy.  Branch-free: all candidate headers are built and selected.
struct Emit {
    Hdr pre, post;
};
MLZ_HD Emit plan_emit(uint32_t lits, uint32_t off, uint32_t len, bool is_repeat) {
    const bool f2 = !is_repeat && lits > 0 && lits <= 4 && off >= kMinCopy2Offset && off <= kMaxCopy2Offset;
    const bool f3 = !is_repeat && lits > 0 && lits <= 3 && off > kMaxCopy2Offset;
    const Hdr none{0, 0};
    const Hdr lh = hdr_sel(lits > 0, lit_header(lits > 0 ? lits : 1), none);
    // one repeat header serves three uses: the repeat itself, the tail of a long fused copy2
    // (len - 11) and the tail of a long copy1 (len - 18)
    const bool c1 = !is_repeat && !f2 && !f3 && off <= kMaxCopy1Offset;
    const uint32_t rarg = is_repeat ? len : f2 ? (len > 11 ? len - 11 : 1) : (len > 18 ? len - 18 : 1);
    const Hdr rh = repeat_header(rarg);
    const LenExt x = len_ext(len);
    // copy3 (plain or fused) and copy2
    const uint32_t o3 = off > 65536 ? off - 65536 : 0;
    const Hdr c3{uint64_t((o3 << 11) | 7u | ((f3 ? lits : 0) << 3) | (x.code << 5)) | (x.extra << 32), 4 + x.extra_n};
    const uint32_t o2 = off >= kMinCopy2Offset ? off - kMinCopy2Offset : 0;
    const Hdr c2{uint64_t((x.code << 2) | 2) | (uint64_t(o2 & 0xffff) << 8) | (x.extra << 24), 3 + x.extra_n};
    const uint32_t l4 = len - 4;
    const Hdr f2h{uint64_t(3u | ((l4 > 7 ? 7u : l4) << 5) | (((lits - 1) & 3) << 3)) | (uint64_t(o2 & 0xffff) << 8), 3};
    const Hdr c1h = copy1_header(off, len, rh);
    Emit e;
    e.pre = hdr_sel(f2, f2h, hdr_sel(f3, c3, lh));
    const Hdr plain = hdr_sel(off > kMaxCopy2Offset, c3, hdr_sel(c1, c1h, c2));
    e.post = hdr_sel(is_repeat, rh, hdr_sel(f2, hdr_sel(len > 11, rh, none), hdr_sel(f3, none, plain)));
    return e;
}

// plan_emit for the common tokens: no length-extension bytes anywhere (pending literals <= 29, repeat
// length <= 29, copy length <= 64).  About a third of the instructions of the general builder; the
// encoder takes it when every token of a batch qualifies (plan_emit_is_short), byte-identical result.
MLZ_HD bool plan_emit_is_short(uint32_t lits, uint32_t len, bool is_repeat) {
    return lits <= 29 && (is_repeat ? len <= 29 : len <= 64);
}
MLZ_HD Emit plan_emit_short(uint32_t lits, uint32_t off, uint32_t len, bool is_repeat) {
    const bool f2 = !is_repeat && lits > 0 && lits <= 4 && off >= kMinCopy2Offset && off <= kMaxCopy2Offset;
    const bool f3 = !is_repeat && lits > 0 && lits <= 3 && off > kMaxCopy2Offset;
    const Hdr none{0, 0};
    const Hdr lh{uint64_t((lits - 1) << 3), lits ? 1u : 0u};
    const uint32_t l4 = len - 4;
    const uint32_t o2 = (off - kMinCopy2Offset) & 0xffff;
    // repeat (len <= 29), or the <= 53-byte tail of a fused copy2 that is longer than 11
    const uint32_t rn = is_repeat ? len : len - 11;
    const Hdr rh = rn <= 29 ? Hdr{uint64_t(((rn - 1) << 3) | 4), 1} : Hdr{uint64_t((29u << 3) | 4) | (uint64_t(rn - 30) << 8), 2};
    const uint32_t o1 = ((off - 1) << 6) & 0xffff;
    const Hdr c1 = len < 19 ? Hdr{uint64_t(o1 | (l4 << 2) | 1), 2} : Hdr{uint64_t(o1 | (15 << 2) | 1) | (uint64_t(len - 18) << 16), 3};
    const Hdr c2{uint64_t((l4 << 2) | 2) | (uint64_t(o2) << 8), 3};
    const Hdr c3{uint64_t(((off - kMinCopy3Offset) << 11) | 7u | ((f3 ? lits : 0) << 3) | (l4 << 5)), 4};
    const Hdr f2h{uint64_t(3u | ((l4 > 7 ? 7u : l4) << 5) | (((lits - 1) & 3) << 3)) | (uint64_t(o2) << 8), 3};
    Emit e;
    e.pre = hdr_sel(f2, f2h, hdr_sel(f3, c3, lh));
    const Hdr plain = hdr_sel(off > kMaxCopy2Offset, c3, hdr_sel(off <= kMaxCopy1Offset, c1, c2));
    e.post = hdr_sel(is_repeat, rh, hdr_sel(f2, hdr_sel(len > 11, rh, none), hdr_sel(f3, none, plain)));
    return e;
}

MLZ_HD uint32_t put_uvarint(uint8_t* dst, uint64_t v) {
    uint32_t i = 0;
    while (v >= 0x80) { dst[i++] = uint8_t(v) | 0x80; v >>= 7; }
    dst[i++] = uint8_t(v);
    return i;
}

// Block header parse = isMinLZ (decode.go:120-156).  Works on any byte-addressable memory.
// Returns MLZ_OK-style code; body = offset of the token stream, dlen = decoded length,
// literals = 1 when the body is to be copied verbatim.
MLZ_HD int parse_block_header(const uint8_t* src, uint64_t n, uint64_t* body, uint64_t* dlen, int* literals) {
    *body = 0; *dlen = 0; *literals = 0;
    if (n == 0) return 1;  // ErrCorrupt
    if (n == 1 && src[0] == 0) { *body = 1; *literals = 1; return 0; }
    if (src[0] != 0) return 3;  // Snappy/S2 block: ErrUnsupported here (fallback out of scope)
    uint64_t x = 0; uint32_t shift = 0; uint64_t i = 1; bool done = false;
    for (; i < n && i <= 10; i++) {  // binary.Uvarint
        uint8_t b = src[i];
        if (b < 0x80) {
            if (i == 10 && b > 1) return 1;
            x |= uint64_t(b) << shift; done = true; i++; break;
        }
        x |= uint64_t(b & 0x7f) << shift;
        shift += 7;
    }
    if (!done) return 1;
    if (x > 0xffffffffull) return 1;
    if (x > kMaxBlockSize) return 2;  // ErrTooLarge
    uint64_t rest = n - i;
    if (rest == 0) return 1;
    if (x == 0) { *body = i; *dlen = rest; *literals = 1; return 0; }
    if (x < rest) return 1;
    *body = i; *dlen = x;
    return 0;
}

}  // namespace mlz
