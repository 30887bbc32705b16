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

// emitLiteral header (asm_none.go:84-122); len >= 1.
MLZ_HD Hdr lit_header(uint32_t len) {
    uint32_t n = len - 1;
    if (n < 29) return {uint64_t(n << 3), 1};
    if (n < 256 + 29) return {uint64_t(29 << 3) | uint64_t(n - 29) << 8, 2};
    if (n < 65536 + 29) return {uint64_t(30 << 3) | uint64_t(n - 29) << 8, 3};
    return {uint64_t(31 << 3) | uint64_t(n - 29) << 8, 4};
}

// emitRepeat (asm_none.go:125-156); len >= 1.
MLZ_HD Hdr repeat_header(uint32_t len) {
    if (len < 30) return {uint64_t(((len - 1) << 3) | 4), 1};
    uint32_t l = len - 30;
    if (l < 256) return {uint64_t((29 << 3) | 4) | uint64_t(l) << 8, 2};
    if (l < 65536) return {uint64_t((30 << 3) | 4) | uint64_t(l) << 8, 3};
    return {uint64_t((31 << 3) | 4) | uint64_t(l) << 8, 4};
}

MLZ_HD Hdr hdr_cat(Hdr a, Hdr b) { return {a.bits | (b.bits << (8 * a.n)), a.n + b.n}; }

// encodeCopy3 (asm_none.go:160-200): 4..7 bytes; lits = fused literal count 0..3.
MLZ_HD Hdr copy3_header(uint32_t off, uint32_t len, uint32_t lits) {
    uint32_t l = len - 4;
    uint32_t enc = ((off - 65536) << 11) | 7u | (lits << 3);
    if (l <= 60) return {uint64_t(enc | (l << 5)), 4};
    l -= 60;
    if (l < 256) return {uint64_t(enc | (61u << 5)) | uint64_t(l) << 32, 5};
    if (l < 65536) return {uint64_t(enc | (62u << 5)) | uint64_t(l) << 32, 6};
    return {uint64_t(enc | (63u << 5)) | uint64_t(l) << 32, 7};
}

// encodeCopy2 (encode.go:247-282): 3..6 bytes.
MLZ_HD Hdr copy2_header(uint32_t off, uint32_t len) {
    uint32_t l = len - 4;
    uint64_t o = uint64_t(off - kMinCopy2Offset) << 8;
    if (l <= 60) return {uint64_t((l << 2) | 2) | o, 3};
    l -= 60;
    if (l < 256) return {uint64_t((61 << 2) | 2) | o | uint64_t(l) << 24, 4};
    if (l < 65536) return {uint64_t((62 << 2) | 2) | o | uint64_t(l) << 24, 5};
    return {uint64_t((63 << 2) | 2) | o | uint64_t(l) << 24, 6};
}

// emitCopy (asm_none.go:207-278): picks copy1 / copy2 / copy3 by offset; long copy1 = copy1(18)+repeat.
MLZ_HD Hdr copy_header(uint32_t off, uint32_t len) {
    if (off > kMaxCopy2Offset) return copy3_header(off, len, 0);
    if (off <= kMaxCopy1Offset) {
        uint32_t o = (off - 1) << 6;
        if (len < 19) return {uint64_t((o | ((len - 4) << 2) | 1) & 0xffff), 2};
        if (len < 274) return {uint64_t((o | (15 << 2) | 1) & 0xffff) | uint64_t(len - 18) << 16, 3};
        return hdr_cat({uint64_t((o | (14 << 2) | 1) & 0xffff), 2}, repeat_header(len - 18));
    }
    return copy2_header(off, len);
}

// emitCopyLits2 header (asm_none.go:284-308): 3 bytes; the 1..4 literals follow, then (for
// len > 11) a repeat for the remainder, returned through *tail.
MLZ_HD Hdr fused2_header(uint32_t off, uint32_t len, uint32_t lits, Hdr* tail) {
    uint32_t l = len - 4;
    uint64_t o = uint64_t(off - kMinCopy2Offset) << 8;
    if (l > 7) {
        *tail = repeat_header(l - 7);
        return {uint64_t(3u | (7u << 5) | ((lits - 1) << 3)) | o, 3};
    }
    *tail = {0, 0};
    return {uint64_t(3u | (l << 5) | ((lits - 1) << 3)) | o, 3};
}

// A complete match emission: [pre][literals][post].  Mirrors the choice made by the reference's
// L1 encoder (encode_l1.go:190-206, :432-446): repeat if the offset equals the previous one,
// fused forms for 1..4 (copy2) / 1..3 (copy3) pending literals with offset >= 64, else
// literal + copy.
struct Emit {
    Hdr pre, post;
};
MLZ_HD Emit plan_emit(uint32_t lits, uint32_t off, uint32_t len, bool is_repeat) {
    Emit e;
    if (is_repeat) {
        e.pre = lits ? lit_header(lits) : Hdr{0, 0};
        e.post = repeat_header(len);
    } else if (lits > 0 && off >= kMinCopy2Offset && off <= kMaxCopy2Offset && lits <= 4) {
        e.pre = fused2_header(off, len, lits, &e.post);
    } else if (lits > 0 && off > kMaxCopy2Offset && lits <= 3) {
        e.pre = copy3_header(off, len, lits);
        e.post = {0, 0};
    } else {
        e.pre = lits ? lit_header(lits) : Hdr{0, 0};
        e.post = copy_header(off, len);
    }
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
