// mlz_toktab.h — "a token starting at this byte": the field-by-field decode (decode_tok) and a table-driven form of it
// (tok_entry / tok_fields) for the index pass, which decodes every token of a stream with a lane each and is bound by
// instruction issue.  Plain C++ so that tools/tok_fields_check.cpp can compile both on the host and compare them.
// Field layout: SPEC.md:68-266 / decode.go:362-582.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define MLZ_HD __host__ __device__ __forceinline__
#define MLZ_HDC __host__ __device__ constexpr
#else
#define MLZ_HD inline
#define MLZ_HDC constexpr
#endif

namespace mlz {

constexpr uint32_t kTokMinCopy2Offset = 64, kTokMinCopy3Offset = 65536;   // encode.go:50, :57 (= kMinCopy2Offset / kMinCopy3Offset)

struct Tok {
    uint32_t hdr, lit, cp, off;  // header bytes, literal bytes, copy bytes, explicit offset (0 = none)
};

// Decode "a token starting at this byte" from the 8 bytes w (little-endian) found there.  Written with selects only: every
// lane of a wavefront decodes a different (mostly bogus) tag, so branches would all be taken anyway.
MLZ_HD Tok decode_tok(uint64_t w) {
    const uint32_t lo = uint32_t(w);
    const uint32_t b = lo & 0xff;
    const uint32_t tag = b & 3;
    // tag 0: literal / repeat.  x < 29: len = x + 1; x = 29/30/31: 1/2/3 length bytes, len = 30 + value
    const uint32_t x = b >> 3;
    const uint32_t e0 = x >= 29 ? x - 28 : 0;
    const uint32_t v0 = (lo >> 8) & (0xffffffu >> (8 * (3 - (e0 ? e0 : 1))));
    const uint32_t len0 = e0 ? 30 + v0 : x + 1;
    const bool rep0 = (b & 4) != 0;
    // tag 1: copy1
    const uint32_t l1 = (b >> 2) & 15;
    const uint32_t cp1 = l1 == 15 ? ((lo >> 16) & 0xff) + 18 : l1 + 4;
    // tag 2: copy2 (length code in the tag byte, extension bytes after the 16-bit offset)
    const uint32_t l2 = b >> 2;
    const uint32_t e2 = l2 > 60 ? l2 - 60 : 0;
    const uint32_t v2 = uint32_t(w >> 24) & (0xffffffu >> (8 * (3 - (e2 ? e2 : 1))));
    const uint32_t cp2 = e2 ? 64 + v2 : l2 + 4;
    // tag 3: fused copy2 (bit 2 clear) or copy3 (bit 2 set)
    const bool c3 = (lo & 4) != 0;
    const uint32_t lits = (lo >> 3) & 3;
    const uint32_t l3 = (lo >> 5) & 63;
    const uint32_t e3 = l3 > 60 ? l3 - 60 : 0;
    const uint32_t v3 = uint32_t(w >> 32) & (0xffffffu >> (8 * (3 - (e3 ? e3 : 1))));
    const uint32_t cp3 = e3 ? 64 + v3 : l3 + 4;
    const uint32_t off16 = ((lo >> 8) & 0xffff) + kTokMinCopy2Offset;
    Tok t;
    t.hdr = tag == 0 ? 1 + e0 : tag == 1 ? 2 + (l1 == 15) : tag == 2 ? 3 + e2 : (c3 ? 4 + e3 : 3);
    t.lit = tag == 0 ? (rep0 ? 0 : len0) : tag == 3 ? (c3 ? lits : lits + 1) : 0;
    t.cp = tag == 0 ? (rep0 ? len0 : 0) : tag == 1 ? cp1 : tag == 2 ? cp2 : (c3 ? cp3 : 4 + ((lo >> 5) & 7));
    t.off = tag == 0 ? 0 : tag == 1 ? ((lo & 0xffff) >> 6) + 1 : tag == 2 ? off16 : (c3 ? (lo >> 11) + kTokMinCopy3Offset : off16);
    return t;
}

// v_bfe_u32 with a width of 0 / 8 / 16 / 24 bits
MLZ_HD uint32_t tok_bfe(uint32_t v, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return width ? (v >> off) & ((1u << width) - 1) : 0;
#endif
}

// The table form: everything the first byte b of a token fixes, in one word.
//   bits  0..5   adv0    stream bytes to the next token, without what the following bytes add
//   bits  6..7   ea      a long literal: that many length bytes follow, and their value is added to adv (not for repeats)
//   bits  8..9   c       copy3 with length code (b >> 5) + 56 when the next byte's low three bits are set: that many length bytes at byte 4
//   bits 10..16  obase   output bytes, without what the following bytes add
//   bits 17..19  xsh     byte offset of the length bytes that add to the output
//   bits 20..21  xn      ... and how many
//   bit  22      c3      copy3: + 8 x (next byte & 7), or the long form
//   bit  23      litall  all output bytes are literals (a literal token)
//   bits 24..26  litfix  else this many are
//   bits 27..28  okind   explicit offset: 0 none, 1 copy1, 2 copy2 / fused copy2, 3 copy3
MLZ_HDC uint32_t tok_entry(uint32_t b) {
    const uint32_t tag = b & 3;
    uint32_t adv = 0, ea = 0, c = 0, obase = 0, xsh = 0, xn = 0, c3 = 0, litall = 0, litfix = 0, okind = 0;
    if (tag == 0) {
        const uint32_t x = b >> 3;
        const bool rep = (b & 4) != 0;
        const uint32_t e0 = x > 28 ? x - 28 : 0;
        obase = e0 ? 30 : x + 1; xsh = 1; xn = e0;
        adv = 1 + e0 + (rep ? 0 : obase); ea = rep ? 0 : e0;
        litall = rep ? 0 : 1;
    } else if (tag == 1) {
        const uint32_t l1 = (b >> 2) & 15;
        adv = 2 + (l1 == 15 ? 1u : 0u);
        obase = l1 == 15 ? 18 : l1 + 4; xsh = 2; xn = l1 == 15 ? 1 : 0;
        okind = 1;
    } else if (tag == 2) {
        const uint32_t l2 = b >> 2, e2 = l2 > 60 ? l2 - 60 : 0;
        adv = 3 + e2;
        obase = e2 ? 64 : l2 + 4; xsh = 3; xn = e2;
        okind = 2;
    } else {
        const uint32_t lits = (b >> 3) & 3;
        adv = 4 + lits;
        if (b & 4) {   // copy3: l3 = (b >> 5) + 8 k, k = the next byte's low three bits
            c3 = 1; litfix = lits; okind = 3;
            obase = lits + (b >> 5) + 4;
            if ((b >> 5) >= 5) c = (b >> 5) - 4;
        } else {       // fused copy2: lits + 1 literals, copy of 4 + (b >> 5)
            litfix = lits + 1; okind = 2;
            obase = lits + 1 + 4 + (b >> 5);
        }
    }
    return adv | (ea << 6) | (c << 8) | (obase << 10) | (xsh << 17) | (xn << 20) | (c3 << 22) | (litall << 23) | (litfix << 24) | (okind << 27);
}
struct TokTable { uint32_t e[256]; };
MLZ_HDC TokTable make_toktab() { TokTable t{}; for (uint32_t b = 0; b < 256; b++) t.e[b] = tok_entry(b); return t; }

// hdr + lit of decode_tok from the token's first four bytes and its entry
MLZ_HD uint32_t tok_adv(uint32_t lo, uint32_t t) {
    uint32_t adv = t & 63;
    adv += tok_bfe(lo, 8, 8 * ((t >> 6) & 3));
    adv += ((lo >> 8) & 7) == 7 ? (t >> 8) & 3 : 0;
    return adv;
}
// lit + cp of decode_tok from the token's first eight bytes and its entry
MLZ_HD uint32_t tok_olen(uint64_t w, uint32_t t) {
    const uint32_t lo = uint32_t(w);
    const uint32_t ext = tok_bfe(uint32_t(w >> (8 * ((t >> 17) & 7))), 0, 8 * ((t >> 20) & 3));
    uint32_t olen = ((t >> 10) & 127) + ext;
    const uint32_t k = (lo >> 8) & 7, c = (t >> 8) & 3;
    const uint32_t c3_long = ((t >> 24) & 7) + 64 + tok_bfe(uint32_t(w >> 32), 0, 8 * c);
    const uint32_t c3_olen = (k == 7 && c) ? c3_long : olen + 8 * k;
    return (t & (1u << 22)) ? c3_olen : olen;
}
MLZ_HD uint32_t tok_lit(uint32_t olen, uint32_t t) { return (t & (1u << 23)) ? olen : (t >> 24) & 7; }
MLZ_HD uint32_t tok_off(uint32_t lo, uint32_t t) {
    const uint32_t kind = (t >> 27) & 3;
    const uint32_t o1 = ((lo & 0xffff) >> 6) + 1, o2 = ((lo >> 8) & 0xffff) + kTokMinCopy2Offset, o3 = (lo >> 11) + kTokMinCopy3Offset;
    return kind == 0 ? 0 : kind == 1 ? o1 : kind == 2 ? o2 : o3;
}

}  // namespace mlz
