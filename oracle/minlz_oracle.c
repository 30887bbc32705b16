/*
 * minlz_oracle.c — CPU restatement of the MinLZ block codec.  TEST INFRASTRUCTURE ONLY
 * (see minlz_oracle.h for the rules and the pinning status).
 *
 * Every function cites the reference file:line it restates (paths relative to the upstream
 * minio/minlz tree).  The code is written for clarity and exact behavioural agreement with the
 * reference's pure-Go path, not for speed.
 */
#include "minlz_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- format constants (encode.go:44-59, minlz.go:93-100) ---- */
#define MAX_COPY1_OFFSET 1024
#define MIN_COPY2_OFFSET 64
#define MAX_COPY2_OFFSET (MIN_COPY2_OFFSET + 65535)
#define COPY2_LIT_MAX_LEN (7 + 4)
#define MAX_COPY2_LITS 4
#define MAX_COPY3_LITS 3
#define MIN_COPY3_OFFSET 65536
#define MAX_COPY3_OFFSET ((2 << 20) + 65535)
#define INPUT_MARGIN 8                 /* encode.go:216 */
#define MIN_NON_LITERAL_BLOCK_SIZE 16  /* encode.go:220 */

#define TAG_LITERAL 0x00
#define TAG_REPEAT 0x04
#define TAG_COPY1 0x01
#define TAG_COPY2 0x02
#define TAG_COPY3 0x07
#define TAG_COPY2_FUSED 0x03

/* ---- unaligned little-endian loads/stores (unsafe_enabled.go:25-57) ---- */
static inline uint32_t ld32(const uint8_t* p, size_t i) { uint32_t v; memcpy(&v, p + i, 4); return v; }
static inline uint64_t ld64(const uint8_t* p, size_t i) { uint64_t v; memcpy(&v, p + i, 8); return v; }
static inline uint16_t ld16(const uint8_t* p, size_t i) { uint16_t v; memcpy(&v, p + i, 2); return v; }
static inline void st16(uint8_t* p, size_t i, uint16_t v) { memcpy(p + i, &v, 2); }
static inline void st32(uint8_t* p, size_t i, uint32_t v) { memcpy(p + i, &v, 4); }

/* ---- hashes (encode_l2.go:25-49, encode_l1.go:26-29) ---- */
static inline uint32_t hash4(uint64_t u, unsigned h) { return ((uint32_t)u * 2654435761u) >> (32 - h); }
static inline uint32_t hash5(uint64_t u, unsigned h) { return (uint32_t)(((u << 24) * 889523592379ull) >> (64 - h)); }
static inline uint32_t hash6(uint64_t u, unsigned h) { return (uint32_t)(((u << 16) * 227718039650203ull) >> (64 - h)); }
static inline uint32_t hash7(uint64_t u, unsigned h) { return (uint32_t)(((u << 8) * 58295818150454627ull) >> (64 - h)); }

/* ======================================================================================
 * Decoder
 * ====================================================================================== */

/* minLZDecodeGo, decode.go:178-622.  The reference has a fast loop (decode.go:188-359) and a
 * checked tail loop (:362-611) with identical results; this restates the checked loop, which
 * is the one that defines the error behaviour, cross-checked against
 * internal/reference/decoder.go:142-369. */
int mlzo_decode_body(uint8_t* dst, size_t dlen, const uint8_t* src, size_t slen) {
    size_t d = 0, s = 0, length = 0, offset = 1; /* decode.go:184-185 */
    while (s < slen) {
        uint8_t b = src[s];
        switch (b & 3) {
        case 0: { /* literal / repeat, decode.go:367-431 */
            unsigned x = b >> 3;
            if (x < 29) { s += 1; length = x + 1; }
            else if (x == 29) { s += 2; if (s > slen) return 1; length = (size_t)src[s - 1] + 30; }
            else if (x == 30) { s += 3; if (s > slen) return 1; length = ((size_t)src[s - 2] | (size_t)src[s - 1] << 8) + 30; }
            else { s += 4; if (s > slen) return 1; length = ((size_t)src[s - 3] | (size_t)src[s - 2] << 8 | (size_t)src[s - 1] << 16) + 30; }
            if (b & 4) break; /* repeat: copy with previous offset */
            if (length > dlen - d || length > slen - s) return 1; /* decode.go:412 */
            memcpy(dst + d, src + s, length);
            d += length; s += length;
            continue;
        }
        case 1: /* copy1, decode.go:433-459 */
            s += 2; if (s > slen) return 1;
            length = (src[s - 2] >> 2) & 15;
            offset = ((size_t)ld16(src, s - 2) >> 6) + 1;
            if (length == 15) { s++; if (s > slen) return 1; length = (size_t)src[s - 1] + 18; }
            else length += 4;
            break;
        case 2: /* copy2, decode.go:460-509 */
            s += 3; if (s > slen) return 1;
            length = src[s - 3] >> 2;
            offset = (size_t)src[s - 2] | (size_t)src[s - 1] << 8;
            if (length <= 60) length += 4;
            else if (length == 61) { s += 1; if (s > slen) return 1; length = (size_t)src[s - 1] + 64; }
            else if (length == 62) { s += 2; if (s > slen) return 1; length = ((size_t)src[s - 2] | (size_t)src[s - 1] << 8) + 64; }
            else { s += 3; if (s > slen) return 1; length = ((size_t)src[s - 3] | (size_t)src[s - 2] << 8 | (size_t)src[s - 1] << 16) + 64; }
            offset += MIN_COPY2_OFFSET;
            break;
        default: { /* fused copy2 / copy3, decode.go:510-582 */
            s += 4; if (s > slen) return 1;
            uint32_t val = ld32(src, s - 4);
            size_t litlen = (val >> 3) & 3;
            if (!(val & 4)) { /* fused copy2: 3-byte header */
                length = 4 + ((val >> 5) & 7);
                offset = ((val >> 8) & 65535) + MIN_COPY2_OFFSET;
                s--; litlen++;
            } else {
                unsigned lt = (val >> 5) & 63;
                offset = (size_t)(val >> 11) + MIN_COPY3_OFFSET;
                if (lt < 61) length = lt + 4;
                else if (lt == 61) { s += 1; if (s > slen) return 1; length = (size_t)src[s - 1] + 64; }
                else if (lt == 62) { s += 2; if (s > slen) return 1; length = ((size_t)src[s - 2] | (size_t)src[s - 1] << 8) + 64; }
                else { s += 3; if (s > slen) return 1; length = ((size_t)src[s - 3] | (size_t)src[s - 2] << 8 | (size_t)src[s - 1] << 16) + 64; }
            }
            if (litlen > 0) { /* literals come BEFORE the copy, decode.go:568-580 */
                if (litlen > dlen - d || s + litlen > slen) return 1;
                memcpy(dst + d, src + s, litlen);
                d += litlen; s += litlen;
            }
            break;
        }
        }
        /* doCopy2, decode.go:584-610 */
        if (offset == 0 || d < offset || length > dlen - d) return 1;
        if (offset > length) {
            memcpy(dst + d, dst + d - offset, length);
        } else { /* forward byte-by-byte: pattern replicate */
            for (size_t i = 0; i < length; i++) dst[d + i] = dst[d - offset + i];
        }
        d += length;
    }
    return d == dlen ? 0 : 1; /* decode.go:615 */
}

/* binary.Uvarint as used by decodedLen (decode.go:160-171): returns header length, 0 = need
 * more input, <0 = overflow. */
static int uvarint(const uint8_t* src, size_t slen, uint64_t* out) {
    uint64_t x = 0; unsigned shift = 0;
    for (size_t i = 0; i < slen; i++) {
        uint8_t b = src[i];
        if (i == 10) return -(int)(i + 1); /* MaxVarintLen64 overflow */
        if (b < 0x80) {
            if (i == 9 && b > 1) return -(int)(i + 1);
            *out = x | (uint64_t)b << shift;
            return (int)i + 1;
        }
        x |= (uint64_t)(b & 0x7f) << shift;
        shift += 7;
    }
    *out = 0;
    return 0;
}

static int put_uvarint(uint8_t* dst, uint64_t v) {
    int i = 0;
    while (v >= 0x80) { dst[i++] = (uint8_t)v | 0x80; v >>= 7; }
    dst[i++] = (uint8_t)v;
    return i;
}

/* decodedLen, decode.go:160-171 */
static int decoded_len_raw(const uint8_t* src, size_t slen, size_t* v, size_t* hdr) {
    uint64_t x; int n = uvarint(src, slen, &x);
    if (n <= 0 || x > 0xffffffffull) return MLZO_ERR_CORRUPT;
    *v = (size_t)x; *hdr = (size_t)n;
    return MLZO_OK;
}

/* isMinLZ, decode.go:120-156 */
int mlzo_is_minlz(const uint8_t* src, size_t slen, int* is_mlz, int* literals, size_t* body_off, size_t* size) {
    *is_mlz = 0; *literals = 0; *body_off = 0; *size = 0;
    if (slen <= 1) {
        if (slen == 0) return MLZO_ERR_CORRUPT;
        if (src[0] == 0) { *is_mlz = 1; *literals = 1; *body_off = 1; *size = 0; return MLZO_OK; }
    }
    if (src[0] != 0) { /* Snappy / S2 block: size is reported, decoding is out of scope */
        size_t v, h; int e = decoded_len_raw(src, slen, &v, &h);
        if (e) return e;
        *size = v; return MLZO_OK;
    }
    size_t v, h; int e = decoded_len_raw(src + 1, slen - 1, &v, &h);
    if (e) return e;
    if (v > MLZO_MAX_BLOCK_SIZE) return MLZO_ERR_TOO_LARGE;
    size_t off = 1 + h, rest = slen - off;
    if (rest == 0) return MLZO_ERR_CORRUPT;
    if (v == 0) { *is_mlz = 1; *literals = 1; *body_off = off; *size = rest; return MLZO_OK; }
    if (v < rest) { *body_off = off; *size = v; return MLZO_ERR_CORRUPT; }
    *is_mlz = 1; *body_off = off; *size = v;
    return MLZO_OK;
}

int mlzo_decoded_len(const uint8_t* src, size_t slen, size_t* dlen) {
    int a, b; size_t off;
    return mlzo_is_minlz(src, slen, &a, &b, &off, dlen);
}

/* Decode, decode.go:50-78 */
int mlzo_decode(const uint8_t* src, size_t slen, uint8_t* dst, size_t dcap, size_t* dlen) {
    int is_mlz, lits; size_t off, size;
    *dlen = 0;
    int e = mlzo_is_minlz(src, slen, &is_mlz, &lits, &off, &size);
    if (e) return e;
    if (lits) {
        if (size > dcap) return MLZO_ERR_DST_TOO_SMALL;
        memcpy(dst, src + off, size); *dlen = size; return MLZO_OK;
    }
    if (!is_mlz) return MLZO_ERR_UNSUPPORTED; /* decode.go:59-68 S2 fallback: out of scope */
    if (size > dcap) return MLZO_ERR_DST_TOO_SMALL;
    *dlen = size;
    if (mlzo_decode_body(dst, size, src + off, slen - off) != 0) return MLZO_ERR_CORRUPT;
    return MLZO_OK;
}

/* ======================================================================================
 * Emitters
 * ====================================================================================== */

/* emitLiteral, asm_none.go:84-122 */
size_t mlzo_emit_literal(uint8_t* dst, const uint8_t* lit, size_t len) {
    if (len == 0) return 0;
    size_t i, n = len - 1;
    if (n < 29) { dst[0] = (uint8_t)(n << 3) | TAG_LITERAL; i = 1; }
    else if (n < (1 << 8) + 29) { dst[1] = (uint8_t)(n - 29); dst[0] = 29 << 3 | TAG_LITERAL; i = 2; }
    else if (n < (1 << 16) + 29) { n -= 29; dst[2] = (uint8_t)(n >> 8); dst[1] = (uint8_t)n; dst[0] = 30 << 3 | TAG_LITERAL; i = 3; }
    else { n -= 29; dst[3] = (uint8_t)(n >> 16); dst[2] = (uint8_t)(n >> 8); dst[1] = (uint8_t)n; dst[0] = 31 << 3 | TAG_LITERAL; i = 4; }
    memcpy(dst + i, lit, len);
    return i + len;
}

/* emitRepeat, asm_none.go:125-156 */
size_t mlzo_emit_repeat(uint8_t* dst, size_t length) {
    if (length < 30) { dst[0] = (uint8_t)((length - 1) << 3) | TAG_REPEAT; return 1; }
    length -= 30;
    if (length < 256) { dst[1] = (uint8_t)length; dst[0] = 29 << 3 | TAG_REPEAT; return 2; }
    if (length < 65536) { dst[2] = (uint8_t)(length >> 8); dst[1] = (uint8_t)length; dst[0] = 30 << 3 | TAG_REPEAT; return 3; }
    dst[3] = (uint8_t)(length >> 16); dst[2] = (uint8_t)(length >> 8); dst[1] = (uint8_t)length; dst[0] = 31 << 3 | TAG_REPEAT;
    return 4;
}

/* encodeCopy3, asm_none.go:160-200 */
static size_t encode_copy3(uint8_t* dst, size_t offset, size_t length, size_t lits) {
    length -= 4;
    uint32_t enc = (uint32_t)(offset - 65536) << 11 | TAG_COPY3 | (uint32_t)(lits << 3);
    if (length <= 60) { enc |= (uint32_t)(length << 5); st32(dst, 0, enc); return 4; }
    length -= 60;
    if (length < 256) { dst[4] = (uint8_t)length; enc |= 61 << 5; st32(dst, 0, enc); return 5; }
    if (length < 65536) { enc |= 62 << 5; dst[5] = (uint8_t)(length >> 8); dst[4] = (uint8_t)length; st32(dst, 0, enc); return 6; }
    enc |= 63 << 5; dst[6] = (uint8_t)(length >> 16); dst[5] = (uint8_t)(length >> 8); dst[4] = (uint8_t)length; st32(dst, 0, enc);
    return 7;
}

/* encodeCopy2, encode.go:247-282 */
static size_t encode_copy2(uint8_t* dst, size_t offset, size_t length) {
    length -= 4; offset -= MIN_COPY2_OFFSET;
    st16(dst, 1, (uint16_t)offset);
    if (length <= 60) { dst[0] = (uint8_t)(length << 2) | TAG_COPY2; return 3; }
    length -= 60;
    if (length < 256) { dst[3] = (uint8_t)length; dst[0] = 61 << 2 | TAG_COPY2; return 4; }
    if (length < 65536) { dst[4] = (uint8_t)(length >> 8); dst[3] = (uint8_t)length; dst[0] = 62 << 2 | TAG_COPY2; return 5; }
    dst[5] = (uint8_t)(length >> 16); dst[4] = (uint8_t)(length >> 8); dst[3] = (uint8_t)length; dst[0] = 63 << 2 | TAG_COPY2;
    return 6;
}

/* emitCopy, asm_none.go:207-278 */
size_t mlzo_emit_copy(uint8_t* dst, size_t offset, size_t length) {
    if (offset > MAX_COPY2_OFFSET) return encode_copy3(dst, offset, length, 0);
    if (offset <= MAX_COPY1_OFFSET) {
        offset--;
        if (length < 15 + 4) { st16(dst, 0, (uint16_t)(offset << 6) | (uint16_t)((length - 4) << 2) | TAG_COPY1); return 2; }
        if (length < 256 + 18) { st16(dst, 0, (uint16_t)(offset << 6) | (15 << 2 | TAG_COPY1)); dst[2] = (uint8_t)(length - 18); return 3; }
        st16(dst, 0, (uint16_t)(offset << 6) | (14 << 2) | TAG_COPY1); /* copy1 of 18 + repeat */
        return 2 + mlzo_emit_repeat(dst + 2, length - 18);
    }
    return encode_copy2(dst, offset, length);
}

/* emitCopyLits2, asm_none.go:284-308 */
size_t mlzo_emit_copy_lits2(uint8_t* dst, const uint8_t* lits, size_t nlits, size_t offset, size_t length) {
    offset -= MIN_COPY2_OFFSET;
    length -= 4;
    const size_t max_raw = COPY2_LIT_MAX_LEN - 4;
    st16(dst, 1, (uint16_t)offset);
    if (length > max_raw) {
        dst[0] = TAG_COPY2_FUSED | (uint8_t)(max_raw << 5) | (uint8_t)((nlits - 1) << 3);
        memcpy(dst + 3, lits, nlits);
        size_t n = nlits + 3;
        return n + mlzo_emit_repeat(dst + n, length - max_raw);
    }
    dst[0] = TAG_COPY2_FUSED | (uint8_t)(length << 5) | (uint8_t)((nlits - 1) << 3);
    memcpy(dst + 3, lits, nlits);
    return nlits + 3;
}

/* emitCopyLits3, asm_none.go:313-323 */
size_t mlzo_emit_copy_lits3(uint8_t* dst, const uint8_t* lits, size_t nlits, size_t offset, size_t length) {
    size_t n = encode_copy3(dst, offset, length, nlits);
    memcpy(dst + n, lits, nlits);
    return n + nlits;
}

/* ======================================================================================
 * Encoders
 * ====================================================================================== */

/* MaxEncodedLen, encode.go:234-244 */
long mlzo_max_encoded_len(size_t n) {
    if (n > MLZO_MAX_BLOCK_SIZE) return -1;
    if (n == 0) return 1;
    return (long)n + 2;
}

static inline int ctz64(uint64_t v) { return __builtin_ctzll(v); }

/* Forward extension used by L1 (encode_l1.go:181-188): 8 bytes at a time while s <= len-8. */
static inline long extend8(const uint8_t* src, long n, long s, long cand) {
    while (s <= n - 8) {
        uint64_t diff = ld64(src, s) ^ ld64(src, cand);
        if (diff) { s += ctz64(diff) >> 3; break; }
        s += 8; cand += 8;
    }
    return s;
}

/* Forward extension used by L2 (encode_l2.go:236-251): runs to the very end of src. */
static inline long extend_full(const uint8_t* src, long n, long s, long cand) {
    while (s < n) {
        if (n - s < 8) {
            if (src[s] == src[cand]) { s++; cand++; continue; }
            break;
        }
        uint64_t diff = ld64(src, s) ^ ld64(src, cand);
        if (diff) { s += ctz64(diff) >> 3; break; }
        s += 8; cand += 8;
    }
    return s;
}

/*
 * L1 "Fastest".  encodeBlockGo (encode_l1.go:39-283; BIG=1: hash6, 15 bits, u32 entries,
 * skipLog 6, <=3 fused literals, copy3 allowed, minSrcPos window check) and encodeBlockGo64K
 * (encode_l1.go:285-524; BIG=0: hash5, 13 bits, u16 entries, skipLog 5, <=4 fused literals).
 */
#define DEFINE_L1(NAME, BIG, TBITS, TTYPE, HASH, SKIPLOG, MAXLITS)                                        \
    static size_t NAME(uint8_t* dst, const uint8_t* src, long n) {                                         \
        TTYPE* table = (TTYPE*)calloc((size_t)1 << TBITS, sizeof(TTYPE));                                  \
        long sLimit = n - INPUT_MARGIN;                                                                    \
        long dstLimit = n - (n >> 5) - 6;                                                                  \
        long nextEmit = 0, s = 1, d = 0, repeat = 1;                                                       \
        uint64_t cv = ld64(src, s);                                                                        \
        for (;;) {                                                                                         \
            long candidate = 0;                                                                            \
            for (;;) {                                                                                     \
                long nextS = s + ((s - nextEmit) >> SKIPLOG) + 4;                                          \
                if (nextS > sLimit) goto emit_remainder;                                                   \
                long minSrcPos = BIG ? s - MAX_COPY3_OFFSET : 0;                                           \
                uint32_t h0 = HASH(cv, TBITS), h1 = HASH(cv >> 8, TBITS);                                  \
                candidate = (long)table[h0];                                                               \
                long candidate2 = (long)table[h1];                                                         \
                table[h0] = (TTYPE)s;                                                                      \
                table[h1] = (TTYPE)(s + 1);                                                                \
                uint32_t h2 = HASH(cv >> 16, TBITS);                                                       \
                /* repeat check one byte ahead (checkRep = 1) */                                           \
                if ((uint32_t)(cv >> 8) == ld32(src, s - repeat + 1)) {                                    \
                    long base = s + 1;                                                                     \
                    for (long i = base - repeat; base > nextEmit && i > 0 && src[i - 1] == src[base - 1];) { i--; base--; } \
                    if (d + (base - nextEmit) > dstLimit) { free(table); return 0; }                       \
                    d += mlzo_emit_literal(dst + d, src + nextEmit, base - nextEmit);                      \
                    long cand = s - repeat + 4 + 1;                                                        \
                    s += 4 + 1;                                                                            \
                    while (s <= sLimit) {                                                                  \
                        uint64_t diff = ld64(src, s) ^ ld64(src, cand);                                    \
                        if (diff) { s += ctz64(diff) >> 3; break; }                                        \
                        s += 8; cand += 8;                                                                 \
                    }                                                                                      \
                    d += mlzo_emit_repeat(dst + d, s - base);                                              \
                    nextEmit = s;                                                                          \
                    if (s >= sLimit) goto emit_remainder;                                                  \
                    cv = ld64(src, s);                                                                     \
                    continue;                                                                              \
                }                                                                                          \
                if (candidate >= minSrcPos && (uint32_t)cv == ld32(src, candidate)) break;                 \
                candidate = (long)table[h2];                                                               \
                if (candidate2 >= minSrcPos && (uint32_t)(cv >> 8) == ld32(src, candidate2)) {             \
                    table[h2] = (TTYPE)(s + 2);                                                            \
                    candidate = candidate2; s++;                                                           \
                    break;                                                                                 \
                }                                                                                          \
                table[h2] = (TTYPE)(s + 2);                                                                \
                if (candidate >= minSrcPos && (uint32_t)(cv >> 16) == ld32(src, candidate)) { s += 2; break; } \
                cv = ld64(src, nextS);                                                                     \
                s = nextS;                                                                                 \
            }                                                                                              \
            while (candidate > 0 && s > nextEmit && src[candidate - 1] == src[s - 1]) { candidate--; s--; } \
            long base = s;                                                                                 \
            repeat = base - candidate;                                                                     \
            s = extend8(src, n, s + 4, candidate + 4);                                                     \
            long length = s - base;                                                                        \
            if (nextEmit != base) {                                                                        \
                if (base - nextEmit > MAXLITS || repeat < MIN_COPY2_OFFSET) {                              \
                    if (d + (s - nextEmit) > dstLimit) { free(table); return 0; }                          \
                    d += mlzo_emit_literal(dst + d, src + nextEmit, base - nextEmit);                      \
                    d += mlzo_emit_copy(dst + d, repeat, length);                                          \
                } else if (!BIG || repeat <= MAX_COPY2_OFFSET) {                                           \
                    d += mlzo_emit_copy_lits2(dst + d, src + nextEmit, base - nextEmit, repeat, length);   \
                } else {                                                                                   \
                    d += mlzo_emit_copy_lits3(dst + d, src + nextEmit, base - nextEmit, repeat, length);   \
                }                                                                                          \
            } else {                                                                                       \
                d += mlzo_emit_copy(dst + d, repeat, length);                                              \
            }                                                                                              \
            for (;;) { /* immediate re-match loop */                                                       \
                nextEmit = s;                                                                              \
                if (s >= sLimit) goto emit_remainder;                                                      \
                uint64_t x = ld64(src, s - 2);                                                             \
                if (d > dstLimit) { free(table); return 0; }                                               \
                uint32_t m2 = HASH(x, TBITS);                                                              \
                x >>= 16;                                                                                  \
                uint32_t cur = HASH(x, TBITS);                                                             \
                candidate = (long)table[cur];                                                              \
                table[m2] = (TTYPE)(s - 2);                                                                \
                table[cur] = (TTYPE)s;                                                                     \
                if ((BIG && s - candidate > MAX_COPY3_OFFSET) || (uint32_t)x != ld32(src, candidate)) {    \
                    cv = ld64(src, s + 1);                                                                 \
                    s++;                                                                                   \
                    break;                                                                                 \
                }                                                                                          \
                repeat = s - candidate;                                                                    \
                base = s;                                                                                  \
                s = extend8(src, n, s + 4, candidate + 4);                                                 \
                d += mlzo_emit_copy(dst + d, repeat, s - base);                                            \
            }                                                                                              \
        }                                                                                                  \
    emit_remainder:                                                                                        \
        if (nextEmit < n) {                                                                                \
            if (d + n - nextEmit > dstLimit) { free(table); return 0; }                                    \
            d += mlzo_emit_literal(dst + d, src + nextEmit, n - nextEmit);                                 \
        }                                                                                                  \
        free(table);                                                                                       \
        return (size_t)d;                                                                                  \
    }

DEFINE_L1(l1_big, 1, 15, uint32_t, hash6, 6, MAX_COPY3_LITS)
DEFINE_L1(l1_64k, 0, 13, uint16_t, hash5, 5, MAX_COPY2_LITS)

/* encodeBlock, asm_none.go:51-59 */
size_t mlzo_encode_block_l1(uint8_t* dst, const uint8_t* src, size_t n) {
    if (n < MIN_NON_LITERAL_BLOCK_SIZE) return 0;
    if (n <= 65536) return l1_64k(dst, src, (long)n);
    return l1_big(dst, src, (long)n);
}

/*
 * L2 "Balanced".  encodeBlockBetterGo (encode_l2.go:61-338; BIG=1: long hash7/17 bits, short
 * hash4/14 bits, u32) and encodeBlockBetterGo64K (encode_l2.go:343-596; BIG=0: hash6/15,
 * hash4/12, u16, no window checks, no far-short-match rejection).
 */
#define DEFINE_L2(NAME, BIG, LBITS, SBITS, TTYPE, LHASH)                                                   \
    static size_t NAME(uint8_t* dst, const uint8_t* src, long n) {                                         \
        long sLimit = n - INPUT_MARGIN;                                                                    \
        TTYPE* lTable = (TTYPE*)calloc((size_t)1 << LBITS, sizeof(TTYPE));                                 \
        TTYPE* sTable = (TTYPE*)calloc((size_t)1 << SBITS, sizeof(TTYPE));                                 \
        long dstLimit = n - (n >> 5) - 6;                                                                  \
        long nextEmit = 0, s = 1, d = 0, repeat = 1;                                                       \
        uint64_t cv = ld64(src, s);                                                                        \
        size_t ret = 0;                                                                                    \
        for (;;) {                                                                                         \
            long candidateL = 0, nextS = 0;                                                                \
            for (;;) {                                                                                     \
                nextS = s + ((s - nextEmit) >> 7) + 1;                                                     \
                if (nextS > sLimit) goto emit_remainder;                                                   \
                long minSrcPos = s - MAX_COPY3_OFFSET + 1;                                                 \
                uint32_t hL = LHASH(cv, LBITS), hS = hash4(cv, SBITS);                                     \
                candidateL = (long)lTable[hL];                                                             \
                long candidateS = (long)sTable[hS];                                                        \
                lTable[hL] = (TTYPE)s;                                                                     \
                sTable[hS] = (TTYPE)s;                                                                     \
                uint64_t valLong = ld64(src, candidateL), valShort = ld64(src, candidateS);                \
                if ((!BIG || candidateL > minSrcPos) && cv == valLong) break;                              \
                /* repeat: 4 bytes at s+1 (checkRep = 1, wantRepeatBytes = 4) */                           \
                const uint64_t repeatMask = 0xffffffffull << 8;                                            \
                if (repeat > 0 && (cv & repeatMask) == (ld64(src, s - repeat) & repeatMask)) {             \
                    long base = s + 1;                                                                     \
                    for (long i = base - repeat; base > nextEmit && i > 0 && src[i - 1] == src[base - 1];) { i--; base--; } \
                    if (d + (base - nextEmit) > dstLimit) goto done;                                       \
                    d += mlzo_emit_literal(dst + d, src + nextEmit, base - nextEmit);                      \
                    long cand = s - repeat + 4 + 1;                                                        \
                    s = extend_full(src, n, s + 4 + 1, cand);                                              \
                    d += mlzo_emit_repeat(dst + d, s - base);                                              \
                    nextEmit = s;                                                                          \
                    if (s >= sLimit) goto emit_remainder;                                                  \
                    long index0 = base + 1, index1 = s - 2;                                                \
                    while (index0 < index1) {                                                              \
                        uint64_t cv0 = ld64(src, index0), cv1 = ld64(src, index1);                         \
                        lTable[LHASH(cv0, LBITS)] = (TTYPE)index0;                                         \
                        sTable[hash4(cv0 >> 8, SBITS)] = (TTYPE)(index0 + 1);                              \
                        lTable[LHASH(cv1, LBITS)] = (TTYPE)index1;                                         \
                        sTable[hash4(cv1 >> 8, SBITS)] = (TTYPE)(index1 + 1);                              \
                        index0 += 2; index1 -= 2;                                                          \
                    }                                                                                      \
                    cv = ld64(src, s);                                                                     \
                    continue;                                                                              \
                }                                                                                          \
                if ((!BIG || candidateL >= minSrcPos) && (uint32_t)cv == (uint32_t)valLong) break;         \
                if ((!BIG || candidateS >= minSrcPos) && (uint32_t)cv == (uint32_t)valShort) {             \
                    hL = LHASH(cv >> 8, LBITS);                                                            \
                    candidateL = (long)lTable[hL];                                                         \
                    lTable[hL] = (TTYPE)(s + 1);                                                           \
                    if ((!BIG || candidateL > minSrcPos) && (uint32_t)(cv >> 8) == ld32(src, candidateL)) { s++; break; } \
                    candidateL = candidateS;                                                               \
                    break;                                                                                 \
                }                                                                                          \
                cv = ld64(src, nextS);                                                                     \
                s = nextS;                                                                                 \
            }                                                                                              \
            while (candidateL > 0 && s > nextEmit && src[candidateL - 1] == src[s - 1]) { candidateL--; s--; } \
            if (d + (s - nextEmit) > dstLimit) goto done;                                                  \
            long base = s, offset = base - candidateL;                                                     \
            s = extend_full(src, n, s + 4, candidateL + 4);                                                \
            if (BIG && offset > 65535 && s - base <= 4 && repeat != offset) {                              \
                s = nextS + 1;                                                                             \
                if (s >= sLimit) goto emit_remainder;                                                      \
                cv = ld64(src, s);                                                                         \
                continue;                                                                                  \
            }                                                                                              \
            long nl = base - nextEmit;                                                                     \
            if (nl > 0) {                                                                                  \
                if (!BIG || offset <= MAX_COPY2_OFFSET) {                                                  \
                    if (nl > MAX_COPY2_LITS || offset < 64) {                                              \
                        d += mlzo_emit_literal(dst + d, src + nextEmit, nl);                               \
                        d += mlzo_emit_copy(dst + d, offset, s - base);                                    \
                    } else d += mlzo_emit_copy_lits2(dst + d, src + nextEmit, nl, offset, s - base);       \
                } else {                                                                                   \
                    if (nl > MAX_COPY3_LITS) {                                                             \
                        d += mlzo_emit_literal(dst + d, src + nextEmit, nl);                               \
                        d += mlzo_emit_copy(dst + d, offset, s - base);                                    \
                    } else d += mlzo_emit_copy_lits3(dst + d, src + nextEmit, nl, offset, s - base);       \
                }                                                                                          \
            } else d += mlzo_emit_copy(dst + d, offset, s - base);                                         \
            repeat = offset;                                                                               \
            nextEmit = s;                                                                                  \
            if (s >= sLimit) goto emit_remainder;                                                          \
            if (d > dstLimit) goto done;                                                                   \
            long index0 = base + 1, index1 = s - 2;                                                        \
            uint64_t cv0 = ld64(src, index0), cv1 = ld64(src, index1);                                     \
            lTable[LHASH(cv0, LBITS)] = (TTYPE)index0;                                                     \
            sTable[hash4(cv0 >> 8, SBITS)] = (TTYPE)(index0 + 1);                                          \
            lTable[LHASH(cv1, LBITS)] = (TTYPE)index1;                                                     \
            sTable[hash4(cv1 >> 8, SBITS)] = (TTYPE)(index1 + 1);                                          \
            index0 += 1; index1 -= 1;                                                                      \
            cv = ld64(src, s);                                                                             \
            long index2 = (index0 + index1 + 1) >> 1;                                                      \
            while (index2 < index1) {                                                                      \
                lTable[LHASH(ld64(src, index0), LBITS)] = (TTYPE)index0;                                   \
                lTable[LHASH(ld64(src, index2), LBITS)] = (TTYPE)index2;                                   \
                index0 += 2; index2 += 2;                                                                  \
            }                                                                                              \
        }                                                                                                  \
    emit_remainder:                                                                                        \
        if (nextEmit < n) {                                                                                \
            if (d + n - nextEmit > dstLimit) goto done;                                                    \
            d += mlzo_emit_literal(dst + d, src + nextEmit, n - nextEmit);                                 \
        }                                                                                                  \
        ret = (size_t)d;                                                                                   \
    done:                                                                                                  \
        free(lTable); free(sTable);                                                                        \
        return ret;                                                                                        \
    }

DEFINE_L2(l2_big, 1, 17, 14, uint32_t, hash7)
DEFINE_L2(l2_64k, 0, 15, 12, uint16_t, hash6)

/* encodeBlockBetter, asm_none.go:68-76 */
size_t mlzo_encode_block_l2(uint8_t* dst, const uint8_t* src, size_t n) {
    if (n < MIN_NON_LITERAL_BLOCK_SIZE) return 0;
    if (n <= (64 << 10)) return l2_64k(dst, src, (long)n);
    return l2_big(dst, src, (long)n);
}

/*
 * L3 "Smallest".  encodeBlockBest (encode_l3.go:38-625) without dictionary support (dict is nil
 * for block encoding, encode.go:101).  Two chained tables (cur | prev<<32): long hash8/20 bits and
 * short hash4/18 bits; every candidate is scored (encode_l3.go:139-160) and the best of up to
 * ~20 candidates at s, s+1, s+2 and "at the end of the best match" wins (bestOf, :338-373).
 */
typedef struct { long offset, s, length, score; int rep, nextrep; } l3match;

static inline long l3_lit_size(long n) { /* emitLiteralSizeN, encode.go:285-299 */
    if (n == 0) return 0;
    if (n <= 29) return 1;
    if (n < 29 + (1 << 8)) return 2;
    if (n < 29 + (1 << 16)) return 3;
    return 4;
}
static inline long l3_repeat_size(long length) { /* emitRepeatSize, encode_l3.go:664-680 */
    if (length <= 0) return 0;
    if (length <= 29) return 1;
    length -= 29;
    if (length <= 256) return 2;
    if (length <= 65536) return 3;
    return 4;
}
static inline long l3_copy2_size(long length) { /* emitCopy2Size, encode_l3.go:684-699 */
    length -= 4;
    if (length <= 60) return 3;
    length -= 60;
    if (length < 256) return 4;
    if (length < 65536) return 5;
    return 6;
}
static inline long bits_len_u(unsigned long v) { long n = 0; while (v) { n++; v >>= 1; } return n; }
static inline long l3_copy_size(long offset, long length) { /* emitCopySize, encode_l3.go:633-660 */
    if (offset > 65536 + 63) {
        length -= 64;
        if (length <= 0) return 4;
        return 4 + (bits_len_u((unsigned long)length) + 7) / 8;
    }
    if (offset <= 1024) {
        if (length <= 18) return 2;
        if (length < 18 + 256) return 3;
        return 2 + l3_repeat_size(length - 18);
    }
    return l3_copy2_size(length);
}
static inline uint32_t hash8(uint64_t u, unsigned h) { return (uint32_t)((u * 0xcf1bbcdcb7a56463ull) >> (64 - h)); }

/*
 * L0 "SuperFast".  encodeFastBlockGo (encode_l0.go:32-279; BIG=1: hash8, 13 bits, u32 entries, skipLog 5,
 * nextS + 5, dstLimit n - n/8 - 6, <=3 fused literals, copy3 allowed, minSrcPos window check) and
 * encodeFastBlockGo64K (encode_l0.go:281-522; BIG=0: 12 bits, u16 entries, skipLog 4, nextS + 4,
 * dstLimit n - n/16 - 32, <=4 fused literals).  Differences from L1: 8-byte candidate checks, no backward
 * extension (the loop is disabled upstream: `for false && ...`, encode_l0.go:133), forward extension from +8.
 * The reference reads cv2 = load64(src, s+2) with s <= len-8, i.e. up to two bytes past the end of src through
 * its unsafe loads (unsafe_enabled.go:42-46); this restatement reads those bytes as zero.
 */
static inline uint64_t ld64z(const uint8_t* src, long n, long i) {
    if (i + 8 <= n) return ld64(src, i);
    uint64_t v = 0;
    if (i < n) memcpy(&v, src + i, (size_t)(n - i));
    return v;
}
#define DEFINE_L0(NAME, BIG, TBITS, TTYPE, SKIPLOG, SKIPADD, DSTLIMIT, MAXLITS)                            \
    static size_t NAME(uint8_t* dst, const uint8_t* src, long n) {                                         \
        TTYPE* table = (TTYPE*)calloc((size_t)1 << TBITS, sizeof(TTYPE));                                  \
        long sLimit = n - INPUT_MARGIN;                                                                    \
        long dstLimit = DSTLIMIT;                                                                          \
        long nextEmit = 0, s = 1, d = 0, repeat = 1;                                                       \
        uint64_t cv = ld64(src, s);                                                                        \
        for (;;) {                                                                                         \
            long candidate = 0;                                                                            \
            for (;;) {                                                                                     \
                long nextS = s + ((s - nextEmit) >> SKIPLOG) + SKIPADD;                                    \
                if (nextS > sLimit) goto emit_remainder;                                                   \
                long minSrcPos = BIG ? s - MAX_COPY3_OFFSET : 0;                                           \
                uint32_t h0 = hash8(cv, TBITS);                                                            \
                uint64_t cv1 = ld64z(src, n, s + 1);                                                       \
                uint32_t h1 = hash8(cv1, TBITS);                                                           \
                candidate = (long)table[h0];                                                               \
                long candidate2 = (long)table[h1];                                                         \
                table[h0] = (TTYPE)s;                                                                      \
                table[h1] = (TTYPE)(s + 1);                                                                \
                uint64_t cv2 = ld64z(src, n, s + 2);                                                       \
                uint32_t h2 = hash8(cv2, TBITS);                                                           \
                if ((uint32_t)cv1 == ld32(src, s - repeat + 1)) {  /* checkRep = 1 */                      \
                    long base = s + 1;                                                                     \
                    for (long i = base - repeat; base > nextEmit && i > 0 && src[i - 1] == src[base - 1];) { i--; base--; } \
                    if (d + (base - nextEmit) > dstLimit) { free(table); return 0; }                       \
                    d += mlzo_emit_literal(dst + d, src + nextEmit, base - nextEmit);                      \
                    long cand = s - repeat + 4 + 1;                                                        \
                    s += 4 + 1;                                                                            \
                    while (s <= sLimit) {                                                                  \
                        uint64_t diff = ld64(src, s) ^ ld64(src, cand);                                    \
                        if (diff) { s += ctz64(diff) >> 3; break; }                                        \
                        s += 8; cand += 8;                                                                 \
                    }                                                                                      \
                    d += mlzo_emit_repeat(dst + d, s - base);                                              \
                    nextEmit = s;                                                                          \
                    if (s >= sLimit) goto emit_remainder;                                                  \
                    cv = ld64(src, s);                                                                     \
                    continue;                                                                              \
                }                                                                                          \
                if (candidate >= minSrcPos && cv == ld64(src, candidate)) break;                           \
                candidate = (long)table[h2];                                                               \
                if (candidate2 >= minSrcPos && cv1 == ld64(src, candidate2)) {                             \
                    table[h2] = (TTYPE)(s + 2);                                                            \
                    candidate = candidate2; s++;                                                           \
                    break;                                                                                 \
                }                                                                                          \
                table[h2] = (TTYPE)(s + 2);                                                                \
                if (candidate >= minSrcPos && cv2 == ld64(src, candidate)) { s += 2; break; }              \
                cv = ld64(src, nextS);                                                                     \
                s = nextS;                                                                                 \
            }                                                                                              \
            long base = s;                                                                                 \
            repeat = base - candidate;                                                                     \
            s = extend8(src, n, s + 8, candidate + 8);                                                     \
            long length = s - base;                                                                        \
            if (nextEmit != base) {                                                                        \
                if (base - nextEmit > MAXLITS || repeat < MIN_COPY2_OFFSET) {                              \
                    if (d + (s - nextEmit) > dstLimit) { free(table); return 0; }                          \
                    d += mlzo_emit_literal(dst + d, src + nextEmit, base - nextEmit);                      \
                    d += mlzo_emit_copy(dst + d, repeat, length);                                          \
                } else if (!BIG || repeat <= MAX_COPY2_OFFSET) {                                           \
                    d += mlzo_emit_copy_lits2(dst + d, src + nextEmit, base - nextEmit, repeat, length);   \
                } else {                                                                                   \
                    d += mlzo_emit_copy_lits3(dst + d, src + nextEmit, base - nextEmit, repeat, length);   \
                }                                                                                          \
            } else {                                                                                       \
                d += mlzo_emit_copy(dst + d, repeat, length);                                              \
            }                                                                                              \
            for (;;) { /* immediate re-match loop */                                                       \
                nextEmit = s;                                                                              \
                if (s >= sLimit) goto emit_remainder;                                                      \
                uint64_t x = ld64(src, s - 2);                                                             \
                if (d > dstLimit) { free(table); return 0; }                                               \
                uint32_t m2 = hash8(x, TBITS);                                                             \
                x = ld64(src, s);                                                                          \
                uint32_t cur = hash8(x, TBITS);                                                            \
                candidate = (long)table[cur];                                                              \
                table[m2] = (TTYPE)(s - 2);                                                                \
                table[cur] = (TTYPE)s;                                                                     \
                if ((BIG && s - candidate > MAX_COPY3_OFFSET) || x != ld64(src, candidate)) {              \
                    cv = ld64z(src, n, s + 1);                                                             \
                    s++;                                                                                   \
                    break;                                                                                 \
                }                                                                                          \
                repeat = s - candidate;                                                                    \
                base = s;                                                                                  \
                s = extend8(src, n, s + 8, candidate + 8);                                                 \
                d += mlzo_emit_copy(dst + d, repeat, s - base);                                            \
            }                                                                                              \
        }                                                                                                  \
    emit_remainder:                                                                                        \
        if (nextEmit < n) {                                                                                \
            if (d + n - nextEmit > dstLimit) { free(table); return 0; }                                    \
            d += mlzo_emit_literal(dst + d, src + nextEmit, n - nextEmit);                                 \
        }                                                                                                  \
        free(table);                                                                                       \
        return (size_t)d;                                                                                  \
    }

DEFINE_L0(l0_big, 1, 13, uint32_t, 5, 5, n - (n >> 3) - 6, MAX_COPY3_LITS)
DEFINE_L0(l0_64k, 0, 12, uint16_t, 4, 4, n - (n >> 4) - 32, MAX_COPY2_LITS)

/* encodeBlockFast, asm_none.go:33-42 */
size_t mlzo_encode_block_l0(uint8_t* dst, const uint8_t* src, size_t n) {
    if (n < MIN_NON_LITERAL_BLOCK_SIZE) return 0;
    if (n <= 65536) return l0_64k(dst, src, (long)n);
    return l0_big(dst, src, (long)n);
}

typedef struct { const uint8_t* src; long n, sLimit, nextEmit; const l3match* best; } l3ctx;

static long l3_score(const l3ctx* c, const l3match* m) { /* encode_l3.go:139-160 */
    long ll = m->s - c->nextEmit;
    long score = m->length - l3_lit_size(ll) - m->s;
    long offset = m->s - m->offset;
    if (m->rep) return score - l3_repeat_size(m->length);
    if (ll > 0 && offset > 1024) {
        if (ll <= MAX_COPY2_LITS && offset < 65536 + 63 && m->length <= COPY2_LIT_MAX_LEN) score++;
        else if (ll <= MAX_COPY3_LITS) score++;
    }
    return score - l3_copy_size(offset, m->length);
}

static l3match l3_extend(const l3ctx* c, long offset, long s, long matched, int rep) {
    const uint8_t* src = c->src;
    l3match m = {offset, s, matched + offset, 0, rep, 0};
    s += matched;
    while (s < c->n) { /* forward, encode_l3.go:177-191 */
        if (c->n - s < 8) {
            if (src[s] == src[m.length]) { m.length++; s++; continue; }
            break;
        }
        uint64_t diff = ld64(src, s) ^ ld64(src, m.length);
        if (diff) { m.length += ctz64(diff) >> 3; break; }
        s += 8; m.length += 8;
    }
    while (m.s > c->nextEmit && m.offset > 0) { /* backward, :193-200 */
        if (src[m.offset - 1] != src[m.s - 1]) break;
        m.s--; m.offset--; m.length++;
    }
    m.length -= offset;
    return m;
}

static l3match l3_match_at(const l3ctx* c, long offset, long s, uint32_t first) { /* matchAt, :162-215 */
    l3match none = {offset, s, 0, 0, 0, 0};
    const l3match* best = c->best;
    if ((best->length != 0 && best->s - best->offset == s - offset) || s - offset >= MAX_COPY3_OFFSET || s <= offset) return none;
    if (ld32(c->src, offset) != first) return none;
    l3match m = l3_extend(c, offset, s, 4, 0);
    m.score = l3_score(c, &m);
    if (m.score <= -m.s) m.length = 0;
    if (m.s + m.length < c->sLimit) {
        long a = m.s + m.length + 1, b = m.offset + m.length + 1;
        m.nextrep = ld32(c->src, a) == ld32(c->src, b);
    }
    return m;
}

static l3match l3_match_repeat(const l3ctx* c, long offset, long s, uint32_t first) { /* matchAtRepeat, :216-263 */
    l3match none = {offset, s, 0, 0, 0, 0};
    if (c->best->rep) return none;
    const uint32_t mask = (1u << 24) - 1;
    if ((ld32(c->src, offset) & mask) != (first & mask)) return none;
    l3match m = l3_extend(c, offset, s, 3, 1);
    if (m.s + m.length < c->sLimit) {
        long a = m.s + m.length + 1, b = m.offset + m.length + 1;
        m.nextrep = ld32(c->src, a) == ld32(c->src, b);
    }
    m.score = l3_score(c, &m);
    return m;
}

static l3match l3_best_of(l3match a, l3match b) { /* bestOf, :338-373 */
    if (b.length == 0) return a;
    if (a.length == 0) return b;
    if (a.score > b.score) return a;
    if (b.score > a.score) return b;
    if (a.s != b.s) return a.s < b.s ? a : b;
    if (a.nextrep != b.nextrep) return a.nextrep ? a : b;
    return a.offset > b.offset ? a : b;
}

size_t mlzo_encode_block_l3(uint8_t* dst, const uint8_t* src, size_t n_) {
    const long n = (long)n_;
    enum { LBITS = 20, SBITS = 18, MAXSKIP = 64 };
    if (n < MIN_NON_LITERAL_BLOCK_SIZE) return 0;
    const long sLimit = n - (8 + 2);
    uint64_t* lTable = (uint64_t*)calloc((size_t)1 << LBITS, 8);
    uint64_t* sTable = (uint64_t*)calloc((size_t)1 << SBITS, 8);
    const long dstLimit = n - 5;
    long nextEmit = 0, s = 1, repeat = 1, d = 0;
    uint64_t cv = ld64(src, s);
    size_t ret = 0;
#define CUR(x) ((long)((x) & 0xffffffffull))
#define PREV(x) ((long)((x) >> 32))
    for (;;) {
        l3match best = {0, 0, 0, 0, 0, 0};
        l3ctx c = {src, n, sLimit, nextEmit, &best};
        for (;;) {
            long nextS = ((s - nextEmit) >> 8) + 1;
            if (nextS > MAXSKIP) nextS = s + MAXSKIP; else nextS += s;
            if (nextS > sLimit) goto emit_remainder;
            const uint32_t hashL = hash8(cv, LBITS), hashS = hash4(cv, SBITS);
            const uint64_t candidateL = lTable[hashL], candidateS = sTable[hashS];
            if (s > 0) {
                best = l3_best_of(l3_match_at(&c, CUR(candidateL), s, (uint32_t)cv), l3_match_at(&c, PREV(candidateL), s, (uint32_t)cv));
                best = l3_best_of(best, l3_match_at(&c, CUR(candidateS), s, (uint32_t)cv));
                best = l3_best_of(best, l3_match_at(&c, PREV(candidateS), s, (uint32_t)cv));
            }
            best = l3_best_of(best, l3_match_repeat(&c, s - repeat, s, (uint32_t)cv));
            best = l3_best_of(best, l3_match_repeat(&c, s - repeat + 1, s + 1, (uint32_t)(cv >> 8)));
            if (best.length > 0) {
                const uint32_t hS1 = hash4(cv >> 8, SBITS);
                uint64_t nextShort = sTable[hS1];
                long sFwd = s + 1;
                uint64_t cv2 = ld64(src, sFwd);
                uint64_t nextLong = lTable[hash8(cv2, LBITS)];
                best = l3_best_of(best, l3_match_at(&c, CUR(nextShort), sFwd, (uint32_t)cv2));
                best = l3_best_of(best, l3_match_at(&c, PREV(nextShort), sFwd, (uint32_t)cv2));
                best = l3_best_of(best, l3_match_at(&c, CUR(nextLong), sFwd, (uint32_t)cv2));
                best = l3_best_of(best, l3_match_at(&c, PREV(nextLong), sFwd, (uint32_t)cv2));
                /* s + 2 */
                sFwd++;
                cv2 = ld64(src, sFwd);
                nextLong = lTable[hash8(cv2, LBITS)];
                best = l3_best_of(best, l3_match_repeat(&c, sFwd - repeat, sFwd, (uint32_t)cv2));
                nextShort = sTable[hash4(cv2, SBITS)];
                best = l3_best_of(best, l3_match_at(&c, CUR(nextShort), sFwd, (uint32_t)cv2));
                best = l3_best_of(best, l3_match_at(&c, PREV(nextShort), sFwd, (uint32_t)cv2));
                best = l3_best_of(best, l3_match_at(&c, CUR(nextLong), sFwd, (uint32_t)cv2));
                best = l3_best_of(best, l3_match_at(&c, PREV(nextLong), sFwd, (uint32_t)cv2));
                /* a match found from the end of the best match, allowing 2 mismatching bytes at its start (:466-498) */
                const long sAt = best.s + best.length - 1;
                if (sAt < sLimit) {
                    const long sBack = best.s + 2 - 1, backL = best.length - 2;
                    cv2 = ld64(src, sBack);
                    uint64_t next = lTable[hash8(ld64(src, sAt), LBITS)];
                    long checkAt = CUR(next) - backL;
                    if (checkAt > 0) best = l3_best_of(best, l3_match_at(&c, checkAt, sBack, (uint32_t)cv2));
                    checkAt = PREV(next) - backL;
                    if (checkAt > 0) best = l3_best_of(best, l3_match_at(&c, checkAt, sBack, (uint32_t)cv2));
                    next = sTable[hash4(ld64(src, sAt), SBITS)];
                    checkAt = CUR(next) - backL;
                    if (checkAt > 0) best = l3_best_of(best, l3_match_at(&c, checkAt, sBack, (uint32_t)cv2));
                    checkAt = PREV(next) - backL;
                    if (checkAt > 0) best = l3_best_of(best, l3_match_at(&c, checkAt, sBack, (uint32_t)cv2));
                }
            }
            lTable[hashL] = (uint64_t)s | candidateL << 32;
            sTable[hashS] = (uint64_t)s | candidateS << 32;
            if (best.length > 0) break;
            cv = ld64(src, nextS);
            s = nextS;
        }
        const long startIdx = s + 1;
        s = best.s;
        if (d + (s - nextEmit) > dstLimit) goto done;
        const long base = s, offset = s - best.offset;
        s += best.length;
        if (!best.rep && best.length <= 4) { /* not worth it, :514-526 */
            if (offset > 65535 || (offset > MAX_COPY1_OFFSET && offset <= MAX_COPY2_OFFSET && base - nextEmit > MAX_COPY2_LITS)) {
                s = startIdx + 1;
                if (s >= sLimit) goto emit_remainder;
                cv = ld64(src, s);
                continue;
            }
        }
        const long nl = base - nextEmit;
        if (best.rep) {
            d += mlzo_emit_literal(dst + d, src + nextEmit, nl);
            d += mlzo_emit_repeat(dst + d, best.length);
        } else if (nl > 0) {
            if (offset <= MAX_COPY2_OFFSET) {
                if (nl > MAX_COPY2_LITS || offset < 64 || (offset <= 1024 && best.length > COPY2_LIT_MAX_LEN)) {
                    d += mlzo_emit_literal(dst + d, src + nextEmit, nl);
                    if (best.length > 18 && best.length <= 64 && offset >= 64) d += encode_copy2(dst + d, offset, best.length);
                    else d += mlzo_emit_copy(dst + d, offset, best.length);
                } else if (best.length > 11) { /* the rest is searched again rather than emitted as a repeat */
                    d += mlzo_emit_copy_lits2(dst + d, src + nextEmit, nl, offset, 11);
                    s = best.s + 11;
                } else {
                    d += mlzo_emit_copy_lits2(dst + d, src + nextEmit, nl, offset, best.length);
                }
            } else if (nl > MAX_COPY3_LITS) {
                d += mlzo_emit_literal(dst + d, src + nextEmit, nl);
                d += mlzo_emit_copy(dst + d, offset, best.length);
            } else {
                d += mlzo_emit_copy_lits3(dst + d, src + nextEmit, nl, offset, best.length);
            }
        } else {
            if (best.length > 18 && best.length <= 64 && offset >= 64 && offset <= MAX_COPY2_OFFSET) d += encode_copy2(dst + d, offset, best.length);
            else d += mlzo_emit_copy(dst + d, offset, best.length);
        }
        repeat = offset;
        nextEmit = s;
        if (s >= sLimit) goto emit_remainder;
        if (d > dstLimit) goto done;
        for (long i = startIdx; i < s; i++) { /* fill tables, :606-612 */
            const uint64_t cv0 = ld64(src, i);
            const uint32_t l0 = hash8(cv0, LBITS), s0 = hash4(cv0, SBITS);
            lTable[l0] = (uint64_t)i | lTable[l0] << 32;
            sTable[s0] = (uint64_t)i | sTable[s0] << 32;
        }
        cv = ld64(src, s);
    }
emit_remainder:
    if (nextEmit < n) {
        const long litLen = n - nextEmit;
        if (d + litLen + l3_lit_size(litLen) > dstLimit) goto done;
        d += mlzo_emit_literal(dst + d, src + nextEmit, litLen);
    }
    ret = (size_t)d;
done:
    free(lTable); free(sTable);
    return ret;
#undef CUR
#undef PREV
}

/* encodeUncompressed, encode.go:223-228 */
static long encode_uncompressed(uint8_t* dst, const uint8_t* src, size_t n) {
    if (n == 0) { dst[0] = 0; return 1; }
    dst[0] = 0; dst[1] = 0; memcpy(dst + 2, src, n);
    return (long)n + 2;
}

size_t mlzo_encode_block_l0(uint8_t* dst, const uint8_t* src, size_t n);

/* Encode, encode.go:74-139.  Levels: -1 superfast, 0 uncompressed, 1 fastest, 2 balanced, 3 smallest. */
long mlzo_encode(uint8_t* dst, size_t dcap, const uint8_t* src, size_t n, int level) {
    long maxlen = mlzo_max_encoded_len(n);
    if (maxlen < 0) return -MLZO_ERR_TOO_LARGE;
    if (dcap < (size_t)maxlen) return -MLZO_ERR_DST_TOO_SMALL;
    if (n < MIN_NON_LITERAL_BLOCK_SIZE) return encode_uncompressed(dst, src, n);
    dst[0] = 0;
    size_t d = 1 + put_uvarint(dst + 1, n);
    size_t m;
    switch (level) {
    case 0: return encode_uncompressed(dst, src, n);
    case -1: m = mlzo_encode_block_l0(dst + d, src, n); break;
    case 1: m = mlzo_encode_block_l1(dst + d, src, n); break;
    case 2: m = mlzo_encode_block_l2(dst + d, src, n); break;
    case 3: m = mlzo_encode_block_l3(dst + d, src, n); break;
    default: return -MLZO_ERR_INVALID_LEVEL;
    }
    if (m > 0) return (long)(d + m);
    return encode_uncompressed(dst, src, n);
}

/* ======================================================================================
 * Stream framing
 * ====================================================================================== */

/* CRC32C (Castagnoli, reflected poly 0x82f63b78), table-driven; then the Snappy-framing mask
 * of minlz.go:137-140. */
static uint32_t crc_tab[8][256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
        crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++) crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xff];
}
uint32_t mlzo_crc(const uint8_t* b, size_t n) {
    pthread_once(&crc_once, crc_init);
    uint32_t c = 0xffffffffu;
    while (n >= 8) {
        uint64_t v = ld64(b, 0) ^ c;
        c = crc_tab[7][v & 0xff] ^ crc_tab[6][(v >> 8) & 0xff] ^ crc_tab[5][(v >> 16) & 0xff] ^ crc_tab[4][(v >> 24) & 0xff] ^
            crc_tab[3][(v >> 32) & 0xff] ^ crc_tab[2][(v >> 40) & 0xff] ^ crc_tab[1][(v >> 48) & 0xff] ^ crc_tab[0][v >> 56];
        b += 8; n -= 8;
    }
    while (n--) c = crc_tab[0][(c ^ *b++) & 0xff] ^ (c >> 8);
    c = ~c;
    return (c >> 15 | c << 17) + 0xa282ead8u;
}

static const uint8_t MAGIC_CHUNK[9] = {0xff, 0x06, 0x00, 0x00, 'M', 'i', 'n', 'L', 'z'}; /* minlz.go:95-100 */

size_t mlzo_stream_bound(size_t n, size_t block_size) {
    size_t blocks = (n + block_size - 1) / block_size;
    return 10 + n + blocks * (8 + 5 + 2) + 4 + 10;
}

static unsigned bits_len(size_t v) { unsigned n = 0; while (v) { n++; v >>= 1; } return n; }

/* Writer sync path: makeHeader (writer.go:1553-1556), per block (writer.go:876-959), EOF chunk
 * (writer.go:1063-1074).  No index, no padding. */
long mlzo_stream_encode_ex(uint8_t* dst, size_t dcap, const uint8_t* src, size_t n, int level, size_t block_size, int add_index);
long mlzo_stream_encode(uint8_t* dst, size_t dcap, const uint8_t* src, size_t n, int level, size_t block_size) {
    return mlzo_stream_encode_ex(dst, dcap, src, n, level, block_size, 0);
}
/* add_index: WriterAddIndex(true) — the index chunk follows the EOF chunk (writer.go:1080-1122). */
long mlzo_stream_encode_ex(uint8_t* dst, size_t dcap, const uint8_t* src, size_t n, int level, size_t block_size, int add_index) {
    if (block_size > MLZO_MAX_BLOCK_SIZE || block_size < (4 << 10)) return -MLZO_ERR_TOO_LARGE; /* writer.go:1238-1246 */
    const size_t nblocks = (n + block_size - 1) / block_size;
    if (dcap < mlzo_stream_bound(n, block_size) + (add_index ? mlzo_index_bound(nblocks) : 0)) return -MLZO_ERR_DST_TOO_SMALL;
    int64_t* ic = add_index ? (int64_t*)malloc(sizeof(int64_t) * (nblocks + 1)) : NULL;
    int64_t* iu = add_index ? (int64_t*)malloc(sizeof(int64_t) * (nblocks + 1)) : NULL;
    size_t nb = 0;
    size_t o = 0;
    if (n > 0) { /* header is written lazily with the first block */
        memcpy(dst, MAGIC_CHUNK, 9); dst[9] = (uint8_t)(bits_len(block_size - 1) - 10); o = 10;
    }
    size_t pos = 0;
    /* The default (concurrent) Writer hands the stream header to its output goroutine like a block, and that goroutine
     * calls index.add(w.written = 0, startOffset = 0) for it (writer.go:236-243); the first block's add(10, 0) is then
     * dropped by index.add's distance rule (index.go:87-90).  The index therefore starts at (0, 0). */
    if (add_index && n > 0) { ic[nb] = 0; iu[nb] = 0; nb++; }
    while (pos < n) {
        size_t bl = n - pos < block_size ? n - pos : block_size;
        const uint8_t* u = src + pos;
        uint32_t checksum = mlzo_crc(u, bl);
        uint8_t* ob = dst + o;
        size_t vn = put_uvarint(ob + 8, bl);
        size_t n2 = 0;
        switch (level) { /* (*Writer).encodeBlock, writer.go:565-602 */
        case 0: n2 = 0; break;
        case 1: n2 = mlzo_encode_block_l1(ob + 8 + vn, u, bl); break;
        case 2: n2 = mlzo_encode_block_l2(ob + 8 + vn, u, bl); break;
        case 3: n2 = mlzo_encode_block_l3(ob + 8 + vn, u, bl); break;
        default: free(ic); free(iu); return -MLZO_ERR_INVALID_LEVEL;
        }
        if (add_index) { ic[nb] = (int64_t)o; iu[nb] = (int64_t)pos; nb++; } /* index.add(w.written, w.uncompWritten), writer.go:945 */
        size_t chunk_len; uint8_t type;
        if (n2 > 0) { type = 0x02; chunk_len = 4 + vn + n2; }
        else { type = 0x01; chunk_len = 4 + bl; memcpy(ob + 8, u, bl); }
        ob[0] = type; ob[1] = (uint8_t)chunk_len; ob[2] = (uint8_t)(chunk_len >> 8); ob[3] = (uint8_t)(chunk_len >> 16);
        st32(ob, 4, checksum);
        o += 4 + chunk_len;
        pos += bl;
    }
    /* EOF chunk: 0x20, len24 = varint length, uvarint(total uncompressed) */
    uint8_t* e = dst + o;
    int vn = put_uvarint(e + 4, n);
    e[0] = 0x20; e[1] = (uint8_t)vn; e[2] = 0; e[3] = 0;
    o += 4 + vn;
    if (add_index) {
        long k = mlzo_index_build(dst + o, dcap - o, ic, iu, nb, block_size, (int64_t)n, (int64_t)o);
        free(ic); free(iu);
        if (k < 0) return k;
        o += (size_t)k;
    }
    return (long)o;
}

/* ---- seek index (index.go:26-414, SPEC.md:477-575) ---- */
#define MAX_INDEX_ENTRIES (1 << 16)
#define MIN_INDEX_DIST (1 << 20)
static const uint8_t INDEX_HEADER[6] = {'s', '2', 'i', 'd', 'x', 0};
static const uint8_t INDEX_TRAILER[6] = {0, 'x', 'd', 'i', '2', 's'};

static size_t put_varint(uint8_t* d, int64_t v) { /* binary.PutVarint: zigzag + uvarint */
    uint64_t u = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
    return put_uvarint(d, u);
}
static int get_varint(const uint8_t* b, size_t n, int64_t* v) { /* binary.Varint; <= 0 on error */
    uint64_t x = 0; unsigned sh = 0;
    for (size_t i = 0; i < n && i < 10; i++) {
        uint8_t c = b[i];
        if (c < 0x80) {
            if (i == 9 && c > 1) return -1;
            x |= (uint64_t)c << sh;
            *v = (int64_t)(x >> 1) ^ -(int64_t)(x & 1);
            return (int)i + 1;
        }
        x |= (uint64_t)(c & 0x7f) << sh; sh += 7;
    }
    return 0;
}

typedef struct { int64_t est; long n; int64_t* c; int64_t* u; } mlzo_index;

static void index_reduce_light(mlzo_index* ix) { /* index.go:176-189 */
    ix->est *= 2;
    long j = 0;
    for (long i = 0; i < ix->n; i++) {
        int64_t bc = ix->c[i], bu = ix->u[i];
        ix->c[j] = bc; ix->u[j] = bu; j++;
        while (i < ix->n && ix->u[i] - bu < ix->est) i++;
    }
    ix->n = j;
}
static void index_add(mlzo_index* ix, int64_t c_off, int64_t u_off) { /* index.go:80-104 */
    if (ix->n > 0 && u_off - ix->u[ix->n - 1] < ix->est) return;
    ix->c[ix->n] = c_off; ix->u[ix->n] = u_off; ix->n++;
    if (ix->n > MAX_INDEX_ENTRIES) index_reduce_light(ix);
}
static void index_reduce(mlzo_index* ix) { /* index.go:150-173 */
    if (ix->n < MAX_INDEX_ENTRIES) return;
    long remove_n = (ix->n + 1) / MAX_INDEX_ENTRIES;
    while (ix->est * (remove_n + 1) < MIN_INDEX_DIST && ix->n / (remove_n + 1) > 1000) remove_n++;
    long j = 0;
    for (long i = 0; i < ix->n; i++) { ix->c[j] = ix->c[i]; ix->u[j] = ix->u[i]; j++; i += remove_n; }
    ix->n = j;
    ix->est += ix->est * remove_n;
}
static size_t index_append(uint8_t* b, mlzo_index* ix, int64_t total_u, int64_t total_c) { /* appendTo, index.go:191-269 */
    index_reduce(ix);
    size_t o = 0;
    b[o++] = 0x40; b[o++] = 0; b[o++] = 0; b[o++] = 0;
    memcpy(b + o, INDEX_HEADER, 6); o += 6;
    o += put_varint(b + o, total_u);
    o += put_varint(b + o, total_c);
    o += put_varint(b + o, ix->est);
    o += put_varint(b + o, ix->n);
    uint8_t has_u = 0;
    for (long i = 0; i < ix->n; i++) {
        if (i == 0) { if (ix->u[0] != 0) { has_u = 1; break; } continue; }
        if (ix->u[i] != ix->u[i - 1] + ix->est) { has_u = 1; break; }
    }
    b[o++] = has_u;
    if (has_u)
        for (long i = 0; i < ix->n; i++) {
            int64_t u = ix->u[i];
            if (i > 0) u -= ix->u[i - 1] + ix->est;
            o += put_varint(b + o, u);
        }
    int64_t predict = ix->est / 2;
    for (long i = 0; i < ix->n; i++) {
        int64_t c = ix->c[i];
        if (i > 0) { c -= ix->c[i - 1] + predict; predict += c / 2; }
        o += put_varint(b + o, c);
    }
    st32(b, o, (uint32_t)(o + 4 + 6)); o += 4;
    memcpy(b + o, INDEX_TRAILER, 6); o += 6;
    size_t chunk_len = o - 4;
    b[1] = (uint8_t)chunk_len; b[2] = (uint8_t)(chunk_len >> 8); b[3] = (uint8_t)(chunk_len >> 16);
    return o;
}

size_t mlzo_index_bound(size_t n_blocks) { return 64 + 20 * (n_blocks < MAX_INDEX_ENTRIES + 1 ? n_blocks : MAX_INDEX_ENTRIES + 1); }

/* Index.reset(block_size) + add(c_off[i], u_off[i]) for every block + appendTo.  Returns index bytes. */
long mlzo_index_build(uint8_t* dst, size_t cap, const int64_t* c_off, const int64_t* u_off, size_t n_blocks, size_t block_size,
                      int64_t total_u, int64_t total_c) {
    if (cap < mlzo_index_bound(n_blocks)) return -MLZO_ERR_DST_TOO_SMALL;
    mlzo_index ix;
    int64_t mb = (int64_t)block_size;
    while (mb < MIN_INDEX_DIST) mb *= 2; /* reset, index.go:56-68 */
    ix.est = mb; ix.n = 0;
    ix.c = (int64_t*)malloc(sizeof(int64_t) * (MAX_INDEX_ENTRIES + 2));
    ix.u = (int64_t*)malloc(sizeof(int64_t) * (MAX_INDEX_ENTRIES + 2));
    for (size_t i = 0; i < n_blocks; i++) index_add(&ix, c_off[i], u_off[i]);
    size_t n = index_append(dst, &ix, total_u, total_c);
    free(ix.c); free(ix.u);
    return (long)n;
}

/* Index.Load, index.go:273-396.  Returns MLZO_* (MLZO_ERR_CORRUPT also for short input); *consumed = bytes of b used. */
int mlzo_index_load(const uint8_t* b, size_t n, int64_t* total_u, int64_t* total_c, int64_t* est, int64_t* c_off, int64_t* u_off,
                    size_t cap, size_t* n_entries, size_t* consumed) {
    size_t p = 0;
    if (n <= 4 + 6 + 6) return MLZO_ERR_CORRUPT;
    if (b[0] != 0x40 && b[0] != 0x99) return MLZO_ERR_CORRUPT;
    size_t chunk_len = (size_t)b[1] | (size_t)b[2] << 8 | (size_t)b[3] << 16;
    p = 4;
    if (n - p < chunk_len) return MLZO_ERR_CORRUPT;
    if (memcmp(b + p, INDEX_HEADER, 6) != 0) return MLZO_ERR_UNSUPPORTED;
    p += 6;
    int64_t v; int k;
    if ((k = get_varint(b + p, n - p, &v)) <= 0 || v < 0) return MLZO_ERR_CORRUPT;
    *total_u = v; p += k;
    if ((k = get_varint(b + p, n - p, &v)) <= 0) return MLZO_ERR_CORRUPT;
    *total_c = v; p += k;
    if ((k = get_varint(b + p, n - p, &v)) <= 0 || v < 0) return MLZO_ERR_CORRUPT;
    *est = v; p += k;
    if ((k = get_varint(b + p, n - p, &v)) <= 0 || v < 0 || v > MAX_INDEX_ENTRIES) return MLZO_ERR_CORRUPT;
    size_t entries = (size_t)v; p += k;
    if (entries > cap) return MLZO_ERR_DST_TOO_SMALL;
    if (n - p < 1) return MLZO_ERR_CORRUPT;
    uint8_t has_u = b[p++];
    if ((has_u & 1) != has_u) return MLZO_ERR_CORRUPT;
    for (size_t i = 0; i < entries; i++) {
        int64_t u = 0;
        if (has_u) { if ((k = get_varint(b + p, n - p, &u)) <= 0) return MLZO_ERR_CORRUPT; p += k; }
        if (i > 0) { int64_t prev = u_off[i - 1]; u += prev + *est; if (u <= prev) return MLZO_ERR_CORRUPT; }
        if (u < 0) return MLZO_ERR_CORRUPT;
        u_off[i] = u;
    }
    int64_t predict = *est / 2;
    for (size_t i = 0; i < entries; i++) {
        int64_t c;
        if ((k = get_varint(b + p, n - p, &c)) <= 0) return MLZO_ERR_CORRUPT;
        p += k;
        if (i > 0) {
            int64_t pn = predict + c / 2, prev = c_off[i - 1];
            c += prev + predict;
            if (c <= prev) return MLZO_ERR_CORRUPT;
            predict = pn;
        }
        if (c < 0) return MLZO_ERR_CORRUPT;
        c_off[i] = c;
    }
    if (n - p < 4 + 6) return MLZO_ERR_CORRUPT;
    p += 4;
    if (memcmp(b + p, INDEX_TRAILER, 6) != 0) return MLZO_ERR_CORRUPT;
    p += 6;
    *n_entries = entries; *consumed = p;
    return MLZO_OK;
}

/* Reader.Read, reader.go:248-543, for MinLZ streams (no Snappy/S2 fallback). */
int mlzo_stream_decode(const uint8_t* src, size_t slen, uint8_t* dst, size_t dcap, size_t* dlen) {
    size_t p = 0, o = 0, max_block = MLZO_MAX_BLOCK_SIZE, stream_out = 0;
    int read_header = 0, want_eof = 0;
    *dlen = 0;
    for (;;) {
        if (p == slen) { *dlen = o; return want_eof ? MLZO_ERR_CORRUPT : MLZO_OK; } /* io.ErrUnexpectedEOF when an EOF chunk is owed */
        if (slen - p < 4) return MLZO_ERR_CORRUPT;
        uint8_t type = src[p];
        size_t clen = (size_t)src[p + 1] | (size_t)src[p + 2] << 8 | (size_t)src[p + 3] << 16;
        p += 4;
        if (!read_header) {
            if (type == 0xff) read_header = 1;
            else if (type <= 0x3f && type != 0x20) return MLZO_ERR_CORRUPT;
        }
        switch (type) {
        case 0x02: case 0x03: { /* reader.go:286-353 */
            if (clen < 4) return MLZO_ERR_CORRUPT;
            if (clen > (size_t)mlzo_max_encoded_len(max_block) + 4) return MLZO_ERR_CORRUPT; /* ensureBufferSize, reader.go:163-167 */
            if (slen - p < clen) return MLZO_ERR_CORRUPT;
            uint32_t checksum = ld32(src, p);
            const uint8_t* buf = src + p + 4; size_t bl = clen - 4;
            size_t nn, hl; int e = decoded_len_raw(buf, bl, &nn, &hl);
            if (e) return e;
            if (nn > max_block) return MLZO_ERR_TOO_LARGE;
            buf += hl; bl -= hl;
            if (nn == 0 || nn < bl) return MLZO_ERR_CORRUPT;
            if (nn > dcap - o) return MLZO_ERR_DST_TOO_SMALL;
            if (mlzo_decode_body(dst + o, nn, buf, bl) != 0) return MLZO_ERR_CORRUPT;
            uint32_t got = type == 0x03 ? mlzo_crc(buf, bl) : mlzo_crc(dst + o, nn);
            if (got != checksum) return MLZO_ERR_CRC;
            o += nn; stream_out += nn; p += clen;
            continue;
        }
        case 0x00: return MLZO_ERR_UNSUPPORTED; /* legacy S2/Snappy chunk: fallback out of scope */
        case 0x01: { /* reader.go:411-464 */
            if (clen < 4) return MLZO_ERR_CORRUPT;
            if (clen > (size_t)mlzo_max_encoded_len(max_block) + 4) return MLZO_ERR_CORRUPT; /* ensureBufferSize, reader.go:163-167 */
            if (slen - p < 4) return MLZO_ERR_CORRUPT;
            uint32_t checksum = ld32(src, p);
            size_t nn = clen - 4;
            if (nn > max_block) return MLZO_ERR_TOO_LARGE;
            if (slen - p - 4 < nn) return MLZO_ERR_CORRUPT;
            if (nn > dcap - o) return MLZO_ERR_DST_TOO_SMALL;
            memcpy(dst + o, src + p + 4, nn);
            if (mlzo_crc(dst + o, nn) != checksum) return MLZO_ERR_CRC;
            o += nn; stream_out += nn; p += clen;
            continue;
        }
        case 0x20: { /* EOF, reader.go:465-499 */
            if (clen > 10) return MLZO_ERR_CORRUPT;
            if (clen != 0) {
                if (slen - p < clen) return MLZO_ERR_CORRUPT;
                uint64_t want; int vn = uvarint(src + p, clen, &want);
                if (vn != (int)clen) return MLZO_ERR_CORRUPT;
                if (want != stream_out) return MLZO_ERR_CORRUPT;
                p += clen;
            }
            want_eof = 0; read_header = 0;
            continue;
        }
        case 0xff: { /* stream identifier, reader.go:500-527 + minLzHeader :994-1027 */
            if (clen != 6) return MLZO_ERR_CORRUPT;
            if (slen - p < 6) return MLZO_ERR_CORRUPT;
            if (memcmp(src + p, "MinLz", 5) != 0) return MLZO_ERR_UNSUPPORTED;
            uint8_t b = src[p + 5];
            if (b & (3 << 6)) return MLZO_ERR_CORRUPT;
            unsigned lg = (b & 15) + 10;
            if (lg > 23) return MLZO_ERR_CORRUPT;
            max_block = (size_t)1 << lg;
            stream_out = 0; want_eof = 1;
            p += 6;
            continue;
        }
        default: break;
        }
        if (type <= 0x3f) return MLZO_ERR_UNSUPPORTED; /* reserved unskippable, reader.go:530-536 */
        if (slen - p < clen) return MLZO_ERR_CORRUPT;   /* skippable chunk */
        p += clen;
    }
}

/* ======================================================================================
 * Multi-threaded block bench (cpu_baseline leg).  One block per thread at a time, as
 * BenchmarkEncodeBlockParallel (benchmarks_test.go:101-107) / mz -bench (cmd/mz/compress.go:647-803).
 * ====================================================================================== */
typedef struct {
    const uint8_t* src; size_t n, block; int level, decode;
    uint8_t** enc; size_t* enc_len; /* per block (decode input) */
    size_t nblocks, items;
    volatile size_t* next;          /* shared work counter: (rep, block) items are taken one at a time */
    pthread_barrier_t* start;       /* released by the timing thread once every worker exists and owns its buffer */
    size_t out_bytes;               /* per thread: compressed bytes produced (encode) */
    double t_end;                   /* when this worker ran out of work */
} bench_arg;

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* Thread creation, buffer allocation and the first touch of the buffer happen BEFORE the clock starts; inside the
 * timed region a worker only takes the next (rep, block) item and runs the codec on it. */
static void* bench_worker(void* p) {
    bench_arg* a = (bench_arg*)p;
    uint8_t* tmp = (uint8_t*)malloc(a->block + 16);
    memset(tmp, 0, a->block + 16);
    pthread_barrier_wait(a->start);
    for (;;) {
        const size_t it = __atomic_fetch_add(a->next, 1, __ATOMIC_RELAXED);
        if (it >= a->items) break;
        size_t b = it % a->nblocks;
        size_t off = b * a->block, bl = a->n - off < a->block ? a->n - off : a->block;
        if (!a->decode) {
            long m = mlzo_encode(tmp, bl + 16, a->src + off, bl, a->level);
            if (m > 0) a->out_bytes += (size_t)m;
        } else {
            size_t dl;
            mlzo_decode(a->enc[b], a->enc_len[b], tmp, a->block + 16, &dl);
        }
    }
    a->t_end = now_s();
    free(tmp);
    return NULL;
}

static double bench_run(const uint8_t* src, size_t n, size_t block, int level, int threads, int reps, int decode,
                        uint8_t** enc, size_t* enc_len, size_t nblocks, size_t* out_bytes) {
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    bench_arg* args = (bench_arg*)malloc(sizeof(bench_arg) * threads);
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)threads + 1);
    volatile size_t next = 0;
    for (int t = 0; t < threads; t++) {
        args[t] = (bench_arg){src, n, block, level, decode, enc, enc_len, nblocks, (size_t)reps * nblocks, &next, &start, 0, 0.0};
        pthread_create(&th[t], NULL, bench_worker, &args[t]);
    }
    pthread_barrier_wait(&start);   /* every worker is parked at the barrier with its buffer: start the clock */
    const double t0 = now_s();
    size_t tot = 0;
    double t1 = t0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); tot += args[t].out_bytes; if (args[t].t_end > t1) t1 = args[t].t_end; }
    pthread_barrier_destroy(&start);
    if (out_bytes) *out_bytes = tot;
    free(th); free(args);
    return t1 - t0;
}

/* reps passes over the blocks of src; *total_out = compressed bytes of ONE pass. */
double mlzo_bench_encode(const uint8_t* src, size_t n, size_t block, int level, int threads, int reps, size_t* total_out) {
    size_t nblocks = (n + block - 1) / block, tot = 0;
    double dt = bench_run(src, n, block, level, threads, reps, 0, NULL, NULL, nblocks, &tot);
    if (total_out) *total_out = reps ? tot / (size_t)reps : 0;
    return dt;
}

double mlzo_bench_decode(const uint8_t* src, size_t n, size_t block, int level, int threads, int reps) {
    size_t nblocks = (n + block - 1) / block;
    uint8_t** enc = (uint8_t**)malloc(sizeof(uint8_t*) * nblocks);
    size_t* enc_len = (size_t*)calloc(nblocks, sizeof(size_t));
    for (size_t b = 0; b < nblocks; b++) { /* untimed: produce the blocks */
        size_t off = b * block, bl = n - off < block ? n - off : block;
        enc[b] = (uint8_t*)malloc(bl + 16);
        long m = mlzo_encode(enc[b], bl + 16, src + off, bl, level);
        enc_len[b] = m > 0 ? (size_t)m : 0;
    }
    double dt = bench_run(src, n, block, level, threads, reps, 1, enc, enc_len, nblocks, NULL);
    for (size_t b = 0; b < nblocks; b++) free(enc[b]);
    free(enc); free(enc_len);
    return dt;
}
