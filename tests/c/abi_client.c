/* abi_client.c — a plain C99 client of include/minlz_hip.h, compiled by gcc the way cgo compiles the preamble of go/minlz_hip.go
 * (tests/test_abi.py builds it; the -m gpu test runs it).  It touches the header only through what a C compiler sees: no C++, no Python.
 * Exit code: 0 = every check passed, 77 = no HIP device (mlz_init failed; the no-GPU run of the test expects exactly this), else the line
 * number of the failing check. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "minlz_hip.h"

#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            fprintf(stderr, "abi_client: line %d: %s\n", __LINE__, #cond); \
            return __LINE__;                                            \
        }                                                               \
    } while (0)

static void fill_text(uint8_t* p, size_t n, unsigned seed) {
    static const char* words[] = {"the ", "quick ", "brown ", "fox ", "jumps ", "over ", "a ", "lazy ", "dog ", "and ", "minlz ", "block ", "stream ", "\n"};
    size_t o = 0;
    unsigned s = seed * 2654435761u + 1u;
    while (o < n) {
        const char* w;
        size_t l;
        s = s * 1664525u + 1013904223u;
        w = words[(s >> 24) % (sizeof(words) / sizeof(words[0]))];
        l = strlen(w);
        if (l > n - o) l = n - o;
        memcpy(p + o, w, l);
        o += l;
    }
}

int main(void) {
    mlz_ctx* ctx = NULL;
    mlz_ctx* two = NULL;
    const size_t n = (size_t)3 << 20;
    uint8_t *src, *enc, *dec, *st, *st2;
    int64_t r, r2, cap;
    int devs[2] = {0, 0};
    char name[128];

    /* host-only entry points first: they need no device */
    CHECK(mlz_version() >= 2);
    CHECK(mlz_max_encoded_len(0) == 1 && mlz_max_encoded_len(100) == 102 && mlz_max_encoded_len((uint64_t)MLZ_MAX_BLOCK_SIZE + 1) == -1);
    {
        const uint8_t stored[6] = {0x00, 0x00, 'a', 'b', 'c', 'd'}; /* `00 00 raw`, encode.go:223-228 */
        const uint8_t snappy[3] = {0x03, 0x08, 'x'};
        CHECK(mlz_decoded_len(stored, sizeof stored) == 4);
        CHECK(mlz_decoded_len(snappy, sizeof snappy) == 3); /* DecodedLen reports the Snappy varint, decode.go:132-136 */
        CHECK(mlz_stream_bound(0, 1u << 20, 0) > 0 && mlz_stream_bound(1, 1000, 0) == -MLZ_ERR_ARG);
    }
    if (mlz_init(-1, &ctx) != 0 || !ctx) return 77;
    CHECK(mlz_device_name(ctx, name, sizeof name) == 0 && name[0]);
    CHECK(mlz_device_count(ctx) == 1 && mlz_device_ctx(ctx, 0) == ctx);
    CHECK(mlz_crc(ctx, (const uint8_t*)"abcd", 4) == 0xb6e61068ll); /* minlz_test.go:1120-1134 */

    src = (uint8_t*)malloc(n);
    enc = (uint8_t*)malloc(n + 64);
    dec = (uint8_t*)malloc(n);
    CHECK(src && enc && dec);
    fill_text(src, n, 7);

    /* minlz.Encode / minlz.Decode */
    r = mlz_encode(ctx, MLZ_LEVEL_FASTEST, src, n, enc, n + 64);
    CHECK(r > 0 && (size_t)r < n / 2);
    CHECK(mlz_decoded_len(enc, (size_t)r) == (int64_t)n);
    CHECK(mlz_decode(ctx, enc, (size_t)r, dec, n) == (int64_t)n && memcmp(dec, src, n) == 0);
    CHECK(mlz_encode(ctx, 3, src, n, enc, n + 64) == -MLZ_ERR_INVALID_LEVEL); /* LevelSmallest stays on the host */
    enc[(size_t)r / 2] ^= 0x55;
    enc[(size_t)r / 2 + 1] ^= 0xff;
    r2 = mlz_decode(ctx, enc, (size_t)r, dec, n);
    CHECK(r2 == -MLZ_ERR_CORRUPT || (r2 == (int64_t)n));

    /* WriterCustomEncoder / minLZDecode contracts */
    r = mlz_encode_block(ctx, MLZ_LEVEL_BALANCED, src, 1u << 20, enc, n);
    CHECK(r > 0);
    memset(dec, 0, 1u << 20);
    CHECK(mlz_decode_block(ctx, enc, (size_t)r, dec, 1u << 20) == 0 && memcmp(dec, src, 1u << 20) == 0);
    CHECK(mlz_decode_block(ctx, enc, (size_t)r - 1, dec, 1u << 20) == 1);

    /* batches */
    {
        const uint8_t* bs[3];
        uint8_t* bd[3];
        size_t bl[3], bc[3];
        int64_t ol[3];
        size_t i;
        for (i = 0; i < 3; i++) {
            bs[i] = src + i * (1u << 20); bl[i] = (1u << 20) - 17 * i;
            bd[i] = enc + i * ((1u << 20) + 16); bc[i] = (1u << 20) + 16;
        }
        CHECK(mlz_encode_batch(ctx, MLZ_LEVEL_FASTEST, 3, bs, bl, bd, bc, ol) == 0);
        for (i = 0; i < 3; i++) CHECK(ol[i] > 0 && mlz_decoded_len(bd[i], (size_t)ol[i]) == (int64_t)bl[i]);
        {
            const uint8_t* ds[3];
            uint8_t* dd[3];
            size_t dl[3], dc[3];
            int64_t dl_out[3];
            for (i = 0; i < 3; i++) { ds[i] = bd[i]; dl[i] = (size_t)ol[i]; dd[i] = dec + i * (1u << 20); dc[i] = bl[i]; }
            CHECK(mlz_decode_batch(ctx, 3, ds, dl, dd, dc, dl_out) == 0);
            for (i = 0; i < 3; i++) CHECK(dl_out[i] == (int64_t)bl[i] && memcmp(dd[i], bs[i], bl[i]) == 0);
        }
    }

    /* streams: one device, then two per-device contexts behind one handle (mlz_init_devices) — byte-identical */
    cap = mlz_stream_bound(n, 1u << 20, MLZ_STREAM_ADD_INDEX);
    CHECK(cap > 0);
    st = (uint8_t*)malloc((size_t)cap);
    st2 = (uint8_t*)malloc((size_t)cap);
    CHECK(st && st2);
    r = mlz_stream_encode(ctx, MLZ_LEVEL_FASTEST, 1u << 20, MLZ_STREAM_ADD_INDEX, src, n, st, (size_t)cap);
    CHECK(r > 0 && memcmp(st, "\xff\x06\x00\x00MinLz", 9) == 0);
    CHECK(mlz_stream_decoded_len(st, (size_t)r) == (int64_t)n);
    memset(dec, 0, n);
    CHECK(mlz_stream_decode(ctx, 0, st, (size_t)r, dec, n) == (int64_t)n && memcmp(dec, src, n) == 0);
    CHECK(mlz_init_devices(devs, 2, &two) == 0 && two && mlz_device_count(two) == 2);
    CHECK(mlz_device_ctx(two, 0) && mlz_device_ctx(two, 1) && !mlz_device_ctx(two, 2));
    r2 = mlz_stream_encode(two, MLZ_LEVEL_FASTEST, 1u << 20, MLZ_STREAM_ADD_INDEX, src, n, st2, (size_t)cap);
    CHECK(r2 == r && memcmp(st, st2, (size_t)r) == 0);
    memset(dec, 0, n);
    CHECK(mlz_stream_decode(two, 0, st2, (size_t)r2, dec, n) == (int64_t)n && memcmp(dec, src, n) == 0);
    st2[(size_t)r2 / 2] ^= 0x04;
    CHECK(mlz_stream_decode(two, 0, st2, (size_t)r2, dec, n) < 0);
    CHECK(mlz_stream_decode(two, MLZ_STREAM_IGNORE_CRC, st, (size_t)r, dec, n) == (int64_t)n);
    CHECK(mlz_get_counter(two, 1) >= 0 && mlz_set_option(two, MLZ_OPT_DEVICE_GROUP, 256) == 0);

    mlz_destroy(two);
    mlz_destroy(ctx);
    free(src); free(enc); free(dec); free(st); free(st2);
    puts("abi_client ok");
    return 0;
}
