/*
 * minlz_oracle.h — CPU restatement of the MinLZ block codec (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X build: a plain-C restatement of the reference's
 * pure-Go algorithms (file:line citations are into /root/reference, the read-only upstream
 * minio/minlz tree).  It is NOT part of the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (minlz_amd/) never links it.
 *
 * Pinning status:
 *   decode  : pinned — bit-exact on testdata/Mark.Twain-Tom.Sawyer.txt.mzb -> .txt
 *             (minlz_test.go:626-660) and on the negative set fuzz/block-corpus-dec.zip.
 *   emitters: pinned — TestEmitLiteral / TestEmitCopy byte tables (minlz_test.go:871-1026).
 *   crc     : pinned — framing KAT crc("abcd") -> 68 10 e6 b6 (minlz_test.go:1120-1134).
 *   L3 enc  : pinned — mlzo_encode(.txt, 3) is byte-identical to the reference's committed
 *             testdata/Mark.Twain-Tom.Sawyer.txt.mzb (8875 bytes).
 *   L1/L2   : "parity unpinned" byte-wise (no Go toolchain here, no reference-encoded L1/L2
 *             artefacts in the tree); pinned only through round-trip against the pinned decoder
 *             and the reference's ratio assertion (minlz_test.go:780-797).
 */
#ifndef MINLZ_ORACLE_H
#define MINLZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLZO_MAX_BLOCK_SIZE (8u << 20) /* minlz.go:84 */

/* error codes of the block API (decode.go:29-40) */
enum {
    MLZO_OK = 0,
    MLZO_ERR_CORRUPT = 1,      /* ErrCorrupt */
    MLZO_ERR_TOO_LARGE = 2,    /* ErrTooLarge */
    MLZO_ERR_UNSUPPORTED = 3,  /* ErrUnsupported (Snappy/S2 fallback is out of scope) */
    MLZO_ERR_INVALID_LEVEL = 4,/* ErrInvalidLevel */
    MLZO_ERR_CRC = 5,          /* ErrCRC */
    MLZO_ERR_DST_TOO_SMALL = 6 /* caller buffer too small (Go would allocate) */
};

/* ---- decode ---- */
/* minLZDecodeGo (decode.go:178-622): tokens -> dst[0..dlen). 0 ok / 1 corrupt. */
int mlzo_decode_body(uint8_t* dst, size_t dlen, const uint8_t* src, size_t slen);
/* isMinLZ (decode.go:120-156). Returns MLZO_* ; on OK fills outputs. */
int mlzo_is_minlz(const uint8_t* src, size_t slen, int* is_mlz, int* literals,
                  size_t* body_off, size_t* size);
/* Decode (decode.go:50-78). dst must hold DecodedLen bytes. */
int mlzo_decode(const uint8_t* src, size_t slen, uint8_t* dst, size_t dcap, size_t* dlen);
/* DecodedLen (decode.go:107-110). */
int mlzo_decoded_len(const uint8_t* src, size_t slen, size_t* dlen);

/* ---- emitters (asm_none.go:84-323, encode.go:247-282) ---- */
size_t mlzo_emit_literal(uint8_t* dst, const uint8_t* lit, size_t n);
size_t mlzo_emit_repeat(uint8_t* dst, size_t length);
size_t mlzo_emit_copy(uint8_t* dst, size_t offset, size_t length);
size_t mlzo_emit_copy_lits2(uint8_t* dst, const uint8_t* lits, size_t nlits, size_t offset, size_t length);
size_t mlzo_emit_copy_lits3(uint8_t* dst, const uint8_t* lits, size_t nlits, size_t offset, size_t length);

/* ---- encoders ---- */
/* MaxEncodedLen (encode.go:234-244): -1 when too large. */
long mlzo_max_encoded_len(size_t n);
/* encodeBlock / encodeBlockBetter (asm_none.go:51-76): token stream only, 0 = incompressible. */
size_t mlzo_encode_block_l0(uint8_t* dst, const uint8_t* src, size_t n);  /* encodeBlockFast (LevelSuperFast), encode_l0.go */
size_t mlzo_encode_block_l1(uint8_t* dst, const uint8_t* src, size_t n);
size_t mlzo_encode_block_l2(uint8_t* dst, const uint8_t* src, size_t n);
/* encodeBlockBest (encode_l3.go:38-625), no dictionary. */
size_t mlzo_encode_block_l3(uint8_t* dst, const uint8_t* src, size_t n);
/* Encode (encode.go:74-139): full block with header. Returns bytes written, <0 = -MLZO_ERR_*. */
long mlzo_encode(uint8_t* dst, size_t dcap, const uint8_t* src, size_t n, int level);

/* ---- stream (writer.go:854-965 sync path, reader.go:248-543) ---- */
uint32_t mlzo_crc(const uint8_t* b, size_t n); /* minlz.go:137-140 masked CRC32C */
/* Bound on stream size for n input bytes at the given block size. */
size_t mlzo_stream_bound(size_t n, size_t block_size);
/* Writer(level, blockSize).EncodeBuffer(src); Close() — no index, no padding. */
long mlzo_stream_encode(uint8_t* dst, size_t dcap, const uint8_t* src, size_t n, int level, size_t block_size);
/* ... with WriterAddIndex(add_index): the seek index chunk (0x40) follows the EOF chunk. */
long mlzo_stream_encode_ex(uint8_t* dst, size_t dcap, const uint8_t* src, size_t n, int level, size_t block_size, int add_index);
/* seek index (index.go:26-414; SPEC.md:477-575).  build = reset(block_size) + add() per block + appendTo. */
size_t mlzo_index_bound(size_t n_blocks);
long mlzo_index_build(uint8_t* dst, size_t cap, const int64_t* c_off, const int64_t* u_off, size_t n_blocks, size_t block_size,
                      int64_t total_u, int64_t total_c);
int mlzo_index_load(const uint8_t* b, size_t n, int64_t* total_u, int64_t* total_c, int64_t* est, int64_t* c_off, int64_t* u_off,
                    size_t cap, size_t* n_entries, size_t* consumed);
/* Reader.Read until EOF. Returns MLZO_*; *dlen = bytes produced. */
int mlzo_stream_decode(const uint8_t* src, size_t slen, uint8_t* dst, size_t dcap, size_t* dlen);

/* multi-threaded helpers for the cpu_baseline leg of bench.py (one block per thread,
 * mirrors BenchmarkEncodeBlockParallel, benchmarks_test.go:101-107). Return seconds. */
double mlzo_bench_encode(const uint8_t* src, size_t n, size_t block_size, int level, int threads,
                         int reps, size_t* total_out);
double mlzo_bench_decode(const uint8_t* src, size_t n, size_t block_size, int level, int threads,
                         int reps);

#ifdef __cplusplus
}
#endif
#endif
