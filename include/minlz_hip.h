/*
 * minlz_hip.h — C ABI of the MI355X (gfx950) MinLZ block codec.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch
 * types, re-entrant, no exceptions.  Each entry point names the reference interface it
 * replaces (file:line into the upstream minio/minlz tree); INTEGRATION.md shows the cgo stub a
 * maintainer adds on the Go side.
 *
 * Two families:
 *   host-pointer calls   — buffers live in host memory for the duration of the call; the
 *                          library stages them through pinned buffers it owns (PCIe-bound).
 *   device-resident calls — src/dst already in HBM; work is queued on the caller's HIP stream
 *                          and nothing is synchronised (this is what bench.py times).
 *
 * Return codes: >= 0 success (or a byte count), < 0 = -MLZ_ERR_*.
 */
#ifndef MINLZ_HIP_H
#define MINLZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLZ_MAX_BLOCK_SIZE (8u << 20) /* minlz.go:84 MaxBlockSize */

/* compression levels, encode.go:25-43.  LevelSmallest (3) is not offered on the device: encode calls
 * return -MLZ_ERR_INVALID_LEVEL for it, which a WriterCustomEncoder treats as "decline". */
#define MLZ_LEVEL_SUPERFAST (-1)
#define MLZ_LEVEL_UNCOMPRESSED 0
#define MLZ_LEVEL_FASTEST 1
#define MLZ_LEVEL_BALANCED 2

/* error codes; 1..5 mirror the reference's sentinel errors (decode.go:29-40) */
#define MLZ_OK 0
#define MLZ_ERR_CORRUPT 1       /* ErrCorrupt */
#define MLZ_ERR_TOO_LARGE 2     /* ErrTooLarge */
#define MLZ_ERR_UNSUPPORTED 3   /* ErrUnsupported: Snappy/S2 fallback blocks (src[0] != 0) */
#define MLZ_ERR_INVALID_LEVEL 4 /* ErrInvalidLevel */
#define MLZ_ERR_CRC 5           /* ErrCRC */
#define MLZ_ERR_DST_TOO_SMALL 6 /* caller's dst cannot hold the result */
#define MLZ_ERR_HIP 7           /* HIP runtime failure: caller should fall back to its CPU path */
#define MLZ_ERR_ARG 8           /* bad argument */

typedef struct mlz_ctx mlz_ctx; /* one per (process, device) from mlz_init, or one over several devices from mlz_init_devices; thread-safe */

/* One block of a batch.  Offsets are relative to the base pointers passed with the batch. */
typedef struct {
    uint64_t src_off; /* start of this block's input  */
    uint64_t src_len; /* its length (encode: <= 8 MiB uncompressed; decode: compressed bytes) */
    uint64_t dst_off; /* start of this block's output */
    uint64_t dst_cap; /* room there (encode: >= mlz_max_encoded_len(src_len)) */
} mlz_block_desc;

/* ---- lifetime ---- */
/* Creates the context on HIP device `device` (-1 = current).  Replaces nothing in the
 * reference (its kernels need no state beyond sync.Pool tables, encode_amd64.go:119-189). */
int mlz_init(int device, mlz_ctx** out);
void mlz_destroy(mlz_ctx* ctx);
/* Several devices behind ONE context, in one process (the reference's Writer and Reader fan the blocks of a stream out to goroutines inside
 * one process: writer.go:501-560, in-order emit writer.go:219-272, reader.go:830-859 — a Go host is one process, so the fan-out over the GPUs
 * of a node sits behind this ABI, not above it).  devices[0 .. n_devices): HIP device ordinals, one per-device context each (an ordinal may be
 * repeated: two contexts on one GPU overlap one's copies with the other's kernels, and it is how the path is tested on a one-GPU box);
 * devices == NULL: every visible device (n_devices > 0: the first n_devices of them).
 * With such a context
 *   mlz_encode_batch / mlz_decode_batch   deal contiguous block ranges of about equal bytes to the devices, one host thread and one PCIe link per
 *                                          device; every result lands at the caller's dst[i] (page-locked destinations are written by the kernels of
 *                                          whichever device ran the block), so there is nothing to gather and no collective;
 *   mlz_stream_encode                      deals the stream's 64 MiB groups of blocks to the devices in turn (group g to device g mod n); a group's chunks go
 *                                          to their final place in dst as soon as the sizes of the groups before it are known (the in-order emit: they
 *                                          belong to the same pipeline step of the other devices), under the kernels of the device's next group;
 *                                          the stream is byte-identical to the one-device call's;
 *   mlz_stream_decode                      deals contiguous chunk ranges of about equal output (every chunk's output offset is known from the chunk walk);
 *   mlz_encode / mlz_decode / *_block / mlz_crc   go to the devices in turn, each with its own combining queue;
 *   the *_batch_device calls               run on the device that holds d_src (-MLZ_ERR_ARG if none of the context's devices does);
 *   mlz_stream_encode_gather_device        (below) is the device-resident form: sources in each device's HBM, the framed stream gathered GPU-to-GPU;
 *   mlz_set_option applies to every device, mlz_get_counter sums (which = 6: the maximum), mlz_get_timers reports the slowest device per family.
 * A context from mlz_init behaves as before everywhere (mlz_device_count = 1).  mlz_device_ctx(ctx, i) is the i-th per-device context — owned by
 * ctx, valid until mlz_destroy(ctx) — for callers that place device-resident work themselves. */
int mlz_init_devices(const int* devices, int n_devices, mlz_ctx** out);
int mlz_device_count(mlz_ctx* ctx);
mlz_ctx* mlz_device_ctx(mlz_ctx* ctx, int i);
const char* mlz_last_error(mlz_ctx* ctx); /* text of the last HIP failure on this context */
/* Library/ABI version and the name of the device the context runs on. */
int mlz_version(void);
int mlz_device_name(mlz_ctx* ctx, char* buf, size_t cap);

/* ---- sizes ---- */
/* MaxEncodedLen (encode.go:234-244): n+2, 1 for n == 0, -1 if n > 8 MiB. */
int64_t mlz_max_encoded_len(uint64_t n);
/* DecodedLen / isMinLZ (decode.go:107-156) on a host buffer: >= 0 decoded size, < 0 error. */
int64_t mlz_decoded_len(const uint8_t* src, size_t n);

/* ---- host-pointer block calls ---- */
/* minlz.Encode(dst, src, level) (encode.go:74-139): full block `00 uvarint(n) tokens`, or the
 * stored form `00 00 raw` when incompressible / n < 16.  Returns bytes written. */
int64_t mlz_encode(mlz_ctx* ctx, int level, const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap);
/* minlz.Decode(dst, src) (decode.go:50-78).  Returns decoded bytes. */
int64_t mlz_decode(mlz_ctx* ctx, const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap);
/* WriterCustomEncoder contract (writer.go:1293-1304) == encodeBlock(dst, src) (asm_none.go:51):
 * token stream only, no header.  > 0 bytes, 0 = incompressible, < 0 = error (decline). */
int64_t mlz_encode_block(mlz_ctx* ctx, int level, const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap);
/* minLZDecode(dst[:n], src) (decode.go:178, decode_amd64.go:28-36): 0 ok, 1 corrupt, < 0 error. */
int mlz_decode_block(mlz_ctx* ctx, const uint8_t* src, size_t c, uint8_t* dst, size_t n);
/* Batched host-pointer forms used by the wrapper's Writer/Reader (writer.go:501-560,
 * reader.go:830-859 fan blocks to goroutines; here one launch covers the batch).
 * out_len[i] receives the bytes produced for block i or -MLZ_ERR_*.
 * Pinned destinations: when every dst[i] .. dst[i] + dst_cap[i] lies in page-locked host memory (hipHostMalloc /
 * hipHostRegister), the kernels store their results straight into it and there is no copy-out.  A block that FAILS
 * (out_len[i] < 0) may then have left partial output in its dst[i] — with pageable destinations a failed block's
 * buffer is not touched.  Nothing outside dst[i][0 .. dst_cap[i]) is ever written either way. */
int mlz_encode_batch(mlz_ctx* ctx, int level, int n_blocks, const uint8_t* const* src, const size_t* src_len,
                     uint8_t* const* dst, const size_t* dst_cap, int64_t* out_len);
int mlz_decode_batch(mlz_ctx* ctx, int n_blocks, const uint8_t* const* src, const size_t* src_len,
                     uint8_t* const* dst, const size_t* dst_cap, int64_t* out_len);

/* ---- device-resident batch calls (asynchronous on `stream`, a hipStream_t; NULL = default) ----
 * d_src / d_dst / d_out_len are device pointers.  d_out_len[i] (int64) receives what
 * mlz_encode / mlz_decode would have returned for block i.  `desc` is a host array (copied).
 * A context owns ONE workspace: calls issued on different streams are ordered on the device (each waits for the
 * previous call's last kernel through an event, recorded when a call arrives on a stream other than the last one's: a stream that was
 * used for a call must stay alive until the context's next call or mlz_destroy), so they are safe but do not overlap; use one context per stream
 * for concurrency.
 * Workspace: a batch runs in internal groups of about 512 MiB of uncompressed data (MLZ_OPT_DEVICE_GROUP; at least one block per group;
 * throughput is flat from 64 blocks of 8 MiB on), one after the other on `stream`, so the workspace is bounded by the group, not by the
 * batch.  It grows to the largest group seen and is kept (mlz_get_counter 3 / 4 report it).  Per group, decode holds about 7 bytes per
 * compressed byte (region exits, the token list — sized for one token per stream byte —, 16-bit token positions) and, for the
 * general-block pass (blocks of other encoders, LevelBalanced's own), 3 bytes per output byte + 8 bytes per stream byte: about 10 bytes
 * per output byte on text-like data, 5.3 GB for a full group.  Encode holds about 3.3 bytes per input byte (token records, piece
 * scratch) plus the far tables (1.5 MiB per 8 MiB block, LevelBalanced 8 MiB): about 3 GB per group.  If the general pass's buffers
 * cannot be allocated, general blocks decode on the exec pass's tile chain instead (slow, correct; mlz_get_counter 5 counts such calls). */
/* A stream that was used for a *_batch_device call and is about to be destroyed: call this first (any time after the last call on it).  The context
 * then records its ordering event on the stream while it is alive; without it the event is recorded lazily, by the context's NEXT call, on the previous
 * call's stream — which must therefore still exist then.  Not needed for streams that outlive the context's use, nor for the host-pointer calls. */
int mlz_release_stream(mlz_ctx* ctx, void* stream);
int mlz_encode_batch_device(mlz_ctx* ctx, void* stream, int level, const uint8_t* d_src, uint8_t* d_dst,
                            const mlz_block_desc* desc, int n_blocks, int64_t* d_out_len);
int mlz_decode_batch_device(mlz_ctx* ctx, void* stream, const uint8_t* d_src, uint8_t* d_dst,
                            const mlz_block_desc* desc, int n_blocks, int64_t* d_out_len);

/* ---- masked CRC32C (stream chunks) ----
 * crc(b) of minlz.go:133-140: Castagnoli CRC, rotated by 15, plus 0xa282ead8; computed over the
 * uncompressed bytes of each block (writer.go:887, reader.go:341-351).
 * The CRC is a pass of its own over the block's bytes (0.059 ms per 100 MB, 4 % of an encode + decode step), not fused into the encoder's read
 * or the decoder's write as SURVEY.md 8(f1) words it: the match kernel is bound by instruction issue and the separate pass also serves the
 * Reader's check and chunk 0x03 (CRC over the token bytes) unchanged.
 * mlz_crc: host buffer, returns the 32-bit value (>= 0) or -MLZ_ERR_*.
 * mlz_crc_batch_device: blocks described by desc[i].src_off/src_len relative to d_base; d_out[i]
 * (uint32, device) receives the masked CRC of block i. */
int64_t mlz_crc(mlz_ctx* ctx, const uint8_t* src, size_t n);
int mlz_crc_batch_device(mlz_ctx* ctx, void* stream, const uint8_t* d_base, const mlz_block_desc* desc, int n_blocks,
                         uint32_t* d_out);

/* ---- whole-buffer streams (host pointers) ----
 * mlz_stream_encode: NewWriter(dst, WriterLevel(level), WriterBlockSize(block_size), WriterAddIndex(flag))
 *   .EncodeBuffer(src) followed by Close() (writer.go:441-563, :854-965, :1051-1126): stream header,
 *   one 0x02 / 0x01 chunk per block with the masked CRC32C of its uncompressed bytes, EOF chunk, and
 *   with MLZ_STREAM_ADD_INDEX the seek index (index.go:191-269).  Returns the stream size.
 * mlz_stream_decode: NewReader(src) read to EOF (reader.go:248-543), MinLZ streams only; CRCs are
 *   verified unless MLZ_STREAM_IGNORE_CRC (ReaderIgnoreCRC).  Returns the decoded size.
 * Copies to and from the device overlap the kernels group by group (~256 MiB). */
#define MLZ_STREAM_ADD_INDEX 1u
#define MLZ_STREAM_IGNORE_CRC 2u
int64_t mlz_stream_bound(uint64_t n, uint32_t block_size, uint32_t flags); /* dst_cap that always suffices */
int64_t mlz_stream_encode(mlz_ctx* ctx, int level, uint32_t block_size, uint32_t flags, const uint8_t* src, size_t n, uint8_t* dst,
                          size_t dst_cap);
int64_t mlz_stream_decoded_len(const uint8_t* src, size_t n); /* host-only chunk walk: total decoded bytes */
int64_t mlz_stream_decode(mlz_ctx* ctx, uint32_t flags, const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap);

/* The device-resident Writer over several devices: range j of the stream lies in HBM at d_src[j] (src_len[j] bytes; every range but the last a whole
 * number of blocks), on any device of the context — normally range j on device j, the concurrent Writer's workers (writer.go:501-560) being the GPUs —
 * and the framed stream (header, the chunks in order, EOF, index with MLZ_STREAM_ADD_INDEX) is assembled in d_dst, a buffer on ONE device.  Every device
 * encodes and checksums its range and frames its run of chunks in its own HBM; 12 bytes per block (size, CRC) visit the host, so that every run's place
 * is known (the in-order emit, writer.go:219-272); the runs then travel GPU to GPU into d_dst (hipMemcpyPeerAsync: xGMI between the GPUs of a node; a run
 * already on d_dst's device is framed in place).  No payload crosses PCIe and there is no collective: a gather of variable-length runs to one consumer
 * is n - 1 point-to-point copies, which is also all RCCL's gather would issue.  Bytes are identical to mlz_stream_encode of the concatenated ranges.
 * Synchronous; returns the stream size.  Works on a one-device context too (n_ranges ranges encoded one after the other). */
int64_t mlz_stream_encode_gather_device(mlz_ctx* ctx, int level, uint32_t block_size, uint32_t flags, const uint8_t* const* d_src, const size_t* src_len,
                                        int n_ranges, uint8_t* d_dst, size_t dst_cap);

/* ---- tuning / introspection (not part of the reference surface) ---- */
#define MLZ_OPT_DECODE_ALGO 1  /* 0 = parallel (default), 1 = serial one-wave-per-block, 3 = parallel with every block on the tile path (cross-checks) */
#define MLZ_OPT_ENCODE_FAR 2   /* 0 = tile-local matches only, 1 = + far matches (default) */
#define MLZ_OPT_L2_FREE 14     /* LevelBalanced: 1 (default) = no tile levels — a copy may read any earlier tile of its window: the ratio of the
                                * reference's encode_l2.go and better (0.93 - 1.05 x its restatement), and the blocks decode, like the reference's own,
                                * through the general-block path; 0 = the four-level tile pattern of rounds 1-3 (1.08 - 1.09 x, level-scheduled decode) */
#define MLZ_OPT_DEVICE_GROUP 17 /* MiB of uncompressed data per internal group of a device batch (default 512): bounds the workspace */
#define MLZ_OPT_L2_GAP 19      /* LevelBalanced without tile levels: a copy from another tile reads at least this many tiles back (default 4; 1 = anywhere,
                                * rounds 4's form).  With K >= 2 the K tiles in front of a tile never feed it, and the decoder settles K (2 or 4) tiles of a
                                * block side by side instead of one: 0.3 - 1.0 % (K = 2) / 0.8 - 2.6 % (K = 4) of output for 2 - 3 x the decode rate of
                                * batches of few large blocks.  The decoder measures the distance itself: any stream that keeps it decodes this way. */
#define MLZ_OPT_FUSED_SERIALIZER 21 /* encode: 1 (default) = the match kernel's waves turn their own token records into token bytes (tile bytes and ring in LDS, records and
                                * far-source lines still in the L2); 0 = the separate serializer kernel of rounds 2-5 on the same records (byte-identical; cross-checks) */
#define MLZ_OPT_LEVEL0_KERNEL 23 /* decode: 1 (default) = a batch's level-0 tiles, when no more than the device has CUs, are decoded by dec_level0_kernel before the exec pass; 0 = by the exec pass (cross-checks) */
#define MLZ_OPT_FOLD_LAYOUT 24  /* encode: 1 (default) = a group whose every block has tiles and room gets its layout (piece offsets, stored-or-not, header, length) from the gather kernel; 0 = always encode_layout_kernel (cross-checks) */
#define MLZ_OPT_INDEX_PASSES 15 /* decode, cross-checks: 1 = the index pass as the three kernels of rounds 2-3 instead of dec_index1 / dec_index2 / dec_viol (default 0) */
/* (debug, timing experiments: option 16 = 1 makes mlz_decode_batch_device return after the index pass, without output) */
#define MLZ_OPT_GEN_SPIN 9     /* patience of the general-block decode with a tile's ready flag, in polls (~0.3 us each; default 2^24); tests */
#define MLZ_OPT_GEN_PACKED 13  /* tests: 1 = general blocks settle through the byte-packed pool (the fallback of tiles whose slots do not fit) */
int mlz_set_option(mlz_ctx* ctx, int opt, int64_t value);
/* Milliseconds spent in each kernel family, measured with HIP events on the caller's stream.
 * mlz_set_option(ctx, MLZ_TIMER_ENABLE, 1): the last *_batch_device call (mlz_get_timers waits for it);
 * (ctx, MLZ_TIMER_ENABLE, 2): running mean over all calls since then — events are read back several calls
 * late, so nothing waits on the device per call; 0 = off. */
#define MLZ_TIMER_ENABLE 100
int mlz_get_timers(mlz_ctx* ctx, float* ms, int cap); /* returns number of timers written */
const char* mlz_timer_name(int idx);
/* Counters of the combining queue behind the single-block host calls (mlz_encode, mlz_encode_block, mlz_decode,
 * mlz_decode_block): concurrent callers — one goroutine per block in the reference's Writer/Reader, writer.go:501-560,
 * reader.go:830-859 — are run as one batched launch.  which: 0 = batches run, 1 = requests served.
 * which = 2: blocks of the last decode call that matched no tile-level pattern of this library's encoder and went through the
 * general-block path (mlz_decode_general.hip.inc): the reference's own blocks, and this library's LevelBalanced ones
 *            (summed over the internal groups a batch ran as).
 * which = 3 / 4: bytes of device workspace the context holds for encoding / decoding (grow-only: the high-water mark so far).
 * which = 5: decode calls whose general blocks fell back to the tile chain because the general pass's buffers could not be allocated.
 * which = 6: workgroups per block (1, 2 or 4) the general-block pass of the last decode call settled with (the largest over its internal groups);
 *            0 = it had no general block. */
int64_t mlz_get_counter(mlz_ctx* ctx, int which);

#ifdef __cplusplus
}
#endif
#endif
