// mlz_hip.hip — C ABI (include/minlz_hip.h) + launch logic of the MI355X MinLZ block codec.
//
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
// No CPU codec lives in this library: every encode/decode call runs the HIP kernels, and a HIP
// failure is reported as -MLZ_ERR_HIP (the Go wrapper then falls back to upstream's CPU path).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/minlz_hip.h"
#include "mlz_format.h"
#include "mlz_kernels.h"

#include "mlz_encode.hip.inc"
#include "mlz_encode2.hip.inc"
#include "mlz_decode_serial.hip.inc"
#include "mlz_toktab.h"
#include "mlz_decode.hip.inc"
#include "mlz_decode_index.hip.inc"
#include "mlz_decode_exec.hip.inc"
#include "mlz_decode_general.hip.inc"
#include "mlz_crc.hip.inc"

using namespace mlz;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = n + n / 4 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() { return static_cast<T*>(p); }
};

constexpr size_t kProfTiles = 4096;                      // debug timeline: tiles recorded by the exec pass
constexpr size_t kProfBytes = 256 + kProfTiles * 32;

// Levels served on the device (encode.go:25-43); LevelSmallest (3) stays on the host side of the boundary.
inline bool valid_level(int level) { return level >= MLZ_LEVEL_SUPERFAST && level <= MLZ_LEVEL_BALANCED; }

enum { T_FAR = 0, T_ENC_TILES, T_ENC_LAYOUT, T_ENC_GATHER, T_DEC_PARSE, T_DEC_CHAIN, T_DEC_INDEX, T_DEC_EXEC, T_DEC_SERIAL, T_CRC, T_ENC_SER, T_DEC_GENERAL, T_COUNT };
const char* kTimerNames[T_COUNT] = {"enc_far_build", "enc_tiles", "enc_layout", "enc_gather", "dec_parse", "dec_chain", "dec_index", "dec_exec", "dec_serial", "crc", "enc_serialize", "dec_general"};

}  // namespace

// One single-block host call waiting to be run (see submit_single below).
struct SingleReq {
    bool encode, with_header, has_dlen;
    int level;
    const uint8_t* src; size_t n;
    uint8_t* dst; size_t cap;
    size_t dlen;      // decode_block: exact decoded length
    int64_t out = 0;  // bytes produced (or a negative per-block error)
    int rc = 0;       // failure of the whole batch
    bool done = false;
    bool same_kind(const SingleReq& o) const { return encode == o.encode && with_header == o.with_header && has_dlen == o.has_dlen && level == o.level; }
};

constexpr size_t kHostGroupEncodeDefault = 32u << 20, kHostGroupDecodeDefault = 48u << 20;  // see host_batch

struct mlz_ctx {
    int device = 0;
    // mlz_init_devices: the per-device contexts this one deals blocks to.  Such a context owns no device state of its own: every entry point
    // either fans out over `kids` (batches, streams), hands a single-block call to one of them in turn, or routes a device-resident call to the
    // kid whose device holds the buffers.
    std::vector<mlz_ctx*> kids;
    std::atomic<uint32_t> rr{0};
    std::mutex mu;
    // combining queue of the single-block host calls
    std::mutex q_mu;
    std::condition_variable q_cv;
    std::vector<SingleReq*> q_pending;
    bool q_leader = false;
    uint64_t q_batches = 0, q_requests = 0;  // mlz_get_counter
    int index_passes = 0;                    // MLZ_OPT_INDEX_PASSES
    int debug_stop = 0;                      // debug option 16: decode stops after the index pass (timing experiments with broken kernel variants)
    void* last_gen = nullptr;                // GenCtl of the last decode call's last group (device memory)
    uint64_t acc_call = 0;                   // the call_seq whose first group has reset d_gen_acc
    std::string err;
    std::string dev_name;
    hipStream_t stream = nullptr;  // used by the host-pointer calls
    // descriptors
    // Block descriptors on the device, one set per kind of call (0 encode / crc, 1 decode): an encode and a decode that
    // alternate (a Writer next to a Reader, bench.py) each find their descriptors already uploaded.
    DevBuf d_blocks_k[2], d_tile_block_k[2], d_seg_block_k[2];
    DevBuf d_place;                          // mlz_stream_encode into pinned memory: placement descriptors
    std::vector<BlockInfo> h_blocks, h_blocks_prev_k[2];
    int dk = 0;  // kind of the call in progress
    DevBuf& d_blocks_cur() { return d_blocks_k[dk]; }
    DevBuf& d_tile_block_cur() { return d_tile_block_k[dk]; }
    DevBuf& d_seg_block_cur() { return d_seg_block_k[dk]; }
    std::vector<uint32_t> h_tile_block, h_seg_block;
    // descriptor staging: two page-locked buffers used in turn, so that the upload of one internal group (device_group below) does not make
    // the host wait for the kernels of the group before it
    void* pinned_k[2] = {nullptr, nullptr};
    size_t pinned_cap_k[2] = {0, 0};
    hipEvent_t upload_done_k[2] = {nullptr, nullptr};
    bool upload_pending_k[2] = {false, false};
    int up_slot = 0;
    // Device batches run in internal groups of about this many bytes of uncompressed data (option MLZ_OPT_DEVICE_GROUP): the workspace is then
    // bounded by the group, not by the batch (round 4: 42 GB of decode workspace for a 4 GiB batch).  Throughput is flat from 64 x 8 MiB blocks on.
    size_t device_group = size_t(512) << 20;
    // One workspace serves every call on this context: a call on another stream first waits (on the device) for the
    // previous call's last launch, so descriptors, scratch and flags are never shared by two calls in flight.
    hipEvent_t ws_done = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_used = false;
    bool ws_recorded = false;   // mlz_release_stream: ws_done already covers the last call (its stream may be gone by the next one)
    // encode workspace
    DevBuf d_scratch, d_tile_size, d_tile_out, d_flags, d_far, d_recs, d_piece_cnt, d_farbin;
    bool farbin_attr = false;
    // decode workspace
    DevBuf d_dec, d_idx;
    DevBuf d_gen_acc;      // two words summed over a call's internal groups by dec_schedule_kernel: general blocks, largest team (mlz_get_counter 2 / 6)
    int general_algo = 0;  // 0 = pointer-jumping pass for general blocks, 1 = tile chain in the exec pass
    size_t host_group_enc = kHostGroupEncodeDefault, host_group_dec = kHostGroupDecodeDefault;  // host-pointer batches: bytes per overlapped group
    int gen_grid = 0;      // workgroups of dec_general_kernel the device holds at once
    uint32_t gen_spin_limit = 1u << 24;  // role S's patience with a tile's ready flag, in polls (~0.3 us each): ~5 s
    int n_cus = 0;
    int far_slices_l2 = 0;     // debug option 18: 1 = LevelBalanced's far tables by far_build_kernel (slice workgroups, round 4) even without level sets
    int gen_settle_cap = 0;    // debug option 20: settling workgroups of dec_general_kernel at most (0 = a quarter of the device)
    int l2_gap = 4;            // option 19: LevelBalanced without levels: a far source lies at least this many tiles back (1 = anywhere; 4: the decoder settles four tiles of a block side by side)
    int l2_free = 1;           // option 14 (default on): LevelBalanced without the tile-level constraint (better ratio; its blocks decode through the general path)
    uint64_t gen_fallbacks = 0;  // decode calls whose general blocks took the tile chain because the general pass's buffers could not be allocated (mlz_get_counter 5)
    int gen_force_packed = 0;  // tests: every tile of a general block takes the byte-packed pool (the fallback path)
    int fold_layout = 1;       // option 24: the encode layout rides in the gather kernel when every block of the group has tiles and room (0: always encode_layout_kernel)
    int level0_by_e = 1;       // option 23: few level-0 tiles are decoded by dec_level0_kernel before the exec pass (0: by the exec pass, rounds 2-5)
    bool l0_attr = false;
    int fuse_ser = 1;          // option 21: the match kernel serializes its pieces itself (0: serialize_pieces_kernel, rounds 2-5; cross-checks)
    // host-pointer staging
    DevBuf d_in, d_out, d_len, d_crc, d_crc_tabs, d_crc_tiles;
    // stream calls: copy-in / copy-out streams, event pool, pinned result buffer
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<hipEvent_t> evpool;
    void* pinned2 = nullptr;
    size_t pinned2_cap = 0;
    // options
    int decode_algo = 0;
    int encode_far = 1;
    bool enc_attrs = false, far_attr = false, dec_attrs = false, gen_attr = false;  // per device: dynamic-LDS limits raised
    uint32_t timer_mask = 0xffffffffu;  // timers that record events (an event pair costs ~10 us of idle device per kernel boundary)
    int timing = 0;  // 0 off, 1 = the last call's kernel times, 2 = running mean over the calls since it was enabled (no sync per call)
    int debug_status = 0;
    bool prof_on = false;
    DevBuf d_prof;
    // A ring of event pairs per timer, resolved kTimerRing uses later (long complete by then), so reading the clock never stalls the caller and
    // launches can run ahead of the device.  A device batch runs as one or more internal groups, each firing the timers: acc_ms sums them and
    // ev_calls counts the API calls a timer fired in.  timing == 1 restarts a timer's sum at the first firing of a new call (the last call's
    // times), timing == 2 keeps the sum over all calls since it was enabled (get_timers divides by the calls).
    static constexpr int kTimerRing = 8;
    hipEvent_t evr[T_COUNT][kTimerRing][2] = {};
    uint64_t ev_cnt[T_COUNT] = {}, ev_res[T_COUNT] = {}, ev_calls[T_COUNT] = {}, ev_last_call[T_COUNT] = {};
    uint64_t call_seq = 0;
    double acc_ms[T_COUNT] = {};
    void resolve_one(int id) {
        const int slot = int(ev_res[id] % kTimerRing);
        float t = 0;
        if (hipEventSynchronize(evr[id][slot][1]) == hipSuccess && hipEventElapsedTime(&t, evr[id][slot][0], evr[id][slot][1]) == hipSuccess) acc_ms[id] += t;
        ev_res[id]++;
    }
};

namespace {

#define HIPCHK(ctx, call)                                                                 \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return -MLZ_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

struct Timer {
    mlz_ctx* c; int id; hipStream_t s;
    bool on;
    Timer(mlz_ctx* c_, int id_, hipStream_t s_) : c(c_), id(id_), s(s_), on(c_->timing != 0 && ((c_->timer_mask >> id_) & 1) != 0) {
        if (!on) return;
        if (c->ev_last_call[id] != c->call_seq) {   // first firing in this API call
            c->ev_last_call[id] = c->call_seq;
            if (c->timing == 1) {
                while (c->ev_res[id] < c->ev_cnt[id]) c->resolve_one(id);
                c->acc_ms[id] = 0; c->ev_calls[id] = 0;
            }
            c->ev_calls[id]++;
        }
        while (c->ev_cnt[id] - c->ev_res[id] >= uint64_t(mlz_ctx::kTimerRing)) c->resolve_one(id);
        (void)hipEventRecord(c->evr[id][c->ev_cnt[id] % mlz_ctx::kTimerRing][0], s);
    }
    ~Timer() {
        if (!on) return;
        (void)hipEventRecord(c->evr[id][c->ev_cnt[id] % mlz_ctx::kTimerRing][1], s);
        c->ev_cnt[id]++;
    }
};

// Orders the calls that share the context's workspace across streams (see mlz_ctx::ws_done).  The event is recorded LAZILY — by the next call
// that arrives on another stream, on the previous call's stream: it then covers everything that stream was given, the previous call included —
// because an event record after every call idles the device for ~5 us at each call boundary (rocprofv3 timeline: 11 us of a 1.41 ms
// encode + decode step), and calls on one stream need no event at all.
struct WorkspaceOrder {
    mlz_ctx* c; hipStream_t s;
    WorkspaceOrder(mlz_ctx* c_, hipStream_t s_) : c(c_), s(s_) {
        if (c->ws_used && (c->ws_recorded || c->ws_stream != s)) {
            if (c->ws_recorded) (void)hipStreamWaitEvent(s, c->ws_done, 0);   // recorded by mlz_release_stream while the stream was alive
            else if (hipEventRecord(c->ws_done, c->ws_stream) == hipSuccess) (void)hipStreamWaitEvent(s, c->ws_done, 0);
            else { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }   // (the caller destroyed that stream: whatever it held is waited for)
        }
    }
    ~WorkspaceOrder() {
        c->ws_stream = s;
        c->ws_used = true;
        c->ws_recorded = false;
    }
};

// Builds BlockInfo / tile map on the host and uploads them when they differ from the last call.
int upload_blocks(mlz_ctx* c, hipStream_t st, const mlz_block_desc* desc, int n, bool tiles_from_dst, uint32_t* total_tiles, uint32_t* total_segs = nullptr,
                  const uint64_t* mirror = nullptr, bool any_length = false /* the CRC pass: tiles for spans of any length */) {
    c->dk = tiles_from_dst ? 1 : 0;
    std::vector<BlockInfo>& prev = c->h_blocks_prev_k[c->dk];
    c->h_blocks.resize(n);
    uint32_t tiles = 0, segs = 0;
    for (int i = 0; i < n; i++) {
        BlockInfo& b = c->h_blocks[i];
        b.src_off = desc[i].src_off; b.src_len = desc[i].src_len; b.dst_off = desc[i].dst_off; b.dst_cap = desc[i].dst_cap;
        b.mirror = mirror ? mirror[i] : 0;
        uint64_t span = tiles_from_dst ? desc[i].dst_cap : desc[i].src_len;
        if (span > kMaxBlockSize && !any_length) span = tiles_from_dst ? kMaxBlockSize : 0;
        b.first_tile = tiles;
        b.n_tiles = uint32_t((span + kTile - 1) >> kTileLog);
        tiles += b.n_tiles;
        b.first_seg = segs;
        b.n_segs = 0;
        if (tiles_from_dst) {  // decode: segments of the token stream
            // a valid token stream is at most twice its output (dec_header_kernel rejects longer ones before any pass walks them)
            uint64_t cl = std::min<uint64_t>(desc[i].src_len, 2 * std::min<uint64_t>(desc[i].dst_cap, kMaxBlockSize) + 32);
            b.n_segs = uint32_t((cl + kSeg - 1) >> kSegLog);
            segs += b.n_segs;
        }
    }
    *total_tiles = tiles;
    if (total_segs) *total_segs = segs;
    const bool same = prev.size() == c->h_blocks.size() && std::memcmp(prev.data(), c->h_blocks.data(), sizeof(BlockInfo) * n) == 0;
    if (same) return 0;
    c->h_tile_block.resize(tiles);
    c->h_seg_block.resize(segs);
    for (int i = 0; i < n; i++) {
        for (uint32_t t = 0; t < c->h_blocks[i].n_tiles; t++) c->h_tile_block[c->h_blocks[i].first_tile + t] = uint32_t(i);
        for (uint32_t t = 0; t < c->h_blocks[i].n_segs; t++) c->h_seg_block[c->h_blocks[i].first_seg + t] = uint32_t(i);
    }
    const size_t nb = sizeof(BlockInfo) * n, nt = sizeof(uint32_t) * tiles, ns = sizeof(uint32_t) * segs;
    const int sl = (c->up_slot ^= 1);
    if (c->upload_pending_k[sl]) { HIPCHK(c, hipEventSynchronize(c->upload_done_k[sl])); c->upload_pending_k[sl] = false; }
    if (nb + nt + ns > c->pinned_cap_k[sl]) {
        if (c->pinned_k[sl]) HIPCHK(c, hipHostFree(c->pinned_k[sl]));
        c->pinned_k[sl] = nullptr;
        c->pinned_cap_k[sl] = (nb + nt + ns) * 2 + 4096;
        HIPCHK(c, hipHostMalloc(&c->pinned_k[sl], c->pinned_cap_k[sl], hipHostMallocDefault));
    }
    char* pin = static_cast<char*>(c->pinned_k[sl]);
    HIPCHK(c, c->d_blocks_cur().ensure(nb + 64));
    HIPCHK(c, c->d_tile_block_cur().ensure(nt + 64));
    HIPCHK(c, c->d_seg_block_cur().ensure(ns + 64));
    std::memcpy(pin, c->h_blocks.data(), nb);
    std::memcpy(pin + nb, c->h_tile_block.data(), nt);
    std::memcpy(pin + nb + nt, c->h_seg_block.data(), ns);
    if (nb) HIPCHK(c, hipMemcpyAsync(c->d_blocks_cur().p, pin, nb, hipMemcpyHostToDevice, st));
    if (nt) HIPCHK(c, hipMemcpyAsync(c->d_tile_block_cur().p, pin + nb, nt, hipMemcpyHostToDevice, st));
    if (ns) HIPCHK(c, hipMemcpyAsync(c->d_seg_block_cur().p, pin + nb + nt, ns, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->upload_done_k[sl], st));
    c->upload_pending_k[sl] = true;
    prev = c->h_blocks;
    return 0;
}

// Blocks [0, n) of a device batch cut into consecutive groups of about c->device_group bytes of uncompressed data (at least one block each):
// calls fn(first, count) per group.  The groups run one after the other on the call's stream and share the workspace.
template <class F> int for_each_group(mlz_ctx* c, const mlz_block_desc* desc, int n, bool by_dst, F fn) {
    int b0 = 0;
    while (b0 < n) {
        uint64_t bytes = 0;
        int b1 = b0;
        while (b1 < n) {
            const uint64_t span = std::min<uint64_t>(by_dst ? desc[b1].dst_cap : desc[b1].src_len, kMaxBlockSize);
            if (b1 > b0 && bytes + span > c->device_group) break;
            bytes += span;
            b1++;
        }
        const int r = fn(b0, b1 - b0);
        if (r) return r;
        b0 = b1;
    }
    return 0;
}

int encode_device_group(mlz_ctx* c, hipStream_t st, int level, const uint8_t* d_src, uint8_t* d_dst, const mlz_block_desc* desc, int n,
                        int64_t* d_out_len, bool with_header, const uint64_t* mirror) {
    uint32_t tiles = 0;
    int r = upload_blocks(c, st, desc, n, false, &tiles, nullptr, mirror);
    if (r) return r;
    // Every level goes through the match + serialize kernels on 8 KiB pieces (mlz_encode2.hip.inc); LevelBalanced is their
    // configuration with larger near tables, denser far tables and a second far probe (kL2*).
    const bool l2new = level == MLZ_LEVEL_BALANCED;
    const uint32_t sub_log = kSubLog;
    const size_t units = size_t(tiles) << sub_log;
    HIPCHK(c, c->d_tile_size.ensure(sizeof(uint32_t) * (units + 1)));
    HIPCHK(c, c->d_tile_out.ensure(sizeof(uint32_t) * (units + 1)));
    HIPCHK(c, c->d_flags.ensure(sizeof(uint32_t) * n));
    const BlockInfo* blocks = c->d_blocks_cur().as<BlockInfo>();
    const uint32_t* tile_block = c->d_tile_block_cur().as<uint32_t>();
    if (level != MLZ_LEVEL_UNCOMPRESSED && tiles > 0) {
        HIPCHK(c, c->d_scratch.ensure(units * kPieceScratch));
        uint64_t maxlen = 0;
        for (int i = 0; i < n; i++) maxlen = std::max<uint64_t>(maxlen, std::min<uint64_t>(desc[i].src_len, kMaxBlockSize));
        const uint32_t epochs = uint32_t((maxlen + (1u << kEpochLog) - 1) >> kEpochLog);
        // LevelBalanced: far matching forced on, both epochs probed and a cost-aware lazy parse (DESIGN.md "Levels").
        // LevelSuperFast: tile-local matches only (no far tables are built or probed).
        const bool far = ((c->encode_far && level != MLZ_LEVEL_SUPERFAST) || level == MLZ_LEVEL_BALANCED) && maxlen > kTile;
        const uint32_t pattern = (level == MLZ_LEVEL_BALANCED && c->l2_free) ? kPatternFree : level_pattern_of(level);   // LevelBalanced: dense (four levels); the faster levels: three (DESIGN.md "Tile levels"); the decoder knows both and round 1's kPatternFast
        bool any_big = false, any_small = false;
        for (int i = 0; i < n; i++) (std::min<uint64_t>(desc[i].src_len, kMaxBlockSize) >= kM2BigBlock ? any_big : any_small) = true;
        if (!c->enc_attrs) {  // per context = per device
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(match_tiles_kernel<true, MLZ_M2_NW, kM2HashBitsBig>), hipFuncAttributeMaxDynamicSharedMemorySize, M2Cfg<kM2HashBitsBig>::kLds));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(match_tiles_kernel<true, MLZ_M2_NW, kM2HashBitsSmall, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, M2Cfg<kM2HashBitsSmall>::kLds));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(match_tiles_kernel<false, MLZ_M2_NW, kM2HashBitsBig>), hipFuncAttributeMaxDynamicSharedMemorySize, M2Cfg<kM2HashBitsBig>::kLds));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(match_tiles_kernel<false, MLZ_M2_NW, kM2HashBitsSmall>), hipFuncAttributeMaxDynamicSharedMemorySize, M2Cfg<kM2HashBitsSmall>::kLds));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(match_tiles_kernel<false, MLZ_M2_NW, kM2HashBitsSuperFast>), hipFuncAttributeMaxDynamicSharedMemorySize, M2Cfg<kM2HashBitsSuperFast>::kLds));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(match_tiles_kernel<true, MLZ_M2_NW, kM2HashBitsSmall, kL2FarBits, true>), hipFuncAttributeMaxDynamicSharedMemorySize, M2Cfg<kM2HashBitsSmall>::kLds));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(match_tiles_kernel<true, MLZ_M2_NW, kM2HashBitsSmall, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, M2Cfg<kM2HashBitsSmall>::kLds));
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(match_tiles_kernel<false, MLZ_M2_NW, kM2HashBitsSmall, kL2FarBits, true>), hipFuncAttributeMaxDynamicSharedMemorySize, M2Cfg<kM2HashBitsSmall>::kLds));
            c->enc_attrs = true;
        }
        // LevelFastest: blocks below 1 MiB have far tables that go with their length (small_far_bits); the tables of a batch
        // are as far apart as its largest block needs
        const int fbits = any_big ? (l2new ? kL2FarBits : kFarBits) : small_far_bits(maxlen) + (l2new ? 1 : 0);
        if (far) {
            Timer t(c, T_FAR, st);
            const size_t words = (size_t(n) * far_sets(pattern) * epochs) << fbits;
            HIPCHK(c, c->d_far.ensure(words * 4));
            if (!c->far_attr) {
                HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(far_build_kernel<kFarBits, kFarStride>), hipFuncAttributeMaxDynamicSharedMemorySize, 4u << kFarSliceBits));
                HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(far_build_kernel<0, kFarStride>), hipFuncAttributeMaxDynamicSharedMemorySize, 4u << kFarSliceBits));
                HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(far_build_kernel<0, kL2FarStride>), hipFuncAttributeMaxDynamicSharedMemorySize, 4u << kFarSliceBits));
                HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(far_build_kernel<kL2FarBits, kL2FarStride>), hipFuncAttributeMaxDynamicSharedMemorySize, 4u << kFarSliceBits));
                c->far_attr = true;
            }
            if (any_big) {
                if (l2new && pattern == kPatternFree && !c->far_slices_l2) {
                    // no level sets: the windows sorted by slice once (far_bin_kernel), every slice workgroup reads its eighth (far_slice_kernel)
                    using FBn = FarBin<kL2FarBits, kL2FarStride>;
                    const size_t units2 = size_t(tiles) * 2;
                    HIPCHK(c, c->d_farbin.ensure(units2 * FBn::kPerUnit * 4 + units2 * FBn::kOffs * 4 + 256));
                    uint32_t* bins = c->d_farbin.as<uint32_t>();
                    uint32_t* binoff = bins + units2 * FBn::kPerUnit;
                    constexpr uint32_t kSliceLds = (4u << kBinSliceBits) + 256 * 4 + 132 * 4;
                    if (!c->farbin_attr) {
                        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(far_bin_kernel<kL2FarBits, kL2FarStride>), hipFuncAttributeMaxDynamicSharedMemorySize, FBn::kLds));
                        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(far_slice_kernel<kL2FarBits, kL2FarStride>), hipFuncAttributeMaxDynamicSharedMemorySize, kSliceLds));
                        c->farbin_attr = true;
                    }
                    hipLaunchKernelGGL((far_bin_kernel<kL2FarBits, kL2FarStride>), dim3(uint32_t(units2)), dim3(256), FBn::kLds, st, d_src, blocks, tile_block, bins, binoff);
                    hipLaunchKernelGGL((far_slice_kernel<kL2FarBits, kL2FarStride>), dim3(FBn::kSlices, epochs, n), dim3(1024), kSliceLds, st, blocks,
                                       bins, binoff, c->d_far.as<uint32_t>(), epochs);
                } else if (l2new)
                    hipLaunchKernelGGL((far_build_kernel<kL2FarBits, kL2FarStride>), dim3(far_slices(kL2FarBits), epochs, n), dim3(1024), 4u << kFarSliceBits, st, d_src,
                                       blocks, c->d_far.as<uint32_t>(), epochs, pattern, uint32_t(fbits), any_small ? 1u : 2u);
                else
                    hipLaunchKernelGGL((far_build_kernel<kFarBits, kFarStride>), dim3(far_slices(kFarBits), epochs, n), dim3(1024), 4u << kFarSliceBits, st, d_src,
                                       blocks, c->d_far.as<uint32_t>(), epochs, pattern, uint32_t(fbits), any_small ? 1u : 2u);
            }
            if (any_small) {
                uint64_t max_small = 0;
                for (int i = 0; i < n; i++) if (desc[i].src_len < kBigBlock) max_small = std::max<uint64_t>(max_small, desc[i].src_len);
                const int sb = small_far_bits(max_small) + (l2new ? 1 : 0);
                const uint32_t lds = 4u << std::min<int>(sb, kFarSliceBits);   // small tables leave room for more workgroups per CU
                const dim3 grid(1u << (std::max<int>(sb, kFarSliceBits) - kFarSliceBits), 1, n);
                if (l2new)
                    hipLaunchKernelGGL((far_build_kernel<0, kL2FarStride>), grid, dim3(1024), lds, st, d_src, blocks, c->d_far.as<uint32_t>(), epochs, pattern,
                                       uint32_t(fbits), any_big ? 0u : 2u);
                else
                    hipLaunchKernelGGL((far_build_kernel<0, kFarStride>), grid, dim3(1024), lds, st, d_src, blocks, c->d_far.as<uint32_t>(), epochs, pattern,
                                       uint32_t(fbits), any_big ? 0u : 2u);
            }
        }
        const uint32_t* ftab = far ? c->d_far.as<uint32_t>() : nullptr;
        // LevelBalanced without levels: far sources at least l2_gap tiles back (MLZ_OPT_L2_GAP)
        const uint32_t far_gap = pattern == kPatternFree && level == MLZ_LEVEL_BALANCED ? uint32_t(c->l2_gap - 1) << kTileLog : 0u;
        {
            HIPCHK(c, c->d_recs.ensure(units * kRecPerPiece * sizeof(uint2)));
            HIPCHK(c, c->d_piece_cnt.ensure(units * sizeof(uint32_t)));
            // the match kernel's waves serialize their pieces themselves (option 21, default on); else serialize_pieces_kernel below
            uint8_t* fuse_scratch = MLZ_M2_FUSE_SER && c->fuse_ser ? c->d_scratch.as<uint8_t>() : nullptr;
            {
                Timer t(c, T_ENC_TILES, st);
                const uint32_t grid = ((tiles + 7) / 8) * 8;  // whole rounds of the eight XCDs (see the kernel's workgroup -> tile map)
#define MLZ_LAUNCH_M2(F, HB, CLS, ...)                                                                                                       \
    hipLaunchKernelGGL((match_tiles_kernel<F, MLZ_M2_NW, HB, ##__VA_ARGS__>), dim3(grid), dim3(256), M2Cfg<HB>::kLds, st, d_src, blocks, tile_block,        \
                       c->d_recs.as<uint2>(), c->d_piece_cnt.as<uint32_t>(), ftab, epochs, pattern, tiles, uint32_t(CLS), uint32_t(fbits), far_gap, fuse_scratch, c->d_tile_size.as<uint32_t>())
                if (level == MLZ_LEVEL_SUPERFAST) MLZ_LAUNCH_M2(false, kM2HashBitsSuperFast, 2);
                else if (l2new && far) {
                    // the same kernel for both block classes, with far tables of the level's size or of the block's
                    if (any_big)
                        hipLaunchKernelGGL((match_tiles_kernel<true, MLZ_M2_NW, kM2HashBitsSmall, kL2FarBits, true>), dim3(grid), dim3(256), M2Cfg<kM2HashBitsSmall>::kLds,
                                           st, d_src, blocks, tile_block, c->d_recs.as<uint2>(), c->d_piece_cnt.as<uint32_t>(), ftab, epochs, pattern, tiles,
                                           any_small ? 1u : 2u, uint32_t(fbits), far_gap, fuse_scratch, c->d_tile_size.as<uint32_t>());
                    if (any_small)
                        hipLaunchKernelGGL((match_tiles_kernel<true, MLZ_M2_NW, kM2HashBitsSmall, 0, true>), dim3(grid), dim3(256), M2Cfg<kM2HashBitsSmall>::kLds, st,
                                           d_src, blocks, tile_block, c->d_recs.as<uint2>(), c->d_piece_cnt.as<uint32_t>(), ftab, epochs, pattern, tiles,
                                           any_big ? 0u : 2u, uint32_t(fbits), far_gap, fuse_scratch, c->d_tile_size.as<uint32_t>());
                }
                else if (l2new)   // blocks of one tile: no far tables, but the same near-table seeding as the level's other blocks
                    hipLaunchKernelGGL((match_tiles_kernel<false, MLZ_M2_NW, kM2HashBitsSmall, kL2FarBits, true>), dim3(grid), dim3(256), M2Cfg<kM2HashBitsSmall>::kLds, st,
                                       d_src, blocks, tile_block, c->d_recs.as<uint2>(), c->d_piece_cnt.as<uint32_t>(), ftab, epochs, pattern, tiles, 2u, uint32_t(fbits), far_gap, fuse_scratch, c->d_tile_size.as<uint32_t>());
                else {
                    // one launch per block class that occurs in the batch (usually one)
                    if (any_big) { if (far) MLZ_LAUNCH_M2(true, kM2HashBitsBig, any_small ? 1 : 2); else MLZ_LAUNCH_M2(false, kM2HashBitsBig, any_small ? 1 : 2); }
                    if (any_small) { if (far) MLZ_LAUNCH_M2(true, kM2HashBitsSmall, any_big ? 0 : 2, 0); else MLZ_LAUNCH_M2(false, kM2HashBitsSmall, any_big ? 0 : 2); }
                }
#undef MLZ_LAUNCH_M2
            }
            if (!fuse_scratch) {
                Timer t(c, T_ENC_SER, st);
                hipLaunchKernelGGL(serialize_pieces_kernel, dim3(tiles), dim3(256), kSerLds, st, d_src, blocks, tile_block, c->d_recs.as<uint2>(),
                                   c->d_piece_cnt.as<uint32_t>(), c->d_scratch.as<uint8_t>(), c->d_tile_size.as<uint32_t>());
            }
        }
    }
    // The layout (piece offsets, stored-or-not, header, length) rides in the gather when every block of the group has tiles and room — the common case;
    // an empty, oversize or too tightly bounded block needs encode_layout_kernel's verdicts (option 24 = 0: always the separate kernel).
    bool fold = c->fold_layout && tiles > 0 && level != MLZ_LEVEL_UNCOMPRESSED;
    for (int i = 0; fold && i < n; i++) {
        const uint64_t len = desc[i].src_len;
        if (len == 0 || len > kMaxBlockSize || desc[i].dst_cap < (with_header ? len + 2 : len)) fold = false;
    }
    if (!fold) {
        Timer t(c, T_ENC_LAYOUT, st);
        hipLaunchKernelGGL(encode_layout_kernel, dim3(n), dim3(64), 0, st, blocks, c->d_tile_size.as<uint32_t>(), c->d_tile_out.as<uint32_t>(), d_dst,
                           d_out_len, c->d_flags.as<uint32_t>(), level, with_header ? 1 : 0, sub_log);
    }
    if (tiles > 0) {
        Timer t(c, T_ENC_GATHER, st);
        if (fold)
            hipLaunchKernelGGL(encode_gather2_kernel<true>, dim3(tiles), dim3(256), 0, st, d_src, blocks, tile_block, c->d_scratch.as<uint8_t>(),
                               c->d_tile_size.as<uint32_t>(), c->d_tile_out.as<uint32_t>(), d_dst, c->d_flags.as<uint32_t>(), with_header ? 1 : 0, level, d_out_len);
        else
            hipLaunchKernelGGL(encode_gather2_kernel<false>, dim3(tiles), dim3(256), 0, st, d_src, blocks, tile_block, c->d_scratch.as<uint8_t>(),
                               c->d_tile_size.as<uint32_t>(), c->d_tile_out.as<uint32_t>(), d_dst, c->d_flags.as<uint32_t>(), with_header ? 1 : 0, level, d_out_len);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

// dec_general_kernel's role S workgroups: up to four per block (teams, GenCtl::team), at most a quarter of the device; a multiple of 4
uint32_t gen_settle_wgs(mlz_ctx* c, int n) {
    const uint32_t grid = c->gen_grid > 0 ? uint32_t(c->gen_grid) : uint32_t(c->n_cus);
    const uint32_t cap = c->gen_settle_cap > 0 ? uint32_t(c->gen_settle_cap) : grid / 4;
    return std::max<uint32_t>(4u, std::min<uint32_t>(4u * uint32_t(n), cap) & ~3u);
}

int decode_parallel(mlz_ctx* c, hipStream_t st, const uint8_t* d_src, uint8_t* d_dst, const mlz_block_desc* desc, int n, int64_t* d_out_len,
                    bool raw_body, const uint64_t* mirror) {
    uint32_t tiles = 0, segs = 0;
    int r = upload_blocks(c, st, desc, n, true, &tiles, &segs, mirror);
    if (r) return r;
    // carve the workspace
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t o_dec = 0;
    const size_t o_exit = o_dec + al(sizeof(DecBlock) * n);
    const size_t o_rexit = o_exit + al(size_t(segs) * kExitKeep * 4);
    const size_t o_entry = o_rexit + al(size_t(segs) * kSeg);
    const size_t o_sout = o_entry + al(size_t(segs) * 4);
    const size_t o_slast = o_sout + al(size_t(segs) * 4);
    const size_t o_tstart = o_slast + al(size_t(segs) * 4);
    const size_t o_rout = o_tstart + al(size_t(tiles) * sizeof(TileStart));
    const size_t reg_words = c->index_passes ? size_t(segs) * kSegThreads : 0;   // per-region records of the three-kernel index pass only
    const size_t o_rlast = o_rout + al(reg_words * 4);
    const size_t o_rentry = o_rlast + al(reg_words * 4);
    const size_t o_sntok = o_rentry + al(reg_words * 4);
    const size_t o_tpos = o_sntok + al(size_t(segs) * 4);                       // token list: a token has at least one stream byte
    const size_t o_rd = o_tpos + al(size_t(segs) * kSeg * 4);
    const size_t o_rr = o_rd + al(size_t(segs) * kSegThreads * 4);              // (per 64 tokens: indexed like the 64-byte chunks)
    const size_t o_order = o_rr + al(size_t(segs) * kSegThreads * 4);
    const size_t o_glist = o_order + al(size_t(tiles) * 4);                     // general blocks of the batch (D3d)
    const size_t o_xcnt = o_glist + al(size_t(n) * 4);                          // per-tile records of dec_general_kernel (GenTile)
    const size_t o_done = o_xcnt + al(size_t(tiles) * sizeof(GenTile));
    const size_t o_ticket = o_done + al(size_t(tiles) * 4);
    const size_t o_gen = o_ticket + 256;  // GenCtl (zeroed with the flags)
    const size_t o_sstate = o_gen + al(sizeof(GenCtl));                          // per segment: aggregate words of the index pass (the segment's, then three prefixes over its quarters; zeroed with the flags)
    const size_t o_sviol = o_sstate + al(size_t(segs) * 32);                    // ... and which level patterns its quarters' copies break (zeroed)
    const size_t o_tok16 = o_sviol + al(size_t(segs) * 4);                    // ... and its token positions inside the segment (16 bits per stream byte; not zeroed)
    const size_t total = o_tok16 + al(c->index_passes ? 0 : size_t(segs) * kSeg * 2);
    HIPCHK(c, c->d_dec.ensure(total));
    uint8_t* ws = c->d_dec.as<uint8_t>();
    DecBlock* dec = reinterpret_cast<DecBlock*>(ws + o_dec);
    uint32_t* exit_tab = reinterpret_cast<uint32_t*>(ws + o_exit);
    uint8_t* rexit_tab = reinterpret_cast<uint8_t*>(ws + o_rexit);
    uint32_t* seg_entry = reinterpret_cast<uint32_t*>(ws + o_entry);
    uint32_t* seg_out = reinterpret_cast<uint32_t*>(ws + o_sout);
    uint32_t* seg_last = reinterpret_cast<uint32_t*>(ws + o_slast);
    TileStart* tile_start = reinterpret_cast<TileStart*>(ws + o_tstart);
    uint32_t* reg_out = reinterpret_cast<uint32_t*>(ws + o_rout);
    uint32_t* reg_last = reinterpret_cast<uint32_t*>(ws + o_rlast);
    uint32_t* reg_entry = reinterpret_cast<uint32_t*>(ws + o_rentry);
    uint32_t* seg_ntok = reinterpret_cast<uint32_t*>(ws + o_sntok);
    uint32_t* tok_pos = reinterpret_cast<uint32_t*>(ws + o_tpos);
    uint32_t* round_d = reinterpret_cast<uint32_t*>(ws + o_rd);
    uint32_t* round_rep = reinterpret_cast<uint32_t*>(ws + o_rr);
    uint32_t* order = reinterpret_cast<uint32_t*>(ws + o_order);
    uint32_t* tile_done = reinterpret_cast<uint32_t*>(ws + o_done);
    uint32_t* ticket = reinterpret_cast<uint32_t*>(ws + o_ticket);
    GenCtl* gen = reinterpret_cast<GenCtl*>(ws + o_gen);
    c->last_gen = gen;
    // General blocks (streams of other encoders) go through dec_general_kernel when its workspace — 2 B of map and 1 B of pool per
    // output byte, 8 B per possible token for the external entries — is affordable (<= 64 GiB) and the current exec pass is in use.
    const size_t map_bytes = (size_t(tiles) << kTileLog) * 3;   // maps, then pools
    const size_t ext_entries = (size_t(segs) << kSegLog) + size_t(tiles) * kExtPerTile + 64 * size_t(n);
    bool jump = c->general_algo == 0 && c->decode_algo == 0 && tiles > 0;
    if (jump && c->d_idx.ensure(map_bytes + ext_entries * sizeof(ExtEnt) + 256) != hipSuccess) {
        // The device cannot hold the pass's buffers (another tenant's memory, a small part): not a reason to fail the call — general blocks
        // then stay on the exec pass's tile chain, which needs none (slow, correct).  Counted, so that a caller can see it: mlz_get_counter(ctx, 5).
        (void)hipGetLastError();
        c->gen_fallbacks++;
        jump = false;
    }
    const BlockInfo* blocks = c->d_blocks_cur().as<BlockInfo>();
    const uint32_t* tile_block = c->d_tile_block_cur().as<uint32_t>();
    const uint32_t* seg_block = c->d_seg_block_cur().as<uint32_t>();
    if (!c->dec_attrs) {
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dec_exit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kExitLds));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dec_index_a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kIndexLds));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dec_index_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kIndexCLds));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dec_index1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kIdxLds));
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dec_exec2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kExecLds));
        c->dec_attrs = true;
    }
    {
        Timer t(c, T_DEC_PARSE, st);
        // the header pass also initialises the workspace words the later passes expect (two memsets = two more launches otherwise)
        const uint32_t n_ff = segs, n_zero = uint32_t((o_tok16 - o_done) / 4);
        const uint32_t hdr_grid = std::max<uint32_t>(uint32_t(n + 63) / 64, std::min<uint32_t>((std::max(n_ff, n_zero) + 255) / 256, 256u));
        hipLaunchKernelGGL(dec_header_kernel, dim3(hdr_grid), dim3(64), 0, st, d_src, blocks, dec, n, raw_body ? 1 : 0, seg_entry, n_ff,
                           reinterpret_cast<uint32_t*>(ws + o_done), n_zero);
        if (segs)
            hipLaunchKernelGGL(dec_exit_kernel, dim3(segs), dim3(kExitThreads), kExitLds, st, d_src, blocks, seg_block, dec, exit_tab, rexit_tab);
    }
    {
        Timer t(c, T_DEC_CHAIN, st);
        hipLaunchKernelGGL(dec_chain_kernel, dim3(n), dim3(kChainThreads), 0, st, blocks, dec, exit_tab, seg_entry, n, d_src, rexit_tab);
    }
    {
        Timer t(c, T_DEC_INDEX, st);
        if (segs && !c->index_passes) {
            unsigned long long* sstate = reinterpret_cast<unsigned long long*>(ws + o_sstate);
            uint16_t* tok16 = reinterpret_cast<uint16_t*>(ws + o_tok16);
            hipLaunchKernelGGL(dec_index1_kernel, dim3(segs), dim3(kIdxThreads), kIdxLds, st, d_src, blocks, seg_block, dec, seg_entry, rexit_tab, sstate, sstate + segs, tok16);
            hipLaunchKernelGGL(dec_index2_kernel, dim3(2 * segs), dim3(kIdx2Threads), 0, st, d_src, blocks, seg_block, dec, sstate, sstate + segs, tok16, tile_start, tok_pos,
                               round_d, round_rep, ws + o_sviol);
        }
        if (c->index_passes) {   // the three-kernel form (cross-checks)
            if (segs)
                hipLaunchKernelGGL(dec_index_a_kernel, dim3(segs), dim3(kSegThreads), kIndexLds, st, d_src, blocks, seg_block, dec, seg_entry, seg_out, seg_last, rexit_tab, reg_out, reg_last, reg_entry, seg_ntok);
            hipLaunchKernelGGL(dec_index_b_kernel, dim3(n), dim3(64), 0, st, blocks, dec, seg_out, seg_last, seg_entry, seg_ntok, n);
            if (segs)
                hipLaunchKernelGGL(dec_index_c_kernel, dim3(segs), dim3(kSegThreads), kIndexCLds, st, d_src, blocks, seg_block, dec, seg_entry, seg_out, seg_last,
                                   tile_start, reg_out, reg_last, reg_entry, seg_ntok, tok_pos, round_d, round_rep,
                                   jump ? &gen->n_general : nullptr);
        }
        if (jump && segs && !c->gen_attr) {
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dec_general_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kGenLds));
            // how many of its workgroups the device holds at once (1024 threads + 132 KiB of LDS: one per CU).  Not a correctness
            // requirement — role E never waits and role S only waits for role E, whose workgroups come first in the grid —: it
            // sizes the grid so that the settling workgroups start beside the explaining ones instead of behind them.
            int per_cu = 0;
            HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(dec_general_kernel), kGenThreads, kGenLds));
            c->gen_grid = per_cu >= 1 ? c->n_cus * per_cu : 0;
            c->gen_attr = true;
        }
        if (segs && !c->index_passes)
            hipLaunchKernelGGL(dec_viol_kernel, dim3((segs + 255) / 256), dim3(256), 0, st, seg_block, ws + o_sviol, dec, jump ? &gen->n_general : nullptr, segs);
        if (tiles) {
            HIPCHK(c, c->d_gen_acc.ensure(64));
            const uint32_t reset = c->acc_call != c->call_seq ? 1u : 0u;
            c->acc_call = c->call_seq;
            hipLaunchKernelGGL(dec_schedule_kernel, dim3(1), dim3(1024), 0, st, blocks, tile_block, dec, order, tiles, uint32_t(n),
                               jump ? reinterpret_cast<uint32_t*>(ws + o_glist) : nullptr, reinterpret_cast<uint32_t*>(gen), gen_settle_wgs(c, n),
                               c->d_gen_acc.as<uint32_t>(), reset);
        }
    }
    if (c->debug_stop) { HIPCHK(c, hipGetLastError()); return 0; }
    {
        Timer t(c, T_DEC_EXEC, st);
        unsigned long long* prof = c->prof_on ? c->d_prof.as<unsigned long long>() : nullptr;
        // level-0 tiles first, by role E of the general pass, when they are few (dec_level0_kernel; option 23 = 0: off)
        uint32_t l0_grid = 0;
        if (tiles && c->level0_by_e && !c->index_passes && c->decode_algo == 0) {
            if (!c->l0_attr) {
                HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(dec_level0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kGenLds));
                c->l0_attr = true;
            }
            l0_grid = std::min<uint32_t>(uint32_t(c->n_cus), tiles);
            hipLaunchKernelGGL(dec_level0_kernel, dim3(l0_grid), dim3(kGenThreads), kGenLds, st, d_src, d_dst, blocks, dec, tok_pos, round_d, round_rep, tile_start,
                               order, tile_block, tile_done, gen);
        }
        if (tiles)
            hipLaunchKernelGGL(dec_exec2_kernel, dim3(tiles), dim3(kExecThreads), kExecLds, st, d_src, d_dst, blocks, tile_block, dec, tile_start, tok_pos,
                               round_d, round_rep, order, tile_done, ticket, tiles, prof, reinterpret_cast<const uint32_t*>(gen), l0_grid);
    }
    {
        Timer tg(c, T_DEC_GENERAL, st);   // (+ the result pass)
        if (jump && segs) {  // returns at once unless D3c flagged a general block
            if (c->gen_grid == 0) { c->err = "dec_general_kernel: the device cannot hold a workgroup"; return -MLZ_ERR_HIP; }
            // role S: one workgroup per general block, at most a quarter of the device (more blocks take turns); role E: the rest
            const uint32_t nS = gen_settle_wgs(c, n);
            const uint32_t nE = std::max<uint32_t>(1u, uint32_t(c->gen_grid) > nS ? uint32_t(c->gen_grid) - nS : 1u);
            hipLaunchKernelGGL(dec_general_kernel, dim3(nE + nS), dim3(kGenThreads), kGenLds, st, d_src, d_dst, blocks, dec, tok_pos, round_d, round_rep, tile_start,
                               reinterpret_cast<const uint32_t*>(ws + o_glist), c->d_idx.as<uint16_t>(), c->d_idx.as<uint8_t>() + (size_t(tiles) << kTileLog) * 2,
                               reinterpret_cast<ExtEnt*>(c->d_idx.as<uint8_t>() + map_bytes), reinterpret_cast<GenTile*>(ws + o_xcnt), tile_done, gen, nE, nS,
                               c->gen_spin_limit, uint32_t(c->gen_force_packed), d_out_len, n, c->debug_status);   // (+ the results: D5)
        } else
            hipLaunchKernelGGL(dec_finish_kernel, dim3((n + 63) / 64), dim3(64), 0, st, dec, d_out_len, n, c->debug_status);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

int encode_device_locked(mlz_ctx* c, hipStream_t st, int level, const uint8_t* d_src, uint8_t* d_dst, const mlz_block_desc* desc, int n,
                         int64_t* d_out_len, bool with_header, const uint64_t* mirror = nullptr) {
    if (!valid_level(level)) return -MLZ_ERR_INVALID_LEVEL;
    if (n <= 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    WorkspaceOrder order(c, st);
    c->call_seq++;
    return for_each_group(c, desc, n, false, [&](int b0, int cnt) {
        return encode_device_group(c, st, level, d_src, d_dst, desc + b0, cnt, d_out_len + b0, with_header, mirror ? mirror + b0 : nullptr);
    });
}

int decode_device_group(mlz_ctx* c, hipStream_t st, const uint8_t* d_src, uint8_t* d_dst, const mlz_block_desc* desc, int n, int64_t* d_out_len,
                        bool raw_body, const uint64_t* mirror) {
    if (c->decode_algo == 1) {
        uint32_t tiles = 0;
        int r = upload_blocks(c, st, desc, n, true, &tiles);
        if (r) return r;
        Timer t(c, T_DEC_SERIAL, st);
        hipLaunchKernelGGL(decode_serial_kernel, dim3(n), dim3(64), 0, st, d_src, d_dst, c->d_blocks_cur().as<BlockInfo>(), d_out_len, raw_body ? 1 : 0);
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    return decode_parallel(c, st, d_src, d_dst, desc, n, d_out_len, raw_body, mirror);
}

int decode_device_locked(mlz_ctx* c, hipStream_t st, const uint8_t* d_src, uint8_t* d_dst, const mlz_block_desc* desc, int n, int64_t* d_out_len,
                         bool raw_body, const uint64_t* mirror = nullptr) {
    if (n <= 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    WorkspaceOrder order(c, st);
    c->call_seq++;
    return for_each_group(c, desc, n, true, [&](int b0, int cnt) {
        return decode_device_group(c, st, d_src, d_dst, desc + b0, cnt, d_out_len + b0, raw_body, mirror ? mirror + b0 : nullptr);
    });
}

int crc_device_locked(mlz_ctx* c, hipStream_t st, const uint8_t* d_base, const mlz_block_desc* desc, int n, uint32_t* d_out) {
    if (n <= 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    WorkspaceOrder order(c, st);
    c->call_seq++;
    uint32_t tiles = 0;
    int r = upload_blocks(c, st, desc, n, false, &tiles, nullptr, nullptr, true);
    if (r) return r;
    static CrcPow pw;
    static CrcTabs tabs;
    static std::once_flag once;
    std::call_once(once, [] {
        uint32_t p = 1u << 30;  // x^1
        pw.x2n[0] = p;
        for (int k = 1; k < 32; k++) pw.x2n[k] = p = crc_mulmod(p, p);
        for (uint32_t t = 0; t < 256; t++) {
            tabs.k128[t] = crc_x2n(pw, 128ull * (255 - t), 3);                 // x^(8 * 128 (255 - t))
            tabs.ktile[t] = crc_x2n(pw, uint64_t(t) << kTileLog, 3);           // x^(8 * 32768 t)
        }
    });
    if (!c->d_crc_tabs.p) {   // per context (= per device): the two tables of constants
        HIPCHK(c, c->d_crc_tabs.ensure(sizeof(CrcTabs)));
        HIPCHK(c, hipMemcpy(c->d_crc_tabs.p, &tabs, sizeof(CrcTabs), hipMemcpyHostToDevice));
    }
    HIPCHK(c, c->d_crc_tiles.ensure(sizeof(uint32_t) * (size_t(tiles) + 1)));
    Timer t(c, T_CRC, st);
    if (tiles)
        hipLaunchKernelGGL(crc_tile_kernel, dim3(tiles), dim3(256), 0, st, d_base, c->d_blocks_cur().as<BlockInfo>(), c->d_tile_block_cur().as<uint32_t>(),
                           c->d_crc_tabs.as<CrcTabs>(), pw, c->d_crc_tiles.as<uint32_t>());
    hipLaunchKernelGGL(crc_block_kernel, dim3(n), dim3(64), 0, st, c->d_blocks_cur().as<BlockInfo>(), c->d_crc_tiles.as<uint32_t>(), c->d_crc_tabs.as<CrcTabs>(), pw,
                       d_out, n);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Device-visible address of the host range [p, p + len) when ALL of it lies in one page-locked (pinned, registered) allocation, else 0:
// such a destination is written by the kernels themselves.  (Both ends must lie in the SAME allocation: a buffer that merely starts
// in one is not ours to write through.)
uint64_t pinned_alias(const void* p, size_t len) {
    hipPointerAttribute_t at, at_end;
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeHost) { (void)hipGetLastError(); return 0; }
    if (len > 1) {
        const uint8_t* last = static_cast<const uint8_t*>(p) + len - 1;
        if (hipPointerGetAttributes(&at_end, last) != hipSuccess || at_end.type != hipMemoryTypeHost ||
            (at.devicePointer && at_end.devicePointer &&
             static_cast<const uint8_t*>(at_end.devicePointer) - static_cast<const uint8_t*>(at.devicePointer) != ptrdiff_t(len - 1))) {
            (void)hipGetLastError(); return 0;
        }
    }
    return reinterpret_cast<uint64_t>(at.devicePointer ? at.devicePointer : const_cast<void*>(p));
}

// ---- host-pointer plumbing: pack blocks into one device buffer, run, copy back ----
int ensure_stream_objects(mlz_ctx* c, size_t n_events, size_t pinned_bytes);  // mlz_stream.hip.inc

// Host-pointer batch: the blocks are cut into groups, and copy-in (stream s_in), kernels (the context's
// stream) and copy-out (s_out) of different groups overlap — the PCIe crossings (63 GB/s spec each way) cost about as
// much as the kernels, so run one after the other they halve the rate a Go caller sees.  Every copy is enqueued up
// front; the host only waits for a group's sizes (a few bytes per block) before it enqueues that group's exact-size
// copy-out, while the next groups' kernels are already queued.
// Group sizes: encode is bound by the copy-in (the kernels of the last group are what is left over when it ends: small
// groups); decode is bound by the kernels' own PCIe writes (every group drains its tail before the next starts: few groups).

int host_batch(mlz_ctx* c, bool encode, int level, int n, const uint8_t* const* src, const size_t* src_len, uint8_t* const* dst, const size_t* dst_cap,
               int64_t* out_len, bool with_header, const size_t* decoded_len /* decode_block only */) {
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<mlz_block_desc> desc(n);
    std::vector<int> gstart{0};
    size_t in_total = 0, out_total = 0, acc = 0;
    for (int i = 0; i < n; i++) {
        desc[i].src_off = in_total; desc[i].src_len = src_len[i];
        in_total += (src_len[i] + 63) & ~size_t(63);
        size_t cap = dst_cap[i];
        if (encode) cap = std::min<size_t>(cap, size_t(kMaxBlockSize) + 16);
        else if (decoded_len) cap = decoded_len[i];
        else cap = std::min<size_t>(cap, kMaxBlockSize);
        desc[i].dst_off = out_total; desc[i].dst_cap = cap;
        out_total += (cap + 63) & ~size_t(63);
        acc += std::max<size_t>(src_len[i], cap);
        if (acc >= (encode ? c->host_group_enc : c->host_group_dec) && i + 1 < n) { gstart.push_back(i + 1); acc = 0; }
    }
    gstart.push_back(n);
    const size_t ngroups = gstart.size() - 1;
    // Pinned (page-locked, device-visible) destinations are written by the kernels themselves: the gather pass of the
    // encoder stores the block straight into the caller's buffer and the decoder's passes store every tile there as
    // well as in HBM, so the batch has no copy-out stage at all (on this ROCm build the copy-out ran as shader copies
    // that queued up behind every kernel of the batch).  Pageable destinations go through the copy-out below.
    std::vector<uint64_t> mirror(size_t(n), 0);
    bool use_mirror = encode || (c->decode_algo == 0 && c->general_algo == 0);
    for (int i = 0; use_mirror && i < n; i++) {
        mirror[size_t(i)] = pinned_alias(dst[i], dst_cap[i]);
        if (!mirror[size_t(i)]) use_mirror = false;
    }
    const uint64_t* mir = use_mirror ? mirror.data() : nullptr;
    HIPCHK(c, c->d_in.ensure(in_total + 64));
    HIPCHK(c, c->d_out.ensure(out_total + 64));
    HIPCHK(c, c->d_len.ensure(sizeof(int64_t) * n));
    int r = ensure_stream_objects(c, 3 * ngroups, sizeof(int64_t) * size_t(n));
    if (r) return r;
    int64_t* h_len = static_cast<int64_t*>(c->pinned2);
    hipStream_t st = c->stream;
    for (size_t g = 0; g < ngroups; g++) {  // every copy-in, back to back on the copy-in stream
        for (int i = gstart[g]; i < gstart[g + 1]; i++)
            if (src_len[i]) HIPCHK(c, hipMemcpyAsync(c->d_in.as<uint8_t>() + desc[i].src_off, src[i], src_len[i], hipMemcpyHostToDevice, c->s_in));
        HIPCHK(c, hipEventRecord(c->evpool[3 * g], c->s_in));
    }
    for (size_t g = 0; g < ngroups; g++) {  // kernels of a group once its bytes are in; its sizes go out behind them
        const int b0 = gstart[g], cnt = gstart[g + 1] - gstart[g];
        HIPCHK(c, hipStreamWaitEvent(st, c->evpool[3 * g], 0));
        r = encode ? encode_device_locked(c, st, level, c->d_in.as<uint8_t>(), c->d_out.as<uint8_t>(), desc.data() + b0, cnt, c->d_len.as<int64_t>() + b0, with_header,
                                          mir ? mir + b0 : nullptr)
                   : decode_device_locked(c, st, c->d_in.as<uint8_t>(), c->d_out.as<uint8_t>(), desc.data() + b0, cnt, c->d_len.as<int64_t>() + b0, !with_header,
                                          mir ? mir + b0 : nullptr);
        if (r) { (void)hipStreamSynchronize(c->s_in); (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(c->s_out); return r; }
        HIPCHK(c, hipEventRecord(c->evpool[3 * g + 1], st));
        HIPCHK(c, hipStreamWaitEvent(c->s_out, c->evpool[3 * g + 1], 0));
        HIPCHK(c, hipMemcpyAsync(h_len + b0, c->d_len.as<int64_t>() + b0, sizeof(int64_t) * size_t(cnt), hipMemcpyDeviceToHost, c->s_out));
        HIPCHK(c, hipEventRecord(c->evpool[3 * g + 2], c->s_out));
    }
    for (size_t g = 0; g < ngroups; g++) {  // exact-size copy-out as soon as a group's sizes are known
        HIPCHK(c, hipEventSynchronize(c->evpool[3 * g + 2]));
        for (int i = gstart[g]; i < gstart[g + 1]; i++) {
            out_len[i] = h_len[i];
            if (out_len[i] > 0) {
                if (size_t(out_len[i]) > dst_cap[i]) { out_len[i] = -MLZ_ERR_DST_TOO_SMALL; continue; }
                if (!use_mirror) HIPCHK(c, hipMemcpyAsync(dst[i], c->d_out.as<uint8_t>() + desc[i].dst_off, size_t(out_len[i]), hipMemcpyDeviceToHost, c->s_out));
            }
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->s_out));
    HIPCHK(c, hipStreamSynchronize(st));
    return 0;
}

// Single-block calls (mlz_encode / mlz_encode_block / mlz_decode / mlz_decode_block) from concurrent threads are
// combined into batched launches: the reference's Writer and Reader call their block codec from one goroutine
// per block (writer.go:501-560, reader.go:830-859), and one 8 MiB block alone cannot fill the device (a tile is a
// serial chain: one block takes about as long as twelve).  Every caller queues its request; whoever finds no
// batch in flight becomes the leader, takes every queued request of the same kind (up to kCombineMax) and runs
// them as ONE host batch while later arrivals queue up behind it; the others sleep until their request is done.
// No timer, no helper thread: a lone caller runs at once, a busy queue batches itself.
constexpr size_t kCombineMax = 64;
void submit_single(mlz_ctx* c, SingleReq& rq) {
    std::unique_lock<std::mutex> lk(c->q_mu);
    c->q_pending.push_back(&rq);
    while (!rq.done) {
        if (c->q_leader) { c->q_cv.wait(lk); continue; }
        c->q_leader = true;
        std::vector<SingleReq*> group;
        {
            const SingleReq& head = *c->q_pending.front();
            std::vector<SingleReq*> rest;
            for (SingleReq* r : c->q_pending) {
                if (group.size() < kCombineMax && r->same_kind(head)) group.push_back(r); else rest.push_back(r);
            }
            c->q_pending.swap(rest);
        }
        lk.unlock();
        const int n = int(group.size());
        std::vector<const uint8_t*> src(n); std::vector<size_t> slen(n), cap(n), dlen(n);
        std::vector<uint8_t*> dst(n); std::vector<int64_t> out(n, 0);
        for (int i = 0; i < n; i++) { src[i] = group[i]->src; slen[i] = group[i]->n; dst[i] = group[i]->dst; cap[i] = group[i]->cap; dlen[i] = group[i]->dlen; }
        const SingleReq& k = *group[0];
        const int r = host_batch(c, k.encode, k.level, n, src.data(), slen.data(), dst.data(), cap.data(), out.data(), k.with_header,
                                 k.has_dlen ? dlen.data() : nullptr);
        lk.lock();
        for (int i = 0; i < n; i++) { group[i]->rc = r; group[i]->out = out[i]; group[i]->done = true; }
        c->q_batches++; c->q_requests += uint64_t(n);
        c->q_leader = false;
        c->q_cv.notify_all();
    }
}

// The contexts a host-pointer call is dealt to: the context itself, or the per-device contexts behind it (mlz_init_devices).
struct Workers {
    mlz_ctx* one;
    mlz_ctx* const* list;
    size_t n;
    explicit Workers(mlz_ctx* c) : one(c), list(&one), n(1) {
        if (!c->kids.empty()) { list = c->kids.data(); n = c->kids.size(); }
    }
    mlz_ctx* next() const { return n == 1 ? list[0] : list[one->rr.fetch_add(1, std::memory_order_relaxed) % n]; }   // single-block calls: the kids in turn
};

// The kid whose device holds `p` (a device-resident call on a several-device context), or the context itself.
mlz_ctx* owner_of(mlz_ctx* c, const void* p) {
    if (c->kids.empty()) return c;
    hipPointerAttribute_t at;
    if (p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice) {
        for (mlz_ctx* k : c->kids) if (k->device == at.device) return k;
        return nullptr;
    }
    (void)hipGetLastError();
    return nullptr;
}

// A host-pointer batch over several devices: contiguous block ranges of about equal weight, one thread and one PCIe link per device,
// every range through host_batch on its own context (the reference fans the blocks of a stream to goroutines, writer.go:501-560,
// reader.go:830-859; results land at the caller's dst[i] whatever device produced them, so there is nothing to gather).
int multi_host_batch(mlz_ctx* c, bool encode, int level, int n, const uint8_t* const* src, const size_t* src_len, uint8_t* const* dst, const size_t* dst_cap,
                     int64_t* out_len) {
    Workers w(c);
    if (w.n == 1) return host_batch(w.list[0], encode, level, n, src, src_len, dst, dst_cap, out_len, true, nullptr);
    std::vector<uint64_t> pre(size_t(n) + 1, 0);
    for (int i = 0; i < n; i++) {
        uint64_t wt = src_len[i];
        if (!encode) { const int64_t dl = mlz_decoded_len(src[i], src_len[i]); if (dl > 0) wt += uint64_t(dl); }
        pre[size_t(i) + 1] = pre[size_t(i)] + wt + 4096;
    }
    const uint64_t total = pre[size_t(n)];
    size_t k = std::min<size_t>(w.n, size_t(n));
    if (total < (uint64_t(4) << 20)) k = 1;    // not worth a second device
    if (k == 1) return host_batch(w.next(), encode, level, n, src, src_len, dst, dst_cap, out_len, true, nullptr);
    std::vector<int> cut(k + 1, n);
    cut[0] = 0;
    {
        size_t j = 1;
        for (int i = 0; i < n && j < k; i++)
            while (j < k && pre[size_t(i) + 1] * k >= total * j) cut[j++] = i + 1;
    }
    std::vector<int> rcs(k, 0);
    auto work = [&](size_t j) {
        const int b0 = cut[j], cnt = cut[j + 1] - cut[j];
        if (cnt > 0) rcs[j] = host_batch(w.list[j], encode, level, cnt, src + b0, src_len + b0, dst + b0, dst_cap + b0, out_len + b0, true, nullptr);
    };
    std::vector<std::thread> th;
    for (size_t j = 1; j < k; j++) th.emplace_back(work, j);
    work(0);
    for (std::thread& t : th) t.join();
    for (size_t j = 0; j < k; j++)
        if (rcs[j]) { std::lock_guard<std::mutex> lk(c->mu); c->err = w.list[j]->err; return rcs[j]; }
    return 0;
}

}  // namespace

extern "C" {

int mlz_version(void) { return 2; }

int mlz_init(int device, mlz_ctx** out) {
    if (!out) return -MLZ_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return -MLZ_ERR_HIP;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return -MLZ_ERR_HIP; }
    if (device >= count) return -MLZ_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return -MLZ_ERR_HIP;
    mlz_ctx* c = new mlz_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->dev_name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
        c->n_cus = prop.multiProcessorCount;
    }
    if (c->n_cus <= 0) c->n_cus = 64;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return -MLZ_ERR_HIP; }
    for (int k = 0; k < 2; k++)
        if (hipEventCreateWithFlags(&c->upload_done_k[k], hipEventDisableTiming) != hipSuccess) { delete c; return -MLZ_ERR_HIP; }
    if (hipEventCreateWithFlags(&c->ws_done, hipEventDisableTiming) != hipSuccess) { delete c; return -MLZ_ERR_HIP; }
    *out = c;
    return 0;
}

int mlz_init_devices(const int* devices, int n_devices, mlz_ctx** out) {
    if (!out || (devices && n_devices <= 0)) return -MLZ_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return -MLZ_ERR_HIP;
    std::vector<int> devs;
    if (devices) devs.assign(devices, devices + n_devices);
    else for (int d = 0; d < count && (n_devices <= 0 || d < n_devices); d++) devs.push_back(d);   // every visible device (or the first n_devices)
    mlz_ctx* p = new mlz_ctx();
    for (int d : devs) {
        mlz_ctx* k = nullptr;
        const int r = (d < 0 || d >= count) ? -MLZ_ERR_ARG : mlz_init(d, &k);
        if (r) { for (mlz_ctx* q : p->kids) mlz_destroy(q); delete p; return r; }
        p->kids.push_back(k);
    }
    p->device = p->kids[0]->device;
    p->n_cus = p->kids[0]->n_cus;
    p->dev_name = std::to_string(p->kids.size()) + " x " + p->kids[0]->dev_name;
    *out = p;
    return 0;
}

int mlz_device_count(mlz_ctx* c) { return !c ? -MLZ_ERR_ARG : c->kids.empty() ? 1 : int(c->kids.size()); }

mlz_ctx* mlz_device_ctx(mlz_ctx* c, int i) {
    if (!c || i < 0) return nullptr;
    if (c->kids.empty()) return i == 0 ? c : nullptr;
    return size_t(i) < c->kids.size() ? c->kids[size_t(i)] : nullptr;
}

void mlz_destroy(mlz_ctx* c) {
    if (!c) return;
    if (!c->kids.empty()) {   // a several-device context: its kids hold everything
        for (mlz_ctx* k : c->kids) mlz_destroy(k);
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (DevBuf* b : {&c->d_place, &c->d_crc, &c->d_crc_tabs, &c->d_crc_tiles, &c->d_prof, &c->d_blocks_k[0], &c->d_blocks_k[1], &c->d_tile_block_k[0], &c->d_tile_block_k[1], &c->d_seg_block_k[0], &c->d_seg_block_k[1], &c->d_scratch, &c->d_tile_size, &c->d_tile_out, &c->d_flags, &c->d_far, &c->d_farbin, &c->d_recs, &c->d_piece_cnt, &c->d_dec, &c->d_idx, &c->d_gen_acc, &c->d_in, &c->d_out, &c->d_len})
        b->release();
    for (int k = 0; k < 2; k++) if (c->pinned_k[k]) (void)hipHostFree(c->pinned_k[k]);
    if (c->pinned2) (void)hipHostFree(c->pinned2);
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    for (hipEvent_t e : c->evpool) (void)hipEventDestroy(e);
    for (int i = 0; i < T_COUNT; i++)
        for (int k = 0; k < mlz_ctx::kTimerRing; k++)
            for (int e = 0; e < 2; e++)
                if (c->evr[i][k][e]) (void)hipEventDestroy(c->evr[i][k][e]);
    for (int k = 0; k < 2; k++) if (c->upload_done_k[k]) (void)hipEventDestroy(c->upload_done_k[k]);
    if (c->ws_done) (void)hipEventDestroy(c->ws_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* mlz_last_error(mlz_ctx* c) {
    if (!c) return "null context";
    if (c->err.empty()) for (mlz_ctx* k : c->kids) if (!k->err.empty()) return k->err.c_str();
    return c->err.c_str();
}

int mlz_device_name(mlz_ctx* c, char* buf, size_t cap) {
    if (!c || !buf || cap == 0) return -MLZ_ERR_ARG;
    std::snprintf(buf, cap, "%s", c->dev_name.c_str());
    return 0;
}

int64_t mlz_max_encoded_len(uint64_t n) {
    if (n > kMaxBlockSize) return -1;
    if (n == 0) return 1;
    return int64_t(n) + 2;
}

int64_t mlz_decoded_len(const uint8_t* src, size_t n) {
    uint64_t body, dlen; int lits;
    int e = parse_block_header(src, n, &body, &dlen, &lits);
    if (e == 3) {  // non-MinLZ block: DecodedLen still reports the Snappy varint (decode.go:132-136)
        uint64_t x = 0; uint32_t shift = 0;
        for (size_t i = 0; i < n && i < 10; i++) {
            uint8_t b = src[i];
            if (b < 0x80) { x |= uint64_t(b) << shift; return x > 0xffffffffull ? -MLZ_ERR_CORRUPT : int64_t(x); }
            x |= uint64_t(b & 0x7f) << shift; shift += 7;
        }
        return -MLZ_ERR_CORRUPT;
    }
    if (e) return -e;
    return int64_t(dlen);
}

int64_t mlz_encode(mlz_ctx* c, int level, const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap) {
    if (!c || (!src && n) || !dst) return -MLZ_ERR_ARG;
    if (n > kMaxBlockSize) return -MLZ_ERR_TOO_LARGE;
    if (int64_t(dst_cap) < mlz_max_encoded_len(n)) return -MLZ_ERR_DST_TOO_SMALL;
    if (!valid_level(level)) return -MLZ_ERR_INVALID_LEVEL;
    SingleReq rq{true, true, false, level, src, n, dst, dst_cap, 0};
    submit_single(Workers(c).next(), rq);
    return rq.rc ? rq.rc : rq.out;
}

int64_t mlz_encode_block(mlz_ctx* c, int level, const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap) {
    if (!c || (!src && n) || !dst) return -MLZ_ERR_ARG;
    if (n > kMaxBlockSize) return -MLZ_ERR_TOO_LARGE;
    if (dst_cap < n) return -MLZ_ERR_DST_TOO_SMALL;
    if (!valid_level(level)) return -MLZ_ERR_INVALID_LEVEL;
    SingleReq rq{true, false, false, level, src, n, dst, dst_cap, 0};
    submit_single(Workers(c).next(), rq);
    return rq.rc ? rq.rc : rq.out;
}

int64_t mlz_decode(mlz_ctx* c, const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap) {
    if (!c || !src || (!dst && dst_cap)) return -MLZ_ERR_ARG;
    int64_t dl = mlz_decoded_len(src, n);
    uint64_t body, dlen; int lits;
    int e = parse_block_header(src, n, &body, &dlen, &lits);
    if (e) return -e;
    if (dlen > dst_cap) return -MLZ_ERR_DST_TOO_SMALL;
    (void)dl;
    if (dlen == 0) return 0;
    SingleReq rq{false, true, false, 0, src, n, dst, dst_cap, 0};
    submit_single(Workers(c).next(), rq);
    return rq.rc ? rq.rc : rq.out;
}

int mlz_decode_block(mlz_ctx* c, const uint8_t* src, size_t clen, uint8_t* dst, size_t n) {
    if (!c || (!src && clen) || (!dst && n)) return -MLZ_ERR_ARG;
    if (n > kMaxBlockSize) return -MLZ_ERR_TOO_LARGE;
    if (n == 0) return clen == 0 ? 0 : 1;
    SingleReq rq{false, false, true, 0, src, clen, dst, n, n};
    submit_single(Workers(c).next(), rq);
    if (rq.rc) return rq.rc;
    if (rq.out == int64_t(n)) return 0;
    // Only a DEVICE failure (-MLZ_ERR_HIP: a launch that failed, a bounded wait that gave up) is passed on as < 0, the shim's cue
    // to run its CPU decoder (go/minlz_hip.go: r < 0, counted in hipFallbacks).  Every other per-block code (corrupt, too large,
    // unsupported, destination too small) is a verdict on the INPUT: minLZDecode's contract has one value for those, 1 = corrupt
    // (decode.go:26), and a healthy device never makes the caller decode a block twice.
    if (rq.out == -MLZ_ERR_HIP) return -MLZ_ERR_HIP;
    return 1;
}

int mlz_encode_batch(mlz_ctx* c, int level, int n, const uint8_t* const* src, const size_t* src_len, uint8_t* const* dst, const size_t* dst_cap,
                     int64_t* out_len) {
    if (!c || n < 0) return -MLZ_ERR_ARG;
    if (n == 0) return 0;
    if (!valid_level(level)) return -MLZ_ERR_INVALID_LEVEL;
    return multi_host_batch(c, true, level, n, src, src_len, dst, dst_cap, out_len);
}

int mlz_decode_batch(mlz_ctx* c, int n, const uint8_t* const* src, const size_t* src_len, uint8_t* const* dst, const size_t* dst_cap, int64_t* out_len) {
    if (!c || n < 0) return -MLZ_ERR_ARG;
    if (n == 0) return 0;
    return multi_host_batch(c, false, 0, n, src, src_len, dst, dst_cap, out_len);
}

int mlz_encode_batch_device(mlz_ctx* c, void* stream, int level, const uint8_t* d_src, uint8_t* d_dst, const mlz_block_desc* desc, int n,
                            int64_t* d_out_len) {
    if (!c || !desc || n < 0 || !d_out_len) return -MLZ_ERR_ARG;
    if (!(c = owner_of(c, d_src))) return -MLZ_ERR_ARG;   // (several devices: the one that holds the buffers)
    std::lock_guard<std::mutex> lk(c->mu);
    return encode_device_locked(c, static_cast<hipStream_t>(stream), level, d_src, d_dst, desc, n, d_out_len, true);
}

int mlz_decode_batch_device(mlz_ctx* c, void* stream, const uint8_t* d_src, uint8_t* d_dst, const mlz_block_desc* desc, int n, int64_t* d_out_len) {
    if (!c || !desc || n < 0 || !d_out_len) return -MLZ_ERR_ARG;
    if (!(c = owner_of(c, d_src))) return -MLZ_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return decode_device_locked(c, static_cast<hipStream_t>(stream), d_src, d_dst, desc, n, d_out_len, false);
}

int mlz_set_option(mlz_ctx* c, int opt, int64_t value) {
    if (!c) return -MLZ_ERR_ARG;
    if (!c->kids.empty()) {   // every device alike (the debug read-backs 5 / 7: the first device's)
        if (opt == 5 || opt == 7) return mlz_set_option(c->kids[0], opt, value);
        for (mlz_ctx* k : c->kids) { const int r = mlz_set_option(k, opt, value); if (r) return r; }
        return 0;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    switch (opt) {
    case MLZ_OPT_DECODE_ALGO: c->decode_algo = int(value); return 0;
    case MLZ_OPT_ENCODE_FAR: c->encode_far = int(value); return 0;
    case 8: c->general_algo = int(value); return 0;  // 0 = pointer-jumping pass for general blocks (default), 1 = tile chain
    case 12: c->timer_mask = uint32_t(value); return 0;  // which timers record events (bit = index of mlz_timer_name)
    case 10: c->host_group_enc = size_t(value > 0 ? value : 1) << 20; return 0;  // tuning: MiB per group of a host-pointer encode batch
    case 11: c->host_group_dec = size_t(value > 0 ? value : 1) << 20; return 0;  // ... of a decode batch
    case 15: c->index_passes = int(value); return 0;  // decode: 1 = the index pass as the three kernels of rounds 2-3 (dec_index_a / _b / _c: cross-checks), 0 = dec_index1 / dec_index2 / dec_viol (default)
    case 16: c->debug_stop = int(value); return 0;  // debug: decode_batch_device returns after the index pass (out_len is not written)
    case MLZ_OPT_DEVICE_GROUP: c->device_group = size_t(value > 0 ? value : 1) << 20; return 0;  // MiB of uncompressed data per internal group of a device batch
    case 18: c->far_slices_l2 = int(value); return 0;  // debug / cross-check: LevelBalanced's far tables by the slice kernel of round 4
    case MLZ_OPT_L2_GAP: if (value < 1 || value > 16) return -MLZ_ERR_ARG; c->l2_gap = int(value); return 0;
    case 24: c->fold_layout = int(value); return 0;  // encode: 1 (default) = layout inside the gather kernel where possible, 0 = always the separate layout kernel (cross-checks)
    case 23: c->level0_by_e = int(value); return 0;  // decode: 1 (default) = level-0 tiles, when no more than CUs, by dec_level0_kernel before the exec pass; 0 = by the exec pass
    case 21: c->fuse_ser = int(value); return 0;  // encode: 1 (default) = the match kernel serializes its pieces itself, 0 = the separate serializer kernel of rounds 2-5 (cross-checks)
    case 20: c->gen_settle_cap = int(value); return 0;  // tuning: role S workgroups of the general pass at most
    case 14: c->l2_free = int(value); return 0;  // LevelBalanced: 1 = no tile levels (ratio of the reference's L2 and better; blocks decode as general blocks)
    case 13: c->gen_force_packed = int(value); return 0;  // tests: general blocks settle through the byte-packed pool (fallback path of dec_general_kernel)
    case 9: c->gen_spin_limit = value > 0 ? uint32_t(value) : 1u; return 0;  // grid-barrier patience of the general-block pass, in polls (tests)
    case 3: c->debug_status = int(value); return 0;  // debug: report failure sites in the error code
    case 4: {  // debug: per-phase cycle counters (16 x u64: 0-7 encode, 8-15 decode)
        c->prof_on = value != 0;
        if (c->prof_on) { if (c->d_prof.ensure(kProfBytes) != hipSuccess) return -MLZ_ERR_HIP; if (hipMemset(c->d_prof.p, 0, kProfBytes) != hipSuccess) return -MLZ_ERR_HIP; }
        return 0;
    }
    case 5: {  // debug: read the counters back into a host buffer whose address is `value`
        if (!c->d_prof.p) return -MLZ_ERR_ARG;
        if (hipDeviceSynchronize() != hipSuccess) return -MLZ_ERR_HIP;
        if (hipMemcpy(reinterpret_cast<void*>(value), c->d_prof.p, 128, hipMemcpyDeviceToHost) != hipSuccess) return -MLZ_ERR_HIP;
        return 0;
    }
    case 7: {  // debug: read the exec pass's per-tile timeline (kProfTiles x 4 u64, 100 MHz clock) into a host buffer
        if (!c->d_prof.p) return -MLZ_ERR_ARG;
        if (hipDeviceSynchronize() != hipSuccess) return -MLZ_ERR_HIP;
        if (hipMemcpy(reinterpret_cast<void*>(value), c->d_prof.as<uint8_t>() + 256, kProfBytes - 256, hipMemcpyDeviceToHost) != hipSuccess) return -MLZ_ERR_HIP;
        return 0;
    }
    case MLZ_TIMER_ENABLE:
        if (value < 0 || value > 2) return -MLZ_ERR_ARG;
        if (value != 0) {
            for (int i = 0; i < T_COUNT; i++)
                for (int k = 0; k < mlz_ctx::kTimerRing; k++)
                    for (int e = 0; e < 2; e++)
                        if (!c->evr[i][k][e]) HIPCHK(c, hipEventCreate(&c->evr[i][k][e]));
        }
        c->timing = int(value);
        for (int i = 0; i < T_COUNT; i++) {   // (events recorded under the old setting are dropped unread)
            c->ev_cnt[i] = c->ev_res[i] = 0; c->ev_calls[i] = 0; c->ev_last_call[i] = ~uint64_t(0); c->acc_ms[i] = 0;
        }
        return 0;
    default: return -MLZ_ERR_ARG;
    }
}

int mlz_release_stream(mlz_ctx* c, void* stream) {
    if (!c) return -MLZ_ERR_ARG;
    if (!c->kids.empty()) {
        for (mlz_ctx* k : c->kids) { const int r = mlz_release_stream(k, stream); if (r) return r; }
        return 0;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->ws_used && !c->ws_recorded && c->ws_stream == static_cast<hipStream_t>(stream)) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipEventRecord(c->ws_done, c->ws_stream));   // covers everything the context put on that stream
        c->ws_recorded = true;
    }
    return 0;
}

int64_t mlz_get_counter(mlz_ctx* c, int which) {
    if (!c) return -MLZ_ERR_ARG;
    if (!c->kids.empty()) {   // sums over the devices (6, a team size: the largest)
        int64_t acc = 0;
        for (mlz_ctx* k : c->kids) {
            const int64_t v = mlz_get_counter(k, which);
            if (v < 0) return v;
            acc = which == 6 ? std::max(acc, v) : acc + v;
        }
        return acc;
    }
    if (which == 2 || which == 6) {
        // 2: blocks of the last decode call that fit no level pattern; 6: workgroups per block (1, 2 or 4) its general pass settled with (0 = no general block).
        // Both over ALL internal groups of the call (sum / maximum, kept by dec_schedule_kernel); waits for the device.
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->d_gen_acc.p) return 0;
        uint32_t v[2] = {0, 0};
        if (hipSetDevice(c->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(v, c->d_gen_acc.p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -MLZ_ERR_HIP;
        return int64_t(which == 2 ? v[0] : (v[0] ? v[1] : 0));
    }
    if (which == 3 || which == 4) {  // device workspace this context holds: 3 = encode side, 4 = decode side (grow-only buffers: the high-water mark of the calls so far)
        std::lock_guard<std::mutex> lk(c->mu);
        size_t e = 0, d = 0;
        for (const DevBuf* b : {&c->d_scratch, &c->d_tile_size, &c->d_tile_out, &c->d_flags, &c->d_far, &c->d_recs, &c->d_piece_cnt, &c->d_farbin}) e += b->cap;
        for (const DevBuf* b : {&c->d_dec, &c->d_idx}) d += b->cap;
        return int64_t(which == 3 ? e : d);
    }
    if (which == 5) { std::lock_guard<std::mutex> lk(c->mu); return int64_t(c->gen_fallbacks); }
    std::lock_guard<std::mutex> lk(c->q_mu);
    return which == 0 ? int64_t(c->q_batches) : which == 1 ? int64_t(c->q_requests) : -MLZ_ERR_ARG;
}

int mlz_get_timers(mlz_ctx* c, float* ms, int cap) {
    if (!c || !ms) return -MLZ_ERR_ARG;
    if (!c->kids.empty()) {   // the devices work side by side: the slowest one's time per kernel family
        int n = 0;
        std::vector<float> t(size_t(std::max(cap, 0)));
        for (int i = 0; i < cap; i++) ms[i] = -1.f;
        for (mlz_ctx* k : c->kids) {
            n = mlz_get_timers(k, t.data(), cap);
            if (n < 0) return n;
            for (int i = 0; i < n; i++) ms[i] = std::max(ms[i], t[size_t(i)]);
        }
        return n;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    int n = std::min<int>(cap, T_COUNT);
    for (int i = 0; i < n; i++) {
        ms[i] = -1.f;
        while (c->ev_res[i] < c->ev_cnt[i]) c->resolve_one(i);
        if (c->ev_calls[i]) ms[i] = float(c->acc_ms[i] / double(c->ev_calls[i]));
    }
    return n;
}

const char* mlz_timer_name(int idx) { return idx >= 0 && idx < T_COUNT ? kTimerNames[idx] : ""; }

int mlz_crc_batch_device(mlz_ctx* c, void* stream, const uint8_t* d_base, const mlz_block_desc* desc, int n, uint32_t* d_out) {
    if (!c || !desc || n < 0 || !d_out) return -MLZ_ERR_ARG;
    if (!(c = owner_of(c, d_base))) return -MLZ_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return crc_device_locked(c, static_cast<hipStream_t>(stream), d_base, desc, n, d_out);
}

int64_t mlz_crc(mlz_ctx* c, const uint8_t* src, size_t n) {
    if (!c || (!src && n)) return -MLZ_ERR_ARG;
    c = Workers(c).next();
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, c->d_in.ensure(n + 64));
    HIPCHK(c, c->d_crc.ensure(64));
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_in.p, src, n, hipMemcpyHostToDevice, c->stream));
    mlz_block_desc d{0, n, 0, 0};
    int r = crc_device_locked(c, c->stream, c->d_in.as<uint8_t>(), &d, 1, c->d_crc.as<uint32_t>());
    if (r) return r;
    uint32_t v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, c->d_crc.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return int64_t(v);
}

#ifdef MLZ_M2_PROF
// debug build only (tools/m2prof.py): per-phase cycle sums of match_tiles_kernel; reset after reading
int mlz_debug_m2prof(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return -MLZ_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mlz::g_m2prof), sizeof(unsigned long long) * 16) != hipSuccess) return -MLZ_ERR_HIP;
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mlz::g_m2prof), z, sizeof(z)) != hipSuccess) return -MLZ_ERR_HIP;
    return 0;
}
#endif

#ifdef MLZ_IDX_PROF
// debug build only (tools/idxprof.py): per-phase time sums of dec_index_kernel; reset after reading
int mlz_debug_idxprof(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return -MLZ_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mlz::g_idxprof), sizeof(unsigned long long) * 16) != hipSuccess) return -MLZ_ERR_HIP;
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mlz::g_idxprof), z, sizeof(z)) != hipSuccess) return -MLZ_ERR_HIP;
    return 0;
}
#endif

}  // extern "C"

#include "mlz_stream.hip.inc"
