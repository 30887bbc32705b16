// mlz_kernels.h — device-side data structures shared between the kernels and the host API.
#pragma once
// Phase cycle counters / per-tile timelines inside the kernels (debug options 4, 5, 7) are compiled in only
// with -DMLZ_PROFILE=1 (tools/exp_build.sh); in the product build every hook folds away.
#ifndef MLZ_PROFILE
#define MLZ_PROFILE 0
#endif
#include <stdint.h>

namespace mlz {

// Tile = the unit of intra-block parallelism (DESIGN.md "Tiles").  Each tile of a block is
// encoded by one wavefront into an independent token sub-stream and decoded by one wavefront.
#ifndef MLZ_TILE_LOG
#define MLZ_TILE_LOG 15
#endif
constexpr int kTileLog = MLZ_TILE_LOG;
constexpr uint32_t kTile = 1u << kTileLog;           // 32 KiB of uncompressed data

struct BlockInfo {
    uint64_t src_off, src_len, dst_off, dst_cap;
    uint32_t first_tile;  // index of this block's first tile in the batch-wide tile arrays
    uint32_t n_tiles;
    uint32_t first_seg;   // decode: index of this block's first token-stream segment
    uint32_t n_segs;
    uint64_t mirror;      // host-pointer calls with pinned destinations: device-visible address of the caller's buffer, written by
                          // the kernels themselves (encode: instead of dst; decode: in addition to it); 0 = none
};

}  // namespace mlz
