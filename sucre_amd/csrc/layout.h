// Workspace layout of one restoration (private to libsucre_hip.so).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace sucre {

constexpr int kTile = 16;            // tile edge in pixels
constexpr int kTilePx = 256;         // pixels per tile = one 64-lane wave x 4 pixels per lane
constexpr int kChunkZ = 1024;        // bytes: 256 float32 ranges
constexpr int kChunkRGB = 768;       // bytes: planar R[256] G[256] B[256] uint8
constexpr int kChunk = kChunkZ + kChunkRGB;  // 1792 bytes per (tile, view)
// Compact observation format (SUCRE_OBS_U16MM, BASELINE config 5): ranges as uint16 millimetres (0 = empty slot),
// 5 bytes per observation.  Only the compact store (what the fit streams) uses it; the dense store stays float32.
constexpr int kChunkZ16 = 512;       // bytes: 256 uint16 ranges
constexpr int kChunk16 = kChunkZ16 + kChunkRGB;  // 1280 bytes per (sorted tile, level)
constexpr float kMmPerM = 1000.0f, kMPerMm = 0.001f;
__host__ __device__ constexpr int chunk_bytes(int fmt) { return fmt ? kChunk16 : kChunk; }
constexpr int kExtChunk = 3 * kChunkZ;  // light model: cP.x, cP.y, cP.z planes of a chunk (extension workspace)
constexpr int kNumSums = 10;         // sB[3], sGZ[3], sBeta[3], cost
constexpr int kSumsPad = 12;
constexpr int kGroup = 32;           // tiles per reduction group (two-level last-arriver reduction)
constexpr int kTicketStride = 16;    // uint32 words between tickets: every counter on its own 64-byte line
constexpr int kFitGrid = 1536;       // persistent fit workgroups (6 per CU x 256 CUs); a constant, so the reduction
                                     // order -- hence every result bit -- does not depend on the device
constexpr int kMaxViews = 4096;

struct Layout {
    int H, W, n_views, tiles_x, tiles_y, n_tiles;
    size_t off_obs;         // uint8  chunks of kChunk bytes, chunk(tile, k) at tile*obs_tile_stride + k*obs_view_stride
    size_t obs_tile_stride, obs_view_stride;
    size_t off_cnt;         // uint16 [n_tiles][n_views]   matches of view k inside the tile
    size_t off_comp;        // uint8  compact store: chunk (sorted tile, level) at tile_off[tile] + level*kChunk
    size_t off_pcount;      // uint16 [n_tiles*256]        observations of every pixel over the kept views
    size_t off_pmask;       // uint64 [n_tiles*256][mask_words]  which views observe the pixel (bit k = view k)
    int mask_words;
    size_t off_blockhist;   // uint32 [256 bins][n_tiles]  counting-sort histograms (bin-major), scanned in place
    size_t off_bin_totals;  // uint32 [256] totals, [256] bin bases
    size_t off_perm;        // uint32 [n_tiles*256]        sorted slot -> dense slot (tile*256 + slot)
    size_t off_invperm;     // uint32 [n_tiles*256]        dense slot  -> sorted slot
    size_t off_levels;      // uint32 [n_tiles]            chunks (levels) of every sorted tile (= largest pixel count in it)
    size_t off_full;        // uint32 [n_tiles]            levels with all 256 slots occupied (= smallest pixel count)
    size_t off_tile_off;    // uint64 [n_tiles]            byte offset of a sorted tile's first chunk in the compact store
    size_t off_total_chunks;// uint64 [1], then uint32 [1]: observation format of the compact store (SUCRE_OBS_*)
    size_t off_view_count;  // uint64 [n_views]
    size_t off_view_keep;   // uint32 [n_views]
    size_t off_n_obs;       // uint64 [1]
    size_t off_n_obs_total; // uint64 [1]
    size_t off_params;      // float  [9] params, [9] exp_avg, [9] exp_avg_sq
    size_t off_sums;        // double [kSumsPad]
    size_t off_ticket;      // uint32 [(1 + n_groups) * kTicketStride]  arrival counters: [0] = groups done, [1+g] = tiles of group g done
    size_t off_gpartials;   // double [kNumSums][n_groups]  per-group sums
    int n_blocks, n_groups; // fit grid (min(n_tiles, kFitGrid)) and its 32-workgroup reduction groups
    size_t off_partials;    // float  [kNumSums][n_blocks]  one partial per fit workgroup
    size_t off_J, off_m, off_v;  // float [n_tiles][3][256]
    size_t total;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline bool make_layout(int H, int W, int n_views, Layout *L) {
    if (H <= 0 || W <= 0 || n_views <= 0 || n_views > kMaxViews || H > 32767 || W > 32767) return false;
    L->H = H; L->W = W; L->n_views = n_views;
    L->tiles_x = (W + kTile - 1) / kTile;
    L->tiles_y = (H + kTile - 1) / kTile;
    L->n_tiles = L->tiles_x * L->tiles_y;
    const size_t nt = (size_t)L->n_tiles, nv = (size_t)n_views;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L->off_obs = take(nt * nv * kChunk);
#ifdef SUCRE_OBS_VIEW_MAJOR
    L->obs_tile_stride = kChunk; L->obs_view_stride = nt * kChunk;      // [view][tile]
#else
    L->obs_tile_stride = nv * kChunk; L->obs_view_stride = kChunk;      // [tile][view]
#endif
    L->off_cnt = take(nt * nv * sizeof(uint16_t));
    L->off_comp = take(nt * nv * kChunk);
    L->off_pcount = take(nt * kTilePx * sizeof(uint16_t));
    L->mask_words = (n_views + 63) / 64;
    L->off_pmask = take(nt * kTilePx * (size_t)L->mask_words * sizeof(uint64_t));
    L->off_blockhist = take(256 * nt * sizeof(uint32_t));
    L->off_bin_totals = take(512 * sizeof(uint32_t));
    L->off_perm = take(nt * kTilePx * sizeof(uint32_t));
    L->off_invperm = take(nt * kTilePx * sizeof(uint32_t));
    L->off_levels = take(nt * sizeof(uint32_t));
    L->off_full = take(nt * sizeof(uint32_t));
    L->off_tile_off = take(nt * sizeof(uint64_t));
    L->off_total_chunks = take(sizeof(uint64_t));
    L->off_view_count = take(nv * sizeof(uint64_t));
    L->off_view_keep = take(nv * sizeof(uint32_t));
    L->off_n_obs = take(sizeof(uint64_t));
    L->off_n_obs_total = take(sizeof(uint64_t));
    L->off_params = take(27 * sizeof(float));
    L->off_sums = take(kSumsPad * sizeof(double));
    L->n_blocks = L->n_tiles < kFitGrid ? L->n_tiles : kFitGrid;
    L->n_groups = (L->n_blocks + kGroup - 1) / kGroup;
    L->off_ticket = take((size_t)(1 + L->n_groups) * kTicketStride * sizeof(uint32_t));
    L->off_gpartials = take((size_t)kNumSums * L->n_groups * sizeof(double));
    L->off_partials = take(nt * kNumSums * sizeof(float));
    L->off_J = take(nt * 3 * kTilePx * sizeof(float));
    L->off_m = take(nt * 3 * kTilePx * sizeof(float));
    L->off_v = take(nt * 3 * kTilePx * sizeof(float));
    L->total = o;
    return true;
}

}  // namespace sucre
