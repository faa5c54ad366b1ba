// Workspace layout of one restoration (private to libsucre_hip.so).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace sucre {

constexpr int kTile = 16;            // tile edge in pixels
constexpr int kTilePx = 256;         // pixels per tile = one 64-lane wave x 4 pixels per lane
constexpr int kChunkZ = 1024;        // bytes: 256 float32 ranges
constexpr int kChunkRGB = 768;       // bytes: 64 lanes x (r[4] g[4] b[4])
constexpr int kChunk = kChunkZ + kChunkRGB;  // 1792 bytes per (tile, view)
constexpr int kNumSums = 10;         // sB[3], sGZ[3], sBeta[3], cost
constexpr int kSumsPad = 12;
constexpr int kMaxViews = 4096;

struct Layout {
    int H, W, n_views, tiles_x, tiles_y, n_tiles;
    size_t off_obs;         // uint8  chunks of kChunk bytes, chunk(tile, k) at tile*obs_tile_stride + k*obs_view_stride
    size_t obs_tile_stride, obs_view_stride;
    size_t off_cnt;         // uint16 [n_tiles][n_views]   matches of view k inside the tile
    size_t off_list;        // uint32 [n_tiles][n_views]   compacted indices of kept, non-empty views
    size_t off_tile_n;      // uint32 [n_tiles]            length of that list
    size_t off_view_count;  // uint64 [n_views]
    size_t off_view_keep;   // uint32 [n_views]
    size_t off_n_obs;       // uint64 [1]
    size_t off_n_obs_total; // uint64 [1]
    size_t off_params;      // float  [9] params, [9] exp_avg, [9] exp_avg_sq
    size_t off_sums;        // double [kSumsPad]
    size_t off_ticket;      // uint32 [1]  arrival counter of the fused last-arriver reduction
    size_t off_partials;    // float  [n_tiles][kNumSums]
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
    L->off_list = take(nt * nv * sizeof(uint32_t));
    L->off_tile_n = take(nt * sizeof(uint32_t));
    L->off_view_count = take(nv * sizeof(uint64_t));
    L->off_view_keep = take(nv * sizeof(uint32_t));
    L->off_n_obs = take(sizeof(uint64_t));
    L->off_n_obs_total = take(sizeof(uint64_t));
    L->off_params = take(27 * sizeof(float));
    L->off_sums = take(kSumsPad * sizeof(double));
    L->off_ticket = take(sizeof(uint32_t));
    L->off_partials = take(nt * kNumSums * sizeof(float));
    L->off_J = take(nt * 3 * kTilePx * sizeof(float));
    L->off_m = take(nt * 3 * kTilePx * sizeof(float));
    L->off_v = take(nt * 3 * kTilePx * sizeof(float));
    L->total = o;
    return true;
}

}  // namespace sucre
