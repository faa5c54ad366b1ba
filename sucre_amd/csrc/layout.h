// Workspace layout of one restoration (private to libsucre_hip.so).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "experiment.h"

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
// Store format 2 (kStoreZ24, chosen ON THE DEVICE for a store the caller asked to be SUCRE_OBS_F32): the float32 ranges of an
// image whose bit patterns all lie within 2^24 - 2 of the smallest one -- any scene whose ranges span less than about a
// factor of four -- are kept as 24-bit codes, code = bits(z) - bits(z_min) + 1 (0 = empty slot): bits(z) = code + offset
// gives every range back exactly, 6 instead of 7 bytes per observation.  A full chunk: 64 x {4 codes (12 B), R word, G word,
// B word} = 1536 B, a lane's 24 bytes side by side (four dwords code | red << 24, then the G and B words); the short last chunk: [code: 64 x r x 3 B][R: 64 x r][G][B].
// Store format 3 (kStoreZ26, round 6; also the device's choice): an image whose ranges span MORE than 2^24 - 2 bit patterns but
// no more than 2^26 - 2 (a factor of up to 256 between the nearest and the farthest range: any real scene) keeps 26-bit codes,
// 6.25 bytes per observation.  A full chunk is a kStoreZ24 chunk -- the low 24 bits of the four codes, the colours -- followed
// by ONE byte per lane holding bits 24-25 of the lane's four codes (level j in bits 2j, 2j+1): 1536 + 64 = 1600 B.  The short
// last chunk (r levels): [low 24 bits: 64 x r x 3 B][R: 64 x r][G][B][bits 24-25: 64 r two-bit fields, field lane r + j, sixteen
// to a dword].  A level is 400 bytes, no multiple of 64, and items are addressed in units of 64 bytes (PlanItem.src64): a
// kStoreZ26 strip therefore STARTS on a multiple of four levels (StripMeta.lvoff counts padded levels) -- its chunks then sit
// at multiples of 1600 = 25 x 64 bytes -- and the padding is never copied (store space, not traffic).
constexpr int kStoreF32 = 0, kStoreU16 = 1, kStoreZ24 = 2, kStoreZ26 = 3;
constexpr int kChunk24 = 6 * 256;    // 1536 bytes
constexpr int kChunk26 = kChunk24 + 64;   // 1600 bytes
__host__ __device__ constexpr int chunk_bytes(int fmt) { return fmt == kStoreU16 ? kChunk16 : fmt == kStoreZ24 ? kChunk24 : fmt == kStoreZ26 ? kChunk26 : kChunk; }
// strip offsets of a store: kStoreZ26 counts every strip's levels rounded up to a multiple of four
__host__ __device__ constexpr uint32_t padded_levels(int fmt, uint32_t levels) { return fmt == kStoreZ26 ? (levels + 3u) & ~3u : levels; }
constexpr int kExtChunk = 3 * kChunkZ;  // light model: cP.x, cP.y, cP.z planes of a chunk (extension workspace)
// Compact store (what the fit streams): the count-sorted pixels are cut into STRIPS of 64 consecutive pixels -- one
// pixel per lane of the wave that owns the strip.  A strip with n levels is stored as ceil(n/4) chunks of 64 pixels x
// r levels, r = 4 except in the last one (r = n - 4 g).  A full chunk: [z: 64 x 4 ranges, pixel-major][64 x {R word, G word,
// B word}: byte j of a word = level j] -- the size of a dense one (1792 B / 1280 B); a lane reads its pixel's four levels
// as one float4 (uint2) + three adjacent dwords.  The short last chunk: [z: 64 x r][R: 64 x r bytes][G][B], kLevelBytes x r.
constexpr int kStripPx = 64;
constexpr int kStripsPerTile = kTilePx / kStripPx;
constexpr int kGroupLv = 4;                       // levels per full chunk
constexpr int kStateFloats = 9 * kStripPx;        // J[3], exp_avg[3], exp_avg_sq[3] planes of a strip (2304 B)
__host__ __device__ constexpr int range_bytes(int fmt) { return fmt == kStoreU16 ? 2 : (fmt == kStoreZ24 || fmt == kStoreZ26) ? 3 : 4; }   // (kStoreZ26: + 2 bits)
__host__ __device__ constexpr int level_bytes(int fmt) { return (range_bytes(fmt) + 3) * kStripPx + (fmt == kStoreZ26 ? kStripPx / 4 : 0); }  // 448 / 320 / 384 / 400
static_assert(kGroupLv * level_bytes(kStoreZ26) == kChunk26 && kChunk26 % 64 == 0, "a 26-bit chunk is four levels and a whole number of 64-byte units");
constexpr int kExtLevelBytes = 3 * 4 * kStripPx;  // extension planes of one level of a strip: 3 floats x 64 pixels

struct StripMeta {
    uint64_t lvoff;   // levels of all earlier strips (kStoreZ26: each rounded up to a multiple of four): the strip's chunks start at comp + lvoff * level_bytes(fmt)
    uint32_t levels;  // largest pixel count in the strip
    uint32_t full;    // smallest pixel count: chunks wholly below it hold 64 x 4 real observations (no select needed)
};
constexpr int kNumSums = 10;         // sB[3], sGZ[3], sBeta[3], cost
constexpr int kSumsPad = 12;
constexpr int kGroup = 32;           // tiles per reduction group (two-level last-arriver reduction)
constexpr int kTicketStride = 16;    // uint32 words between tickets: every counter on its own 64-byte line
// (SUCRE_FIT_WAVES / SUCRE_CLOSED_WAVES: experiment.h.  5: 82-94 VGPRs -- at 6 waves (80) the kernels spill, and scratch
// traffic both counts in the hand-counted vmcnt waits and made results depend on what else ran on the GPU, round 2)
constexpr int kFitWaves = SUCRE_FIT_WAVES;        // waves per SIMD of fit_grad_kernel / group_iter_kernel (<= 96 VGPRs)
constexpr int kClosedWaves = SUCRE_CLOSED_WAVES;  // ... of fit_closed_kernel (27 accumulators per lane: 118 VGPRs)
constexpr int kFitGrid = 256 * kFitWaves;         // persistent fit workgroups (kFitWaves per CU x 256 CUs: all resident
                                     // at once); constants, so the reduction order -- hence every result bit -- does
                                     // not depend on the device
constexpr int kClosedGrid = 256 * kClosedWaves;
constexpr int kMaxViews = 4096;
constexpr int kMinStripsPerWave = SUCRE_MIN_STRIPS;   // strips a fit wave gets at least, when the image has them (make_layout)

// ---------------------------------------------------------------------------------------------------------------
// The deal: which strips a fit wave works on (static, so that every sum is formed in the same order on every run).
//
// A full fit grid keeps G = kFitWaves (kClosedWaves) workgroups resident on every CU, i.e. G waves on every SIMD, dispatched
// in workgroup order: workgroup b is the (b / 256)-th arrival on its CU -- its GENERATION.  The SIMD issues the OLDEST ready wave
// first, so the generations do not run at one speed: with equal shares generation 0 was done after 54 us of a 126 us launch,
// generation 4 after 123 us, and for the last 40 us one or two waves per SIMD could not keep the memory system busy
// (tools/exp/wave_times.py).  The shares are therefore unequal.  The strips (sorted heaviest first) are
// dealt in ROUNDS: the waves of the generations taking part in round r get consecutive strips, in wave order on even rounds and
// in reverse on odd ones.  Generation 0 takes part in every round; generation g takes part whenever that keeps its work (in
// items: a strip's chunks + 2, judged by the round's first strip) nearest to p[g] / kDealDen of generation 0's.  The late rounds
// hold the light strips, so the shares are met to within a few items.  p = kDealDen for every generation is the plain
// boustrophedon deal (strip r W + wid / r W + W - 1 - wid), used whenever the grid is not the full one.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kDealDen = 64;
constexpr int kMaxGen = 8;
struct DealShares { uint32_t G; uint32_t p[kMaxGen]; };

__host__ __device__ inline DealShares deal_shares(int mode, uint32_t blocks) {
    const uint32_t fit[kMaxGen] = {SUCRE_DEAL_FIT}, closed[kMaxGen] = {SUCRE_DEAL_CLOSED};
    DealShares sh;
    const uint32_t G = (uint32_t)(mode ? kClosedWaves : kFitWaves);
    bool weighted = blocks == 256u * G && G <= (uint32_t)kMaxGen;
    for (uint32_t g = 0; weighted && g < G; ++g) weighted = (mode ? closed[g] : fit[g]) >= 1u && (mode ? closed[g] : fit[g]) <= kDealDen;
    weighted = weighted && (mode ? closed[0] : fit[0]) == kDealDen;   // somebody takes part in every round
    bool all_equal = true;   // equal shares ARE the plain deal (the product default): one generation, so that deal_rounds -- hence
    for (uint32_t g = 0; g < G && g < (uint32_t)kMaxGen; ++g) all_equal = all_equal && (mode ? closed[g] : fit[g]) == kDealDen;   // the plan space -- is not G x too large
    weighted = weighted && !all_equal;
    sh.G = weighted ? G : 1u;
    for (int g = 0; g < kMaxGen; ++g) sh.p[g] = weighted && (uint32_t)g < G ? (mode ? closed[g] : fit[g]) : kDealDen;
    return sh;
}

// The same for a kernel that sizes its grid at run time (light.hip): G resident workgroups per CU, shares by generation from the
// table the fit kernels' measurements suggest (G = 4: the closed-form kernel's, G = 5: the J-parameter kernel's); any other grid
// gets the plain deal.
__host__ __device__ inline DealShares deal_shares_resident(uint32_t blocks) {
    const uint32_t four[kMaxGen] = {SUCRE_DEAL_CLOSED}, five[kMaxGen] = {SUCRE_DEAL_FIT};
    DealShares sh;
    const bool g4 = blocks == 1024u && kClosedWaves == 4, g5 = blocks == 1280u && kFitWaves == 5;
    sh.G = g4 ? 4u : g5 ? 5u : 1u;
    bool all_equal = true;
    for (int g = 0; g < kMaxGen; ++g) {
        sh.p[g] = (g4 && g < 4) ? four[g] : (g5 && g < 5) ? five[g] : kDealDen;
        all_equal = all_equal && sh.p[g] == kDealDen;
    }
    if (all_equal || sh.p[0] != kDealDen) { sh.G = 1u; for (int g = 0; g < kMaxGen; ++g) sh.p[g] = kDealDen; }
    return sh;
}

// Calls f(k, strip) for the wave's strips in its working order; returns how many there are.  levels(strip) = the strip's level
// count (StripMeta.levels; device data: the deal is made where the plan is written).  W % G == 0 (deal_shares).
template <class L, class F>
__host__ __device__ inline uint32_t deal_walk(uint32_t wid, uint32_t W, uint32_t n_strips, const DealShares &sh, L &&levels, F &&f) {
    const uint32_t Wg = W / sh.G, g = wid / Wg, j = wid - g * Wg;
    uint32_t base = 0u, k = 0u;
    uint32_t work[kMaxGen] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // per wave of the generation, in items
    for (uint32_t r = 0u; base < n_strips; ++r) {
        const uint32_t items = sh.G > 1u ? ((levels(base) + 3u) >> 2) + 2u : 0u;
        uint32_t below = 0u, total = 0u;
        bool mine = false;
        for (uint32_t q = 0u; q < sh.G; ++q) {
            // generation q > 0 sits the round out if taking part would put it further above its share than it is below now
            if (q > 0u && (uint64_t)(2u * work[q] + items) * kDealDen > (uint64_t)2u * sh.p[q] * work[0]) continue;
            work[q] += items;
            ++total;
            below += q < g ? 1u : 0u;
            mine = mine || q == g;
        }
        const uint32_t n_r = total * Wg;
        if (mine) {
            const uint32_t rank = below * Wg + j, strip = base + ((r & 1u) ? n_r - 1u - rank : rank);
            if (strip < n_strips) { f(k, strip); ++k; }
        }
        base += n_r;
    }
    return k;
}

// Rounds of a deal = strips of its busiest wave, at most (generation 0 takes part in every round).
__host__ __device__ inline uint32_t deal_rounds(uint32_t W, uint32_t n_strips, const DealShares &sh) {
    const uint32_t per_round = sh.G > 1u ? W / sh.G : W;
    return (n_strips + per_round - 1u) / per_round;
}

// One entry of a wave's item stream (fit.hip, plan_kernel): what to copy into the LDS ring.  What the item IS follows
// from its position: a wave's stream is, strip after strip, [J plane][full chunks, the unmasked ones first][short last
// chunk][moments (J-parameter mode)], and StripEntry says how many of each the strip has.
struct PlanItem {
    uint32_t src64;   // source address: workspace base + 64 * src64
    uint32_t shape;   // lanes of the two DMA instructions (fit.hip, item_shape)
};
struct StripEntry {
    uint32_t strip;   // index of the strip (state block, StripMeta)
    uint32_t counts;  // [11:0] full chunks without an empty slot, [23:12] full chunks that need the z > 0 test, [26:24] levels of the short last chunk
};
// A strip has at most kMaxViews levels = kMaxViews / 4 full chunks: both chunk counts must hold that many (until round 4 the
// fields were 8 bits wide: a strip of 1024 levels or more decoded as nu = 0, nm = 1 and the wave's item stream desynchronised).
constexpr uint32_t kCountBits = 12;
static_assert(kMaxViews / kGroupLv < (1 << kCountBits), "StripEntry.counts: a strip's full chunks must fit their field");
__host__ __device__ constexpr uint32_t strip_counts(uint32_t unmasked, uint32_t masked, uint32_t tail) { return unmasked | (masked << kCountBits) | (tail << (2 * kCountBits)); }
__host__ __device__ constexpr uint32_t counts_unmasked(uint32_t c) { return c & ((1u << kCountBits) - 1u); }
__host__ __device__ constexpr uint32_t counts_masked(uint32_t c) { return (c >> kCountBits) & ((1u << kCountBits) - 1u); }
__host__ __device__ constexpr uint32_t counts_tail(uint32_t c) { return (c >> (2 * kCountBits)) & 7u; }
static_assert(counts_unmasked(strip_counts(kMaxViews / kGroupLv, 0, 3)) == kMaxViews / kGroupLv && counts_masked(strip_counts(0, kMaxViews / kGroupLv, 3)) == kMaxViews / kGroupLv
              && counts_masked(strip_counts(kMaxViews / kGroupLv, 0, 3)) == 0 && counts_tail(strip_counts(1023, 1, 3)) == 3, "StripEntry.counts round trip at the largest strip");

struct Layout {
    int H, W, n_views, tiles_x, tiles_y, n_tiles;
    size_t off_obs;         // uint8  chunks of kChunk bytes, chunk(tile, k) at tile*obs_tile_stride + k*obs_view_stride
    size_t obs_tile_stride, obs_view_stride;
    size_t off_cnt;         // uint16 [n_tiles][n_views]   matches of view k inside the tile
    size_t off_comp;        // uint8  compact store: strip s at lvoff[s] * level_bytes(fmt), see StripMeta
    size_t off_pcount;      // uint16 [n_tiles*256]        observations of every pixel over the kept views
    size_t off_pmask;       // uint64 [n_tiles*256][mask_words]  which kept views observe the pixel (bit k = view k)
    int mask_words;
    size_t off_vbits;       // uint64 [n_tiles][n_views][4]  which pixels of the tile view k observes: word j, bit l = slot 64 j + l
    size_t off_blockhist;   // uint32 [256 bins][n_tiles]  counting-sort histograms (bin-major), scanned in place
    size_t off_bin_totals;  // uint32 [256] pixels per bin (the bin bases are their suffix sums: permute_kernel, strip_table_kernel)
    size_t off_perm;        // uint32 [n_tiles*256]        sorted slot -> dense slot (tile*256 + slot)
    size_t off_invperm;     // uint32 [n_tiles*256]        dense slot  -> sorted slot
    int n_strips;           // n_tiles * 4 strips of 64 sorted pixels
    size_t off_strip_meta;  // StripMeta [n_strips]
    size_t off_total_chunks;// uint64 [1] total (padded) levels over all strips, then uint32 [5]: format of the compact store (kStore*), the
                            // offset of its range codes (kStoreZ24 / kStoreZ26), smallest / largest range bits of the dense store
    size_t off_zrange;      // uint2  [n_tiles][n_views]   smallest / largest float32 bit pattern of ranges of the tile (0xffffffff / 0: none); an imported
                            // view's entry holds its own, a matched view's the ranges of ALL the views its wave walked (or none)
    size_t off_zpart;       // uint2  [ceil(n_tiles / 32)] the same over 32 tiles and all views
    size_t off_view_count;  // uint64 [n_views]
    size_t off_view_keep;   // uint32 [n_views]
    size_t off_view_partial;// uint32 [ceil(n_tiles / 32)][n_views]  match counts of 32 tiles per view (finalize)
    size_t off_n_obs;       // uint64 [1]
    size_t off_n_obs_total; // uint64 [1]
    size_t off_params;      // float  [9] params, [9] exp_avg, [9] exp_avg_sq
    size_t off_sums;        // double [kSumsPad]
    size_t off_ticket;      // uint32 [(1 + n_groups) * kTicketStride]  arrival counters: [0] = groups done, [1+g] = tiles of group g done
    size_t off_gpartials;   // double [kNumSums][n_groups]  per-group sums
    int n_blocks, n_groups; // largest fit grid (min(n_tiles, kFitGrid)) and its 32-workgroup reduction groups
    int fit_blocks[2], fit_groups[2];  // grid / reduction groups of the J-parameter [0] and closed-form [1] kernels
    size_t off_partials;    // float  [kNumSums][n_blocks]  one partial per fit workgroup
    size_t off_state;       // float [n_strips][9][64]: J, exp_avg, exp_avg_sq (three channel planes each) of every strip
    size_t off_plan[2];     // PlanItem [4 n_blocks][plan_stride]: item streams of the fit waves (J-parameter / closed-form)
    size_t off_plan_strips[2]; // StripEntry [4 n_blocks][plan_kmax]: the strips of every wave, in the order it works on them
    size_t off_plan_count[2];  // uint32 [4 n_blocks] strips of every wave
    size_t plan_stride[2];  // items reserved per wave (its strips' items + the never-consumed trailing items + one spare)
    size_t plan_kmax[2];    // strips reserved per wave
    size_t total;
};

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

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
    L->obs_tile_stride = nv * kChunk; L->obs_view_stride = kChunk;      // [tile][view]
    L->off_cnt = take(nt * nv * sizeof(uint16_t));
    // (the 26-bit store rounds every strip up to whole chunks: at most n_strips * (n_views + 3) levels of 400 bytes)
    const size_t comp_f32 = nt * nv * kChunk, comp_z26 = nt * (nv + 3) * kChunk26;
    L->off_comp = take(comp_f32 > comp_z26 ? comp_f32 : comp_z26);
    L->off_pcount = take(nt * kTilePx * sizeof(uint16_t));
    L->mask_words = (n_views + 63) / 64;
    L->off_pmask = take(nt * kTilePx * (size_t)L->mask_words * sizeof(uint64_t));
    L->off_vbits = take(nt * nv * 4 * sizeof(uint64_t));
    L->off_blockhist = take(256 * nt * sizeof(uint32_t));
    L->off_bin_totals = take(512 * sizeof(uint32_t));
    L->off_perm = take(nt * kTilePx * sizeof(uint32_t));
    L->off_invperm = take(nt * kTilePx * sizeof(uint32_t));
    L->n_strips = L->n_tiles * kStripsPerTile;
    L->off_strip_meta = take((size_t)L->n_strips * sizeof(StripMeta));
    L->off_total_chunks = take(sizeof(uint64_t));
    L->off_zrange = take(nt * nv * 2 * sizeof(uint32_t));
    L->off_zpart = take((nt + 31) / 32 * 2 * sizeof(uint32_t));
    L->off_view_count = take(nv * sizeof(uint64_t));
    L->off_view_keep = take(nv * sizeof(uint32_t));
    L->off_view_partial = take((nt + 31) / 32 * nv * sizeof(uint32_t));
    L->off_n_obs = take(sizeof(uint64_t));
    L->off_n_obs_total = take(sizeof(uint64_t));
    L->off_params = take(27 * sizeof(float));
    L->off_sums = take(kSumsPad * sizeof(double));
    // The fit grid of an image (the workgroups of its own launches; the waves its plan is written for; the shape of its reduction
    // trees): the persistent grid, or -- for a small image -- as many workgroups as give every wave kMinStripsPerWave strips.
    // (Until round 5: one strip per wave for images of fewer tiles than the grid.  What a wave pays per image -- descriptors, the
    // start and the end of its item stream, the tree over its ten sums -- is then paid per 64 pixels: a launch over 32 images of
    // 640x480 x 5 views spent more on that than on their observations, DESIGN.md section 4.7.  A batch launch keeps the whole
    // GPU busy whatever this number is: its workgroups take (image, workgroup-of-the-image) pairs in turn.)
    auto grid_of = [&](int full) { const int want = (L->n_tiles + kMinStripsPerWave - 1) / kMinStripsPerWave; return want < 1 ? 1 : (want < full ? want : full); };
    L->n_blocks = grid_of(kFitGrid);
    L->n_groups = (L->n_blocks + kGroup - 1) / kGroup;
    L->fit_blocks[0] = L->n_blocks;
    L->fit_blocks[1] = grid_of(kClosedGrid);
    for (int m = 0; m < 2; ++m) L->fit_groups[m] = (L->fit_blocks[m] + kGroup - 1) / kGroup;
    L->off_ticket = take((size_t)(1 + L->n_groups) * kTicketStride * sizeof(uint32_t));
    L->off_gpartials = take((size_t)kNumSums * L->n_groups * sizeof(double));
    L->off_partials = take(nt * kNumSums * sizeof(float));
    L->off_state = take((size_t)L->n_strips * kStateFloats * sizeof(float));
    {   // a wave gets at most one strip per round of the deal, each of at most ceil(n_views / 4) chunks + J plane + moments
        for (int m = 0; m < 2; ++m) {
            const size_t waves = (size_t)L->fit_blocks[m] * 4;
            L->plan_kmax[m] = deal_rounds((uint32_t)waves, (uint32_t)L->n_strips, deal_shares(m, (uint32_t)L->fit_blocks[m]));
            L->plan_stride[m] = L->plan_kmax[m] * (((size_t)n_views + kGroupLv - 1) / kGroupLv + 2) + SUCRE_RING;   // + kAhead trailing items + a spare
            L->off_plan[m] = take(waves * L->plan_stride[m] * sizeof(PlanItem));
            L->off_plan_strips[m] = take(waves * L->plan_kmax[m] * sizeof(StripEntry));
            L->off_plan_count[m] = take(waves * sizeof(uint32_t));
        }
    }
    L->total = o;
    return true;
}

}  // namespace sucre
