// Count-sorted per-pixel compaction of the observation store (gfx950).
//
// The match kernel writes one dense chunk per (tile, view).  Inside a covered region ~10 % of the slots are
// empty (pixels that fail the forward/backward consistency test of sfm.py:171-175 are scattered), and every empty
// slot costs the fit as much VALU time and HBM traffic as a real observation.  The fit never needs to know WHICH
// view an observation came from (sucre.py:79-82 only reads z and I), so after matching we
//   1. count the valid observations of every pixel over the kept views,
//   2. sort the pixels by that count, descending, with a deterministic stable counting sort
//      (block histograms -> per-bin scan over blocks -> in-block stable rank),
//   3. stack each pixel's observations, in view order, into "levels": chunk (tile', level) of the compact store
//      holds the level-th observation of the 256 pixels of sorted tile tile'.
// A sorted tile's pixels have (nearly) equal counts, so it has max-count levels and almost no empty slot; tiles
// come out heaviest first, which is also the better dispatch order.  J and the Adam moments live in the sorted
// pixel order; fit_init / export_J translate through perm / invperm.  Everything is fixed-order: results stay
// bitwise reproducible.
#include "launch.h"

namespace sucre {

constexpr int kMaxBins = 256;

__host__ __device__ inline int num_bins(int n_views) { return (n_views < kMaxBins - 1 ? n_views : kMaxBins - 1) + 1; }

// count -> bin (monotone; identity while n_views < 255)
__device__ __forceinline__ int bin_of(uint32_t count, int n_views) {
    if (n_views < kMaxBins - 1) return (int)count;
    return (int)(((uint64_t)count * (kMaxBins - 1) + n_views - 1) / n_views);
}

// The views a tile has observations in (kept by min_cover, non-empty in this tile), compacted into LDS by one
// wave with ballot prefix sums; returns how many.  All 256 threads call it.
__device__ __forceinline__ int tile_view_list(const uint16_t *__restrict__ cnt, const uint32_t *__restrict__ view_keep,
                                              int tile, int n_views, uint16_t *vlist, int *vn) {
    const int t = threadIdx.x;
    if (t < 64) {
        int n = 0;
        for (int base = 0; base < n_views; base += 64) {
            const int k = base + t;
            const bool f = k < n_views && view_keep[k] != 0 && cnt[(size_t)tile * n_views + k] > 0;
            const unsigned long long m = __ballot(f);
            if (f) vlist[n + __builtin_popcountll(m & ((1ull << t) - 1ull))] = (uint16_t)k;
            n += __builtin_popcountll(m);
        }
        if (t == 0) *vn = n;
    }
    __syncthreads();
    return *vn;
}

constexpr int kBatch = 8;  // views loaded per thread before any dependent work: independent loads in flight

// 1. per-pixel observation count over the kept views + per-block histogram (blockhist is bin-major).
__global__ __launch_bounds__(256) void pixel_count_kernel(const uint8_t *__restrict__ obs,
                                                          const uint16_t *__restrict__ cnt,
                                                          const uint32_t *__restrict__ view_keep, int n_views,
                                                          int n_tiles, size_t tile_stride, size_t view_stride,
                                                          uint16_t *__restrict__ pcount,
                                                          uint64_t *__restrict__ pmask, int mask_words,
                                                          uint32_t *__restrict__ blockhist) {
    __shared__ uint32_t hist[kMaxBins];
    __shared__ uint16_t vlist[kMaxViews];
    __shared__ int vn;
    const int tile = blockIdx.x, t = threadIdx.x;
    hist[t] = 0;
    const int n = tile_view_list(cnt, view_keep, tile, n_views, vlist, &vn);
    const uint8_t *tbase = obs + (size_t)tile * tile_stride;
    uint64_t *mask = pmask + ((size_t)tile * kTilePx + t) * mask_words;
    for (int wv = 0; wv < mask_words; ++wv) mask[wv] = 0ull;
    uint32_t c = 0;
    unsigned long long word = 0ull;
    int word_idx = 0;
    for (int i0 = 0; i0 < n; i0 += kBatch) {
        float z[kBatch];
        int kk[kBatch];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            kk[b] = i0 + b < n ? vlist[i0 + b] : -1;
            z[b] = kk[b] >= 0 ? reinterpret_cast<const float *>(tbase + (size_t)kk[b] * view_stride)[t] : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            if (kk[b] >= 0 && (kk[b] >> 6) != word_idx) {  // views come in increasing order: flush the finished word
                mask[word_idx] = word;
                word = 0ull;
                word_idx = kk[b] >> 6;
            }
            if (z[b] > 0.0f) { word |= 1ull << (kk[b] & 63); ++c; }
        }
    }
    mask[word_idx] = word;
    pcount[(size_t)tile * kTilePx + t] = (uint16_t)c;
    atomicAdd(&hist[bin_of(c, n_views)], 1u);  // integer LDS atomics: order-independent result
    __syncthreads();
    if (t < num_bins(n_views)) blockhist[(size_t)t * n_tiles + tile] = hist[t];
}

// 2a. exclusive scan of every bin's column over the blocks (one workgroup per bin); totals[bin] = column sum.
__global__ __launch_bounds__(256) void bin_scan_kernel(uint32_t *__restrict__ blockhist, int n_tiles,
                                                       uint32_t *__restrict__ totals) {
    __shared__ uint32_t part[256];
    uint32_t *col = blockhist + (size_t)blockIdx.x * n_tiles;
    const int t = threadIdx.x;
    const int per = (n_tiles + 255) / 256;
    const int lo = t * per, hi = min(lo + per, n_tiles);
    uint32_t s = 0;
    for (int i = lo; i < hi; ++i) s += col[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 256; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; }
        totals[blockIdx.x] = run;
    }
    __syncthreads();
    uint32_t run = part[t];
    for (int i = lo; i < hi; ++i) { const uint32_t v = col[i]; col[i] = run; run += v; }
}

// 2b. start of every bin in the sorted order: bins with more observations first.
__global__ void bin_base_kernel(const uint32_t *__restrict__ totals, int bins, uint32_t *__restrict__ bin_base) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t run = 0;
        for (int b = bins - 1; b >= 0; --b) { bin_base[b] = run; run += totals[b]; }
    }
}

// 2c. stable destination of every pixel: bin start + pixels of the same bin in earlier blocks + earlier threads
//     of this block in the same bin.
__global__ __launch_bounds__(256) void permute_kernel(const uint16_t *__restrict__ pcount,
                                                      const uint32_t *__restrict__ blockhist,
                                                      const uint32_t *__restrict__ bin_base, int n_views, int n_tiles,
                                                      uint32_t *__restrict__ perm, uint32_t *__restrict__ invperm) {
    __shared__ int bins_of[kTilePx];
    const int tile = blockIdx.x, t = threadIdx.x;
    const uint32_t src = (uint32_t)tile * kTilePx + t;
    const int b = bin_of(pcount[src], n_views);
    bins_of[t] = b;
    __syncthreads();
    uint32_t rank = 0;
    for (int i = 0; i < t; ++i) rank += bins_of[i] == b ? 1u : 0u;  // lock-step broadcast reads
    const uint32_t dst = bin_base[b] + blockhist[(size_t)b * n_tiles + tile] + rank;
    perm[dst] = src;
    invperm[src] = dst;
}

// 3a. levels of every sorted tile = the largest pixel count in it; the levels below the smallest count are
//     completely full (every slot a real observation), which lets the fit skip the validity select there.
__global__ __launch_bounds__(256) void tile_levels_kernel(const uint16_t *__restrict__ pcount,
                                                          const uint32_t *__restrict__ perm,
                                                          uint32_t *__restrict__ levels, uint32_t *__restrict__ full) {
    __shared__ uint32_t mx[256], mn[256];
    const int t = threadIdx.x;
    mx[t] = mn[t] = pcount[perm[(size_t)blockIdx.x * kTilePx + t]];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) { mx[t] = max(mx[t], mx[t + w]); mn[t] = min(mn[t], mn[t + w]); }
        __syncthreads();
    }
    if (t == 0) { levels[blockIdx.x] = mx[0]; full[blockIdx.x] = mn[0]; }
}

// 3b. byte offset of every sorted tile's first chunk (exclusive scan of levels; one workgroup).
__global__ __launch_bounds__(256) void tile_offset_kernel(const uint32_t *__restrict__ levels, int n_tiles,
                                                          uint64_t *__restrict__ tile_off,
                                                          uint64_t *__restrict__ total_chunks, int fmt) {
    __shared__ unsigned long long part[256];
    const int t = threadIdx.x;
    const int per = (n_tiles + 255) / 256;
    const int lo = t * per, hi = min(lo + per, n_tiles);
    unsigned long long s = 0;
    for (int i = lo; i < hi; ++i) s += levels[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 256; ++i) { const unsigned long long v = part[i]; part[i] = run; run += v; }
        *total_chunks = run;
        *reinterpret_cast<uint32_t *>(total_chunks + 1) = (uint32_t)fmt;  // what the fit kernels must be told
    }
    __syncthreads();
    unsigned long long run = part[t];
    for (int i = lo; i < hi; ++i) { tile_off[i] = run * (unsigned long long)chunk_bytes(fmt); run += levels[i]; }
}

// 3c. the compaction itself, driven from the DENSE side.  One workgroup owns one dense tile: it stages the tile's
//     chunks through LDS a group of views at a time -- every chunk is read from HBM exactly once, fully coalesced --
//     and each thread (pixel) then walks its view bitmask (next set bit = next level) and emits its observations to
//     its slot of the sorted order.  Pixels of one dense tile that fall into the same count bin are neighbours in the
//     sorted order (the sort is stable), so these writes form runs.  (History: the first version gathered from the
//     sorted side, one thread per sorted slot reading 4 + 3 x 1 bytes per observation from a different (tile, view)
//     chunk; every dense chunk was re-read by the ~7 sorted tiles holding some of its pixels -- 3.9 GB fetched for
//     0.55 GB of observations (rocprofv3 FETCH_SIZE), HBM-bound at 1.13 ms against 0.72 ms now.)
//     kFmt = SUCRE_OBS_U16MM: the range is stored as uint16 millimetres, rint(1000 z) clamped to [1, 65535]
//     (0 stays the empty-slot marker), colours unchanged -> 1280-byte chunks.
//     kExt: the three float32 extension planes of every chunk (camera points of the light model / float32
//     colours) ride along, 3072 more bytes per staged view.
template <bool kExt>
struct StageCfg {
    static constexpr int kViews = kExt ? 16 : 32;                 // 16 x 4864 B = 76 KB / 32 x 1792 B = 56 KB of LDS
    static constexpr int kBytes = kChunk + (kExt ? kExtChunk : 0);
};

template <int kFmt, bool kExt>
__global__ __launch_bounds__(256) void scatter_kernel(const uint8_t *__restrict__ obs, size_t tile_stride,
                                                      size_t view_stride, const uint32_t *__restrict__ invperm,
                                                      const uint64_t *__restrict__ pmask, int mask_words,
                                                      const uint32_t *__restrict__ levels,
                                                      const uint64_t *__restrict__ tile_off,
                                                      uint8_t *__restrict__ comp, int n_views,
                                                      const uint8_t *__restrict__ ext_dense,
                                                      uint8_t *__restrict__ ext_comp) {
    constexpr int kStageViews = StageCfg<kExt>::kViews, kStageBytes = StageCfg<kExt>::kBytes;
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageViews][kStageBytes];
    __shared__ uint32_t present;
    __shared__ uint8_t vl[kStageViews];
    const int tile = blockIdx.x, t = threadIdx.x;
    const uint32_t dst = invperm[(size_t)tile * kTilePx + t];
    const uint32_t dtile = dst / kTilePx, dslot = dst % kTilePx;
    const uint32_t nl = levels[dtile];
    uint8_t *out = comp + tile_off[dtile];
    float *eout = kExt ? reinterpret_cast<float *>(ext_comp + (tile_off[dtile] / kChunk) * kExtChunk) : nullptr;
    const uint64_t *mask = pmask + ((size_t)tile * kTilePx + t) * mask_words;
    const uint8_t *tbase = obs + (size_t)tile * tile_stride;
    constexpr int cb = kFmt ? kChunk16 : kChunk, zb = kFmt ? kChunkZ16 : kChunkZ;
    constexpr int kUnits = kStageBytes / 16;  // 16-byte pieces of a staged view
    constexpr int kMainUnits = kChunk / 16;
    uint32_t lv = 0;
    for (int g0 = 0; g0 < n_views; g0 += kStageViews) {
        const uint32_t bits = (uint32_t)(mask[g0 >> 6] >> (g0 & 63)) &
                              (kStageViews == 32 ? 0xffffffffu : ((1u << (kStageViews & 31)) - 1u));  // views g0 .. g0+kStageViews-1
        if (t == 0) present = 0u;
        __syncthreads();
        if (bits) atomicOr(&present, bits);
        __syncthreads();
        const uint32_t p = present;  // views of the group that some pixel of the tile uses (workgroup-uniform)
        if (p == 0u) continue;
        if (t < kStageViews && ((p >> t) & 1u)) vl[__builtin_popcount(p & ((1u << t) - 1u))] = (uint8_t)t;
        __syncthreads();
        const int total = __builtin_popcount(p) * kUnits;
        for (int u = t; u < total; u += 256) {
            const int i = vl[u / kUnits], w = u % kUnits;
            const uint8_t *src = (!kExt || w < kMainUnits)
                ? tbase + (size_t)(g0 + i) * view_stride + (size_t)w * 16
                : ext_dense + ((size_t)tile * n_views + (g0 + i)) * kExtChunk + (size_t)(w - kMainUnits) * 16;
            *reinterpret_cast<uint4 *>(&stage[i][w * 16]) = *reinterpret_cast<const uint4 *>(src);
        }
        __syncthreads();
        uint32_t m = bits;
        while (m) {
            const int i = __builtin_ctz(m);
            m &= m - 1u;
            const float z = reinterpret_cast<const float *>(&stage[i][0])[t];
            uint8_t *o = out + (size_t)lv * cb;
            if (kFmt) {
                const float mm = fminf(fmaxf(rintf(z * kMmPerM), 1.0f), 65535.0f);
                reinterpret_cast<uint16_t *>(o)[dslot] = z > 0.0f ? (uint16_t)mm : (uint16_t)0;
            } else {
                reinterpret_cast<float *>(o)[dslot] = z;
            }
            o[zb + dslot] = stage[i][kChunkZ + t];
            o[zb + kTilePx + dslot] = stage[i][kChunkZ + kTilePx + t];
            o[zb + 2 * kTilePx + dslot] = stage[i][kChunkZ + 2 * kTilePx + t];
            if (kExt) {
                const float *se = reinterpret_cast<const float *>(&stage[i][kChunk]);
                float *de = eout + (size_t)lv * (kExtChunk / 4);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) de[pl * kTilePx + dslot] = se[pl * kTilePx + t];
            }
            ++lv;
        }
        __syncthreads();  // everyone is done with the stage before the next group overwrites it
    }
    for (; lv < nl; ++lv) {  // padding slots of this pixel: the sorted tile has more levels than it has observations
        uint8_t *o = out + (size_t)lv * cb;
        if (kFmt) reinterpret_cast<uint16_t *>(o)[dslot] = 0;
        else reinterpret_cast<float *>(o)[dslot] = 0.0f;
        o[zb + dslot] = 0;
        o[zb + kTilePx + dslot] = 0;
        o[zb + 2 * kTilePx + dslot] = 0;
        if (kExt) {
            float *de = eout + (size_t)lv * (kExtChunk / 4);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) de[pl * kTilePx + dslot] = 0.0f;
        }
    }
}

hipError_t launch_compact(const Layout &L, uint8_t *ws, hipStream_t s, const uint8_t *ext_dense, uint8_t *ext_comp,
                          int fmt) {
    auto *cnt = reinterpret_cast<const uint16_t *>(ws + L.off_cnt);
    auto *keep = reinterpret_cast<const uint32_t *>(ws + L.off_view_keep);
    auto *pcount = reinterpret_cast<uint16_t *>(ws + L.off_pcount);
    auto *pmask = reinterpret_cast<uint64_t *>(ws + L.off_pmask);
    auto *blockhist = reinterpret_cast<uint32_t *>(ws + L.off_blockhist);
    auto *totals = reinterpret_cast<uint32_t *>(ws + L.off_bin_totals);
    auto *bin_base = totals + kMaxBins;
    auto *perm = reinterpret_cast<uint32_t *>(ws + L.off_perm);
    auto *invperm = reinterpret_cast<uint32_t *>(ws + L.off_invperm);
    auto *levels = reinterpret_cast<uint32_t *>(ws + L.off_levels);
    auto *tile_off = reinterpret_cast<uint64_t *>(ws + L.off_tile_off);
    const int bins = num_bins(L.n_views);
    hipLaunchKernelGGL(pixel_count_kernel, dim3(L.n_tiles), dim3(256), 0, s, ws + L.off_obs, cnt, keep, L.n_views,
                       L.n_tiles, L.obs_tile_stride, L.obs_view_stride, pcount, pmask, L.mask_words, blockhist);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(bins), dim3(256), 0, s, blockhist, L.n_tiles, totals);
    hipLaunchKernelGGL(bin_base_kernel, dim3(1), dim3(64), 0, s, totals, bins, bin_base);
    hipLaunchKernelGGL(permute_kernel, dim3(L.n_tiles), dim3(256), 0, s, pcount, blockhist, bin_base, L.n_views,
                       L.n_tiles, perm, invperm);
    hipLaunchKernelGGL(tile_levels_kernel, dim3(L.n_tiles), dim3(256), 0, s, pcount, perm, levels,
                       reinterpret_cast<uint32_t *>(ws + L.off_full));
    hipLaunchKernelGGL(tile_offset_kernel, dim3(1), dim3(256), 0, s, levels, L.n_tiles, tile_off,
                       reinterpret_cast<uint64_t *>(ws + L.off_total_chunks), fmt);
    const dim3 grid(L.n_tiles), block(256);
    const uint8_t *obs = ws + L.off_obs;
    uint8_t *comp = ws + L.off_comp;
    if (ext_dense)  // light model / float32 colours: float32 store only
        hipLaunchKernelGGL((scatter_kernel<0, true>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                           pmask, L.mask_words, levels, tile_off, comp, L.n_views, ext_dense, ext_comp);
    else if (fmt)
        hipLaunchKernelGGL((scatter_kernel<1, false>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                           pmask, L.mask_words, levels, tile_off, comp, L.n_views, ext_dense, ext_comp);
    else
        hipLaunchKernelGGL((scatter_kernel<0, false>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                           pmask, L.mask_words, levels, tile_off, comp, L.n_views, ext_dense, ext_comp);
    return hipGetLastError();
}

}  // namespace sucre
