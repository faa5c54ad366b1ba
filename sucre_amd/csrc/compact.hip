// Count-sorted per-pixel compaction of the observation store (gfx950).
//
// The match kernel writes one dense chunk per (tile, view).  Inside a covered region ~10 % of the slots are
// empty (pixels that fail the forward/backward consistency test of sfm.py:171-175 are scattered), and every empty
// slot costs the fit as much VALU time and HBM traffic as a real observation.  The fit never needs to know WHICH
// view an observation came from (sucre.py:79-82 only reads z and I), so after matching we
//   1. count the valid observations of every pixel over the kept views,
//   2. sort the pixels by that count, descending, with a deterministic stable counting sort
//      (block histograms -> per-bin scan over blocks -> in-block stable rank),
//   3. stack each pixel's observations, in view order, into "levels", and cut the sorted pixels into strips of 64:
//      a strip is stored as chunks of 64 pixels x 4 levels (layout.h, StripMeta), one pixel per lane of the wave
//      that fits the strip.
// A strip's pixels have (nearly) equal counts, so it has max-count levels and almost no empty slot; strips come out
// heaviest first.  J and the Adam moments live in the sorted pixel order; fit_init / export_J translate through
// perm / invperm.  Everything is fixed-order: results stay bitwise reproducible.
#include "launch.h"

namespace sucre {

constexpr int kMaxBins = 256;

__host__ __device__ inline int num_bins(int n_views) { return (n_views < kMaxBins - 1 ? n_views : kMaxBins - 1) + 1; }

// count -> bin (monotone; identity while n_views < 255)
__device__ __forceinline__ int bin_of(uint32_t count, int n_views) {
    if (n_views < kMaxBins - 1) return (int)count;
    return (int)(((uint64_t)count * (kMaxBins - 1) + n_views - 1) / n_views);
}

// Bit k of keep[] = view k passed the min_cover rule (sfm.py:136); built by ballots, word w by wave w (mod 4).
// All 256 threads call it; ends with a barrier.
__device__ __forceinline__ void keep_words(const uint32_t *__restrict__ view_keep, int n_views, int mask_words, uint64_t *keep) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int wv = wave; wv < mask_words; wv += 4) {
        const int k = wv * 64 + lane;
        const unsigned long long m = __ballot(k < n_views && view_keep[k] != 0);
        if (lane == 0) keep[wv] = m;
    }
    __syncthreads();
}

// 64 x 64 bit-matrix transpose across the wave: lane i holds row i, and gets column i back (bit r of the result of
// lane c = bit c of the input of lane r).  Six butterfly stages: at distance s the lanes pair up, and the off-diagonal
// blocks of every 2s x 2s block swap.
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x) {
    const int lane = threadIdx.x & 63;
    constexpr uint64_t kLow[6] = {0x00000000ffffffffull, 0x0000ffff0000ffffull, 0x00ff00ff00ff00ffull,
                                  0x0f0f0f0f0f0f0f0full, 0x3333333333333333ull, 0x5555555555555555ull};
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int s = 32 >> st;
        const uint64_t m = kLow[st];
        const uint64_t xp = __shfl_xor((unsigned long long)x, s, 64);
        x = (lane & s) ? (((xp >> s) & m) | (x & ~m)) : ((x & m) | ((xp & m) << s));
    }
    return x;
}

// 1. per-pixel observation count over the kept views + per-block histogram (blockhist is bin-major), from the views'
//    pixel bits (`vbits`: 32 bytes per (tile, view), written by whatever filled view k: match_kernel, or
//    count_view_kernel after an import) -- the dense ranges are not read again (that pass took 129 us per image).
//    Also writes every pixel's mask of kept views (`pmask`), which drives the scatter.  Wave j owns the pixels
//    64 j .. 64 j + 63 (word j of the views' bits): lane i fetches word j of view 64 w + i, a bit-matrix transpose hands
//    lane l the 64 views' bits of pixel 64 j + l = word w of its mask.  (Collecting the word bit by bit from a copy of
//    the bits in LDS cost ~650 instructions per thread and 64 views: 39 us per image.)
__global__ __launch_bounds__(256) void pixel_count_kernel(const uint64_t *__restrict__ vbits, uint64_t *__restrict__ pmask,
                                                          int mask_words, const uint32_t *__restrict__ view_keep,
                                                          int n_views, int n_tiles, uint16_t *__restrict__ pcount,
                                                          uint32_t *__restrict__ blockhist) {
    __shared__ uint32_t hist[kMaxBins];
    __shared__ uint64_t keep[kMaxViews / 64];
    const int tile = blockIdx.x, t = threadIdx.x;
    const int l = t & 63, j = t >> 6;   // slot t = 64 j + l: bit l of word j
    const uint64_t *rows = vbits + (size_t)tile * n_views * 4 + j;
    uint64_t x0 = l < n_views ? rows[(size_t)l * 4] : 0ull;   // in flight while the keep words are built
    hist[t] = 0;
    keep_words(view_keep, n_views, mask_words, keep);
    uint64_t *mask = pmask + ((size_t)tile * kTilePx + t) * mask_words;
    uint32_t c = 0;
    for (int wv = 0; wv < mask_words; ++wv) {
        const int next = (wv + 1) * 64 + l;
        const uint64_t x1 = next < n_views ? rows[(size_t)next * 4] : 0ull;
        const uint64_t word = wave_transpose64(x0) & keep[wv];
        mask[wv] = word;
        c += (uint32_t)__builtin_popcountll(word);
        x0 = x1;
    }
    pcount[(size_t)tile * kTilePx + t] = (uint16_t)c;
    atomicAdd(&hist[bin_of(c, n_views)], 1u);  // integer LDS atomics: order-independent result
    __syncthreads();
    if (t < num_bins(n_views)) blockhist[(size_t)t * n_tiles + tile] = hist[t];
}

// Inclusive scan of one value per lane over the wave.
template <class T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// 2a. exclusive scan of every bin's column over the blocks (one workgroup per bin); totals[bin] = column sum.
//     The column passes through LDS 8192 elements at a time: coalesced in, every thread scans a run of 32 consecutive
//     elements (index i sits at i + i/32: the runs start in different banks), the run totals are scanned by shuffles,
//     coalesced out.  (The first version read and rewrote the runs in global memory, one dependent load after the
//     other, and scanned the 256 run totals on one thread: 16.6 us per image.)
constexpr int kScanRun = 32;
constexpr int kScanChunk = 256 * kScanRun;

__global__ __launch_bounds__(256) void bin_scan_kernel(uint32_t *__restrict__ blockhist, int n_tiles,
                                                       uint32_t *__restrict__ totals) {
    __shared__ uint32_t buf[kScanChunk + kScanChunk / 32];
    __shared__ uint32_t wsum[4];
    uint32_t *col = blockhist + (size_t)blockIdx.x * n_tiles;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t carry = 0;
    for (int c0 = 0; c0 < n_tiles; c0 += kScanChunk) {
        const int n = min(kScanChunk, n_tiles - c0);
#pragma unroll
        for (int r = 0; r < kScanRun; ++r) {
            const int i = t + 256 * r;
            buf[i + (i >> 5)] = i < n ? col[c0 + i] : 0u;
        }
        __syncthreads();
        uint32_t v[kScanRun], s = 0;
#pragma unroll
        for (int r = 0; r < kScanRun; ++r) { v[r] = buf[33 * t + r]; s += v[r]; }
        const uint32_t incl = wave_inclusive_scan(s);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = carry + incl - s;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
#pragma unroll
        for (int r = 0; r < kScanRun; ++r) { buf[33 * t + r] = run; run += v[r]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kScanRun; ++r) {
            const int i = t + 256 * r;
            if (i < n) col[c0 + i] = buf[i + (i >> 5)];
        }
        __syncthreads();
    }
    if (t == 0) totals[blockIdx.x] = carry;
}

// 2b. stable destination of every pixel: bin start + pixels of the same bin in earlier blocks + earlier threads
//     of this block in the same bin.  Bin starts: bins with more observations first = the suffix sums of the bins'
//     totals, which every workgroup works out for itself (<= 256 values; a launch of its own took 5 us).  Rank inside
//     the block: the wave walks its distinct bins (neighbouring pixels: a handful), one ballot each -- a lane's rank is
//     the lanes below it in its bin's ballot, plus the bin's pixels in the waves before it.  (A loop over all earlier
//     threads took 26 us per image.)
__global__ __launch_bounds__(256) void permute_kernel(const uint16_t *__restrict__ pcount,
                                                      const uint32_t *__restrict__ blockhist,
                                                      const uint32_t *__restrict__ totals, int n_views, int n_tiles,
                                                      uint32_t *__restrict__ perm, uint32_t *__restrict__ invperm) {
    __shared__ uint32_t base[kMaxBins];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t wcount[4][kMaxBins];
    const int tile = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bins = num_bins(n_views);
    const uint32_t src = (uint32_t)tile * kTilePx + t;
    const int b = bin_of(pcount[src], n_views);
    const uint32_t before = blockhist[(size_t)b * n_tiles + tile];
#pragma unroll
    for (int w = 0; w < 4; ++w) wcount[w][t] = 0;
    {   // base[b] = pixels of the bins above b: thread t takes bin bins-1-t
        const uint32_t v = t < bins ? totals[bins - 1 - t] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - v;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        if (t < bins) base[bins - 1 - t] = run;
    }
    uint32_t rank = 0;
    unsigned long long todo = __ballot(1);
    while (todo) {   // wave-uniform
        const int bb = __shfl(b, __ffsll((long long)todo) - 1, 64);
        const unsigned long long same = __ballot(b == bb);
        if (b == bb) {
            const unsigned long long below = same & ((1ull << lane) - 1ull);
            rank = (uint32_t)__builtin_popcountll(below);
            if (below == 0ull) wcount[wave][bb] = (uint32_t)__builtin_popcountll(same);
        }
        todo &= ~same;
    }
    __syncthreads();
    for (int w = 0; w < wave; ++w) rank += wcount[w][b];
    const uint32_t dst = base[b] + before + rank;
    perm[dst] = src;
    invperm[src] = dst;
}

// 3a. levels of every strip (64 consecutive sorted pixels) = the largest pixel count in it; the chunks wholly below
//     the smallest count hold real observations only, which lets the fit skip the validity select there.
//     One workgroup per sorted tile = four strips, one per wave; also the tile's total, for the offsets below.
__device__ __forceinline__ uint32_t choose_store_format(uint32_t lo, uint32_t hi, int fmt_req, int allow);

__global__ __launch_bounds__(256) void strip_levels_kernel(const uint16_t *__restrict__ pcount,
                                                           const uint32_t *__restrict__ perm,
                                                           StripMeta *__restrict__ meta, uint32_t *__restrict__ tile_levels,
                                                           const uint64_t *__restrict__ total_levels, int fmt_req, int allow) {
    __shared__ uint32_t lv[kStripsPerTile];
    // (the store's format is a function of the ranges' span, which view_total_kernel has left behind: every workgroup works it
    // out for itself; tile_offset_kernel records it)
    const uint32_t *fw = reinterpret_cast<const uint32_t *>(total_levels + 1);
    const int fmt = (int)choose_store_format(fw[2], fw[3], fmt_req, allow);
    const int t = threadIdx.x;
    uint32_t mx = pcount[perm[(size_t)blockIdx.x * kTilePx + t]], mn = mx;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, off, 64));
    }
    if ((t & 63) == 0) {
        StripMeta *m = meta + (size_t)blockIdx.x * kStripsPerTile + (t >> 6);
        m->levels = mx;
        m->full = mn;
        lv[t >> 6] = mx;
    }
    __syncthreads();
    if (t == 0) tile_levels[blockIdx.x] = padded_levels(fmt, lv[0]) + padded_levels(fmt, lv[1]) + padded_levels(fmt, lv[2]) + padded_levels(fmt, lv[3]);
}

// What the store will be: the format the caller asked for, except that float32 ranges become range CODES (layout.h) when the
// caller allows it: 24-bit ones (kStoreZ24) when every range of the image lies within 2^24 - 2 bit patterns of the smallest
// one, 26-bit ones (kStoreZ26) within 2^26 - 2 (view_total_kernel left both ends behind the format word).  allow: bit 0 = 24-bit
// codes, bit 1 = 26-bit codes may be chosen.
__device__ __forceinline__ uint32_t choose_store_format(uint32_t lo, uint32_t hi, int fmt_req, int allow) {
    if (fmt_req != kStoreF32 || !(hi >= lo && lo >= 2u)) return (uint32_t)fmt_req;
    const uint32_t span = hi - lo;
    if ((allow & 1) && span <= 0xfffffdu) return (uint32_t)kStoreZ24;
    if ((allow & 2) && span <= 0x3fffffdu) return (uint32_t)kStoreZ26;
    return (uint32_t)kStoreF32;
}

// One thread records it; the scatter, the plan and the fit kernels read the two words.
__device__ __forceinline__ void decide_store_format(uint64_t *total_levels, int fmt_req, int allow) {
    uint32_t *w = reinterpret_cast<uint32_t *>(total_levels + 1);   // [0] format, [1] code offset, [2] smallest, [3] largest range bits
    const uint32_t f = choose_store_format(w[2], w[3], fmt_req, allow);
    w[0] = f;
    w[1] = (f == (uint32_t)kStoreZ24 || f == (uint32_t)kStoreZ26) ? w[2] - 1u : 0u;
}

// 3b. where every strip's chunks start, in levels: exclusive scan of the sorted tiles' totals by one workgroup of 1024
//     (coalesced reads: thread t takes a run of consecutive tiles), then every tile spreads its offset over its four
//     strips.  (Scanning the 16-byte StripMeta records themselves in one workgroup took 62 us per image.)
__global__ __launch_bounds__(1024) void tile_offset_kernel(const uint32_t *__restrict__ tile_levels, int n_tiles,
                                                           uint64_t *__restrict__ tile_off, uint64_t *__restrict__ total_levels,
                                                           int fmt, int allow) {
    __shared__ unsigned long long wave_base[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (n_tiles + 1023) / 1024;
    const int lo = min(t * per, n_tiles), hi = min(lo + per, n_tiles);
    unsigned long long s = 0;
    for (int i0 = lo; i0 < hi; i0 += 8) {   // eight loads in flight, not one after the other
        uint32_t v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = i0 + r < hi ? tile_levels[i0 + r] : 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += v[r];
    }
    const unsigned long long incl = wave_inclusive_scan(s);
    if (lane == 63) wave_base[wave] = incl;
    __syncthreads();
    unsigned long long run = incl - s, total = 0;
    for (int w = 0; w < 16; ++w) {
        const unsigned long long v = wave_base[w];
        if (w < wave) run += v;
        total += v;
    }
    if (t == 0) {
        *total_levels = total;
        decide_store_format(total_levels, fmt, allow);  // what the scatter, the plan and the fit kernels must be told
    }
    for (int i0 = lo; i0 < hi; i0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = i0 + r < hi ? tile_levels[i0 + r] : 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (i0 + r < hi) tile_off[i0 + r] = run;
            run += v[r];
        }
    }
}

__global__ __launch_bounds__(256) void strip_offset_kernel(StripMeta *__restrict__ meta, const uint64_t *__restrict__ tile_off,
                                                           int n_tiles, const uint32_t *__restrict__ store_fmt) {
    const int tile = blockIdx.x * 256 + threadIdx.x;
    if (tile >= n_tiles) return;
    const int fmt = (int)store_fmt[0];   // (tile_offset_kernel has recorded it)
    uint64_t run = tile_off[tile];
#pragma unroll
    for (int i = 0; i < kStripsPerTile; ++i) {
        StripMeta *m = meta + (size_t)tile * kStripsPerTile + i;
        m->lvoff = run;
        run += padded_levels(fmt, m->levels);
    }
}

// 3 (fewer than 255 views).  A pixel's bin IS its count then, so the sorted order's counts follow from the bins' totals
//     alone: bin b occupies the sorted positions [start_b, start_b + totals_b), a strip's largest count is the bin of
//     its first pixel, its smallest the bin of its last, and the strips that START inside bin b all have b levels -- the
//     offset of strip s is (levels of the strips starting in heavier bins) + (s - first strip of its bin) * b.  One launch,
//     one thread per strip, a 256-entry table per workgroup: replaces strip_levels (a gather through perm), the scan of
//     the tiles' totals and strip_offset (8 + 13 + 5 us per image).  With 255 views or more the bins are coarser than
//     the counts and the three kernels above do the work.
__global__ __launch_bounds__(256) void strip_table_kernel(const uint32_t *__restrict__ totals, int n_views, int n_strips,
                                                          StripMeta *__restrict__ meta, uint64_t *__restrict__ total_levels,
                                                          int fmt_req, int allow) {
    __shared__ uint32_t start[kMaxBins];              // entry i: the bin of count bins-1-i (heaviest first)
    // the store's format follows from the span of the ranges (in place since view_total_kernel): every workgroup works it out
    // for itself, workgroup 0 records it.  It decides how the strips are laid out: a kStoreZ26 strip starts on a whole chunk.
    const int fmt = (int)choose_store_format(reinterpret_cast<const uint32_t *>(total_levels + 1)[2], reinterpret_cast<const uint32_t *>(total_levels + 1)[3], fmt_req, allow);
    __shared__ uint32_t first[kMaxBins];              // first strip that starts inside the bin
    __shared__ unsigned long long before[kMaxBins];   // levels of all strips starting in heavier bins
    __shared__ uint32_t wsum[4];
    __shared__ unsigned long long wsum64[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bins = num_bins(n_views);
    {
        const uint32_t v = t < bins ? totals[bins - 1 - t] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t st = incl - v;
        for (int w = 0; w < wave; ++w) st += wsum[w];
        const uint32_t fs = (st + kStripPx - 1) / kStripPx, fe = (st + v + kStripPx - 1) / kStripPx;
        const unsigned long long lv = t < bins ? (unsigned long long)(fe - fs) * padded_levels(fmt, (uint32_t)(bins - 1 - t)) : 0ull;
        const unsigned long long incl64 = wave_inclusive_scan(lv);
        if (lane == 63) wsum64[wave] = incl64;
        __syncthreads();
        unsigned long long bf = incl64 - lv, total = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) bf += wsum64[w];
            total += wsum64[w];
        }
        start[t] = t < bins ? st : 0xffffffffu;
        first[t] = fs;
        before[t] = bf;
        if (blockIdx.x == 0 && t == 0) {
            *total_levels = total;
            decide_store_format(total_levels, fmt_req, allow);  // what the scatter, the plan and the fit kernels must be told
        }
        __syncthreads();
    }
    const int s = blockIdx.x * 256 + t;
    if (s >= n_strips) return;
    auto bin_at = [&](uint32_t p) {   // last i with start[i] <= p
        int lo = 0, hi = kMaxBins;    // start[lo] <= p < start[hi] (start[0] = 0; entries past the bins are UINT_MAX)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (start[mid] <= p) lo = mid; else hi = mid;
        }
        return lo;
    };
    const uint32_t p = (uint32_t)s * kStripPx;
    const int i = bin_at(p), j = bin_at(p + kStripPx - 1);
    const uint32_t levels = (uint32_t)(bins - 1 - i);
    StripMeta m;
    m.lvoff = before[i] + (unsigned long long)((uint32_t)s - first[i]) * padded_levels(fmt, levels);
    m.levels = levels;
    m.full = (uint32_t)(bins - 1 - j);
    meta[s] = m;
}

// 3c. the compaction itself, driven from the DENSE side.  One workgroup owns one dense tile: it stages the tile's
//     chunks through LDS a group of views at a time -- every chunk is read from HBM exactly once, fully coalesced --
//     and each thread (pixel) then walks its view bitmask (next set bit = next level), collects four levels at a
//     time and writes them to its pixel's place in its strip: one 16-byte store of ranges and three dwords of
//     colours per full chunk (StripMeta / layout.h).  Pixels of one dense tile that fall into the same count bin are
//     neighbours in the sorted order (the sort is stable), so these writes form runs.  (History: the first version
//     gathered from the sorted side, one thread per sorted slot reading 4 + 3 x 1 bytes per observation from a
//     different (tile, view) chunk; every dense chunk was re-read by the ~7 sorted tiles holding some of its pixels
//     -- 3.9 GB fetched for 0.55 GB of observations (rocprofv3 FETCH_SIZE), HBM-bound at 1.13 ms.)
//     kFmt = SUCRE_OBS_U16MM: the range is stored as uint16 millimetres, rint(1000 z) clamped to [1, 65535]
//     (0 stays the empty-slot marker), colours unchanged.
//     kExt = 1, 2: that many sets of three float32 extension planes per chunk ride along, 3072 more bytes per set and
//     staged view (one set: the light model's camera points or float32 colours; two: both).
template <int kExt>
struct StageCfg {
    // 16 views of 1792 B = 28 KB of LDS (5 workgroups per CU; 32 views = 2 workgroups per CU were 0.07 ms slower per
    // image, 8 views no better) / 16 x 4864 B = 76 KB / 8 x 7936 B = 62 KB
    static constexpr int kViews = kExt == 0 ? 16 : (kExt == 1 ? 16 : 8);
    static constexpr int kBytes = kChunk + kExt * kExtChunk;
};

// Four consecutive levels of one pixel on their way to the strip store.
template <int kExt>
struct LevelGroup {
    float z[kGroupLv];
    uint32_t c[3];              // four bytes per colour plane, level j in byte j
    float e[kExt ? 3 * kExt : 1][kGroupLv];
};

template <int kFmt>
__device__ __forceinline__ uint32_t range_mm(float z) {
    const float mm = fminf(fmaxf(rintf(z * kMmPerM), 1.0f), 65535.0f);
    return z > 0.0f ? (uint32_t)mm : 0u;
}

// Code of a range (layout.h, kStoreZ24 / kStoreZ26): bits - offset, 0 for an empty slot.
__device__ __forceinline__ uint32_t range_code(float z, uint32_t zoff) { return z > 0.0f ? __float_as_uint(z) - zoff : 0u; }

// Writes chunk g (r of its four levels exist in the strip) of pixel `lane` of a store of range codes (the device's choice for
// a float32 store without extension planes): kZ = 24 or 26 bits.
template <int kZ>
__device__ __forceinline__ void store_group_codes(uint8_t *strip, uint32_t g, uint32_t r, uint32_t lane, const float (&z)[kGroupLv],
                                                  const uint32_t (&c)[3], uint32_t zoff) {
    uint8_t *ch = strip + (size_t)g * (kZ == 26 ? kChunk26 : kChunk24);
    if (r == kGroupLv) {
        // the lane's 24 bytes: four dwords {low 24 bits of the code of level j | red of level j << 24} (the reader takes them with
        // one v_mad_u32_u24 and the red byte with v_cvt_f32_ubyte3), then the G and B words
        const uint32_t k0 = range_code(z[0], zoff), k1 = range_code(z[1], zoff), k2 = range_code(z[2], zoff), k3 = range_code(z[3], zoff);
        uint32_t *p = reinterpret_cast<uint32_t *>(ch + lane * 24);
        *reinterpret_cast<uint4 *>(p) = make_uint4((k0 & 0xffffffu) | (c[0] << 24), (k1 & 0xffffffu) | ((c[0] >> 8) << 24),
                                                   (k2 & 0xffffffu) | ((c[0] >> 16) << 24), (k3 & 0xffffffu) | ((c[0] >> 24) << 24));
        *reinterpret_cast<uint2 *>(p + 4) = make_uint2(c[1], c[2]);
        if (kZ == 26) ch[kChunk24 + lane] = (uint8_t)((k0 >> 24) | ((k1 >> 24) << 2) | ((k2 >> 24) << 4) | ((k3 >> 24) << 6));   // bits 24-25 of the four codes
    } else {
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)kGroupLv - 1u; ++j) {
            if (j >= r) break;
            const uint32_t code = range_code(z[j], zoff);
            uint8_t *zp = ch + (lane * r + j) * 3u;
            zp[0] = (uint8_t)code; zp[1] = (uint8_t)(code >> 8); zp[2] = (uint8_t)(code >> 16);
            uint8_t *cb = ch + 3u * kStripPx * r;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) cb[pl * kStripPx * r + lane * r + j] = (uint8_t)(c[pl] >> (8 * j));
            if (kZ == 26 && (code >> 24)) {
                // bits 24-25: two-bit field lane r + j, sixteen fields to a dword shared by several pixels (of this and of other
                // workgroups): OR-ed into words tail_bits_clear_kernel has zeroed
                const uint32_t f = lane * r + j;
                atomicOr(reinterpret_cast<uint32_t *>(ch + 6u * kStripPx * r) + (f >> 4), (code >> 24) << (2u * (f & 15u)));
            }
        }
    }
}

// kStoreZ26: the two-bit fields of every strip's short last chunk start out as zeros (the scatter ORs into them).  One thread
// per strip; returns at once for any other store.
__global__ __launch_bounds__(256) void tail_bits_clear_kernel(const StripMeta *__restrict__ meta, int n_strips, uint8_t *__restrict__ comp,
                                                              const uint32_t *__restrict__ store_fmt) {
    if (store_fmt[0] != (uint32_t)kStoreZ26) return;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_strips) return;
    const StripMeta m = meta[s];
    const uint32_t r = m.levels & 3u;
    if (r == 0u) return;
    uint32_t *w = reinterpret_cast<uint32_t *>(comp + (m.lvoff + (m.levels & ~3u)) * (uint64_t)level_bytes(kStoreZ26) + 6u * kStripPx * r);
    for (uint32_t i = 0; i < 4u * r; ++i) w[i] = 0u;   // 64 r fields of two bits = 16 r bytes
}

template <int kFmt, int kExt>
__device__ __forceinline__ void store_group(uint8_t *strip, float *const (&estrip)[2], uint32_t g, uint32_t r, uint32_t lane,
                                            const LevelGroup<kExt> &q) {
    uint8_t *ch = strip + (size_t)g * (kGroupLv * level_bytes(kFmt));
    constexpr uint32_t zb = kFmt ? 2 : 4;   // bytes per range
    if (r == kGroupLv) {
        if (kFmt) {
            *reinterpret_cast<uint2 *>(ch + lane * 8) =
                make_uint2(range_mm<kFmt>(q.z[0]) | (range_mm<kFmt>(q.z[1]) << 16), range_mm<kFmt>(q.z[2]) | (range_mm<kFmt>(q.z[3]) << 16));
        } else {
            *reinterpret_cast<float4 *>(ch + lane * 16) = make_float4(q.z[0], q.z[1], q.z[2], q.z[3]);
        }
        // the lane's three colour words side by side: one 12-byte store (three 4-byte stores to three planes until round 3:
        // the kernel's stores go 64 scattered lanes at a time through the same texture addresser as its loads)
        uint32_t *cp = reinterpret_cast<uint32_t *>(ch + zb * kStripPx * kGroupLv) + 3 * lane;
        cp[0] = q.c[0]; cp[1] = q.c[1]; cp[2] = q.c[2];
#pragma unroll
        for (int set = 0; set < kExt; ++set) {
            float *e = estrip[set] + (size_t)g * (kGroupLv * kExtLevelBytes / 4);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<float4 *>(e + pl * (kStripPx * kGroupLv) + lane * 4) =
                    make_float4(q.e[3 * set + pl][0], q.e[3 * set + pl][1], q.e[3 * set + pl][2], q.e[3 * set + pl][3]);
        }
    } else {   // the strip's last chunk: r < 4 levels, same arrangement with rows of r
#pragma unroll   // constant trip count: q stays in registers (a loop up to r indexes it dynamically -> scratch)
        for (uint32_t j = 0; j < (uint32_t)kGroupLv - 1u; ++j) {
            if (j >= r) break;
            if (kFmt) reinterpret_cast<uint16_t *>(ch)[lane * r + j] = (uint16_t)range_mm<kFmt>(q.z[j]);
            else reinterpret_cast<float *>(ch)[lane * r + j] = q.z[j];
            uint8_t *cb = ch + zb * kStripPx * r;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) cb[pl * kStripPx * r + lane * r + j] = (uint8_t)(q.c[pl] >> (8 * j));
#pragma unroll
            for (int set = 0; set < kExt; ++set) {
                float *e = estrip[set] + (size_t)g * (kGroupLv * kExtLevelBytes / 4);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) e[pl * kStripPx * r + lane * r + j] = q.e[3 * set + pl][j];
            }
        }
    }
}

template <int kExt>
__device__ __forceinline__ void clear_group(LevelGroup<kExt> &q) {
#pragma unroll
    for (int j = 0; j < kGroupLv; ++j) {
        q.z[j] = 0.0f;
#pragma unroll
        for (int pl = 0; pl < (kExt ? 3 * kExt : 1); ++pl) q.e[pl][j] = 0.0f;
    }
    q.c[0] = q.c[1] = q.c[2] = 0u;
}

// Position (0..15) of the (n+1)-th set bit of a 16-bit mask that has more than n bits set.
__device__ __forceinline__ int nth_set_bit(uint32_t p, int n) {
    int pos = 0;
    int c = __builtin_popcount(p & 0xffu);
    if (n >= c) { n -= c; pos = 8; p >>= 8; }
    c = __builtin_popcount(p & 0xfu);
    if (n >= c) { n -= c; pos += 4; p >>= 4; }
    c = __builtin_popcount(p & 0x3u);
    if (n >= c) { n -= c; pos += 2; p >>= 2; }
    return pos + ((n >= (int)(p & 1u)) ? 1 : 0);
}

__device__ unsigned long long g_exp_scatter_clock[2];   // (experiment build) shader cycles / 100 MHz ticks, summed over the workgroups

// kZ = 24 / 26: the instantiations that write range codes.  Which of the three forms a float32 store takes is decided on the
// device (decide_store_format), so all allowed instantiations are launched and the ones whose form it is not return at once (a
// few microseconds per image; one kernel with two store paths in it held 100 registers instead of 92 and lost a wave per SIMD).
template <int kFmt, int kExt, int kZ = 0>
__global__ __launch_bounds__(256) void scatter_kernel(const uint8_t *__restrict__ obs, size_t tile_stride,
                                                      size_t view_stride, const uint32_t *__restrict__ invperm,
                                                      const uint64_t *__restrict__ pmask, int mask_words,
                                                      const StripMeta *__restrict__ meta,
                                                      uint8_t *__restrict__ comp, int n_views,
                                                      const uint8_t *__restrict__ ext_dense,
                                                      uint8_t *__restrict__ ext_comp,
                                                      const uint8_t *__restrict__ ext2_dense,
                                                      uint8_t *__restrict__ ext2_comp, const uint32_t *__restrict__ store_fmt) {
    static_assert(kZ == 0 || ((kZ == 24 || kZ == 26) && kFmt == 0 && kExt == 0), "range codes: float32 store without extension planes");
    // the other instantiations' store (kernel-uniform): kZ = 0 writes whatever is no store of codes
    if (kZ ? store_fmt[0] != (uint32_t)(kZ == 24 ? kStoreZ24 : kStoreZ26) : (store_fmt[0] == (uint32_t)kStoreZ24 || store_fmt[0] == (uint32_t)kStoreZ26)) return;
    const unsigned long long exp_c0 = kExpWaveTimes ? clock64() : 0ull, exp_t0 = kExpWaveTimes ? wall_clock64() : 0ull;
    constexpr int kStageViews = StageCfg<kExt>::kViews, kStageBytes = StageCfg<kExt>::kBytes;
    constexpr uint32_t kStageMask = (1u << kStageViews) - 1u;
    static_assert(kStageViews == 8 || kStageViews == 16, "a stage group is a byte or a half word of the 32-bit presence words");
    __shared__ __attribute__((aligned(16))) uint8_t stage[kStageViews][kStageBytes];
    __shared__ uint32_t tpres[kMaxViews / 32];   // bit k: some pixel of the tile has a kept observation in view k
    const int tile = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < 2 * mask_words; i += 256) tpres[i] = 0u;
    __syncthreads();
    const uint32_t dst = invperm[(size_t)tile * kTilePx + t];
    const uint32_t dlane = dst % kStripPx;
    const StripMeta sm = meta[dst / kStripPx];
    const uint32_t nl = sm.levels;
    // a float32 store without extension planes may have been made a store of 24-bit codes (decide_store_format)
    constexpr bool z24 = kZ != 0;   // (a store of range codes, 24 or 26 bits)
    const uint32_t zoff = store_fmt[1];
    uint8_t *out = comp + sm.lvoff * (uint64_t)(kZ == 24 ? level_bytes(kStoreZ24) : kZ == 26 ? level_bytes(kStoreZ26) : level_bytes(kFmt));
    float *const eout[2] = {kExt >= 1 ? reinterpret_cast<float *>(ext_comp + sm.lvoff * (uint64_t)kExtLevelBytes) : nullptr,
                            kExt >= 2 ? reinterpret_cast<float *>(ext2_comp + sm.lvoff * (uint64_t)kExtLevelBytes) : nullptr};
    const uint64_t *mask = pmask + ((size_t)tile * kTilePx + t) * mask_words;
    const uint8_t *tbase = obs + (size_t)tile * tile_stride;
    for (int wv = 0; wv < mask_words; ++wv) {
        const uint64_t m = mask[wv];
        if ((uint32_t)m) atomicOr(&tpres[2 * wv], (uint32_t)m);
        if ((uint32_t)(m >> 32)) atomicOr(&tpres[2 * wv + 1], (uint32_t)(m >> 32));
    }
    __syncthreads();
    constexpr int kUnits = kStageBytes / 16;  // 16-byte pieces of a staged view
    constexpr int kMainUnits = kChunk / 16;
    constexpr int kExtUnits = kExtChunk / 16;
    // The views of a group that some pixel of the tile uses are staged through LDS: every dense chunk is read once,
    // coalesced.  Without extension planes the NEXT group's pieces are fetched into registers while this group is being
    // scattered (7 x 16 bytes per thread), so the fetch latency hides behind the stores instead of preceding them.
    constexpr bool kPrefetch = kExt == 0;
    constexpr int kIters = (kStageViews * kUnits + 255) / 256;
    auto present = [&](int g0) -> uint32_t { return g0 < n_views ? (tpres[g0 >> 5] >> (g0 & 31)) & kStageMask : 0u; };
    auto next_group = [&](int g0) { while (g0 < n_views && present(g0) == 0u) g0 += kStageViews; return g0; };   // workgroup-uniform
    auto piece = [&](int g0, uint32_t p, int u) -> const uint8_t * {
        const int sl = u / kUnits, w = u - sl * kUnits;
        const int k = g0 + nth_set_bit(p, sl);
        return (!kExt || w < kMainUnits)
            ? tbase + (size_t)k * view_stride + (size_t)w * 16
            : (kExt < 2 || w < kMainUnits + kExtUnits)
                ? ext_dense + ((size_t)tile * n_views + k) * kExtChunk + (size_t)(w - kMainUnits) * 16
                : ext2_dense + ((size_t)tile * n_views + k) * kExtChunk + (size_t)(w - kMainUnits - kExtUnits) * 16;
    };
    static_assert(!kPrefetch || kIters <= 8, "eight named prefetch registers");
    uint4 r0, r1, r2, r3, r4, r5, r6, r7;   // named, not an array: an indexed uint4 array ended up in scratch memory
    r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = make_uint4(0u, 0u, 0u, 0u);
#define SUCRE_EACH_REG(X) X(0, r0) X(1, r1) X(2, r2) X(3, r3) X(4, r4) X(5, r5) X(6, r6) X(7, r7)
    uint32_t lv = 0;
    LevelGroup<kExt> q;
    clear_group(q);
    int g0 = next_group(0);
    if (kPrefetch && g0 < n_views) {
        const uint32_t pf = present(g0);
        const int total_f = __builtin_popcount(pf) * kUnits;
#define SUCRE_X(IT, R) if (IT < kIters && t + 256 * IT < total_f) R = *reinterpret_cast<const uint4 *>(piece(g0, pf, t + 256 * IT));
        SUCRE_EACH_REG(SUCRE_X)
#undef SUCRE_X
    }
    while (g0 < n_views) {
        const uint32_t p = present(g0);
        if (kPrefetch) {
            __syncthreads();  // everyone is done with the stage of the previous group
            const int total_s = __builtin_popcount(p) * kUnits;   // staged slot sl holds the (sl+1)-th present view of the group
#define SUCRE_X(IT, R) if (IT < kIters && t + 256 * IT < total_s) *reinterpret_cast<uint4 *>(&stage[(t + 256 * IT) / kUnits][((t + 256 * IT) % kUnits) * 16]) = R;
            SUCRE_EACH_REG(SUCRE_X)
#undef SUCRE_X
        } else {
            __syncthreads();
            const int total = __builtin_popcount(p) * kUnits;
            for (int u = t; u < total; u += 256)
                *reinterpret_cast<uint4 *>(&stage[u / kUnits][(u % kUnits) * 16]) = *reinterpret_cast<const uint4 *>(piece(g0, p, u));
        }
        __syncthreads();
        const int gn = next_group(g0 + kStageViews);
        if (kPrefetch && gn < n_views) {
            const uint32_t pf = present(gn);
            const int total_f = __builtin_popcount(pf) * kUnits;
#define SUCRE_X(IT, R) if (IT < kIters && t + 256 * IT < total_f) R = *reinterpret_cast<const uint4 *>(piece(gn, pf, t + 256 * IT));
            SUCRE_EACH_REG(SUCRE_X)
#undef SUCRE_X
        }
        uint32_t m = (uint32_t)(mask[g0 >> 6] >> (g0 & 63)) & kStageMask;   // this pixel's views of the group
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1u;
            const int i = __builtin_popcount(p & ((1u << b) - 1u));   // the staged slot of view g0 + b
            const uint32_t j = lv & 3u;
            // select-style updates (no dynamic register indexing)
            const float z = reinterpret_cast<const float *>(&stage[i][0])[t];
            const uint32_t cr = stage[i][kChunkZ + t], cg = stage[i][kChunkZ + kTilePx + t], cb = stage[i][kChunkZ + 2 * kTilePx + t];
#pragma unroll
            for (int jj = 0; jj < kGroupLv; ++jj)
                if (j == (uint32_t)jj) {
                    q.z[jj] = z;
                    if (kExt) {
                        const float *se = reinterpret_cast<const float *>(&stage[i][kChunk]);   // the sets follow each other
#pragma unroll
                        for (int pl = 0; pl < 3 * kExt; ++pl) q.e[pl][jj] = se[pl * kTilePx + t];
                    }
                }
            q.c[0] |= cr << (8 * j); q.c[1] |= cg << (8 * j); q.c[2] |= cb << (8 * j);
            ++lv;
            if ((lv & 3u) == 0u) {   // four levels collected: this pixel's share of chunk lv/4 - 1 (a full one)
                if (z24) store_group_codes<kZ ? kZ : 24>(out, (lv >> 2) - 1u, kGroupLv, dlane, q.z, q.c, zoff);
                else store_group<kFmt, kExt>(out, eout, (lv >> 2) - 1u, kGroupLv, dlane, q);
                clear_group(q);
            }
        }
        g0 = gn;
    }
    // the open group (zero-filled past the pixel's last observation), then the all-zero groups up to the strip's
    // level count: the strip has as many levels as its richest pixel
    for (uint32_t g = lv >> 2; g * kGroupLv < nl; ++g) {
        const uint32_t r = min((uint32_t)kGroupLv, nl - g * kGroupLv);
        if (z24) store_group_codes<kZ ? kZ : 24>(out, g, r, dlane, q.z, q.c, zoff);
        else store_group<kFmt, kExt>(out, eout, g, r, dlane, q);
        clear_group(q);
    }
#undef SUCRE_EACH_REG
    if (kExpWaveTimes && t == 0) {   // experiment.h SUCRE_EXP_WAVE_TIMES: the shader clock the kernel ran at (tools/exp/scatter_clock.py)
        atomicAdd(&g_exp_scatter_clock[0], clock64() - exp_c0);
        atomicAdd(&g_exp_scatter_clock[1], wall_clock64() - exp_t0);
    }
}

}  // namespace sucre
SUCRE_EXP_EXPORT int sucre_exp_scatter_clock(unsigned long long *out) {   // exported by the experiment build only; reads and clears
    const int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sucre::g_exp_scatter_clock), sizeof(sucre::g_exp_scatter_clock));
    const unsigned long long zero[2] = {0ull, 0ull};
    return rc ? rc : (int)hipMemcpyToSymbol(HIP_SYMBOL(sucre::g_exp_scatter_clock), zero, sizeof(zero));
}
namespace sucre {

hipError_t launch_compact(const Layout &L, uint8_t *ws, hipStream_t s, const uint8_t *ext_dense, uint8_t *ext_comp,
                          int fmt, const uint8_t *ext2_dense, uint8_t *ext2_comp) {
    auto *keep = reinterpret_cast<const uint32_t *>(ws + L.off_view_keep);
    auto *pcount = reinterpret_cast<uint16_t *>(ws + L.off_pcount);
    auto *pmask = reinterpret_cast<uint64_t *>(ws + L.off_pmask);
    auto *blockhist = reinterpret_cast<uint32_t *>(ws + L.off_blockhist);
    auto *totals = reinterpret_cast<uint32_t *>(ws + L.off_bin_totals);
    auto *perm = reinterpret_cast<uint32_t *>(ws + L.off_perm);
    auto *invperm = reinterpret_cast<uint32_t *>(ws + L.off_invperm);
    auto *meta = reinterpret_cast<StripMeta *>(ws + L.off_strip_meta);
    const int bins = num_bins(L.n_views);
    // fmt: what the caller asked for (include/sucre_hip.h): SUCRE_OBS_F32 = float32 ranges, kept as 24-bit codes when the image
    // allows it and nothing rides along in extension planes; SUCRE_OBS_U16MM; SUCRE_OBS_F32_PLAIN = float32 ranges as they are
    const int store_req = fmt == SUCRE_OBS_U16MM ? kStoreU16 : kStoreF32;
    // range codes the device may choose: bit 0 = 24-bit, bit 1 = 26-bit.  The default store takes the 24-bit codes or the words:
    // the 26-bit codes (SUCRE_OBS_F32_Z26: those or the words) are built, bit-exact and measured SLOWER than the words they would
    // replace -- 139.3 against 135.9 us per launch at 1080p x 65 views, 90.0 against 86.3 us on a scene whose ranges span 0.7-8 m
    // (profiles/r06_jparam_f32z26_*, r06_jparam_deep_*): eight more vector instructions per chunk cost more than 192 fewer bytes return
    const int allow = ext_dense ? 0 : fmt == SUCRE_OBS_F32 ? 1 : fmt == SUCRE_OBS_F32_Z26 ? 2 : 0;
    const uint32_t *store_fmt = reinterpret_cast<const uint32_t *>(ws + L.off_total_chunks) + 2;
    hipLaunchKernelGGL(pixel_count_kernel, dim3(L.n_tiles), dim3(256), 0, s, reinterpret_cast<const uint64_t *>(ws + L.off_vbits),
                       pmask, L.mask_words, keep, L.n_views, L.n_tiles, pcount, blockhist);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(bins), dim3(256), 0, s, blockhist, L.n_tiles, totals);
    hipLaunchKernelGGL(permute_kernel, dim3(L.n_tiles), dim3(256), 0, s, pcount, blockhist, totals, L.n_views,
                       L.n_tiles, perm, invperm);
    if (L.n_views < kMaxBins - 1) {
        hipLaunchKernelGGL(strip_table_kernel, dim3((L.n_strips + 255) / 256), dim3(256), 0, s, totals, L.n_views, L.n_strips,
                           meta, reinterpret_cast<uint64_t *>(ws + L.off_total_chunks), store_req, allow);
    } else {
        // the counting sort is done with its histograms: their space holds the sorted tiles' totals and offsets
        auto *tile_levels = blockhist;
        auto *tile_off = reinterpret_cast<uint64_t *>(blockhist + align_up((size_t)L.n_tiles, 2));
        hipLaunchKernelGGL(strip_levels_kernel, dim3(L.n_tiles), dim3(256), 0, s, pcount, perm, meta, tile_levels,
                           reinterpret_cast<const uint64_t *>(ws + L.off_total_chunks), store_req, allow);
        hipLaunchKernelGGL(tile_offset_kernel, dim3(1), dim3(1024), 0, s, tile_levels, L.n_tiles, tile_off,
                           reinterpret_cast<uint64_t *>(ws + L.off_total_chunks), store_req, allow);
        hipLaunchKernelGGL(strip_offset_kernel, dim3((L.n_tiles + 255) / 256), dim3(256), 0, s, meta, tile_off, L.n_tiles, store_fmt);
    }
    const dim3 grid(L.n_tiles), block(256);
    const uint8_t *obs = ws + L.off_obs;
    uint8_t *comp = ws + L.off_comp;
    if (ext_dense && ext2_dense)  // light model on float32 colours: camera points and colours ride along
        hipLaunchKernelGGL((scatter_kernel<0, 2>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                           pmask, L.mask_words, meta, comp, L.n_views, ext_dense, ext_comp, ext2_dense, ext2_comp, store_fmt);
    else if (ext_dense)  // light model / float32 colours: float32 store only
        hipLaunchKernelGGL((scatter_kernel<0, 1>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                           pmask, L.mask_words, meta, comp, L.n_views, ext_dense, ext_comp, ext2_dense, ext2_comp, store_fmt);
    else if (fmt == SUCRE_OBS_U16MM)
        hipLaunchKernelGGL((scatter_kernel<1, 0>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                           pmask, L.mask_words, meta, comp, L.n_views, ext_dense, ext_comp, ext2_dense, ext2_comp, store_fmt);
    else {
        hipLaunchKernelGGL((scatter_kernel<0, 0, 0>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                           pmask, L.mask_words, meta, comp, L.n_views, ext_dense, ext_comp, ext2_dense, ext2_comp, store_fmt);
        if (allow & 1)
            hipLaunchKernelGGL((scatter_kernel<0, 0, 24>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                               pmask, L.mask_words, meta, comp, L.n_views, ext_dense, ext_comp, ext2_dense, ext2_comp, store_fmt);
        if (allow & 2) {
            hipLaunchKernelGGL(tail_bits_clear_kernel, dim3((L.n_strips + 255) / 256), dim3(256), 0, s, meta, L.n_strips, comp, store_fmt);
            hipLaunchKernelGGL((scatter_kernel<0, 0, 26>), grid, block, 0, s, obs, L.obs_tile_stride, L.obs_view_stride, invperm,
                               pmask, L.mask_words, meta, comp, L.n_views, ext_dense, ext_comp, ext2_dense, ext2_comp, store_fmt);
        }
    }
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    return launch_plan(L, ws, s);   // the fit waves' item streams over the store just written (fit.hip); it reads the store's format
}

}  // namespace sucre
