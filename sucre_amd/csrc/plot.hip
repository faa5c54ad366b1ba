// Output stage on the device: exact order statistics of the restored image for SUCRe.plot_J (sucre.py:84-95).
//
// plot_J stretches every channel between its 1st and 99th percentile over the valid pixels (no NaN in any channel).
// numpy's percentile is a linear interpolation between two order statistics; the two ranks follow from the number of
// valid pixels on the host, in numpy's own arithmetic, and this file finds the values at those ranks without sorting
// and without moving the image: a most-significant-byte-first radix select on the order-preserving integer image of
// the float32 bit pattern, four passes of 256-bin histograms.  The interpolation itself stays on the host
// (sucre_amd/sucre.py), so the result is numpy's number bit for bit.
#include "launch.h"

namespace sucre {

constexpr int kMaxRanks = 8;

struct SelectState {
    uint32_t hist[3][kMaxRanks][256];   // pass 0 uses [c][0] for every rank of channel c
    uint32_t prefix[3][kMaxRanks];      // key bytes fixed so far, right-aligned
    uint64_t remaining[3][kMaxRanks];   // rank among the keys that share the prefix
};

__device__ __forceinline__ uint32_t order_key(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone: a < b  <=>  key(a) < key(b)
}

__device__ __forceinline__ float key_value(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// One pass: histogram of byte (3 - kPass) of every valid pixel's key whose higher bytes equal the rank's prefix.
template <int kPass>
__global__ __launch_bounds__(256) void select_pass_kernel(const float *__restrict__ J, long long n_px, int n_ranks,
                                                          SelectState *__restrict__ st) {
    __shared__ uint32_t h[3][kMaxRanks][256];
    const int nr = kPass == 0 ? 1 : n_ranks;
    for (int i = threadIdx.x; i < 3 * kMaxRanks * 256; i += 256) (&h[0][0][0])[i] = 0u;
    __shared__ uint32_t pre[3][kMaxRanks];
    if (threadIdx.x < 3 * kMaxRanks) (&pre[0][0])[threadIdx.x] = (&st->prefix[0][0])[threadIdx.x];
    __syncthreads();
    constexpr int shift = 24 - 8 * kPass;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n_px; p += (long long)gridDim.x * 256) {
        const float x0 = J[p * 3], x1 = J[p * 3 + 1], x2 = J[p * 3 + 2];
        if (x0 != x0 || x1 != x1 || x2 != x2) continue;   // np.all(~np.isnan(J), axis=2), sucre.py:87
        const uint32_t key[3] = {order_key(x0), order_key(x1), order_key(x2)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t byte = (key[c] >> shift) & 255u;
            for (int r = 0; r < nr; ++r)
                if (kPass == 0 || (key[c] >> (shift + 8)) == pre[c][r]) atomicAdd(&h[c][r][byte], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * kMaxRanks * 256; i += 256) {
        const uint32_t v = (&h[0][0][0])[i];
        if (v) atomicAdd(&st->hist[0][0][0] + i, v);
    }
}

// After a pass: the byte under which every rank falls; the histograms are cleared for the next pass.  One block.
template <int kPass>
__global__ __launch_bounds__(256) void select_locate_kernel(SelectState *__restrict__ st, int n_ranks, float *__restrict__ out) {
    const int t = threadIdx.x;
    if (t < 3 * n_ranks) {
        const int c = t / n_ranks, r = t % n_ranks;
        const uint32_t *h = st->hist[c][kPass == 0 ? 0 : r];
        uint64_t rem = st->remaining[c][r];
        uint32_t b = 0;
        for (; b < 255u; ++b) {
            if (rem < h[b]) break;
            rem -= h[b];
        }
        const uint32_t prefix = (kPass == 0 ? 0u : st->prefix[c][r] << 8) | b;
        st->prefix[c][r] = prefix;
        st->remaining[c][r] = rem;
        if (kPass == 3) out[c * n_ranks + r] = key_value(prefix);
    }
    __syncthreads();
    for (int i = t; i < 3 * kMaxRanks * 256; i += 256) (&st->hist[0][0][0])[i] = 0u;
}

__global__ void select_init_kernel(SelectState *__restrict__ st, int n_ranks, uint64_t r0, uint64_t r1, uint64_t r2, uint64_t r3,
                                   uint64_t r4, uint64_t r5, uint64_t r6, uint64_t r7) {
    const uint64_t ranks[kMaxRanks] = {r0, r1, r2, r3, r4, r5, r6, r7};
    for (int i = threadIdx.x; i < 3 * kMaxRanks * 256; i += blockDim.x) (&st->hist[0][0][0])[i] = 0u;
    if (threadIdx.x < 3 * kMaxRanks) {
        const int c = threadIdx.x / kMaxRanks, r = threadIdx.x % kMaxRanks;
        st->prefix[c][r] = 0u;
        st->remaining[c][r] = r < n_ranks ? ranks[r] : 0ull;
    }
}

size_t select_scratch_bytes() { return sizeof(SelectState); }

hipError_t launch_select_ranks(const float *J, int H, int W, int n_ranks, const uint64_t *ranks, float *out, void *scratch,
                               hipStream_t s) {
    auto *st = static_cast<SelectState *>(scratch);
    uint64_t r[kMaxRanks] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_ranks; ++i) r[i] = ranks[i];
    const long long n_px = (long long)H * W;
    const int grid = (int)((n_px + 255) / 256 < 1024 ? (n_px + 255) / 256 : 1024);
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, s, st, n_ranks, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    hipLaunchKernelGGL(select_pass_kernel<0>, dim3(grid), dim3(256), 0, s, J, n_px, n_ranks, st);
    hipLaunchKernelGGL(select_locate_kernel<0>, dim3(1), dim3(256), 0, s, st, n_ranks, out);
    hipLaunchKernelGGL(select_pass_kernel<1>, dim3(grid), dim3(256), 0, s, J, n_px, n_ranks, st);
    hipLaunchKernelGGL(select_locate_kernel<1>, dim3(1), dim3(256), 0, s, st, n_ranks, out);
    hipLaunchKernelGGL(select_pass_kernel<2>, dim3(grid), dim3(256), 0, s, J, n_px, n_ranks, st);
    hipLaunchKernelGGL(select_locate_kernel<2>, dim3(1), dim3(256), 0, s, st, n_ranks, out);
    hipLaunchKernelGGL(select_pass_kernel<3>, dim3(grid), dim3(256), 0, s, J, n_px, n_ranks, st);
    hipLaunchKernelGGL(select_locate_kernel<3>, dim3(1), dim3(256), 0, s, st, n_ranks, out);
    return hipGetLastError();
}

// The rest of plot_J (sucre.py:88-94) in one pass: clip to [lo, hi], shift, scale, times 255, truncate to uint8; pixels
// with a NaN in any channel come out black.  After the clip the smallest value IS lo and the largest IS hi (a
// percentile lies between the extreme order statistics, and np.clip returns the bound itself), so np.min of the clipped
// values is lo, np.max of the shifted ones is the float32 difference hi - lo: no reduction is needed, and every
// operation below is the same IEEE float32 operation numpy performs (this file is built without contraction).
__global__ __launch_bounds__(256) void plot_stretch_kernel(const float *__restrict__ J, long long n_px, float lo0, float lo1,
                                                           float lo2, float hi0, float hi1, float hi2,
                                                           uint8_t *__restrict__ out) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_px) return;
    const float v[3] = {J[p * 3], J[p * 3 + 1], J[p * 3 + 2]};
    const float lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
    const bool valid = !(__builtin_isnan(v[0]) || __builtin_isnan(v[1]) || __builtin_isnan(v[2]));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float x = v[c];
        x = x < lo[c] ? lo[c] : x;          // np.clip = minimum(maximum(x, lo), hi)
        x = x > hi[c] ? hi[c] : x;
        x = (x - lo[c]) / (hi[c] - lo[c]);
        out[p * 3 + c] = valid ? (uint8_t)(x * 255.0f) : (uint8_t)0;
    }
}

// Number of valid pixels (no NaN in any channel, sucre.py:87) -> *count (uint64, zeroed here).
__global__ __launch_bounds__(256) void count_valid_kernel(const float *__restrict__ J, long long n_px,
                                                          unsigned long long *__restrict__ count) {
    unsigned int mine = 0;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n_px; p += (long long)gridDim.x * 256)
        mine += !(__builtin_isnan(J[p * 3]) || __builtin_isnan(J[p * 3 + 1]) || __builtin_isnan(J[p * 3 + 2]));
    __shared__ unsigned int total;
    if (threadIdx.x == 0) total = 0u;
    __syncthreads();
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && total) atomicAdd(count, (unsigned long long)total);
}

hipError_t launch_plot_stretch(const float *J, int H, int W, const float *lo, const float *hi, uint8_t *out, hipStream_t s) {
    const long long n_px = (long long)H * W;
    hipLaunchKernelGGL(plot_stretch_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, s, J, n_px, lo[0], lo[1], lo[2],
                       hi[0], hi[1], hi[2], out);
    return hipGetLastError();
}

hipError_t launch_count_valid(const float *J, int H, int W, uint64_t *count, hipStream_t s) {
    const long long n_px = (long long)H * W;
    if (hipError_t e = hipMemsetAsync(count, 0, sizeof(uint64_t), s); e != hipSuccess) return e;
    const int grid = (int)((n_px + 255) / 256 < 1024 ? (n_px + 255) / 256 : 1024);
    hipLaunchKernelGGL(count_valid_kernel, dim3(grid), dim3(256), 0, s, J, n_px, reinterpret_cast<unsigned long long *>(count));
    return hipGetLastError();
}

}  // namespace sucre
