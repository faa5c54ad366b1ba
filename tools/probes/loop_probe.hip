// Probe: the fit kernel's inner loop (accumulate_view on LDS-resident chunks) in isolation, at 1..8 waves per SIMD.
// Answers: is the loop VALU-throughput bound, and at what cost per chunk?   hipcc -O3 --offload-arch=gfx950 ...
#include "../../sucre_amd/csrc/fit.hip"
#include <cstdio>
using namespace sucre;

// MODE 0: the product's accumulate_view<kPassGradJ, false>; 1: same without the two exp (a, g from one fma each);
// 2: masked variant; 3: closed-form one-pass accumulate (AccOne)
template <int MODE>
__device__ __forceinline__ void body(const float4 z4, const uint3 c3, const Water &w, const float (&J)[3][4], Acc &acc) {
    if (MODE == 0) accumulate_view<kPassGradJ, false>(z4, c3, w, J, acc);
    if (MODE == 2) accumulate_view<kPassGradJ, true>(z4, c3, w, J, acc);
    if (MODE == 1) {
        const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
        const uint32_t cc[3] = {c3.x, c3.y, c3.z};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = zz[j];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t k = (cc[c] >> (8 * j)) & 255u;
                const float a = __builtin_fmaf(z, w.nb[c], 1.0f);
                const float g = __builtin_fmaf(z, w.ng[c], 1.0f);
                const float omg = 1.0f - g;
                const float bt = w.B[c] * omg;
                const float Ihat = __builtin_fmaf(J[c][j], a, bt);
                float r = __builtin_fmaf((float)k, kInv255, -Ihat);
                const float rz = r * z;
                acc.cost = __builtin_fmaf(r, r, acc.cost);
                acc.pa[c][j] = __builtin_fmaf(r, a, acc.pa[c][j]);
                acc.pb[c][j] = __builtin_fmaf(rz, a, acc.pb[c][j]);
                acc.sB[c] = __builtin_fmaf(r, omg, acc.sB[c]);
                acc.sGZ[c] = __builtin_fmaf(rz, g, acc.sGZ[c]);
            }
        }
    }
}

// MODE 3: per pixel: 6 products, 6 exps, then the arithmetic (exps batched, results consumed later)
// MODE 4: all 24 products, all 24 exps, then the arithmetic
template <int MODE>
__device__ __forceinline__ void body_batched(const float4 z4, const uint3 c3, const Water &w, const float (&J)[3][4], Acc &acc) {
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
    const uint32_t cc[3] = {c3.x, c3.y, c3.z};
    float a[4][3], g[4][3];
    if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[j][c] = zz[j] * w.nb[c]; g[j][c] = zz[j] * w.ng[c]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[j][c] = fast_exp2(a[j][c]); g[j][c] = fast_exp2(g[j][c]); }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float z = zz[j];
        if (MODE == 3) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[j][c] = z * w.nb[c]; g[j][c] = z * w.ng[c]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 3; ++c) { a[j][c] = fast_exp2(a[j][c]); g[j][c] = fast_exp2(g[j][c]); }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t k = (cc[c] >> (8 * j)) & 255u;
            const float omg = 1.0f - g[j][c];
            const float bt = w.B[c] * omg;
            const float Ihat = __builtin_fmaf(J[c][j], a[j][c], bt);
            float r = __builtin_fmaf((float)k, kInv255, -Ihat);
            const float rz = r * z;
            acc.cost = __builtin_fmaf(r, r, acc.cost);
            acc.pa[c][j] = __builtin_fmaf(r, a[j][c], acc.pa[c][j]);
            acc.pb[c][j] = __builtin_fmaf(rz, a[j][c], acc.pb[c][j]);
            acc.sB[c] = __builtin_fmaf(r, omg, acc.sB[c]);
            acc.sGZ[c] = __builtin_fmaf(rz, g[j][c], acc.sGZ[c]);
        }
        if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

// MODE 5: exps only (24 mul + 24 exp + 24 adds to keep them alive); MODE 6: 24 exps of independent inputs, no mul
template <int MODE>
__device__ __forceinline__ void body_exponly(const float4 z4, const uint3 c3, const Water &w, Acc &acc) {
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc.pa[c][j] += fast_exp2(zz[j] * w.nb[c]);
            acc.pb[c][j] += fast_exp2(zz[j] * w.ng[c]);
        }
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, const float *params) {
    __shared__ FitLds lds;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int sl = 0; sl < kRing; ++sl) {
        *reinterpret_cast<float4 *>(&lds.u.ring[wave][sl][lane * 16]) =
            make_float4(2.5f + 0.001f * lane, 2.75f + 0.01f * sl, 3.0f, 3.25f + 0.002f * lane);
        if (lane < 48)
            *reinterpret_cast<uint4 *>(&lds.u.ring[wave][sl][kChunkZ + lane * 16]) =
                make_uint4(0x10203040u + lane, 0x50607080u + sl, 0x11223344u, 0x55667788u);
    }
    __syncthreads();
    const Water w = load_water(params);
    float J[3][4];
    for (int c = 0; c < 3; ++c) for (int j = 0; j < 4; ++j) J[c][j] = 0.3f + 0.01f * (c + j) + 0.001f * lane;
    Acc acc;
    zero_acc(acc);
    uint32_t slot = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");  // LDS contents are 'new' every iteration: no hoisting
        const uint8_t *sp = &lds.u.ring[wave][slot][0];
        const float4 z4 = *reinterpret_cast<const float4 *>(sp + lane * 16);
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(sp + kChunkZ) + lane;
        const uint3 c3 = make_uint3(cp[0], cp[64], cp[128]);
        if (MODE <= 2) body<MODE>(z4, c3, w, J, acc);
        else if (MODE <= 4) body_batched<MODE>(z4, c3, w, J, acc);
        else body_exponly<MODE>(z4, c3, w, acc);
        slot = slot + 1 == kRing ? 0 : slot + 1;
    }
    float s = acc.cost;
    for (int c = 0; c < 3; ++c) { s += acc.sB[c] + acc.sGZ[c]; for (int j = 0; j < 4; ++j) s += acc.pa[c][j] + acc.pb[c][j]; }
    out[blockIdx.x * 256 + t] = s;
}

template <int MODE>
void run(const char *name, int wgs_per_cu, const float *params, float *out) {
    const int iters = 4000;
    const int blocks = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, params);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, params);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // chunks per SIMD = waves per SIMD * iters
    const double ns_per_chunk_simd = ms * 1e6 / ((double)wgs_per_cu * iters);
    printf("%-28s waves/SIMD %d: %8.3f ms  -> %7.1f ns per chunk per SIMD  (C2: 301 chunks/SIMD -> %.1f us)\n", name, wgs_per_cu, ms,
           ns_per_chunk_simd, ns_per_chunk_simd * 301.2 / 1e3);
}

int main() {
    float hp[9] = {0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f};
    float *params, *out;
    hipMalloc(&params, sizeof(hp));
    hipMemcpy(params, hp, sizeof(hp), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4, 6, 8}) {
        run<0>("grad select-free", w, params, out);
        run<1>("grad without exp", w, params, out);
        run<3>("grad exps batched per pixel", w, params, out);
        run<4>("grad exps batched per chunk", w, params, out);
        run<5>("24 x (mul, exp, add) only", w, params, out);
    }
    return 0;
}
