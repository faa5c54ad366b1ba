// Probe: the J-parameter chunk arithmetic of csrc/fit.hip (accumulate_chunk<false>: 4 levels x 3 channels, 24 exponentials
// back to back) on REGISTER data -- no LDS reads, no DMA, no plan -- at 1..6 waves per SIMD: how many shader cycles does a
// chunk cost a SIMD when nothing but the arithmetic runs?  (round 5; build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct Water { float B[3], nb[3], ng[3]; };
struct Acc { float pa[3], pb[3], sB[3], sGZ[3], cost; };
constexpr float kInv255 = (float)(1.0 / 255.0);

template <int kVariant>
__device__ __forceinline__ void chunk(const float (&zz)[4], const uint32_t (&cc)[3], const Water &w, const float (&J)[3], Acc &acc) {
    float ea[4][3], eg[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { ea[j][c] = zz[j] * w.nb[c]; eg[j][c] = zz[j] * w.ng[c]; }
    if (kVariant != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (kVariant == 1) { ea[j][c] = ea[j][c] * 0.5f + 1.0f; eg[j][c] = eg[j][c] * 0.5f + 1.0f; }   // exponentials replaced by FMAs
            else { ea[j][c] = __builtin_amdgcn_exp2f(ea[j][c]); eg[j][c] = __builtin_amdgcn_exp2f(eg[j][c]); }
        }
    if (kVariant != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float z = zz[j];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t k = (cc[c] >> (8 * j)) & 255u;
            const float a = ea[j][c], g = eg[j][c];
            const float omg = 1.0f - g;
            const float Ihat = __builtin_fmaf(J[c], a, w.B[c] * omg);
            float r = __builtin_fmaf((float)k, kInv255, -Ihat);
            const float rz = r * z;
            acc.cost = __builtin_fmaf(r, r, acc.cost);
            acc.pa[c] = __builtin_fmaf(r, a, acc.pa[c]);
            acc.pb[c] = __builtin_fmaf(rz, a, acc.pb[c]);
            acc.sB[c] = __builtin_fmaf(r, omg, acc.sB[c]);
            acc.sGZ[c] = __builtin_fmaf(rz, g, acc.sGZ[c]);
        }
    }
}

template <int kVariant, int kWaves>
__global__ __launch_bounds__(256, kWaves) void k(float *out, int iters, const float *params) {
    Water w;
    for (int c = 0; c < 3; ++c) { w.B[c] = params[c]; w.nb[c] = params[3 + c]; w.ng[c] = params[6 + c]; }
    float J[3] = {0.3f + threadIdx.x * 1e-4f, 0.4f, 0.5f};
    Acc acc = {};
    float zz[4] = {3.0f + threadIdx.x * 1e-3f, 3.1f, 3.2f, 3.3f};
    uint32_t cc[3] = {0x10203040u + threadIdx.x, 0x50607080u, 0x11223344u};
    for (int it = 0; it < iters; ++it) {
        chunk<kVariant>(zz, cc, w, J, acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) zz[j] += 1e-4f;   // (four more instructions per chunk; keeps the work in the loop)
        cc[0] += 0x01010101u;
    }
    float s = acc.cost;
    for (int c = 0; c < 3; ++c) s += acc.pa[c] + acc.pb[c] + acc.sB[c] + acc.sGZ[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int kVariant, int kWaves>
void run(const char *name, const float *params, float *out) {
    const int iters = 4000, blocks = 256 * kWaves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<kVariant, kWaves>), dim3(blocks), dim3(256), 0, 0, out, iters, params); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<kVariant, kWaves>), dim3(blocks), dim3(256), 0, 0, out, iters, params);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double chunks_per_simd = (double)kWaves * iters;
    printf("%-34s %d waves/SIMD: %7.3f ms -> %6.1f ns per chunk per SIMD (= %5.0f cycles at 2.2 GHz; the kernel's loop: ~811)\n", name, kWaves, ms,
           ms * 1e6 / chunks_per_simd, ms * 1e6 / chunks_per_simd * 2.2);
}

int main() {
    float hp[9] = {0.1f, 0.1f, 0.1f, -0.144f, -0.144f, -0.144f, -0.2f, -0.2f, -0.2f}, *params, *out;
    hipMalloc(&params, sizeof(hp)); hipMemcpy(params, hp, sizeof(hp), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0, 1>("chunk (24 exp back to back)", params, out); run<0, 2>("chunk (24 exp back to back)", params, out);
    run<0, 4>("chunk (24 exp back to back)", params, out); run<0, 5>("chunk (24 exp back to back)", params, out);
    run<0, 6>("chunk (24 exp back to back)", params, out);
    run<2, 5>("chunk, compiler's own schedule", params, out);
    run<1, 5>("chunk, exponentials -> FMAs", params, out);
    return 0;
}
