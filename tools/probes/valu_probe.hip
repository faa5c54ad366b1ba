// Probe: sustained issue rate of v_fma_f32 / v_pk_fma_f32 / v_exp_f32 / v_cndmask at several waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[8]; f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.f}; }
    const float m = 0.999f, c = 0.001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);
                if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], f2{m, m}, f2{c, c});
                if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);
                if (MODE == 3) { a[i] = __builtin_fmaf(a[i], m, c); a[i] = __builtin_amdgcn_exp2f(a[i]); }
                if (MODE == 4) a[i] = (a[i] > 0.5f) ? a[i] * m : c;
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int blocks_per_cu, int instr_per_iter) {
    const int iters = 2000; float *out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25f); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = blocks_per_cu (waves/SIMD) * iters * instr_per_iter
    const double wi = (double)blocks_per_cu * iters * instr_per_iter;
    printf("%-22s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instr per SIMD = %.2f cycles @2.4GHz\n", name, blocks_per_cu, ms, ms * 1e6 / wi, ms * 1e6 / wi * 2.4);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, 32); run<1>("v_pk_fma_f32", w, 32); run<2>("v_exp_f32", w, 32); run<3>("fma+exp (dep)", w, 64); run<4>("cmp+mul+cndmask", w, 96);
    }
    return 0;
}
