// Probe: does a v_fma_f32 with THREE different vector-register operands (what the fit kernels' accumulations are:
// acc = fma(r, a, acc)) issue as fast as the two-register form tools/probes/exp_probe.hip measured (x = fma(x, k, k))?
// And what does a dependent chain cost when the 16 accumulators of a wave are updated four times in a row, like the four
// levels of a chunk?   hipcc -O3 --offload-arch=gfx950 tools/probes/fma3_probe.hip -o /tmp/fma3_probe && /tmp/fma3_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define FMA2(R) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(R) : "v"(k));
#define FMA3(R, A, B) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(R) : "v"(A), "v"(B));
#define MUL(R, A, B) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(R) : "v"(A), "v"(B));

// MODE 0: 32 two-register FMAs; 1: 32 three-register FMAs (16 accumulators, 16 + 16 distinct sources);
// 2: 16 muls into temporaries + 16 three-register FMAs reading them (one dependent pair per accumulator)
template <int MODE>
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *cycles, int iters) {
    float x[16], a[16], b[16], t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = 0.5f + threadIdx.x * 1e-3f + i * .01f; a[i] = 0.999f - i * 1e-4f; b[i] = 1e-3f * (i + 1); t[i] = 0.f; }
    const float k = 0.999f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) FMA2(x[i])
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) FMA3(x[i], a[i], b[(i + r) & 15])
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) MUL(t[i], a[i], b[i])
#pragma unroll
            for (int i = 0; i < 16; ++i) FMA3(x[i], t[i], a[(i + 1) & 15])
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + t[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *what, int waves_per_simd) {
    const int blocks = 1024 * waves_per_simd, iters = 4000, per_iter = 32;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * 64 * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= blocks;
    // wall-clock cross-check: instructions per SIMD / (ms * clock) -- the clock is whatever the box runs at (printed as GHz if 2.3)
    const double instr_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("%-40s waves/SIMD %d: %.2f s_memtime ticks per instruction per SIMD slot; kernel %.3f ms = %.2f ns per instruction per SIMD\n",
           what, waves_per_simd, mean / ((double)iters * per_iter) / waves_per_simd, ms, ms * 1e6 / instr_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4, 5, 8}) {
        run<0>("32 v_fma_f32, two registers", w);
        run<1>("32 v_fma_f32, three registers", w);
        run<2>("16 v_mul_f32 + 16 dependent v_fma_f32", w);
    }
    return 0;
}
