// Probe: does a v_mfma_f32_4x4x1_16b_f32 cost VALU issue time?  The fit kernels are bound by VALU issue at the sustained shader
// clock (kernel time x clock = constant, tools/exp/wave_times.py), and 15 of their ~45 plain instructions per observation are
// accumulations acc = fma(x, y, acc) of per-lane products -- the diagonal of a 4x4x1 MFMA (D[j][j] += A[j] B[j]).  If the matrix
// pipe took them beside the VALU, a third of the loop's issue time would go.
//   MODE 0: 32 v_fma_f32 per iteration (16 accumulators);  1: 32 v_fma_f32 + 16 MFMAs interleaved 2:1;  2: 16 MFMAs;
//   3: 32 v_fma_f32 + 32 MFMAs 1:1.   Prints shader cycles per iteration per SIMD at 1, 2, 4, 5 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_coissue_probe mfma_coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

#define FMA(R) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(R) : "v"(a), "v"(b));
#define MF(C) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b));

template <int MODE>
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *cycles, int iters) {
    float x[16];
    f4 c[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = f4{0.f, 0.f, 0.f, 0.f};
    const float a = 0.999f + threadIdx.x * 1e-6f, b = 1e-3f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { FMA(x[0]) FMA(x[1]) FMA(x[2]) FMA(x[3]) FMA(x[4]) FMA(x[5]) FMA(x[6]) FMA(x[7]) FMA(x[8]) FMA(x[9]) FMA(x[10]) FMA(x[11]) FMA(x[12]) FMA(x[13]) FMA(x[14]) FMA(x[15]) }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { FMA(x[0]) FMA(x[1]) MF(c[0]) FMA(x[2]) FMA(x[3]) MF(c[1]) FMA(x[4]) FMA(x[5]) MF(c[2]) FMA(x[6]) FMA(x[7]) MF(c[3]) FMA(x[8]) FMA(x[9]) MF(c[4]) FMA(x[10]) FMA(x[11]) MF(c[5]) FMA(x[12]) FMA(x[13]) MF(c[6]) FMA(x[14]) FMA(x[15]) MF(c[7]) }
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { MF(c[0]) MF(c[1]) MF(c[2]) MF(c[3]) MF(c[4]) MF(c[5]) MF(c[6]) MF(c[7]) }
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) { FMA(x[0]) MF(c[0]) FMA(x[1]) MF(c[1]) FMA(x[2]) MF(c[2]) FMA(x[3]) MF(c[3]) FMA(x[4]) MF(c[4]) FMA(x[5]) MF(c[5]) FMA(x[6]) MF(c[6]) FMA(x[7]) MF(c[7]) FMA(x[8]) MF(c[0]) FMA(x[9]) MF(c[1]) FMA(x[10]) MF(c[2]) FMA(x[11]) MF(c[3]) FMA(x[12]) MF(c[4]) FMA(x[13]) MF(c[5]) FMA(x[14]) MF(c[6]) FMA(x[15]) MF(c[7]) }
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd, iters = 4000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 100);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += (double)v; m /= blocks;
    printf("%-34s %d waves/SIMD: %7.1f cycles per iteration per wave, %7.1f per SIMD;  %.1f us\n", name, waves_per_simd, m / iters, m / iters * 1.0, ms * 1e3);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4, 5}) {
        run<0>("32 fma", w);
        run<2>("16 mfma 4x4x1", w);
        run<1>("32 fma + 16 mfma", w);
        run<3>("32 fma + 32 mfma", w);
    }
    return 0;
}
