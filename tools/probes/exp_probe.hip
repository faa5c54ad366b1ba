// Probe: what do the fit kernel's two kinds of vector instructions cost to issue on gfx950 -- v_fma_f32 and v_exp_f32 --
// alone, mixed in the kernel's own ratio (13 plain : 2 exp per observation-channel), and when one wave of a SIMD runs
// exponentials while another runs FMAs (does the transcendental unit overlap with the plain pipe across waves?).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/exp_probe.hip -o /tmp/exp_probe && /tmp/exp_probe
// Prints shader cycles (s_memtime) per instruction per SIMD.  Behind DESIGN.md section 4.2 ("instruction floor").
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define FMA(R) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(R) : "v"(k));
#define EXP(R) asm volatile("v_exp_f32 %0, %0" : "+v"(R));
#define X16(OP) OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7) OP(x8) OP(x9) OP(x10) OP(x11) OP(x12) OP(x13) OP(x14) OP(x15)

// MODE 0: 32 FMAs per iteration; 1: 32 exps; 2: 26 FMAs + 4 exps (the fit's mix, exps back to back);
// 3: even waves of a SIMD run MODE 1's body, odd waves MODE 0's (needs >= 2 waves per SIMD)
template <int MODE>
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *cycles, int iters) {
    float x0 = 0.5f + threadIdx.x * 1e-3f, x1 = x0 + .01f, x2 = x0 + .02f, x3 = x0 + .03f, x4 = x0 + .04f, x5 = x0 + .05f,
          x6 = x0 + .06f, x7 = x0 + .07f, x8 = x0 + .08f, x9 = x0 + .09f, x10 = x0 + .1f, x11 = x0 + .11f, x12 = x0 + .12f,
          x13 = x0 + .13f, x14 = x0 + .14f, x15 = x0 + .15f;
    const float k = 0.999f;
    const bool exps = MODE == 1 || (MODE == 3 && ((blockIdx.x >> 10) & 1) == 0);   // blocks are dealt over CUs/SIMDs first
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 2) {
            X16(FMA) FMA(x0) FMA(x1) FMA(x2) FMA(x3) FMA(x4) FMA(x5) FMA(x6) FMA(x7) FMA(x8) FMA(x9)
            EXP(x10) EXP(x11) EXP(x12) EXP(x13)
        } else if (exps) {
            X16(EXP) X16(EXP)
        } else {
            X16(FMA) X16(FMA)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + x8 + x9 + x10 + x11 + x12 + x13 + x14 + x15;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *what, int waves_per_simd, int per_iter) {
    const int blocks = 1024 * waves_per_simd, iters = 4000;
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
    // a wave's own time / its instructions = cycles per instruction as that wave sees them; x waves sharing = per SIMD
    printf("%-44s waves/SIMD %d: %.2f cycles per instruction per wave, %.2f per SIMD slot, kernel %.3f ms\n", what,
           waves_per_simd, mean / ((double)iters * per_iter), mean / ((double)iters * per_iter) / waves_per_simd, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 5}) {
        run<0>("32 v_fma_f32", w, 32);
        run<1>("32 v_exp_f32", w, 32);
        run<2>("26 v_fma_f32 + 4 v_exp_f32 (the fit's mix)", w, 30);
    }
    for (int w : {2, 4}) run<3>("half the waves exps, half FMAs", w, 32);
    return 0;
}
