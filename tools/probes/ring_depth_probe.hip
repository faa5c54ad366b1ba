// Probe: would a DEEPER per-wave LDS-DMA ring at the same LDS budget pay?  The fit kernels keep kAhead = 2 items of <= 1.5 KiB in
// flight per wave in three 2 KiB slots (the second DMA's surplus lanes write up to 512 B past a 1536-byte chunk).  If the second
// DMA ran with the surplus lanes masked out of EXEC (two s_mov_b64), a slot would be 1536 B and FOUR slots (three items ahead)
// would cost the same 24 KiB per workgroup.  This probe is the fit kernels' skeleton (5120 waves, items of 1536 B, strip-major)
// with the ring depth, the slot size, the masking and an amount of per-item arithmetic (dependent-free FMAs on 16 accumulators,
// `work` rounds of 16) as parameters.
// Measured (MI355X, 503 MB per launch): 6.9 TB/s at ring depth 3, 4 and 5 without arithmetic, and equal times with it (79 / 85 /
// 104 us at 96 / 160 / 224 FMAs per item) -- the depth is not what a launch waits for.  The same stream by ORDINARY loads, one
// 4608-byte chunk ahead in registers (light.hip's way): 6.45-6.55 TB/s without arithmetic, equal with it -- moving the light
// kernels onto the ring would buy 5 % at most.
//   hipcc -w -O3 --offload-arch=gfx950 -o ring_depth_probe ring_depth_probe.hip;  run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int kItem = 1536;

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)); }

template <bool kMaskB>
__device__ __forceinline__ void dma(const uint8_t *src, uint32_t slotA, uint32_t slotB, uint32_t va, uint32_t vb) {
    unsigned keep;
    if (kMaskB)
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_mov_b64 exec, 0xffffffff\n\t"
            "global_load_lds_dwordx4 %5, %2 nt\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(va), "s"(src), "s"(slotA), "s"(slotB), "v"(vb) : "memory");
    else
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %5, %2 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(va), "s"(src), "s"(slotA), "s"(slotB), "v"(vb) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// kDma = false: the same loop without the copies and waits (the arithmetic alone; the ring holds whatever LDS held).
// clk: per workgroup {shader cycles, 100 MHz ticks} of wave 0 -> the shader clock the launch really ran at.
template <int kRing, int kSlot, bool kMaskB, int kWgPerCu, bool kDma = true>
__global__ __launch_bounds__(256, kWgPerCu) void stream(const uint8_t *__restrict__ buf, int items, int work, float *out, unsigned long long *clk) {
    const unsigned long long c0 = clock64(), t0 = wall_clock64();
    constexpr int kAhead = kRing - 1;
    __shared__ __attribute__((aligned(16))) uint8_t ring[4][kRing][kSlot];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t w = blockIdx.x * 4u + wave;
    const uint32_t ring0 = lds_addr(&ring[wave][0][0]);
    const uint32_t va = lane * 16u, vb = (64u + (kMaskB ? (uint32_t)lane : min((uint32_t)lane, 31u))) * 16u;
    const uint8_t *base = buf + (uint64_t)w * items * kItem;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
    for (int k = 0; k < kAhead; ++k) if (kDma) dma<kMaskB>(base + (size_t)k * kItem, ring0 + k * kSlot, ring0 + k * kSlot + 1024, va, vb);
    uint32_t cs = 0;
    for (int i = 0; i < items; ++i) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t is = cs == 0 ? (kRing - 1) * kSlot : cs - kSlot;
        const int nx = i + kAhead < items ? i + kAhead : items - 1;   // (the last ones re-read the last item: never consumed)
        if (kDma) dma<kMaskB>(base + (size_t)nx * kItem, ring0 + is, ring0 + is + 1024, va, vb);
        if (kDma) wait_vm<2 * kAhead>();
        const uint4 a = *reinterpret_cast<const uint4 *>(&ring[wave][0][0] + cs + lane * 24);
        const uint2 b = *reinterpret_cast<const uint2 *>(&ring[wave][0][0] + cs + lane * 24 + 16);
        const float x = __uint_as_float((a.x & 0x007fffffu) | 0x3f000000u), y = __uint_as_float((b.y & 0x007fffffu) | 0x3f000000u);
        acc[0] += __uint_as_float(a.y & 0x3fffffffu) + __uint_as_float(a.z & 0x3fffffffu) + __uint_as_float(a.w & 0x3fffffffu) + __uint_as_float(b.x & 0x3fffffffu);
        for (int r = 0; r < work; ++r) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = __builtin_fmaf(acc[k], x, y);
        }
        cs = cs == (kRing - 1) * kSlot ? 0 : cs + kSlot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[k];
    if (s == 12345.678f) out[w] = s;   // keeps the reads alive
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = clock64() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - t0; }
}

template <int kRing, int kSlot, bool kMaskB, int kWgPerCu, bool kDma = true>
void run(const uint8_t *buf, float *out, int items, int work) {
    static unsigned long long *clk = nullptr;
    if (!clk) hipMalloc(&clk, 2 * 2048 * sizeof(unsigned long long));
    const uint32_t n_wg = 256 * kWgPerCu;
    const int per_wave = items * 5 / kWgPerCu;     // the same bytes per launch whatever the grid
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((stream<kRing, kSlot, kMaskB, kWgPerCu, kDma>), dim3(n_wg), dim3(256), 0, 0, buf, per_wave, work, out, clk); };
    for (int k = 0; k < 3; ++k) launch();
    hipEventRecord(e0);
    const int launches = 20;
    for (int k = 0; k < launches; ++k) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / launches, bytes = (double)n_wg * 4 * per_wave * kItem;
    static unsigned long long h[2 * 2048];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, tick = 0;
    for (uint32_t b = 0; b < n_wg; ++b) { cyc += (double)h[2 * b]; tick += (double)h[2 * b + 1]; }
    printf("ring %d slot %4d mask %d wg/cu %d dma %d work %3d: %7.1f us per launch, %.2f TB/s, shader clock %.0f MHz (%s)\n", kRing, kSlot, (int)kMaskB,
           kWgPerCu, (int)kDma, work, us, kDma ? bytes / us * 1e-6 : 0.0, cyc / tick * 100.0, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

// The light kernel's way (light.hip): ordinary loads, ONE chunk of three items (4608 B here) ahead, in registers.
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
template <int kWgPerCu>
__global__ __launch_bounds__(256, kWgPerCu) void stream_plain(const uint8_t *__restrict__ buf, int items, int work, float *out, unsigned long long *clk) {
    const unsigned long long c0 = clock64(), t0 = wall_clock64();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x * 4u + wave;
    const uint8_t *base = buf + (uint64_t)w * items * kItem;
    const int chunks = items / 3;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    auto load = [&](int c, u4v (&r)[5]) {
        const uint8_t *p = base + (size_t)c * 3 * kItem;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p + q * 1024 + lane * 16));
        r[4] = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p + 4096 + (lane & 31) * 16));
    };
    u4v cur[5], nxt[5];
    load(0, cur);
    for (int c = 0; c < chunks; ++c) {
        load(c + 1 < chunks ? c + 1 : c, nxt);
        const float x = __uint_as_float((cur[0].x & 0x007fffffu) | 0x3f000000u), y = __uint_as_float((cur[4].y & 0x007fffffu) | 0x3f000000u);
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q] += __uint_as_float(cur[q].y & 0x3fffffffu) + __uint_as_float(cur[q].z & 0x3fffffffu) + __uint_as_float(cur[q].w & 0x3fffffffu);
        for (int r = 0; r < 3 * work; ++r) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = __builtin_fmaf(acc[k], x, y);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) cur[q] = nxt[q];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += acc[k];
    if (s == 12345.678f) out[w] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = clock64() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - t0; }
}

template <int kWgPerCu>
void run_plain(const uint8_t *buf, float *out, int items, int work) {
    static unsigned long long *clk = nullptr;
    if (!clk) hipMalloc(&clk, 2 * 2048 * sizeof(unsigned long long));
    const uint32_t n_wg = 256 * kWgPerCu;
    const int per_wave = items * 5 / kWgPerCu / 3 * 3;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((stream_plain<kWgPerCu>), dim3(n_wg), dim3(256), 0, 0, buf, per_wave, work, out, clk); };
    for (int k = 0; k < 3; ++k) launch();
    hipEventRecord(e0);
    const int launches = 20;
    for (int k = 0; k < launches; ++k) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / launches, bytes = (double)n_wg * 4 * per_wave * kItem;
    printf("plain loads, one 4608-byte chunk ahead, wg/cu %d work %3d: %7.1f us per launch, %.2f TB/s (%s)\n", kWgPerCu, work, us, bytes / us * 1e-6, hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    const int items = 64;                         // 5120 x 64 x 1536 B = 503 MB per launch
    const size_t bytes = (size_t)5120 * items * kItem * 5 / 4 + 4096;
    uint8_t *buf; float *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 8192 * 4);
    hipMemset(buf, 1, bytes); hipDeviceSynchronize();
    printf("alloc %s\n", hipGetErrorString(hipGetLastError())); fflush(stdout);
    for (int rep = 0; rep < 2; ++rep)
        for (int work : {0, 6, 10}) {             // 0 / 96 / 160 / 224 FMAs per item (the fit kernel: ~180 + ~85 of bookkeeping)
            run_plain<4>(buf, out, items, work);
            run_plain<5>(buf, out, items, work);
            run<3, 2048, false, 4>(buf, out, items, work);   // the ring at four workgroups per CU
            run<3, 2048, false, 5>(buf, out, items, work);   // today
            run<3, 2048, false, 5, false>(buf, out, items, work);   // its arithmetic alone
            run<3, 1536, true, 5>(buf, out, items, work);    // masking alone
            run<4, 1536, true, 5>(buf, out, items, work);    // same LDS, three items ahead
            run<5, 1536, true, 5>(buf, out, items, work);    // 150 KiB per CU
        }
    return 0;
}
