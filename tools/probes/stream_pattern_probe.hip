// Probe: what the memory side sustains for the fit kernels' access pattern, and whether another placement of the strips'
// chunks would sustain more.  5120 waves (1280 workgroups x 4, the fit grid) each stream ITEMS of 1536 B (a 24-bit chunk) by two
// LDS-DMA instructions into a private ring of three 2 KiB slots, two items ahead, vmcnt(4) before an item is touched -- the fit
// kernels' skeleton without their arithmetic.  Patterns (where item i of wave w lives):
//   0  strip-major   base(w) + i * 1536                      (today: a wave walks its own contiguous strip)
//   1  interleaved over the 4 waves of a workgroup: wgbase + (i * 4 + w % 4) * 1536
//   2  interleaved over 32 consecutive waves (8 workgroups: one per XCD)
// Measured (MI355X, 503 MB per launch): 76.5 / 75.1 / 78.1 us = 6.58 / 6.70 / 6.45 TB/s -- the skeleton itself sustains 0.82 of the
// 8 TB/s peak in today's strip-major placement, and interleaving the chunks of neighbouring strips changes it by +-2 %.
// build: hipcc -O3 --offload-arch=gfx950 -o stream_pattern_probe stream_pattern_probe.hip;  run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int kItem = 1536, kSlot = 2048, kRing = 3;

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)); }

__device__ __forceinline__ void dma(const uint8_t *src, uint32_t slotA, uint32_t slotB, uint32_t va, uint32_t vb) {
    unsigned keep;
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

__global__ __launch_bounds__(256, 5) void stream(const uint8_t *__restrict__ buf, int items, int pattern, uint32_t n_waves, float *out) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[4][kRing][kSlot];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t w = blockIdx.x * 4u + wave;
    const uint32_t ring0 = lds_addr(&ring[wave][0][0]);
    const uint32_t va = lane * 16u, vb = (64u + min((uint32_t)lane, 31u)) * 16u;
    auto addr = [&](uint32_t i) -> const uint8_t * {
        uint64_t idx;
        if (pattern == 0) idx = (uint64_t)w * items + i;
        else if (pattern == 1) idx = (uint64_t)(w / 4u) * 4u * items + (uint64_t)i * 4u + (w % 4u);
        else if (pattern == 2) idx = (uint64_t)(w / 32u) * 32u * items + (uint64_t)i * 32u + (w % 32u);
        else idx = (uint64_t)w * items + i;
        return buf + idx * kItem;
    };
    float acc = 0.f;
    dma(addr(0), ring0, ring0 + 1024, va, vb);
    dma(addr(1), ring0 + kSlot, ring0 + kSlot + 1024, va, vb);
    uint32_t cs = 0;
    for (int i = 0; i < items; ++i) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t is = cs == 0 ? 2 * kSlot : cs - kSlot;
        const int nx = i + 2 < items ? i + 2 : items - 1;   // (the last two re-read the last item: never consumed)
        dma(addr(nx), ring0 + is, ring0 + is + 1024, va, vb);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        const uint2 a = *reinterpret_cast<const uint2 *>(&ring[wave][0][0] + cs + lane * 24);
        acc += __uint_as_float(a.x & 0x3fffffffu) + __uint_as_float(a.y & 0x3fffffffu);
        cs = cs == 2 * kSlot ? 0 : cs + kSlot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.678f) out[w] = acc;   // keeps the reads alive
}

int main() {
    const uint32_t n_waves = 5120;
    const int items = 64;                         // 5120 x 64 x 1536 B = 503 MB per launch
    const size_t bytes = (size_t)n_waves * items * kItem + 4096;
    uint8_t *buf; float *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, n_waves * 4);
    printf("alloc %s\n", hipGetErrorString(hipGetLastError())); hipMemset(buf, 1, bytes); hipDeviceSynchronize(); printf("memset %s\n", hipGetErrorString(hipGetLastError())); fflush(stdout);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int pattern = 0; pattern < 3; ++pattern) {
            for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(stream, dim3(n_waves / 4), dim3(256), 0, 0, buf, items, pattern, n_waves, out);
            hipEventRecord(e0);
            const int launches = 20;
            for (int k = 0; k < launches; ++k) hipLaunchKernelGGL(stream, dim3(n_waves / 4), dim3(256), 0, 0, buf, items, pattern, n_waves, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / launches;
            printf("pattern %d: %.1f us per launch, %.2f TB/s (%s)\n", pattern, us, (double)n_waves * items * kItem / us * 1e-6, hipGetErrorString(hipGetLastError())); fflush(stdout);
        }
    return 0;
}
