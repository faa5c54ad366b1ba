// Probe: where do global_load_lds_dwordx4 / dwordx3 (saddr form, M0 = LDS byte offset) put their bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)); }

__global__ __launch_bounds__(256) void probe(const uint8_t *src, uint32_t *out, uint32_t *addr_out, int mode) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][4096];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = t; i < 4096; i += 256) reinterpret_cast<uint32_t *>(&lds[0][0])[i] = 0xDEAD0000u | i;
    __syncthreads();
    const uint8_t *chunk = src + (size_t)wave * 1792;
    const uint32_t slot = lds_addr(&lds[wave][0]);
    if (lane == 0) addr_out[wave] = slot;
    const uint32_t voff_z = lane * 16, voff_c = 1024 + lane * 12;
    unsigned keep;
    const uint32_t slot_c = slot + 1024;
    if (mode == 0) {
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx3 %5, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(voff_z), "s"(chunk), "s"(slot), "s"(slot_c), "v"(voff_c) : "memory");
    } else {
        // vaddr 64-bit form
        const uint8_t *pz = chunk + voff_z, *pc = chunk + voff_c;
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx3 %4, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(pz), "s"(slot), "s"(slot_c), "v"(pc) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = t; i < 4096; i += 256) out[i] = reinterpret_cast<uint32_t *>(&lds[0][0])[i];
}

int main() {
    std::vector<uint8_t> h(4 * 1792);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)((i * 7 + (i >> 8)) & 0xFF);
    uint8_t *src; uint32_t *out, *addr;
    hipMalloc(&src, h.size()); hipMalloc(&out, 16384); hipMalloc(&addr, 16);
    hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(out, 0, 16384);
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, src, out, addr, mode);
        hipError_t e = hipDeviceSynchronize();
        std::vector<uint32_t> o(4096); uint32_t a[4];
        hipMemcpy(o.data(), out, 16384, hipMemcpyDeviceToHost); hipMemcpy(a, addr, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %s  slot addrs %u %u %u %u\n", mode, hipGetErrorString(e), a[0], a[1], a[2], a[3]);
        const uint8_t *ob = reinterpret_cast<const uint8_t *>(o.data());
        for (int w = 0; w < 4; ++w) {
            // expected: lds[w][0..1791] == h[w*1792 ..]
            size_t good = 0; long first_bad = -1;
            for (int i = 0; i < 1792; ++i) { if (ob[w * 4096 + i] == h[w * 1792 + i]) ++good; else if (first_bad < 0) first_bad = i; }
            printf("  wave %d: %zu/1792 bytes match, first mismatch at %ld; dwords at slot[0..3]=%08x %08x %08x %08x  slot[256..257]=%08x %08x slot[448]=%08x\n",
                   w, good, first_bad, o[w * 1024 + 0], o[w * 1024 + 1], o[w * 1024 + 2], o[w * 1024 + 3], o[w * 1024 + 256], o[w * 1024 + 257], o[w*1024+448]);
        }
        if (mode == 0) {
            printf("  src dwords [256..271]: ");
            for (int i = 256; i < 272; ++i) { uint32_t v; memcpy(&v, &h[i * 4], 4); printf("%08x ", v); }
            printf("\n  lds dwords [256..275]: ");
            for (int i = 256; i < 276; ++i) printf("%08x ", o[i]);
            printf("\n");
            // locate each source dword of the rgb part in LDS
            for (int i = 256; i < 448; i += 1) { uint32_t v; memcpy(&v, &h[i * 4], 4); int found = -1; for (int j = 0; j < 1024; ++j) if (o[j] == v) { found = j; break; } if (i < 270 || i % 48 == 0) printf("   src dword %d -> lds dword %d\n", i, found); }
        }
        // where did wave 1's first dword land?
        uint32_t want; memcpy(&want, &h[1792], 4);
        for (int i = 0; i < 4096; ++i) if (o[i] == want) printf("  wave1 first dword found at lds dword %d (byte %d)\n", i, i * 4);
    }
    return 0;
}
