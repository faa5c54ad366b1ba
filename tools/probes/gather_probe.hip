// Probe: what does a 64-lane gather cost on gfx950 as a function of the cache lines it touches?  Behind DESIGN.md
// section 4.1: match_kernel was bound by its gathers -- with four adjacent pixels per lane every gather spread over 16
// rows of the view; with sixteen adjacent lanes on sixteen adjacent pixels it touches 4 short row segments.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
// Every wave issues `iters` x 8 independent dword gathers from an L2-resident image (8 MB); the lanes of one gather read
// `rows` row segments of 64 / rows adjacent pixels each (rows = 1: one 256-byte run ... rows = 64: every lane its own row).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kW = 2048, kH = 1024;   // float image, 8 MB

__global__ __launch_bounds__(256) void probe(const float *__restrict__ img, float *out, unsigned long long *cycles, int rows,
                                             int iters) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int per = 64 / rows;                       // adjacent pixels per row segment
    const int r = lane / per, c = lane % per;        // this lane's row segment and position in it
    unsigned x0 = (wave * 97u) % (kW - 64), y0 = (wave * 31u) % (kH - 80);
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {                // 8 independent gathers in flight
            const unsigned y = (y0 + r + 8u * k + i) % (kH - 1), x = (x0 + c + 3u * i) % (kW - 64);
            v[k] = img[(size_t)y * kW + x];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0) cycles[wave] = t1 - t0;
}

int main() {
    float *img, *out;
    unsigned long long *cyc;
    const int blocks = 256 * 8, iters = 200;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD, like match_kernel
    hipMalloc(&img, sizeof(float) * kW * kH);
    hipMemset(img, 0, sizeof(float) * kW * kH);
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    std::vector<unsigned long long> h(blocks * 4);
    for (int rows : {1, 2, 4, 8, 16, 32, 64}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, img, out, cyc, rows, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto c : h) mean += (double)c;
        mean /= h.size();
        const double gathers = (double)blocks * 4 * iters * 8;
        printf("%2d row segments x %2d adjacent pixels per gather: %.3f ms, %6.1f ns per gather per CU, %7.0f cycles per gather as a wave sees it\n",
               rows, 64 / rows, ms, ms * 1e6 / (gathers / 256), mean / (iters * 8.0));
    }
    return 0;
}
