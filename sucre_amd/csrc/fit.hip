// Exponential-model fit for gfx950.
//
// One iteration of sucre.adam (sucre.py:138-148) = ONE launch of fit_grad_kernel (J as a parameter) or
// fit_closed_kernel (--use-closed-form): it streams the compact observation store once, updates J in place, reduces
// the ten global sums and -- in the workgroup that arrives last -- applies Adam to B, beta, gamma and logs the row.
// The split form (launch_fit_grad + launch_fit_step) leaves the sums in the workspace for a host all-reduce.
//
// Work decomposition: 1536 persistent 256-thread workgroups, each looping over 16x16-pixel tiles of the sorted
// compact store (csrc/compact.hip).  Phase 1 is level-parallel: the tile's levels are dealt round-robin to the four
// waves, each lane owning 4 pixels x 3 channels; a chunk is copied HBM -> LDS by two LDS-DMA instructions into a
// per-wave ring, so the per-pixel sums need no atomics and prefetch depth costs no registers.  Phase 2 adds the four
// waves' per-pixel sums through LDS in a fixed order; phase 3 is pixel-parallel (one pixel per thread):
// torch.optim.Adam on J.  The ten global sums go wave shuffle -> LDS -> one float32 partial per workgroup ->
// float64 fixed-order two-level reduction, so results are bitwise reproducible.
//
// Model (sucre.py:79-82, l = 1):  Ihat = J a + B (1 - g),  a = exp(-beta z),  g = exp(-gamma z),  r = I - Ihat.
// With L = sum r^2 / (3 n_obs) and s = (1/3)/n_obs (sucre.py:145):
//   dL/dJ[p]  = -2 s sum_k r a          dL/dB     = -2 s sum r (1 - g)
//   dL/dbeta  = +2 s sum_p J sum_k r a z    dL/dgamma = -2 s B sum r g z        (per channel)
#include "fit_math.h"

namespace sucre {

struct Water {
    float B[3], nb[3], ng[3];  // B, -beta*log2(e), -gamma*log2(e)
};

__device__ __forceinline__ Water load_water(const float *__restrict__ params) {
    Water w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        w.B[c] = params[c];
        w.nb[c] = -params[3 + c] * kLog2e;
        w.ng[c] = -params[6 + c] * kLog2e;
    }
    return w;
}

// What one pass over a view accumulates.
enum Pass { kPassGradJ = 0, kPassClosedJ = 1 };

// The kernel is VALU-issue bound (SQ_ACTIVE_INST_VALU ~ 89 % of the launch): 15 instructions per
// observation-channel in the select-free loop, 4 cycles each, 8 for v_exp_f32.  Packed v_pk_*_f32 issues at half
// rate on gfx950 (tools/probes/valu_probe.hip), so writing the loop two-wide saved instructions but no time (and
// cost 20 VGPRs); it stays scalar.
struct Acc {
    float pa[3][4];  // per pixel-channel: sum r a           | closed-form numerator   sum (I - b) a
    float pb[3][4];  // per pixel-channel: sum r a z         | closed-form denominator sum a^2
    float sB[3];     // sum r (1 - g)
    float sGZ[3];    // sum r g z
    float cost;      // sum r^2
};

constexpr float kInv255 = (float)(1.0 / 255.0);

// kMasked = false: every slot of the chunk is a real observation (levels below the tile's smallest pixel count in
// the compact store), so the z > 0 test and the selects are compiled out.
template <int kPass, bool kMasked>
__device__ __forceinline__ void accumulate_view(const float4 z4, const uint3 c3, const Water &w,
                                                const float (&J)[3][4], Acc &acc) {
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
    const uint32_t cc[3] = {c3.x, c3.y, c3.z};
#ifdef SUCRE_EXP_NOCOMPUTE  // experiment build only (tools/microbench.py): touch the data, skip the model
    acc.cost += (z4.x + z4.y) + (z4.z + z4.w) + (float)(c3.x ^ c3.y ^ c3.z);
    return;
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float z = zz[j];
        const bool valid = !kMasked || z > 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t k = (cc[c] >> (8 * j)) & 255u;
            const float a = fast_exp2(z * w.nb[c]);
            const float g = fast_exp2(z * w.ng[c]);
            const float omg = 1.0f - g;
            const float bt = w.B[c] * omg;
            if (kPass == kPassClosedJ) {
                // sucre.py:73-76: numerator += (I - backscatter) * absorption ; denominator += absorption^2
                const float I = unit_from_u8(k);
                const float y = valid ? (I - bt) : 0.0f;
                const float a2 = valid ? a * a : 0.0f;
                acc.pa[c][j] = __builtin_fmaf(y, a, acc.pa[c][j]);
                acc.pb[c][j] += a2;
            } else {
                const float Ihat = __builtin_fmaf(J[c][j], a, bt);
                // I = k/255 folded into the residual: one rounding instead of two, two VALU ops fewer
                float r = __builtin_fmaf((float)k, kInv255, -Ihat);
                r = valid ? r : 0.0f;  // select, not multiply: J may be NaN where unobserved
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

// Closed-form mode in ONE pass over the observations (sucre.py:141 + 142-147 with J a constant of the backward pass).
// Write y = I - B(1-g) (free of J) and let Jp be the pixel's J of the previous iteration.  With the provisional
// residual p = y - Jp a, the re-solved J = sum y a / sum a^2 is Jp + dJ, dJ = N/D, N = sum p a, D = sum a^2, the true
// residual is r = p - dJ a, and every sum the gradient needs factors through per-pixel sums that do not contain dJ:
//   sum r (1-g) = S1 - dJ S2      S1 = sum p (1-g)   S2 = sum a (1-g)
//   sum r z a   = S3 - dJ S4      S3 = sum p z a     S4 = sum z a^2
//   sum r z g   = S5 - dJ S6      S5 = sum p z g     S6 = sum a z g
//   sum r^2     = S7 - dJ N       S7 = sum p^2                      (dJ^2 D = dJ N)
// so the observations are streamed once (the two-pass form: J first, then the gradient, is 2x the traffic and 1.4x
// the VALU work).  Measuring from Jp matters: J moves by ~1e-3 per iteration, so p is already at the scale of r and
// the corrections dJ S' are small -- with p = y the differences S - J S' cancel three to four digits when a pixel
// has few observations (seen as 1e-4 relative noise on the cost and the gradients of 2-view scenes).
__device__ __forceinline__ float finite_or_zero(float x) { return __builtin_isfinite(x) ? x : 0.0f; }

struct AccOne {
    float q[9][3][4];  // N, D, S1..S7 per pixel-channel
};

template <int kPass, bool kMasked>
__device__ __forceinline__ void accumulate_view(const float4 z4, const uint3 c3, const Water &w,
                                                const float (&Jp)[3][4], AccOne &acc) {
    const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
    const uint32_t cc[3] = {c3.x, c3.y, c3.z};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float z = zz[j];
        const bool valid = !kMasked || z > 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t k = (cc[c] >> (8 * j)) & 255u;
            const float a = fast_exp2(z * w.nb[c]);
            const float g = fast_exp2(z * w.ng[c]);
            const float omg = 1.0f - g;
            const float y = unit_from_u8(k) - w.B[c] * omg;
            float p = __builtin_fmaf(-Jp[c][j], a, y);
            p = valid ? p : 0.0f;  // a padding slot contributes nothing (its Jp a is not zero)
            const float za = z * a, zg = z * g;  // a padding slot has z = 0, g = 1: it only touches D (masked below)
            acc.q[0][c][j] = __builtin_fmaf(p, a, acc.q[0][c][j]);
            if (kMasked) acc.q[1][c][j] += valid ? a * a : 0.0f;
            else acc.q[1][c][j] = __builtin_fmaf(a, a, acc.q[1][c][j]);
            acc.q[2][c][j] = __builtin_fmaf(p, omg, acc.q[2][c][j]);
            acc.q[3][c][j] = __builtin_fmaf(a, omg, acc.q[3][c][j]);
            acc.q[4][c][j] = __builtin_fmaf(p, za, acc.q[4][c][j]);
            acc.q[5][c][j] = __builtin_fmaf(a, za, acc.q[5][c][j]);
            acc.q[6][c][j] = __builtin_fmaf(p, zg, acc.q[6][c][j]);
            acc.q[7][c][j] = __builtin_fmaf(a, zg, acc.q[7][c][j]);
            acc.q[8][c][j] = __builtin_fmaf(p, p, acc.q[8][c][j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Observation streaming: per-wave LDS-DMA ring.
//
// Each wave consumes levels wave, wave+4, ... of its (sorted) tile in the compact store (csrc/compact.hip).  A chunk is copied HBM -> LDS by
// two LDS-DMA instructions (global_load_lds_dwordx4: 64 lanes = 1 KiB of ranges, 48 lanes = 768 B of colours)
// into one of kRing private slots; kAhead = kRing-1 chunks stay in flight behind the one being consumed.
// The DMAs have no VGPR destination, so prefetch depth costs LDS, not registers, and hipcc can neither sink them
// next to their use nor drain them early: they live in inline asm and are waited for by hand-counted
// s_waitcnt vmcnt(2 * chunks still allowed in flight) (vmcnt retires in issue order; cdna_hip_programming.md 5.7).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRing = 3;
constexpr int kAhead = kRing - 1;
constexpr int kSlot = kChunk;  // LDS image of one chunk = byte-exact copy

struct __attribute__((aligned(16))) FitLds {
    union {
        uint8_t ring[4][kRing][kSlot];  // phase 1: per-wave chunk ring
        float red[4][6][kTilePx];        // phase 2: per-pixel sums of the four waves (ring is dead by then)
    } u;
    double stot[kSumsPad];
    float wsum[4][kNumSums];
    int is_last, is_last_total;  // one flag word per arrive_last level: no wave can see the second verdict as the first
};

struct __attribute__((aligned(16))) FitLdsOne {
    union {
        uint8_t ring[4][kRing][kSlot];
        float red[4][9][kTilePx];        // one round = one triple of quantities x 3 channels of the four waves
    } u;
    double stot[kSumsPad];
    float wsum[4][kNumSums];
    int is_last, is_last_total;  // one flag word per arrive_last level: no wave can see the second verdict as the first
};

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));  // low 32 bits of a flat LDS address = LDS offset
}

#ifndef SUCRE_DMA_POLICY
#define SUCRE_DMA_POLICY " nt"  // chunks are read once per launch: streaming policy (measured -20 % vs default)
#endif

// chunk (wave-uniform global address) -> LDS slot (wave-uniform LDS byte address); lane offsets are loop constants.
// A dwordx4 LDS-DMA writes lane i's 16 bytes at slot + 16 i (a dwordx3 one also strides by 16, leaving holes --
// measured, tools/probes/lds_dma_probe.hip), so the 768 colour bytes are moved by the first 48 lanes of a second
// dwordx4: the LDS image is a byte-exact copy of the 1792-byte chunk.  EXEC is all ones here (whole workgroup
// runs this code) and is restored inside the statement; M0 is written in the statement that reads it.
template <int kFmt>
__device__ __forceinline__ void dma_chunk(const uint8_t *chunk, uint32_t slot, uint32_t voff) {
#ifndef SUCRE_EXP_NOLOAD
    unsigned keep;
    const uint32_t slot_c = slot + kChunkZ;
    if (kFmt == 0) {
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2" SUCRE_DMA_POLICY "\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_mov_b32 exec_hi, 0xffff\n\t"
            "global_load_lds_dwordx4 %5, %2" SUCRE_DMA_POLICY "\n\t"
            "s_mov_b32 exec_hi, -1\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(chunk), "s"(slot), "s"(slot_c), "v"(voff + kChunkZ)
            : "memory");
    } else {
        // 1280-byte chunk (512 B of uint16 ranges + 768 B of colours): 64 lanes move the first KiB, 16 lanes the rest
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2" SUCRE_DMA_POLICY "\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_mov_b64 exec, 0xffff\n\t"
            "global_load_lds_dwordx4 %5, %2" SUCRE_DMA_POLICY "\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(chunk), "s"(slot), "s"(slot_c), "v"(voff + kChunkZ)
            : "memory");
    }
#endif
}

// Waits until at most 2*ahead DMA instructions are outstanding.
template <int kAheadNow>
__device__ __forceinline__ void wait_chunks() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kAheadNow) : "memory");
}

// Number of levels (wave, wave+4, ...) this wave consumes.
__device__ __forceinline__ uint32_t views_of_wave(uint32_t n, int wave) {
    return n > (uint32_t)wave ? (n - (uint32_t)wave + 3u) / 4u : 0u;
}

// Ring prologue: put the first kAhead chunks in flight.
template <int kFmt, class FitLds>
__device__ __forceinline__ void stream_begin(FitLds &lds, const uint8_t *__restrict__ tile_obs, uint32_t r, int wave,
                                             int lane) {
    const uint32_t voff = lane * 16;
    const uint32_t ring0 = lds_addr(&lds.u.ring[wave][0][0]);
#ifdef SUCRE_EXP_NOLOAD  // experiment build only: no DMA; the ring holds plausible constant data instead
    for (int sl = 0; sl < kRing; ++sl) {
        *reinterpret_cast<float4 *>(&lds.u.ring[wave][sl][lane * 16]) = make_float4(2.5f, 2.75f, 3.0f, 3.25f);
        if (lane < 48) *reinterpret_cast<uint4 *>(&lds.u.ring[wave][sl][kChunkZ + lane * 16]) =
            make_uint4(0x10203040u, 0x50607080u, 0x11223344u, 0x55667788u);
    }
#endif
#pragma unroll
    for (uint32_t d = 0; d < (uint32_t)kAhead; ++d)
        if (d < r) dma_chunk<kFmt>(tile_obs + (size_t)(wave + 4u * d) * chunk_bytes(kFmt), ring0 + d * kSlot, voff);
}

// Ring steady state for this wave's levels v0 <= v < v1 (of r); slot / slot_in carry the ring position across calls.
template <int kPass, bool kMasked, int kFmt, class FitLds, class AccT>
__device__ __forceinline__ void stream_range(FitLds &lds, const uint8_t *__restrict__ tile_obs, uint32_t r, uint32_t v0,
                                             uint32_t v1, int wave, int lane, const Water &w, const float (&J)[3][4],
                                             AccT &acc, uint32_t &slot, uint32_t &slot_in) {
    const uint32_t voff = lane * 16;
    const uint32_t ring0 = lds_addr(&lds.u.ring[wave][0][0]);
    for (uint32_t v = v0; v < v1; ++v) {
        if (v + kAhead < r)
            dma_chunk<kFmt>(tile_obs + (size_t)(wave + 4u * (v + kAhead)) * chunk_bytes(kFmt), ring0 + slot_in * kSlot, voff);
        const uint32_t ahead = min((uint32_t)kAhead, r - 1u - v);  // chunks allowed to stay in flight
        if (ahead >= (uint32_t)kAhead) wait_chunks<kAhead>();
        else if (kAhead > 2 && ahead == 2u) wait_chunks<2>();
        else if (ahead == 1u) wait_chunks<1>();
        else wait_chunks<0>();
        const uint8_t *sp = &lds.u.ring[wave][slot][0];
        float4 z4;
        if (kFmt == 0) {
            z4 = *reinterpret_cast<const float4 *>(sp + lane * 16);
        } else {  // uint16 millimetres -> metres, the same float32 product the CPU restatement forms
            const uint2 q = *reinterpret_cast<const uint2 *>(sp + lane * 8);
            z4 = make_float4((float)(q.x & 0xffffu) * kMPerMm, (float)(q.x >> 16) * kMPerMm,
                             (float)(q.y & 0xffffu) * kMPerMm, (float)(q.y >> 16) * kMPerMm);
        }
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(sp + (kFmt ? kChunkZ16 : kChunkZ)) + lane;  // planar R | G | B
        const uint3 c3 = make_uint3(cp[0], cp[64], cp[128]);
        accumulate_view<kPass, kMasked>(z4, c3, w, J, acc);
        slot = slot + 1 == kRing ? 0 : slot + 1;
        slot_in = slot_in + 1 == kRing ? 0 : slot_in + 1;
    }
}

// Ring steady state + drain; stream_begin must have been called for the same (tile, wave).  The levels below
// nfull (the tile's smallest pixel count) hold 256 real observations each: they run the select-free loop; the
// few levels above it run the masked one.  Two separate loops on purpose: as one loop with a uniform branch hipcc
// if-converted both bodies into one (41 selects, 127 VGPRs).
template <int kPass, int kFmt, class FitLds, class AccT>
__device__ __forceinline__ void stream_views(FitLds &lds, const uint8_t *__restrict__ tile_obs, uint32_t r,
                                             uint32_t nfull, int wave, int lane, const Water &w,
                                             const float (&J)[3][4], AccT &acc) {
    uint32_t slot = 0, slot_in = kAhead;  // slot_in = (v + kAhead) % kRing
    const uint32_t rf = min(r, views_of_wave(nfull, wave));
    stream_range<kPass, false, kFmt>(lds, tile_obs, r, 0u, rf, wave, lane, w, J, acc, slot, slot_in);
    stream_range<kPass, true, kFmt>(lds, tile_obs, r, rf, r, wave, lane, w, J, acc, slot, slot_in);
    wait_chunks<0>();  // nothing of ours is in flight past this point
}
static_assert(kAhead == 2 || kAhead == 3, "wait ladder in stream_views covers kAhead 2 and 3");

__device__ __forceinline__ void zero_acc(Acc &a) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { a.pa[c][j] = 0.f; a.pb[c][j] = 0.f; }
        a.sB[c] = 0.f;
        a.sGZ[c] = 0.f;
    }
    a.cost = 0.f;
}

// Adds the four waves' per-pixel sums (fixed order) and returns, for pixel slot `t`, the six totals.
// The leading barrier retires every wave's ring before `red` (which overlays it) is written.
template <class FitLds>
__device__ __forceinline__ void reduce_pixels(FitLds &lds, const Acc &acc, int wave, int lane, int t, float out[6]) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        *reinterpret_cast<float4 *>(&lds.u.red[wave][c][lane * 4]) =
            make_float4(acc.pa[c][0], acc.pa[c][1], acc.pa[c][2], acc.pa[c][3]);
        *reinterpret_cast<float4 *>(&lds.u.red[wave][3 + c][lane * 4]) =
            make_float4(acc.pb[c][0], acc.pb[c][1], acc.pb[c][2], acc.pb[c][3]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 6; ++q)
        out[q] = ((lds.u.red[0][q][t] + lds.u.red[1][q][t]) + lds.u.red[2][q][t]) + lds.u.red[3][q][t];
}

// Two-level, fixed-order float64 reduction of the per-tile partials (layout [kNumSums][n_tiles]):
//   gpart[q][g] = sum of the 32 tiles of group g       (one load per lane + fixed-shape shuffle tree)
//   sums[q]     = sum over the groups                  (4 loads per lane + the same tree)
// The same two functions run in the fused tail (group-last / global-last workgroup) and in the split-path
// kernel, so both paths produce the same bits.  All hand-off data move with agent-scope (sc1) accesses.
__device__ __forceinline__ double wave_sum_fixed(double x) {  // fixed-shape tree: same bits on every run
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

// 256 threads: wave w reduces quantities q = w, w+4, w+8; lane l holds tile 32 g + l (lanes >= 32 hold 0).
__device__ __forceinline__ void reduce_group(const float *partials, int n_tiles, int g, double *gpart, int n_groups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = g * kGroup + lane;
    for (int q = wave; q < kNumSums; q += 4) {
        double x = 0.0;
        if (lane < kGroup && tile < n_tiles)
            x = (double)__hip_atomic_load(partials + (size_t)q * n_tiles + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = wave_sum_fixed(x);
        if (lane == 0) __hip_atomic_store(gpart + (size_t)q * n_groups + g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// 256 threads: wave w reduces quantities q = w, w+4, w+8 over all groups (lane l takes groups l, l+64, ...).
__device__ __forceinline__ void reduce_total(const double *gpart, int n_groups, double *stot, double *__restrict__ sums) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = wave; q < kNumSums; q += 4) {
        double x = 0.0;
        for (int g = lane; g < n_groups; g += 64)
            x += __hip_atomic_load(gpart + (size_t)q * n_groups + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = wave_sum_fixed(x);
        if (lane == 0) {
            sums[q] = x;  // global copy: read by the host all-reduce in shared-water runs
            stot[q] = x;  // LDS copy: read by water_step of the same workgroup
        }
    }
    __syncthreads();
}

// Arrival on a counter (relaxed agent-scope fetch_add by one lane behind the wave's drained stores); returns
// true in the workgroup that arrived last, which has then done its agent acquire and re-armed the counter.
__device__ __forceinline__ bool arrive_last(unsigned *counter, unsigned expected, int *flag) {
    const int t = threadIdx.x;
    if (t < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0) {
            const unsigned got = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (got == expected - 1u) ? 1 : 0;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
            }
            *flag = last;
        }
    }
    __syncthreads();
    return *flag != 0;
}

// torch.optim.Adam step on B, beta, gamma (sucre.py:148) from the reduced sums, by lanes 0..8 of one wave;
// logs cost + parameters (sucre.py:149-152) into one trace row.  All lanes of the wave must call it.
__device__ __forceinline__ void water_step(const double *__restrict__ sums, float *__restrict__ pstate,
                                           const uint64_t *__restrict__ n_obs_total, const AdamCoef &co,
                                           double *__restrict__ trace_row) {
    const int q = threadIdx.x & 63;
    const float scale = (1.0f / 3.0f) / (float)(*n_obs_total);
    float p = 0.f, m = 0.f, v = 0.f;
    double g = 0.0;
    if (q < 9) {
        const int c = q % 3;
        if (q < 3) g = -2.0 * (double)scale * sums[c];                            // dL/dB
        else if (q < 6) g = 2.0 * (double)scale * sums[6 + c];                    // dL/dbeta
        else g = -2.0 * (double)scale * (double)pstate[c] * sums[3 + c];           // dL/dgamma (B before its step)
        p = pstate[q];
        m = pstate[9 + q];
        v = pstate[18 + q];
    }
    __builtin_amdgcn_wave_barrier();  // B is read (above) by lanes 6..8 before lanes 0..2 overwrite it (below)
    if (q < 9) {
        adam_update(p, m, v, (float)g, co);
        pstate[q] = p;
        pstate[9 + q] = m;
        pstate[18 + q] = v;
    }
    if (trace_row) {
        if (q < 9) trace_row[1 + q] = (double)p;
        if (q == 9) trace_row[0] = sums[9];
    }
}

// End of a fit launch: the workgroup's ten sums -> one float32 partial each -> (fused form) two-level last-arriver
// reduction in float64 and the Adam step on B, beta, gamma by the workgroup that arrives last.
template <bool kFused, class FitLds>
__device__ __forceinline__ void finish_launch(FitLds &lds, float (&s)[kNumSums], float *partials, const AdamCoef &co,
                                              unsigned *ticket, double *gpart, int n_groups, double *sums,
                                              float *pstate, const uint64_t *__restrict__ n_obs_total,
                                              double *trace_row) {
    const int n_blocks = gridDim.x;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // ten workgroup sums: wave shuffle tree, then the four waves in fixed order
#pragma unroll
    for (int q = 0; q < kNumSums; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kNumSums; ++q) lds.wsum[wave][q] = s[q];
    }
    __syncthreads();
    // Publish the workgroup's partials with agent-scope write-through (sc1) stores: they need no release fence
    // (a release = L2 write-back in EVERY workgroup measured +260 us per launch; cdna_hip_programming.md section 5,
    // 'In-launch split-K reduction').
    if (t < kNumSums)
        __hip_atomic_store(partials + (size_t)t * n_blocks + blockIdx.x,
                           ((lds.wsum[0][t] + lds.wsum[1][t]) + lds.wsum[2][t]) + lds.wsum[3][t], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);

    if (kFused) {
        // Two-level last-arriver tail, so an iteration is ONE launch and no counter sees more than 48 arrivals
        // (one word shared by 8160 arrivals saturated at ~88 returning atomics/us, measured).  Hand-off per
        // Guideline 16, sc1 form: payload stored write-through, the storing wave drains vmcnt, one lane does a relaxed
        // agent-scope fetch_add; the last arriver acquires once.  Nobody spins.
        const int g = blockIdx.x / kGroup;
        const unsigned gsize = (unsigned)(min((g + 1) * kGroup, n_blocks) - g * kGroup);
        if (arrive_last(ticket + (size_t)(1 + g) * kTicketStride, gsize, &lds.is_last)) {  // workgroup-uniform
            reduce_group(partials, n_blocks, g, gpart, n_groups);
            if (arrive_last(ticket, (unsigned)n_groups, &lds.is_last_total)) {
                reduce_total(gpart, n_groups, lds.stot, sums);
                // every other workgroup has finished (it arrived after its last use of the parameters)
                if (t < 64) water_step(lds.stot, pstate, n_obs_total, co, trace_row);
            }
        }
    }
}

template <bool kFused, int kFmt>
__global__ __launch_bounds__(256) void fit_grad_kernel(const uint8_t *__restrict__ comp,
                                                       const uint64_t *__restrict__ tile_off,
                                                       const uint32_t *__restrict__ levels,
                                                       const uint32_t *__restrict__ full, int n_tiles,
                                                       float *pstate, const uint64_t *__restrict__ n_obs_total,
                                                       float *__restrict__ Jt, float *__restrict__ mt,
                                                       float *__restrict__ vt, float *partials, const AdamCoef co,
                                                       unsigned *ticket, double *gpart, int n_groups, double *sums,
                                                       double *trace_row, const uint32_t *__restrict__ obs_format) {
    __shared__ FitLds lds;  // 24.9 KB: 6 workgroups per CU
    const int n_blocks = gridDim.x;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const Water w = load_water(pstate);
    const float gscale = -2.0f * ((1.0f / 3.0f) / (float)(*n_obs_total));  // (loss / n_obs / 3).backward(), sucre.py:145

    // The workgroup is persistent over tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the compact store's tiles are
    // sorted heaviest first, so this deal is balanced).  The per-thread global sums simply keep accumulating across
    // its tiles; the shuffle reduction, the sc1 publish and the ticket hand-off -- measured at ~40 us per launch when
    // done once per tile (bisected with early-return builds) -- happen once per workgroup.
    Acc acc;
    zero_acc(acc);
    float sBetaAcc[3] = {0.f, 0.f, 0.f};
    // a store compacted in the other format is not read at all; the logged cost turns NaN instead
    const bool fmt_ok = *obs_format == (uint32_t)kFmt;
    if (!fmt_ok) acc.cost = __builtin_nanf("");

#ifdef SUCRE_EXP_SNAKE
    for (int stripe = 0; stripe * n_blocks < n_tiles; ++stripe) {
        const int tile = stripe * n_blocks + ((stripe & 1) ? n_blocks - 1 - (int)blockIdx.x : (int)blockIdx.x);
        if (tile >= n_tiles) continue;
#else
    for (int tile = blockIdx.x; tile < n_tiles; tile += n_blocks) {
#endif
        const uint32_t n = fmt_ok ? levels[tile] : 0u, nfull = fmt_ok ? full[tile] : 0u;
        const uint8_t *tile_obs = comp + tile_off[tile];
        float *Jtile = Jt + (size_t)tile * 3 * kTilePx;
        const uint32_t r = views_of_wave(n, wave);
        float J[3][4];
        float tot[6];
        stream_begin<kFmt>(lds, tile_obs, r, wave, lane);
        {
            // J of this lane's four pixels: ordinary loads issued BEHIND the ring prologue.  hipcc does not count
            // the asm DMAs, so the wait it emits for J (vmcnt(0)) would also drain whatever DMA is in flight at J's
            // first use; the empty asm makes that first use happen here, where only the prologue (issued at the same
            // time, hence landing at the same time) is outstanding: one shared start-up latency per tile.
            float4 jv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) jv[c] = *reinterpret_cast<const float4 *>(Jtile + c * kTilePx + lane * 4);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                asm volatile("" : "+v"(jv[c].x), "+v"(jv[c].y), "+v"(jv[c].z), "+v"(jv[c].w));
                J[c][0] = jv[c].x; J[c][1] = jv[c].y; J[c][2] = jv[c].z; J[c][3] = jv[c].w;
            }
        }

        // per-pixel sums restart with every tile; the global sums (sB, sGZ, cost) carry on
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc.pa[c][j] = 0.f; acc.pb[c][j] = 0.f; }
        stream_views<kPassGradJ, kFmt>(lds, tile_obs, r, nfull, wave, lane, w, J, acc);
        reduce_pixels(lds, acc, wave, lane, t, tot);

        // pixel-parallel tail: this thread owns pixel slot t
        {
            float *mtile = mt + (size_t)tile * 3 * kTilePx;
            float *vtile = vt + (size_t)tile * 3 * kTilePx;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float Jc = Jtile[c * kTilePx + t], m = mtile[c * kTilePx + t], v = vtile[c * kTilePx + t];
                // unobserved pixels: zero sum, and J (possibly NaN) must not leak into the beta gradient
                sBetaAcc[c] += (tot[3 + c] == 0.0f) ? 0.0f : Jc * tot[3 + c];
                adam_update(Jc, m, v, gscale * tot[c], co);
                Jtile[c * kTilePx + t] = Jc;
                mtile[c * kTilePx + t] = m;
                vtile[c * kTilePx + t] = v;
            }
        }
        __syncthreads();  // `red` retired before the next tile's ring prologue overwrites the LDS
    }

    float s[kNumSums] = {acc.sB[0], acc.sB[1], acc.sB[2], acc.sGZ[0], acc.sGZ[1], acc.sGZ[2],
                         sBetaAcc[0], sBetaAcc[1], sBetaAcc[2], acc.cost};
    finish_launch<kFused>(lds, s, partials, co, ticket, gpart, n_groups, sums, pstate, n_obs_total, trace_row);
}

// Closed-form mode, one observation pass per iteration (see AccOne).
template <bool kFused, int kFmt>
__global__ __launch_bounds__(256) void fit_closed_kernel(const uint8_t *__restrict__ comp,
                                                         const uint64_t *__restrict__ tile_off,
                                                         const uint32_t *__restrict__ levels,
                                                         const uint32_t *__restrict__ full, int n_tiles,
                                                         float *pstate, const uint64_t *__restrict__ n_obs_total,
                                                         float *__restrict__ Jt, float *partials, const AdamCoef co,
                                                         unsigned *ticket, double *gpart, int n_groups, double *sums,
                                                         double *trace_row, const uint32_t *__restrict__ obs_format) {
    __shared__ FitLdsOne lds;  // 37 KB; the kernel is register-bound (3 workgroups per CU) before it is LDS-bound
    const int n_blocks = gridDim.x;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const Water w = load_water(pstate);
    const bool fmt_ok = *obs_format == (uint32_t)kFmt;
    float sB[3] = {0.f, 0.f, 0.f}, sGZ[3] = {0.f, 0.f, 0.f}, sBeta[3] = {0.f, 0.f, 0.f};
    float cost = fmt_ok ? 0.f : __builtin_nanf("");

    for (int tile = blockIdx.x; tile < n_tiles; tile += n_blocks) {
        const uint32_t n = fmt_ok ? levels[tile] : 0u, nfull = fmt_ok ? full[tile] : 0u;
        const uint8_t *tile_obs = comp + tile_off[tile];
        const uint32_t r = views_of_wave(n, wave);
        AccOne acc;
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc.q[q][c][j] = 0.f;
        float *Jtile = Jt + (size_t)tile * 3 * kTilePx;
        stream_begin<kFmt>(lds, tile_obs, r, wave, lane);
        // previous J of this lane's four pixels: ordinary loads issued behind the ring prologue (see fit_grad_kernel)
        float Jp[3][4];
        {
            float4 jv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) jv[c] = *reinterpret_cast<const float4 *>(Jtile + c * kTilePx + lane * 4);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                asm volatile("" : "+v"(jv[c].x), "+v"(jv[c].y), "+v"(jv[c].z), "+v"(jv[c].w));
                // a pixel that had no J so far (NaN: never observed, or a warm start without it) is measured from 0
                Jp[c][0] = finite_or_zero(jv[c].x); Jp[c][1] = finite_or_zero(jv[c].y);
                Jp[c][2] = finite_or_zero(jv[c].z); Jp[c][3] = finite_or_zero(jv[c].w);
            }
        }
        stream_views<kPassClosedJ, kFmt>(lds, tile_obs, r, nfull, wave, lane, w, Jp, acc);

        // the four waves' per-pixel sums, three quantities (x 3 channels) per round through `red`
        float tot[9][3];
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            __syncthreads();  // ring (round 0) / previous round retired
#pragma unroll
            for (int qq = 0; qq < 3; ++qq)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    *reinterpret_cast<float4 *>(&lds.u.red[wave][qq * 3 + c][lane * 4]) =
                        make_float4(acc.q[round * 3 + qq][c][0], acc.q[round * 3 + qq][c][1],
                                    acc.q[round * 3 + qq][c][2], acc.q[round * 3 + qq][c][3]);
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < 3; ++qq)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int pl = qq * 3 + c;
                    tot[round * 3 + qq][c] =
                        ((lds.u.red[0][pl][t] + lds.u.red[1][pl][t]) + lds.u.red[2][pl][t]) + lds.u.red[3][pl][t];
                }
        }
        // pixel-parallel tail: this thread owns pixel slot t
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float N = tot[0][c], D = tot[1][c];
            const float dJ = N / D;                          // 0/0 = NaN where nothing was observed (sucre.py:77)
            const float Jc = finite_or_zero(Jtile[c * kTilePx + t]) + dJ;    // = sum y a / sum a^2
            Jtile[c * kTilePx + t] = Jc;
            if (D != 0.0f) {
                sB[c] += __builtin_fmaf(-dJ, tot[3][c], tot[2][c]);
                sBeta[c] += Jc * __builtin_fmaf(-dJ, tot[5][c], tot[4][c]);
                sGZ[c] += __builtin_fmaf(-dJ, tot[7][c], tot[6][c]);
                cost += __builtin_fmaf(-dJ, N, tot[8][c]);
            }
        }
        __syncthreads();  // `red` retired before the next tile's ring prologue overwrites the LDS
    }
    float s[kNumSums] = {sB[0], sB[1], sB[2], sGZ[0], sGZ[1], sGZ[2], sBeta[0], sBeta[1], sBeta[2], cost};
    finish_launch<kFused>(lds, s, partials, co, ticket, gpart, n_groups, sums, pstate, n_obs_total, trace_row);
}

// SUCRe.update_J alone (sucre.py:66-77, 156)
template <int kFmt>
__global__ __launch_bounds__(256) void update_J_kernel(const uint8_t *__restrict__ comp,
                                                       const uint64_t *__restrict__ tile_off,
                                                       const uint32_t *__restrict__ levels,
                                                       const uint32_t *__restrict__ full,
                                                       const float *__restrict__ params, float *__restrict__ Jt,
                                                       const uint32_t *__restrict__ obs_format) {
    __shared__ FitLds lds;
    const int tile = blockIdx.x;
    if (*obs_format != (uint32_t)kFmt) {  // wrong format announced by the caller: poison instead of misreading
        for (int c = 0; c < 3; ++c) Jt[((size_t)tile * 3 + c) * kTilePx + threadIdx.x] = __builtin_nanf("");
        return;
    }
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const Water w = load_water(params);
    const uint32_t r = views_of_wave(levels[tile], wave), nfull = full[tile];
    const uint8_t *tile_obs = comp + tile_off[tile];
    float J[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) J[c][j] = 0.f;
    Acc acc;
    zero_acc(acc);
    stream_begin<kFmt>(lds, tile_obs, r, wave, lane);
    stream_views<kPassClosedJ, kFmt>(lds, tile_obs, r, nfull, wave, lane, w, J, acc);
    float tot[6];
    reduce_pixels(lds, acc, wave, lane, t, tot);
#pragma unroll
    for (int c = 0; c < 3; ++c) Jt[((size_t)tile * 3 + c) * kTilePx + t] = tot[c] / tot[3 + c];
}

__global__ __launch_bounds__(256) void reduce_groups_kernel(const float *partials, int n_tiles, double *gpart,
                                                            int n_groups) {
    reduce_group(partials, n_tiles, blockIdx.x, gpart, n_groups);
}

__global__ __launch_bounds__(256) void reduce_sums_kernel(const double *gpart, int n_groups, double *__restrict__ sums) {
    __shared__ double stot[kSumsPad];
    reduce_total(gpart, n_groups, stot, sums);
}

__global__ __launch_bounds__(64) void param_step_kernel(const double *__restrict__ sums, float *__restrict__ pstate,
                                                        const uint64_t *__restrict__ n_obs_total, const AdamCoef co,
                                                        double *__restrict__ trace_row) {
    water_step(sums, pstate, n_obs_total, co, trace_row);
}

struct Params9 { float v[9]; };

__global__ __launch_bounds__(256) void fit_init_kernel(const uint8_t *__restrict__ rgb1,
                                                       const float *__restrict__ depth1,
                                                       const float *__restrict__ J0, int H, int W, int tiles_x,
                                                       const uint32_t *__restrict__ perm, float *__restrict__ Jt, float *__restrict__ mt,
                                                       float *__restrict__ vt, float *__restrict__ pstate,
                                                       unsigned *__restrict__ ticket, int n_tickets,
                                                       const Params9 p0) {
    const int tile = blockIdx.x, t = threadIdx.x;  // sorted tile / slot; perm gives the pixel that lives there
    const uint32_t src = perm[(size_t)tile * kTilePx + t];
    const int stile = src / kTilePx, sslot = src % kTilePx;
    const int ty = stile / tiles_x, tx = stile - ty * tiles_x;
    const int v = ty * kTile + (sslot >> 4), u = tx * kTile + (sslot & 15);
    const bool inside = v < H && u < W;
    const size_t o = inside ? (size_t)v * W + u : 0;
    const bool valid = inside && !(depth1[o] <= 0.0f);  // self.J[depth <= 0] = nan, sucre.py:48
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float Jc = __builtin_nanf("");
        if (valid) Jc = J0 ? J0[o * 3 + c] : unit_from_u8(rgb1[o * 3 + c]);
        const size_t i = ((size_t)tile * 3 + c) * kTilePx + t;
        Jt[i] = Jc;
        mt[i] = 0.f;
        vt[i] = 0.f;
    }
    if (tile == 0 && t < 27) pstate[t] = t < 9 ? p0.v[t] : 0.f;
    for (int i = tile * 256 + t; i < n_tickets; i += gridDim.x * 256) ticket[i] = 0u;
}

__global__ __launch_bounds__(256) void export_J_kernel(const float *__restrict__ Jt, int H, int W, int tiles_x,
                                                       const uint32_t *__restrict__ invperm, float *__restrict__ J) {
    const int tile = blockIdx.x, t = threadIdx.x;  // image tile / slot; invperm says where the pixel was sorted to
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int v = ty * kTile + (t >> 4), u = tx * kTile + (t & 15);
    if (v >= H || u >= W) return;
    const uint32_t dst = invperm[(size_t)tile * kTilePx + t];
    const size_t o = ((size_t)v * W + u) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) J[o + c] = Jt[((size_t)(dst / kTilePx) * 3 + c) * kTilePx + dst % kTilePx];
}

__global__ void set_n_obs_total_kernel(uint64_t *dst, uint64_t v) { *dst = v; }

hipError_t launch_fit_init(const Layout &L, uint8_t *ws, const uint8_t *rgb1, const float *depth1,
                           const float *params0, const float *J0, hipStream_t s) {
    Params9 p0;
    for (int i = 0; i < 9; ++i) p0.v[i] = params0[i];
    hipLaunchKernelGGL(fit_init_kernel, dim3(L.n_tiles), dim3(256), 0, s, rgb1, depth1, J0, L.H, L.W, L.tiles_x,
                       reinterpret_cast<const uint32_t *>(ws + L.off_perm), reinterpret_cast<float *>(ws + L.off_J), reinterpret_cast<float *>(ws + L.off_m),
                       reinterpret_cast<float *>(ws + L.off_v), reinterpret_cast<float *>(ws + L.off_params),
                       reinterpret_cast<unsigned *>(ws + L.off_ticket), (1 + L.n_groups) * kTicketStride, p0);
    return hipGetLastError();
}

template <bool kFused, int kFmt>
static void launch_grad_fmt(const Layout &L, uint8_t *ws, const AdamCoef &co, double *trace_row, hipStream_t s) {
    hipLaunchKernelGGL((fit_grad_kernel<kFused, kFmt>), dim3(L.n_blocks), dim3(256), 0, s, ws + L.off_comp,
                       reinterpret_cast<const uint64_t *>(ws + L.off_tile_off),
                       reinterpret_cast<const uint32_t *>(ws + L.off_levels),
                       reinterpret_cast<const uint32_t *>(ws + L.off_full), L.n_tiles,
                       reinterpret_cast<float *>(ws + L.off_params),
                       reinterpret_cast<const uint64_t *>(ws + L.off_n_obs_total),
                       reinterpret_cast<float *>(ws + L.off_J), reinterpret_cast<float *>(ws + L.off_m),
                       reinterpret_cast<float *>(ws + L.off_v), reinterpret_cast<float *>(ws + L.off_partials), co,
                       reinterpret_cast<unsigned *>(ws + L.off_ticket),
                       reinterpret_cast<double *>(ws + L.off_gpartials), L.n_groups,
                       reinterpret_cast<double *>(ws + L.off_sums), trace_row,
                       reinterpret_cast<const uint32_t *>(ws + L.off_total_chunks + sizeof(uint64_t)));
}

template <bool kFused, int kFmt>
static void launch_closed_fmt(const Layout &L, uint8_t *ws, const AdamCoef &co, double *trace_row, hipStream_t s) {
    hipLaunchKernelGGL((fit_closed_kernel<kFused, kFmt>), dim3(L.n_blocks), dim3(256), 0, s, ws + L.off_comp,
                       reinterpret_cast<const uint64_t *>(ws + L.off_tile_off),
                       reinterpret_cast<const uint32_t *>(ws + L.off_levels),
                       reinterpret_cast<const uint32_t *>(ws + L.off_full), L.n_tiles,
                       reinterpret_cast<float *>(ws + L.off_params),
                       reinterpret_cast<const uint64_t *>(ws + L.off_n_obs_total),
                       reinterpret_cast<float *>(ws + L.off_J), reinterpret_cast<float *>(ws + L.off_partials), co,
                       reinterpret_cast<unsigned *>(ws + L.off_ticket),
                       reinterpret_cast<double *>(ws + L.off_gpartials), L.n_groups,
                       reinterpret_cast<double *>(ws + L.off_sums), trace_row,
                       reinterpret_cast<const uint32_t *>(ws + L.off_total_chunks + sizeof(uint64_t)));
}

// J-parameter mode: gradient pass + Adam on J; closed-form mode: the one-pass kernel (J re-solved, then constant).
template <bool kFused>
static void launch_grad_variant(const Layout &L, uint8_t *ws, const AdamCoef &co, unsigned flags, double *trace_row,
                                hipStream_t s) {
    const bool u16 = (flags & SUCRE_FIT_OBS_U16MM) != 0;
    if (flags & SUCRE_FIT_CLOSED_FORM) {
        if (u16) launch_closed_fmt<kFused, 1>(L, ws, co, trace_row, s);
        else launch_closed_fmt<kFused, 0>(L, ws, co, trace_row, s);
    } else {
        if (u16) launch_grad_fmt<kFused, 1>(L, ws, co, trace_row, s);
        else launch_grad_fmt<kFused, 0>(L, ws, co, trace_row, s);
    }
}

// One whole iteration in a single launch (gradient pass + last-arriver reduction + water-parameter step).
hipError_t launch_fit_iter_fused(const Layout &L, uint8_t *ws, const AdamCoef &co, unsigned flags,
                                 double *trace_row, hipStream_t s) {
    launch_grad_variant<true>(L, ws, co, flags, trace_row, s);
    return hipGetLastError();
}

// Split form for multi-GPU shared-water runs: gradient pass + reduction, sums left at off_sums for the host's
// all-reduce; launch_fit_step applies them.
hipError_t launch_fit_grad(const Layout &L, uint8_t *ws, const AdamCoef &co, unsigned flags, hipStream_t s) {
    launch_grad_variant<false>(L, ws, co, flags, nullptr, s);
    hipLaunchKernelGGL(reduce_groups_kernel, dim3(L.n_groups), dim3(256), 0, s,
                       reinterpret_cast<const float *>(ws + L.off_partials), L.n_blocks,
                       reinterpret_cast<double *>(ws + L.off_gpartials), L.n_groups);
    hipLaunchKernelGGL(reduce_sums_kernel, dim3(1), dim3(256), 0, s,
                       reinterpret_cast<const double *>(ws + L.off_gpartials), L.n_groups,
                       reinterpret_cast<double *>(ws + L.off_sums));
    return hipGetLastError();
}

hipError_t launch_fit_step(const Layout &L, uint8_t *ws, const AdamCoef &co, double *trace_row, hipStream_t s) {
    hipLaunchKernelGGL(param_step_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const double *>(ws + L.off_sums),
                       reinterpret_cast<float *>(ws + L.off_params),
                       reinterpret_cast<const uint64_t *>(ws + L.off_n_obs_total), co, trace_row);
    return hipGetLastError();
}

template <int kFmt>
static void launch_update_J_fmt(const Layout &L, uint8_t *ws, hipStream_t s) {
    hipLaunchKernelGGL(update_J_kernel<kFmt>, dim3(L.n_tiles), dim3(256), 0, s, ws + L.off_comp,
                       reinterpret_cast<const uint64_t *>(ws + L.off_tile_off),
                       reinterpret_cast<const uint32_t *>(ws + L.off_levels),
                       reinterpret_cast<const uint32_t *>(ws + L.off_full),
                       reinterpret_cast<const float *>(ws + L.off_params), reinterpret_cast<float *>(ws + L.off_J),
                       reinterpret_cast<const uint32_t *>(ws + L.off_total_chunks + sizeof(uint64_t)));
}

hipError_t launch_update_J(const Layout &L, uint8_t *ws, int fmt, hipStream_t s) {
    if (fmt) launch_update_J_fmt<1>(L, ws, s);
    else launch_update_J_fmt<0>(L, ws, s);
    return hipGetLastError();
}

hipError_t launch_export_J(const Layout &L, const uint8_t *ws, float *J, hipStream_t s) {
    hipLaunchKernelGGL(export_J_kernel, dim3(L.n_tiles), dim3(256), 0, s,
                       reinterpret_cast<const float *>(ws + L.off_J), L.H, L.W, L.tiles_x,
                       reinterpret_cast<const uint32_t *>(ws + L.off_invperm), J);
    return hipGetLastError();
}

hipError_t launch_set_n_obs_total(const Layout &L, uint8_t *ws, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(set_n_obs_total_kernel, dim3(1), dim3(1), 0, s,
                       reinterpret_cast<uint64_t *>(ws + L.off_n_obs_total), n);
    return hipGetLastError();
}

}  // namespace sucre
