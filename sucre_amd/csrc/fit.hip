// Exponential-model fit for gfx950.
//
// One iteration of sucre.adam (sucre.py:138-148) = ONE launch of fit_grad_kernel (J as a parameter) or
// fit_closed_kernel (--use-closed-form): it streams the compact observation store once, updates J in place, reduces
// the ten global sums and -- in the workgroup that arrives last -- applies Adam to B, beta, gamma and logs the row.
// The split form (launch_fit_grad + launch_fit_step) leaves the sums in the workspace for a host all-reduce.
//
// Work decomposition: 256 x kFitWaves (1280) persistent 256-thread workgroups; every WAVE works alone on strips of 64 count-sorted
// pixels (csrc/compact.hip, layout.h), one pixel per lane, and never synchronises with the other waves until the
// launch's final reduction.  A strip is a sequence of ITEMS -- its J plane (768 B), its observation chunks (64 pixels
// x 4 levels: 1792 B of float32 ranges + colours, 1536 B when the ranges are kept as 24-bit codes, 1280 B as uint16
// millimetres; the last one may hold fewer levels) and, in J-parameter mode, its Adam moments (1536 B) --
// and the strips of a wave follow each other without a gap, so a wave sees ONE stream of items.  Every item is
// copied HBM -> LDS by two LDS-DMA instructions into a private ring of kRing slots, kAhead items ahead of the one
// being consumed, across strip boundaries: prefetch depth costs LDS, not registers, no latency is exposed between
// strips, and the only ordinary memory instructions of the loop are the stores of J (and the moments) at a strip's
// end.  Per-pixel sums live in the lane that owns the pixel: no atomics, no cross-wave reduction, and Adam on J
// runs in the same lane (fit_math.h adam_update_J: torch.optim.Adam's update with the hardware square root and
// reciprocals, 1 ulp each, in place of the IEEE sequences -- held to the fit's tolerance, not to bit parity; the nine water
// parameters take the op-for-op IEEE form, adam_update).  The ten global sums go
// lane -> wave shuffle -> LDS -> one float32 partial per workgroup -> float64 fixed-order two-level reduction, so
// results are bitwise reproducible.
//
// Model (sucre.py:79-82, l = 1):  Ihat = J a + B (1 - g),  a = exp(-beta z),  g = exp(-gamma z),  r = I - Ihat.
// With L = sum r^2 / (3 n_obs) and s = (1/3)/n_obs (sucre.py:145):
//   dL/dJ[p]  = -2 s sum_k r a          dL/dB     = -2 s sum r (1 - g)
//   dL/dbeta  = +2 s sum_p J sum_k r a z    dL/dgamma = -2 s B sum r g z        (per channel)
#include <cstddef>

#include "experiment.h"
#include "fit_math.h"
#include "handoff.h"

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

// The same for a caller that keeps other things in vector registers across the pass: every value made wave-uniform
// explicitly, so that the nine parameters live in scalar registers.
__device__ __forceinline__ float uniform_f(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
__device__ __forceinline__ Water load_water_uniform(const float *params) {
    Water w = load_water(params);
#pragma unroll
    for (int c = 0; c < 3; ++c) { w.B[c] = uniform_f(w.B[c]); w.nb[c] = uniform_f(w.nb[c]); w.ng[c] = uniform_f(w.ng[c]); }
    return w;
}

constexpr float kInv255 = (float)(1.0 / 255.0);

// ---------------------------------------------------------------------------------------------------------------
// Arithmetic of one chunk: four levels of this lane's pixel.
//
// The loop's arithmetic is what the launch waits for next to the stream: 13 plain instructions and 2 v_exp_f32 per
// observation-channel (an exponential issues at 1.7x the cost of a plain instruction and overlaps with nothing:
// tools/probes/exp_probe.hip; a three-register v_fma_f32 every 2.4-2.6 shader cycles per SIMD at 5-8 waves:
// tools/probes/fma3_probe.hip -- the loop as it runs reaches about half of that).  The 24 exponentials of a chunk are issued back to back, their arguments before them and their
// uses after them: interleaved with their dependent arithmetic the loop was 14 % slower (337 -> 296 ns per chunk per SIMD).
// ---------------------------------------------------------------------------------------------------------------
struct Acc {
    float pa[3];   // this pixel: sum r a
    float pb[3];   // this pixel: sum r a z
    float sB[3];   // lane's share of sum r (1 - g)
    float sGZ[3];  // lane's share of sum r g z
    float cost;    // lane's share of sum r^2
};

struct Exps { float a[kGroupLv][3], g[kGroupLv][3]; };

__device__ __forceinline__ void chunk_exps(const float (&zz)[kGroupLv], const Water &w, Exps &e) {
#pragma unroll
    for (int j = 0; j < kGroupLv; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { e.a[j][c] = zz[j] * w.nb[c]; e.g[j][c] = zz[j] * w.ng[c]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < kGroupLv; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { e.a[j][c] = fast_exp2(e.a[j][c]); e.g[j][c] = fast_exp2(e.g[j][c]); }
    __builtin_amdgcn_sched_barrier(0);
}

// kMasked = false: every level of the chunk is a real observation of every pixel of the strip (chunks wholly below
// the strip's smallest pixel count), so the z > 0 test and the selects are compiled out.
template <bool kMasked>
__device__ __forceinline__ void accumulate_chunk(const float (&zz)[kGroupLv], const uint32_t (&cc)[3], const Water &w,
                                                 const float (&J)[3], Acc &acc) {
    if (kExpNoCompute) {  // ablation build only (experiment.h): touch the data, skip the model
        acc.cost += (zz[0] + zz[1]) + (zz[2] + zz[3]) + (float)(cc[0] ^ cc[1] ^ cc[2]);
        return;
    }
    Exps e;
    chunk_exps(zz, w, e);
#pragma unroll
    for (int j = 0; j < kGroupLv; ++j) {
        const float z = zz[j];
        const bool valid = !kMasked || z > 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t k = (cc[c] >> (8 * j)) & 255u;
            const float a = e.a[j][c], g = e.g[j][c];
            const float omg = 1.0f - g;
            const float Ihat = __builtin_fmaf(J[c], a, w.B[c] * omg);
            // I = k/255 folded into the residual: one rounding instead of two, two VALU ops fewer
            float r = __builtin_fmaf((float)k, kInv255, -Ihat);
            r = valid ? r : 0.0f;  // select, not multiply: J may be NaN where unobserved
            const float rz = r * z;
            acc.cost = __builtin_fmaf(r, r, acc.cost);
            acc.pa[c] = __builtin_fmaf(r, a, acc.pa[c]);
            acc.pb[c] = __builtin_fmaf(rz, a, acc.pb[c]);
            acc.sB[c] = __builtin_fmaf(r, omg, acc.sB[c]);
            acc.sGZ[c] = __builtin_fmaf(rz, g, acc.sGZ[c]);
        }
    }
}

// One level (a strip's last, short chunk): same arithmetic, always masked.
__device__ __forceinline__ void accumulate_level(float z, const uint32_t (&k)[3], const Water &w, const float (&J)[3],
                                                 Acc &acc) {
    const bool valid = z > 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = fast_exp2(z * w.nb[c]);
        const float g = fast_exp2(z * w.ng[c]);
        const float omg = 1.0f - g;
        const float Ihat = __builtin_fmaf(J[c], a, w.B[c] * omg);
        float r = __builtin_fmaf((float)k[c], kInv255, -Ihat);
        r = valid ? r : 0.0f;
        const float rz = r * z;
        acc.cost = __builtin_fmaf(r, r, acc.cost);
        acc.pa[c] = __builtin_fmaf(r, a, acc.pa[c]);
        acc.pb[c] = __builtin_fmaf(rz, a, acc.pb[c]);
        acc.sB[c] = __builtin_fmaf(r, omg, acc.sB[c]);
        acc.sGZ[c] = __builtin_fmaf(rz, g, acc.sGZ[c]);
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
    float q[9][3];  // N, D, S1..S7 of this pixel, per channel
};

__device__ __forceinline__ void closed_terms(float z, float a, float g, uint32_t k, bool valid, float Bc, float Jp,
                                             float &q0, float &q1, float &q2, float &q3, float &q4, float &q5,
                                             float &q6, float &q7, float &q8) {
    const float omg = 1.0f - g;
    // I = k/255 folded into y: one rounding instead of two and two VALU operations fewer, like the J-parameter loop (the
    // kernel is instruction-limited: 79 instructions per observation, profiles/r03_closed_summary.txt)
    const float y = __builtin_fmaf((float)k, kInv255, -(Bc * omg));
    float p = __builtin_fmaf(-Jp, a, y);
    p = valid ? p : 0.0f;  // a padding slot contributes nothing (its Jp a is not zero)
    const float za = z * a, zg = z * g;  // a padding slot has z = 0, g = 1: it only touches D (masked below)
    q0 = __builtin_fmaf(p, a, q0);
    q1 = __builtin_fmaf(a, valid ? a : 0.0f, q1);   // one operation (the file is built with -ffp-contract=off)
    q2 = __builtin_fmaf(p, omg, q2);
    q3 = __builtin_fmaf(a, omg, q3);
    q4 = __builtin_fmaf(p, za, q4);
    q5 = __builtin_fmaf(a, za, q5);
    q6 = __builtin_fmaf(p, zg, q6);
    q7 = __builtin_fmaf(a, zg, q7);
    q8 = __builtin_fmaf(p, p, q8);
}

template <bool kMasked>
__device__ __forceinline__ void accumulate_chunk(const float (&zz)[kGroupLv], const uint32_t (&cc)[3], const Water &w,
                                                 const float (&Jp)[3], AccOne &acc) {
    if (kExpNoCompute) {  // ablation build only (experiment.h): touch the data, skip the model
        acc.q[8][0] += (zz[0] + zz[1]) + (zz[2] + zz[3]) + (float)(cc[0] ^ cc[1] ^ cc[2]);
        return;
    }
    if (kExpHalfExps) {   // (experiment.h: the chunk's exponentials in two batches of twelve -- twelve registers fewer in flight)
#pragma unroll
        for (int h = 0; h < kGroupLv; h += 2) {
            float ea[2][3], eg[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) { ea[j][c] = zz[h + j] * w.nb[c]; eg[j][c] = zz[h + j] * w.ng[c]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) { ea[j][c] = fast_exp2(ea[j][c]); eg[j][c] = fast_exp2(eg[j][c]); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float z = zz[h + j];
                const bool valid = !kMasked || z > 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    closed_terms(z, ea[j][c], eg[j][c], (cc[c] >> (8 * (h + j))) & 255u, valid, w.B[c], Jp[c], acc.q[0][c], acc.q[1][c],
                                 acc.q[2][c], acc.q[3][c], acc.q[4][c], acc.q[5][c], acc.q[6][c], acc.q[7][c], acc.q[8][c]);
            }
        }
        return;
    }
    Exps e;
    chunk_exps(zz, w, e);
#pragma unroll
    for (int j = 0; j < kGroupLv; ++j) {
        const float z = zz[j];
        const bool valid = !kMasked || z > 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            closed_terms(z, e.a[j][c], e.g[j][c], (cc[c] >> (8 * j)) & 255u, valid, w.B[c], Jp[c], acc.q[0][c], acc.q[1][c],
                         acc.q[2][c], acc.q[3][c], acc.q[4][c], acc.q[5][c], acc.q[6][c], acc.q[7][c], acc.q[8][c]);
    }
}

__device__ __forceinline__ void accumulate_level(float z, const uint32_t (&k)[3], const Water &w, const float (&Jp)[3],
                                                 AccOne &acc) {
    if (kExpNoCompute) {
        acc.q[8][0] += z + (float)(k[0] ^ k[1] ^ k[2]);
        return;
    }
    const bool valid = z > 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        closed_terms(z, fast_exp2(z * w.nb[c]), fast_exp2(z * w.ng[c]), k[c], valid, w.B[c], Jp[c], acc.q[0][c], acc.q[1][c],
                     acc.q[2][c], acc.q[3][c], acc.q[4][c], acc.q[5][c], acc.q[6][c], acc.q[7][c], acc.q[8][c]);
}

// ---------------------------------------------------------------------------------------------------------------
// Item stream: per-wave LDS-DMA ring.
//
// An item is copied by two LDS-DMA instructions (global_load_lds_dwordx4: lane i moves 16 bytes from src + voff_i to
// slot + 16 i -- a dwordx3 one also strides by 16 and leaves holes, tools/probes/lds_dma_probe.hip).  ALL 64 lanes are
// active in both (round 4; until then the lanes beyond the item were masked out of EXEC: two 64-bit masks computed and
// EXEC written four times per item, ~25 scalar instructions of the ~85 an item cost next to its 180 of arithmetic):
//   A  lanes 0 .. nA-1 copy the item's first 16 nA bytes (nA = 64 for items of a KiB or more); the lanes beyond them
//      re-read lane nA-1's 16 bytes into the slot's bytes past 16 nA;
//   B  lanes 0 .. nB-1 copy the remaining 16 nB bytes to slot + 16 nA; the lanes beyond them re-read lane nB-1's 16 bytes
//      into what follows.  B is issued after A and A's surplus bytes only exist when nB = 0, so no surplus byte ever
//      lands on a real one; a slot is 2 KiB (16 nA + 1024 <= 2048), the ring's only cost: 24 KiB of LDS per workgroup.
// The surplus lanes read bytes some other lane of the same instruction reads: no extra traffic beyond the L1.
// The DMAs have no VGPR destination: hipcc can neither sink them next to their use nor drain them early; they live in
// inline asm and are waited for by hand-counted s_waitcnt vmcnt (vmcnt retires in issue order; cdna_hip_programming.md
// 5.7).  What may still be outstanding when item q is needed: the items issued after it (two instructions each) and, if
// they were issued after it, the stores of the previous strip's end.
// ---------------------------------------------------------------------------------------------------------------
// waves per SIMD the fit kernels are compiled for (register budget 512 / waves) = workgroups per CU of their
// persistent grids: kFitWaves / kClosedWaves (layout.h)
constexpr int kRing = SUCRE_RING;
constexpr int kAhead = kRing - 1;
constexpr int kSlot = 2048;  // largest item (a full float32 chunk, 1792 B) + room for the second DMA's surplus lanes

struct __attribute__((aligned(16))) FitLds {
    uint8_t ring[4][kRing][kSlot];  // per-wave item ring
    double stot[kSumsPad];
    float wsum[4][kNumSums];
    double wtotal[4][kNumSums];   // group launches: every wave's sums over the images it has walked so far
    int is_last, is_last_total;  // one flag word per arrive_last level: no wave can see the second verdict as the first
};

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));  // low 32 bits of a flat LDS address = LDS offset
}

// PlanItem.shape: [6:0] nA - 1, [14:8] max(nB, 1) - 1, [22:16] B's first 16-byte piece (nA, or nA - 1 when nB = 0: B then
// re-reads the item's last piece), [31] a full chunk (nA = 64 and nB = kFullB: both offsets are per-lane constants).
constexpr uint32_t kShapeFull = 1u << 31;
constexpr uint32_t kShapeTrail = 1u << 30;   // one of the trailing items behind a wave's last strip (nobody consumes it; a batch
                                             // launch copies the NEXT image's first items in its place: StreamChain)
__host__ __device__ constexpr uint32_t full_chunk_b(int fmt) { return (uint32_t)(chunk_bytes(fmt) - 1024) / 16u; }   // 48 / 16 / 32 lanes
__host__ __device__ __forceinline__ uint32_t item_shape(uint32_t bytes, bool full_chunk) {
    const uint32_t pieces = bytes / 16u;   // every item is a multiple of 16 bytes (static_asserts below)
    const uint32_t nA = pieces < 64u ? pieces : 64u, nB = pieces - nA;
    return (nA - 1u) | ((nB ? nB - 1u : 0u) << 8) | ((nB ? nA : nA - 1u) << 16) | (full_chunk ? kShapeFull : 0u);
}
static_assert(level_bytes(0) % 16 == 0 && level_bytes(1) % 16 == 0 && level_bytes(2) % 16 == 0 && level_bytes(3) % 16 == 0 && kChunk % 16 == 0 && kChunk16 % 16 == 0 && kChunk24 % 16 == 0 && kChunk26 % 16 == 0, "items are copied in 16-byte pieces");
static_assert(kChunk <= 2048 && kChunk >= 1024 && kChunk16 >= 1024 && kChunk24 >= 1024 && kChunk26 >= 1024 && kChunk26 <= kChunk, "a full chunk is one whole DMA + a partial one");
static_assert(level_bytes(0) % 64 == 0 && level_bytes(1) % 64 == 0 && level_bytes(2) % 64 == 0 && (kGroupLv * level_bytes(3)) % 64 == 0, "items start on 64-byte units (PlanItem.src64; a kStoreZ26 strip starts on a whole chunk)");

// src (wave-uniform global address) -> LDS bytes [slotA, ...) and [slotB, ...) (wave-uniform LDS byte addresses) through the
// per-lane byte offsets voffA / voffB.  EXEC is all ones (whole waves run this code) and stays so; M0 is written in the
// statement that reads it.  s_nop 4: a VMEM instruction must not read an SGPR a VALU instruction (v_readfirstlane) wrote
// in the five preceding cycles, and the compiler's hazard recogniser does not look inside an asm statement.
__device__ __forceinline__ void dma_item(const uint8_t *src, uint32_t slotA, uint32_t slotB, uint32_t voffA, uint32_t voffB) {
    if (kExpNoLoad) return;  // ablation build only (experiment.h): the ring keeps whatever LDS holds
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2" SUCRE_DMA_POLICY "\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %2" SUCRE_DMA_POLICY "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voffA), "s"(src), "s"(slotA), "s"(slotB), "v"(voffB)
        : "memory");
}

// Waits until at most n vector-memory instructions of this wave are outstanding.
__device__ __forceinline__ void wait_vm(uint32_t n) {
#define SUCRE_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        SUCRE_W(0) SUCRE_W(1) SUCRE_W(2) SUCRE_W(3) SUCRE_W(4) SUCRE_W(5) SUCRE_W(6) SUCRE_W(7) SUCRE_W(8) SUCRE_W(9)
        SUCRE_W(10) SUCRE_W(11) SUCRE_W(12) SUCRE_W(13) SUCRE_W(14) SUCRE_W(15) SUCRE_W(16) SUCRE_W(17) SUCRE_W(18)
        SUCRE_W(19) SUCRE_W(20) SUCRE_W(21)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // never weaker than asked
    }
#undef SUCRE_W
}
static_assert(2 * kAhead + 9 <= 63, "s_waitcnt vmcnt takes six bits");

// ---------------------------------------------------------------------------------------------------------------
// The plan: every wave's item stream, written once per image (launch_plan, called by the compaction) so that the
// per-iteration kernels do no bookkeeping.  (History: the first strip kernel walked the strips itself with two
// cursors; 136 scalar instructions per item on the CU's one scalar unit put a 74 us floor under the launch,
// measured with both the loads and the arithmetic compiled out.  Rounds 2-3 listed self-describing items -- kind, wait
// count, end flag -- that one loop dispatched on: ~85 scalar instructions and 13-26 register copies per item, because
// every item walked a compare-and-branch chain to its kind and to its s_waitcnt immediate and the accumulators changed
// registers between the branches.  Round 4: the stream has a fixed grammar per strip, so the consumer is straight-line
// code per strip with the full chunks in two tight loops, and every wait is an immediate known from the position.)
//
// A wave's stream, strip after strip (StripEntry says how many): [J plane 768 B][full chunks of four levels, first the
// ones without an empty slot, then the ones that need the z > 0 test][the short last chunk, r < 4 levels][J-parameter
// mode: the Adam moments 1536 B].  After the last strip's items come kAhead items that are issued and never consumed (they
// re-read the wave's first J plane), so that EVERY consumed item has exactly kAhead items issued behind it: the wait
// of an item is vmcnt(2 kAhead), plus the strip-end stores for the first kAhead items of every strip but the first.
// ---------------------------------------------------------------------------------------------------------------
// kMode 0: J plane, chunks, moments (J-parameter iteration, 9 stores at a strip's end); 1: J plane, chunks
// (closed-form iteration and update_J, 3 stores).  One thread per (fit wave, strip of that wave): it finds where its
// strip's items start in the wave's list from the level counts of the wave's earlier strips (at most a handful) and
// writes them.  (The first version replayed the issue order serially, one thread per wave over all its strips: 90 + 60 us
// per image for the two modes.)
__device__ __forceinline__ uint32_t strip_items(uint32_t levels, int mode) { return 1u + ((levels + 3u) >> 2) + (mode == 0 ? 1u : 0u); }

__global__ __launch_bounds__(256) void plan_kernel(const StripMeta *__restrict__ meta, int n_strips, const uint32_t *__restrict__ store_fmt, uint32_t W0, uint32_t W1,
                                                   uint32_t Kmax, uint32_t stride0, uint32_t stride1, uint32_t kmax0, uint32_t kmax1,
                                                   PlanItem *__restrict__ plan0, PlanItem *__restrict__ plan1,
                                                   StripEntry *__restrict__ strips0, StripEntry *__restrict__ strips1,
                                                   uint32_t *__restrict__ count0, uint32_t *__restrict__ count1, uint64_t comp_off,
                                                   uint64_t state_off) {
    const int mode = blockIdx.y;   // both plans in one launch
    const int fmt = (int)store_fmt[0];   // kStoreF32 / kStoreU16 / kStoreZ24 / kStoreZ26: what the compaction has just written (decide_store_format)
    const uint32_t W = mode ? W1 : W0, stride = mode ? stride1 : stride0, kmax = mode ? kmax1 : kmax0;
    PlanItem *plan = mode ? plan1 : plan0;
    StripEntry *strips = mode ? strips1 : strips0;
    uint32_t *count = mode ? count1 : count0;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    const uint32_t wid = idx / Kmax, k = idx - wid * Kmax;
    if (wid >= W) return;
    // the wave's strips up to its k-th (layout.h, deal_walk): where that one's items start in the wave's list, and how many strips
    // the wave has in all
    const DealShares sh = deal_shares(mode, W / 4u);
    uint32_t start = 0u, strip = 0u, strip0 = 0u;
    const uint32_t K = deal_walk(wid, W, (uint32_t)n_strips, sh, [&](uint32_t s) { return meta[s].levels; }, [&](uint32_t kk, uint32_t s) {
        if (kk == 0u) strip0 = s;
        if (kk < k) start += strip_items(meta[s].levels, mode);
        if (kk == k) strip = s;
    });
    if (k == 0u) count[wid] = K;
    if (k >= K) return;
    const StripMeta m = meta[strip];
    const uint32_t nfull = m.levels >> 2, r = m.levels & 3u;   // full chunks, levels of the short last one
    const uint32_t nu = min(nfull, m.full >> 2);               // chunks wholly below the strip's smallest pixel count
    const uint32_t lb = (uint32_t)level_bytes(fmt);
    const uint64_t st = state_off + (uint64_t)strip * (kStateFloats * 4);
    PlanItem *out = plan + (size_t)wid * stride + start;
    strips[(size_t)wid * kmax + k] = StripEntry{strip, strip_counts(nu, nfull - nu, r)};
    *out++ = PlanItem{(uint32_t)(st >> 6), item_shape(3u * kStripPx * 4u, false)};
    for (uint32_t g = 0; g < nfull; ++g)
        *out++ = PlanItem{(uint32_t)((comp_off + (m.lvoff + (uint64_t)g * kGroupLv) * lb) >> 6), item_shape((uint32_t)chunk_bytes(fmt), true)};
    if (r) *out++ = PlanItem{(uint32_t)((comp_off + (m.lvoff + (uint64_t)nfull * kGroupLv) * lb) >> 6), item_shape(r * lb, false)};
    if (mode == 0) *out++ = PlanItem{(uint32_t)((st + 3 * kStripPx * 4) >> 6), item_shape(6u * kStripPx * 4u, false)};
    if (k + 1u == K) {   // the kAhead trailing items nobody consumes (+ one spare the descriptor prefetch may touch)
        const uint64_t st0 = state_off + (uint64_t)strip0 * (kStateFloats * 4);
        for (int j = 0; j < kAhead + 1; ++j) *out++ = PlanItem{(uint32_t)(st0 >> 6), item_shape(3u * kStripPx * 4u, false) | kShapeTrail};
    }
}

// Plan items are read through the constant address space: for it the compiler emits scalar loads (s_load) whenever the
// address is wave-uniform and tracks their completion itself.  A vector load here would count in vmcnt and be waited
// for with vmcnt(0), draining the ring.  (A hand-issued s_load in inline asm with the wait in a second statement is
// NOT safe: the compiler may copy the destination registers between the two -- e.g. at the unrolled loop's back
// edge -- before the data has landed; seen as strips ended twice or never once two processes shared the GPU.)
typedef uint32_t ItemRegs __attribute__((ext_vector_type(2)));   // PlanItem: src64, shape / StripEntry: strip, counts
typedef const __attribute__((address_space(4))) ItemRegs *ConstItems;

// A wave-uniform pointer that the compiler may hold in vector registers -> scalar registers.
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<T *>(((uint64_t)hi << 32) | lo);
}

template <class T>
__device__ __forceinline__ ConstItems const_items(const T *p) {
    static_assert(sizeof(T) == 8, "two dwords per entry");
    return (ConstItems)(reinterpret_cast<uintptr_t>(uniform_ptr(p)));
}

// What a kernel instantiated for float32 ranges finds in the store: the float32 words, or their 24- or 26-bit codes
// (layout.h, kStoreZ24 / kStoreZ26; chosen on the device by the compaction): bits(z) = code + zoff, code 0 = empty slot.
// fullB: lanes of the second DMA of a full chunk of the store as it is (full_chunk_b).
struct RangeCodes { bool z24, z26; uint32_t zoff, fullB; };

template <int kFmt>
__device__ __forceinline__ RangeCodes range_codes_of(uint32_t f0, uint32_t f1) {   // f0, f1: the store's format word and code offset, wave-uniform
    RangeCodes rc;
    rc.z24 = kFmt == 0 && f0 == (uint32_t)kStoreZ24;
    rc.z26 = kFmt == 0 && f0 == (uint32_t)kStoreZ26;
    rc.zoff = f1;
    rc.fullB = rc.z24 ? full_chunk_b(kStoreZ24) : rc.z26 ? full_chunk_b(kStoreZ26) : full_chunk_b(kFmt);
    return rc;
}

template <int kFmt>
__device__ __forceinline__ RangeCodes range_codes(const uint32_t *__restrict__ store_fmt) {
    return range_codes_of<kFmt>(__builtin_amdgcn_readfirstlane(store_fmt[0]), __builtin_amdgcn_readfirstlane(store_fmt[1]));
}

// Is the store one the instantiation can read?  (kFmt 0: float32 ranges, as words or as range codes; 1: uint16 millimetres.)
__host__ __device__ constexpr bool format_readable(int kFmt, uint32_t f) {
    return kFmt == kStoreU16 ? f == (uint32_t)kStoreU16 : (f == (uint32_t)kStoreF32 || f == (uint32_t)kStoreZ24 || f == (uint32_t)kStoreZ26);
}
template <int kFmt>
__device__ __forceinline__ bool store_matches(const uint32_t *__restrict__ store_fmt) {
    return format_readable(kFmt, __builtin_amdgcn_readfirstlane(store_fmt[0]));
}

// A full chunk in a ring slot -> this lane's four ranges and three colour dwords.  masked: the chunk may hold empty slots.
template <int kFmt>
__device__ __forceinline__ void read_chunk(const uint8_t *sp, int lane, float (&zz)[kGroupLv], uint32_t (&cc)[3], const RangeCodes &rc, bool masked) {
    if (kFmt == 0 && rc.z26) {   // wave-uniform: a kStoreZ24 chunk + one byte per lane with bits 24-25 of its four codes
        const uint2 a = *reinterpret_cast<const uint2 *>(sp + lane * 24), b = *reinterpret_cast<const uint2 *>(sp + lane * 24 + 8),
                    c = *reinterpret_cast<const uint2 *>(sp + lane * 24 + 16);
        const uint32_t h = sp[kChunk24 + lane];
        const uint32_t w[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
        for (int j = 0; j < kGroupLv; ++j) {
            uint32_t bits;
            asm("v_mad_u32_u24 %0, %1, 1, %2" : "=v"(bits) : "v"(w[j]), "s"(rc.zoff));   // low 24 bits of the code + offset
            bits += ((h >> (2 * j)) & 3u) << 24;                                            // v_bfe_u32 + v_lshl_add_u32
            zz[j] = __uint_as_float(bits);
            if (masked) zz[j] = bits != rc.zoff ? zz[j] : 0.0f;   // code 0: an empty slot reads as range 0, like in the float32 store
        }
        cc[0] = __builtin_amdgcn_perm(a.y, a.x, 0x0c0c0703u) | __builtin_amdgcn_perm(b.y, b.x, 0x07030c0cu);
        cc[1] = c.x; cc[2] = c.y;
        return;
    }
    if (kFmt == 0 && rc.z24) {   // wave-uniform: the lane's 24 bytes side by side -- {code | red << 24} x 4, G word, B word
        const uint2 a = *reinterpret_cast<const uint2 *>(sp + lane * 24), b = *reinterpret_cast<const uint2 *>(sp + lane * 24 + 8),
                    c = *reinterpret_cast<const uint2 *>(sp + lane * 24 + 16);
        const uint32_t w[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
        for (int j = 0; j < kGroupLv; ++j) {
            // bits(z) = code + offset in one instruction: v_mad_u32_u24 multiplies the LOW 24 BITS of its operands
            uint32_t bits;
            asm("v_mad_u32_u24 %0, %1, 1, %2" : "=v"(bits) : "v"(w[j]), "s"(rc.zoff));
            zz[j] = __uint_as_float(bits);
            if (masked) zz[j] = (w[j] & 0xffffffu) ? zz[j] : 0.0f;   // an empty slot reads as range 0, like in the float32 store
        }
        // the red bytes sit in byte 3 of the four dwords: gathered into one word so that the arithmetic below is format-blind
        cc[0] = __builtin_amdgcn_perm(a.y, a.x, 0x0c0c0703u) | __builtin_amdgcn_perm(b.y, b.x, 0x07030c0cu);
        cc[1] = c.x; cc[2] = c.y;
        return;
    }
    if (kFmt == 0) {
        const float4 z4 = *reinterpret_cast<const float4 *>(sp + lane * 16);
        zz[0] = z4.x; zz[1] = z4.y; zz[2] = z4.z; zz[3] = z4.w;
    } else {  // uint16 millimetres -> metres, the same float32 product the CPU restatement forms
        const uint2 q = *reinterpret_cast<const uint2 *>(sp + lane * 8);
        zz[0] = (float)(q.x & 0xffffu) * kMPerMm; zz[1] = (float)(q.x >> 16) * kMPerMm;
        zz[2] = (float)(q.y & 0xffffu) * kMPerMm; zz[3] = (float)(q.y >> 16) * kMPerMm;
    }
    const uint32_t *cp = reinterpret_cast<const uint32_t *>(sp + (kFmt ? 2 : 4) * kStripPx * kGroupLv) + 3 * lane;
    cc[0] = cp[0]; cc[1] = cp[1]; cc[2] = cp[2];   // R, G, B words of the lane's four levels, side by side (layout.h)
}

// Level j of a short chunk (r < 4 levels, rows of r) in a ring slot.
template <int kFmt>
__device__ __forceinline__ void read_level(const uint8_t *sp, int lane, uint32_t r, uint32_t j, float &z, uint32_t (&k)[3], const RangeCodes &rc) {
    const bool z24 = kFmt == 0 && (rc.z24 || rc.z26);   // (range codes: three bytes per level, then the colours)
    if (z24) {
        const uint8_t *zp = sp + (lane * r + j) * 3u;
        uint32_t code = (uint32_t)zp[0] | ((uint32_t)zp[1] << 8) | ((uint32_t)zp[2] << 16);
        if (rc.z26) {   // bits 24-25: two-bit field lane r + j behind the colour planes
            const uint32_t f = (uint32_t)lane * r + j;
            code |= ((reinterpret_cast<const uint32_t *>(sp + 6u * kStripPx * r)[f >> 4] >> (2u * (f & 15u))) & 3u) << 24;
        }
        z = code ? __uint_as_float(code + rc.zoff) : 0.0f;
    } else if (kFmt == 0) z = reinterpret_cast<const float *>(sp)[lane * r + j];
    else z = (float)reinterpret_cast<const uint16_t *>(sp)[lane * r + j] * kMPerMm;
    const uint8_t *cb = sp + (z24 ? 3 : kFmt ? 2 : 4) * kStripPx * r;
#pragma unroll
    for (int c = 0; c < 3; ++c) k[c] = cb[c * kStripPx * r + lane * r + j];
}

// The streaming skeleton shared by the fit kernels: the wave's strips in plan order, kAhead items in flight at any time.
//   on_J(slot pointer)                      a strip begins: its J plane has landed
//   on_chunk(slot pointer, masked)          a full chunk
//   on_tail(slot pointer, r)                the strip's short last chunk
//   on_end(strip, slot pointer or nullptr)  the strip is complete (kMoments: its moments have landed); must issue exactly
//                                           kStores vector stores per lane (9 with the moments, 3 without)
// One STEP per item: wait for the LDS reads of the item whose slot is about to be overwritten, issue item i + kAhead into
// it, fetch the descriptor of item i + kAhead + 1 (scalar load, used one step later), wait for item i.  The wait is an
// immediate: vmcnt(2 kAhead), or vmcnt(2 kAhead + kStores) for the first kAhead items behind a strip's stores.
// One item -> ring slot at byte offset slot_off of the wave's ring.
template <int kFmt>
__device__ __forceinline__ void issue_item(const ItemRegs it, uint32_t ring0, uint32_t slot_off, const uint8_t *__restrict__ ws, int lane, uint32_t fullB) {
    // per-lane byte offsets of the two DMAs of a full chunk (constants of the launch; fullB = RangeCodes.fullB)
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t voffB_full = (64u + min((uint32_t)lane, fullB - 1u)) * 16u;
    const uint8_t *src = ws + ((uint64_t)it.x << 6);
    const uint32_t slot = ring0 + slot_off;
    if (it.y & kShapeFull) {
        dma_item(src, slot, slot + 1024u, lane16, voffB_full);
    } else {
        const uint32_t a_last = it.y & 127u, b_last = (it.y >> 8) & 127u, b_base = (it.y >> 16) & 127u;
        dma_item(src, slot, slot + ((a_last + 1u) << 4), min((uint32_t)lane, a_last) << 4, (b_base + min((uint32_t)lane, b_last)) << 4);
    }
}

// A wave's streams of consecutive images as ONE stream (batch launches).  A stream on its own ends with kAhead trailing copies
// nobody reads and a drain (its last strip's stores and those copies: a memory round trip with nothing in flight), and the next
// one starts with another (its first items).  Chained, the trailing items' places are taken by the NEXT image's first items,
// the ring position and the count of items issued ahead of the last stores carry over, and nothing is waited for between two
// images: the wave goes from one image's last strip to the next one's first like from strip to strip.
//   head[0 .. kAhead-1], head[kAhead], se0   THIS stream's first items, the one behind them and its first strip entry (asked for
//                                            one image ahead: no scalar round trip at the start)
//   next_on, next_head, next_ws, next_fullB  the stream that follows, if this wave has one (else: trailing copies + drain)
//   cs, behind                               in: where this stream's first item sits (its first kAhead items are in flight);
//                                            out: the same for the next stream -- 0, 0 after a drain
struct NoChain { static constexpr bool kOn = false; };
struct StreamChain {
    static constexpr bool kOn = true;
    ItemRegs head[kAhead + 1], se0;
    bool next_on;
    ItemRegs next_head[kAhead];
    const uint8_t *next_ws;
    uint32_t next_fullB;
    uint32_t cs, behind;
};

template <int kFmt, int kStores, bool kMoments, bool kPrimed = false, class Chain = NoChain, class OnJ, class OnChunk, class OnTail, class OnEnd>
__device__ __forceinline__ void stream_strips(FitLds &lds, const PlanItem *__restrict__ plan, const StripEntry *__restrict__ strip_list,
                                              uint32_t K, const uint8_t *__restrict__ ws, int wave, int lane, uint32_t fullB, OnJ on_J,
                                              OnChunk on_chunk, OnTail on_tail, OnEnd on_end, Chain *chain = nullptr) {
    if (K == 0) return;
    constexpr bool kChain = Chain::kOn;
    static_assert(!kChain || kPrimed, "a chained stream's first items are in flight when it starts");
    const uint32_t ring0 = lds_addr(&lds.ring[wave][0][0]);
    const uint8_t *ringp = &lds.ring[wave][0][0];
    const ConstItems items = const_items(plan);
    const ConstItems strips = const_items(strip_list);
    uint32_t trail = 0u;               // (chained) how many of the next stream's first items have been issued
    auto issue = [&](const ItemRegs it, uint32_t slot_off) {
        if constexpr (kChain) {
            if (chain->next_on && (it.y & kShapeTrail)) {   // wave-uniform
                ItemRegs f = chain->next_head[0];
#pragma unroll
                for (int q = 1; q < kAhead; ++q) f = trail == (uint32_t)q ? chain->next_head[q] : f;
                ++trail;
                issue_item<kFmt>(f, ring0, slot_off, chain->next_ws, lane, chain->next_fullB);
                return;
            }
        }
        issue_item<kFmt>(it, ring0, slot_off, ws, lane, fullB);
    };
    // the first kAhead items (a wave with a strip has at least 1 + kAhead + 1 entries: its items, the trailing ones, the spare)
    if (!kPrimed) {
#pragma unroll
        for (int q = 0; q < kAhead; ++q) issue(items[q], (uint32_t)(q * kSlot));
    }
    ItemRegs nxt;                      // item i + kAhead, issued at the top of step i
    if constexpr (kChain) nxt = chain->head[kAhead]; else nxt = items[kAhead];
    uint32_t i = 0;                    // the item being consumed
    uint32_t cs = 0u;                  // byte offset of its slot in the wave's ring; item i + kAhead goes to the slot before it
    uint32_t behind = 0u;              // how many of the next items were issued before the previous strip's stores
    if constexpr (kChain) { cs = chain->cs; behind = chain->behind; }
    const uint32_t gen = blockIdx.x >> 8;   // (experiment) which of a CU's resident workgroups this one is, oldest first
    constexpr int kPrio = kChain ? 0 : kExpPrio;   // (the priority experiments were on one image's launch; a batch kernel has no register to spare)
    auto set_prio = [&](uint32_t p) {
        switch (p & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    };
    auto step = [&]() -> const uint8_t * {
        // The slot that takes item i + kAhead held item i - 1.  Its LDS reads must have RETURNED before the DMA may overwrite
        // it: a chunk's reads have (the arithmetic consumed them), but a J plane is read into registers that are first used
        // one step later -- without this wait the DMA raced those reads (rare, and only under load: wrong J, first channel
        // first, seen when other work shared the GPU).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t is = cs == 0u ? (uint32_t)((kRing - 1) * kSlot) : cs - (uint32_t)kSlot;   // slot of item i - 1 = of item i + kAhead
        issue(nxt, is);
        nxt = items[i + (uint32_t)kAhead + 1u];
        if (behind) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kAhead + kStores) : "memory");
            --behind;
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kAhead) : "memory");
        }
        const uint8_t *sp = ringp + cs;
        cs = cs == (uint32_t)((kRing - 1) * kSlot) ? 0u : cs + (uint32_t)kSlot;
        ++i;
        if (kPrio == 3) set_prio(gen + i);
        return sp;
    };
    ItemRegs se;
    if constexpr (kChain) se = chain->se0; else se = strips[0];
    if (kPrio == 2) set_prio(gen >= 3u ? 3u : gen);
    for (uint32_t k = 0; k < K; ++k) {
        if (kPrio == 1) set_prio(gen + k);
        const uint32_t strip = se.x, counts = se.y;
        if (k + 1u < K) se = strips[k + 1u];
        const uint32_t nu = counts_unmasked(counts), nm = counts_masked(counts), r = counts_tail(counts);
        on_J(step());
        for (uint32_t g = 0; g < nu; ++g) on_chunk(step(), false);
        for (uint32_t g = 0; g < nm; ++g) on_chunk(step(), true);
        if (r) on_tail(step(), r);
        if (kMoments) on_end(strip, step());
        else on_end(strip, (const uint8_t *)nullptr);
        behind = (uint32_t)kAhead;
    }
    if constexpr (kChain) {
        if (chain->next_on) {   // the next stream's first items are in flight in the slots behind this one's last
            chain->cs = cs;
            chain->behind = (uint32_t)kAhead;
            return;
        }
        chain->cs = 0u;
        chain->behind = 0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of ours is in flight past this point (the trailing items included)
}

// Two-level, fixed-order float64 reduction of the per-workgroup partials (layout [kNumSums][n_blocks]):
//   gpart[q][g] = sum of the 32 workgroups of group g  (one load per lane + fixed-shape shuffle tree)
//   sums[q]     = sum over the groups                  (4 loads per lane + the same tree)
// The same two functions run in the fused tail (group-last / global-last workgroup) and in the split-path
// kernel, so both paths produce the same bits.  All hand-off data move with agent-scope (sc1) accesses.
__device__ __forceinline__ double wave_sum_fixed(double x) {  // fixed-shape tree: same bits on every run (lane 0 holds the sum)
    if (kExpShflSums) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        return x;
    }
    return wave_sum_lane0(x);   // the same additions, register to register (handoff.h)
}

// 256 threads: wave w reduces quantities q = w, w+4, w+8; lane l holds workgroup 32 g + l (lanes >= 32 hold 0).
__device__ __forceinline__ void reduce_group(const float *partials, int n_blocks, int g, double *gpart, int n_groups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = g * kGroup + lane;
    for (int q = wave; q < kNumSums; q += 4) {
        double x = 0.0;
        if (lane < kGroup && b < n_blocks)
            x = (double)__hip_atomic_load(partials + (size_t)q * n_blocks + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
            __hip_atomic_store(sums + q, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global copy: read by the host all-reduce
            stot[q] = x;  // LDS copy: read by water_step of the same workgroup
        }
    }
    __syncthreads();
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

// The lanes' ten sums -> lane 0 holds the wave's: x_l + x_(l+32), then + (l+16), ... + (l+1) -- the additions of the
// __shfl_down tree that lane 0 depends on, in its association (the other lanes end with values nobody reads).
__device__ __forceinline__ void wave_sums(float (&s)[kNumSums]) {
    if (kExpShflSums || kExpPlainWaveSums) {
#pragma unroll
        for (int q = 0; q < kNumSums; ++q) {
            if (kExpShflSums) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off, 64);
            } else {
                s[q] = wave_sum_lane0(s[q]);
            }
        }
        return;
    }
    // The same ten trees, several to a register (round 6: 38 instructions instead of 80 -- a small image's stream in a batch launch is
    // a few hundred).  v_permlane32_swap a, b leaves [a.lo | b.lo] and [a.hi | b.hi]: their sum holds x_l + x_(l+32) of quantity q in
    // its lower half and of quantity q + 5 in its upper one; v_permlane16_swap of two such registers and one addition leaves the
    // 16-lane sums of FOUR quantities in the four rows of one register; the row shifts then serve all four at once.  Every addition
    // is the one lane 0's __shfl_down tree makes, operand for operand; the totals come back by v_readlane_b32 (wave-uniform).
    static_assert(kNumSums == 10, "five pairs");
    float h[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s[q]), __float_as_uint(s[q + 5]), false, false);
        h[q] = __uint_as_float(r[0]) + __uint_as_float(r[1]);                     // [q | q + 5]
    }
    float R[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const float a = h[2 * p], b = h[p < 2 ? 2 * p + 1 : 4];
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        R[p] = __uint_as_float(r[0]) + __uint_as_float(r[1]);                     // rows: [2p, 2p + 1, 2p + 5, 2p + 6]  (p = 2: [4, 4, 9, 9])
        R[p] += lane_down<8>(R[p]);
        R[p] += lane_down<4>(R[p]);
        R[p] += lane_down<2>(R[p]);
        R[p] += lane_down<1>(R[p]);
    }
    auto at = [](float v, int lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); };
    s[0] = at(R[0], 0); s[1] = at(R[0], 16); s[5] = at(R[0], 32); s[6] = at(R[0], 48);
    s[2] = at(R[1], 0); s[3] = at(R[1], 16); s[7] = at(R[1], 32); s[8] = at(R[1], 48);
    s[4] = at(R[2], 0); s[9] = at(R[2], 32);
}

template <bool kFused, bool kStep>
__device__ __forceinline__ void finish_from_wave_sums(FitLds &lds, float *partials, const AdamCoef &co, unsigned *ticket,
                                                      double *gpart, int n_groups, double *sums, float *pstate,
                                                      const uint64_t *__restrict__ n_obs_total, double *trace_row);

// End of a fit launch: the lanes' ten sums -> one float32 partial per workgroup -> (fused form) two-level
// last-arriver reduction in float64 and the Adam step on B, beta, gamma by the workgroup that arrives last.
template <bool kFused, bool kStep = kFused>
__device__ __forceinline__ void finish_launch(FitLds &lds, float (&s)[kNumSums], float *partials, const AdamCoef &co,
                                              unsigned *ticket, double *gpart, int n_groups, double *sums,
                                              float *pstate, const uint64_t *__restrict__ n_obs_total,
                                              double *trace_row) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // ten workgroup sums: wave shuffle tree, then the four waves in fixed order
    wave_sums(s);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kNumSums; ++q) lds.wsum[wave][q] = s[q];
    }
    finish_from_wave_sums<kFused, kStep>(lds, partials, co, ticket, gpart, n_groups, sums, pstate, n_obs_total, trace_row);
}

// ... from the four waves' sums in lds.wsum on.
template <bool kFused, bool kStep>
__device__ __forceinline__ void finish_from_wave_sums(FitLds &lds, float *partials, const AdamCoef &co, unsigned *ticket,
                                                      double *gpart, int n_groups, double *sums, float *pstate,
                                                      const uint64_t *__restrict__ n_obs_total, double *trace_row) {
    const int n_blocks = gridDim.x;
    const int t = threadIdx.x;
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
                if (kStep && t < 64) water_step(lds.stot, pstate, n_obs_total, co, trace_row);
            }
        }
    }
}

// One wave's share of a J-parameter iteration on one image (sucre.py:142-148 with J among the parameters): streams
// the wave's items, steps J of every strip, and keeps adding the lane's shares of the global sums to acc / sBeta.
template <int kFmt, bool kPrimed = false, class Chain = NoChain>
__device__ __forceinline__ void grad_pass(FitLds &lds, const PlanItem *__restrict__ plan, const StripEntry *__restrict__ strips, uint32_t n_strips_wave,
                                          const uint8_t *__restrict__ ws, float *__restrict__ state, int wave, int lane,
                                          const Water &w, float gscale, const AdamCoef &co, Acc &acc, float (&sBeta)[3], const RangeCodes &rc,
                                          Chain *chain = nullptr) {
    float J[3] = {0.f, 0.f, 0.f};
    stream_strips<kFmt, 9, true, kPrimed, Chain>(
        lds, plan, strips, n_strips_wave, ws, wave, lane, rc.fullB,
        [&](const uint8_t *sp) {  // J plane
            const float *f = reinterpret_cast<const float *>(sp);
#pragma unroll
            for (int c = 0; c < 3; ++c) { J[c] = f[c * kStripPx + lane]; acc.pa[c] = 0.f; acc.pb[c] = 0.f; }
        },
        [&](const uint8_t *sp, bool masked) {  // chunk of four levels
            float zz[kGroupLv];
            uint32_t cc[3];
            read_chunk<kFmt>(sp, lane, zz, cc, rc, masked);
            if (masked) accumulate_chunk<true>(zz, cc, w, J, acc);
            else accumulate_chunk<false>(zz, cc, w, J, acc);
        },
        [&](const uint8_t *sp, uint32_t r) {  // the strip's short last chunk
            for (uint32_t j = 0; j < r; ++j) {
                float z;
                uint32_t k[3];
                read_level<kFmt>(sp, lane, r, j, z, k, rc);
                accumulate_level(z, k, w, J, acc);
            }
        },
        [&](uint32_t strip, const uint8_t *sp) {  // moments landed: torch.optim.Adam on this pixel's J
            const float *f = reinterpret_cast<const float *>(sp);
            // (kExpStoreLocal, timing only: every wave writes ONE strip's place over and over -- the stores are issued and
            // acknowledged like the product's, but the lines stay in L2)
            float *st = state + (size_t)(kExpStoreLocal ? blockIdx.x * 4u + (uint32_t)wave : strip) * kStateFloats;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float Jc = J[c], m = f[c * kStripPx + lane], v = f[(3 + c) * kStripPx + lane];
                // unobserved pixels: zero sum, and J (possibly NaN) must not leak into the beta gradient
                sBeta[c] += (acc.pb[c] == 0.0f) ? 0.0f : Jc * acc.pb[c];
                adam_update_J(Jc, m, v, gscale * acc.pa[c], co);
                if (kExpNoStore) {   // ablation build only: the step is computed (the asm keeps it alive), nothing is written
                    asm volatile("" ::"v"(Jc), "v"(m), "v"(v));
                    continue;
                }
                if (kStoreNt == 1 || (kStoreNt == 2 && Chain::kOn)) {
                    __builtin_nontemporal_store(Jc, &st[c * kStripPx + lane]);
                    __builtin_nontemporal_store(m, &st[(3 + c) * kStripPx + lane]);
                    __builtin_nontemporal_store(v, &st[(6 + c) * kStripPx + lane]);
                    continue;
                }
                st[c * kStripPx + lane] = Jc;
                st[(3 + c) * kStripPx + lane] = m;
                st[(6 + c) * kStripPx + lane] = v;
            }
        },
        chain);
}

__device__ __forceinline__ void zero_acc(Acc &acc) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { acc.pa[c] = 0.f; acc.pb[c] = 0.f; acc.sB[c] = 0.f; acc.sGZ[c] = 0.f; }
    acc.cost = 0.f;
}

// experiment.h SUCRE_EXP_WAVE_TIMES (tools/exp/wave_times.py): when every wave of the last fit launch entered and left its
// strips, in ticks of the 100 MHz wall clock.
constexpr uint32_t kExpWaveSlots = 8192;
__device__ unsigned long long g_exp_wave_times[5][kExpWaveTimes ? kExpWaveSlots : 1];   // strips begun, strips done, kernel entered, kernel left (100 MHz ticks); shader cycles in the kernel
}  // namespace sucre
SUCRE_EXP_EXPORT int sucre_exp_wave_times(unsigned long long *out) {   // exported by the experiment build only
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sucre::g_exp_wave_times), sizeof(sucre::g_exp_wave_times));
}
namespace sucre {

// J-parameter iteration of one image: one launch.
template <bool kFused, int kFmt>
__global__ __launch_bounds__(256, kFitWaves) void fit_grad_kernel(const uint8_t *__restrict__ ws,
                                                       const PlanItem *__restrict__ plan, const StripEntry *__restrict__ plan_strips, const uint32_t *__restrict__ plan_count,
                                                       uint32_t plan_stride, uint32_t plan_kmax,
                                                       float *pstate, const uint64_t *__restrict__ n_obs_total,
                                                       float *__restrict__ state, float *partials, const AdamCoef co,
                                                       unsigned *ticket, double *gpart, int n_groups, double *sums,
                                                       double *trace_row, const uint32_t *__restrict__ obs_format) {
    __shared__ FitLds lds;  // 21.9 KB
    const unsigned long long exp_t_in = kExpWaveTimes ? wall_clock64() : 0ull, exp_c_in = kExpWaveTimes ? clock64() : 0ull;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const Water w = load_water(pstate);
    const float gscale = -2.0f * ((1.0f / 3.0f) / (float)(*n_obs_total));  // (loss / n_obs / 3).backward(), sucre.py:145
    // a store compacted in the other format is not read at all; the logged cost turns NaN instead
    const RangeCodes rc = range_codes<kFmt>(obs_format);
    const bool fmt_ok = store_matches<kFmt>(obs_format);
    const uint32_t wid = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t n_mine = fmt_ok ? plan_count[wid] : 0u;   // strips of this wave

    // The lane's shares of the global sums keep accumulating across its strips; the per-pixel sums restart with
    // every strip.
    Acc acc;
    zero_acc(acc);
    if (!fmt_ok) acc.cost = __builtin_nanf("");
    float sBeta[3] = {0.f, 0.f, 0.f};
    if (kExpWaveTimes && lane == 0 && wid < kExpWaveSlots) g_exp_wave_times[0][wid] = wall_clock64();
    grad_pass<kFmt>(lds, plan + (size_t)wid * plan_stride, plan_strips + (size_t)wid * plan_kmax, n_mine, ws, state, wave, lane, w, gscale, co, acc, sBeta, rc);
    if (kExpWaveTimes && lane == 0 && wid < kExpWaveSlots) g_exp_wave_times[1][wid] = wall_clock64();

    float s[kNumSums] = {acc.sB[0], acc.sB[1], acc.sB[2], acc.sGZ[0], acc.sGZ[1], acc.sGZ[2],
                         sBeta[0], sBeta[1], sBeta[2], acc.cost};
    finish_launch<kFused>(lds, s, partials, co, ticket, gpart, n_groups, sums, pstate, n_obs_total, trace_row);
    if (kExpWaveTimes && lane == 0 && wid < kExpWaveSlots) { g_exp_wave_times[2][wid] = exp_t_in; g_exp_wave_times[3][wid] = wall_clock64(); g_exp_wave_times[4][wid] = clock64() - exp_c_in; }
}

// One wave's share of a closed-form iteration on one image, one observation pass (see AccOne); kJOnly:
// SUCRe.update_J alone (sucre.py:66-77, 156): J = sum (I - b) a / sum a^2 from the current parameters, nothing else.
struct ClosedSums { float sB[3], sGZ[3], sBeta[3], cost; };

template <int kFmt, bool kJOnly, bool kPrimed = false, class Chain = NoChain>
__device__ __forceinline__ void closed_pass(FitLds &lds, const PlanItem *__restrict__ plan, const StripEntry *__restrict__ strips, uint32_t n_strips_wave,
                                            const uint8_t *__restrict__ ws, float *__restrict__ state, int wave, int lane,
                                            const Water &w, bool fmt_ok, ClosedSums &cs, const RangeCodes &rc, Chain *chain = nullptr) {
    AccOne acc;
    float Jp[3] = {0.f, 0.f, 0.f};
    stream_strips<kFmt, 3, false, kPrimed, Chain>(
        lds, plan, strips, n_strips_wave, ws, wave, lane, rc.fullB,
        [&](const uint8_t *sp) {  // previous J of this pixel
            const float *f = reinterpret_cast<const float *>(sp);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // a pixel that had no J so far (NaN: never observed, or a warm start without it) is measured from 0;
                // update_J measures from 0 always (J = N / D exactly as sucre.py:77 forms it)
                Jp[c] = kJOnly ? 0.0f : finite_or_zero(f[c * kStripPx + lane]);
            }
#pragma unroll
            for (int q = 0; q < 9; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc.q[q][c] = 0.f;
        },
        [&](const uint8_t *sp, bool masked) {
            float zz[kGroupLv];
            uint32_t cc[3];
            read_chunk<kFmt>(sp, lane, zz, cc, rc, masked);
            if (masked) accumulate_chunk<true>(zz, cc, w, Jp, acc);
            else accumulate_chunk<false>(zz, cc, w, Jp, acc);
        },
        [&](const uint8_t *sp, uint32_t r) {
            for (uint32_t j = 0; j < r; ++j) {
                float z;
                uint32_t k[3];
                read_level<kFmt>(sp, lane, r, j, z, k, rc);
                accumulate_level(z, k, w, Jp, acc);
            }
        },
        [&](uint32_t strip, const uint8_t *) {  // all levels seen: re-solve J, form the pixel's share of the sums
            float *st = state + (size_t)strip * kStateFloats;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float N = acc.q[0][c], D = acc.q[1][c];
                const float dJ = N / D;                          // 0/0 = NaN where nothing was observed (sucre.py:77)
                const float Jc = fmt_ok ? Jp[c] + dJ : __builtin_nanf("");   // = sum y a / sum a^2
                if (kStoreNt == 1 || (kStoreNt == 2 && Chain::kOn)) __builtin_nontemporal_store(Jc, &st[c * kStripPx + lane]);
                else st[c * kStripPx + lane] = Jc;
                if (!kJOnly && D != 0.0f) {
                    cs.sB[c] += __builtin_fmaf(-dJ, acc.q[3][c], acc.q[2][c]);
                    cs.sBeta[c] += Jc * __builtin_fmaf(-dJ, acc.q[5][c], acc.q[4][c]);
                    cs.sGZ[c] += __builtin_fmaf(-dJ, acc.q[7][c], acc.q[6][c]);
                    cs.cost += __builtin_fmaf(-dJ, N, acc.q[8][c]);
                } else if (!kJOnly && N != 0.0f) {
                    // An OBSERVED pixel whose every a^2 underflowed (ranges of hundreds of metres): the reference's J is
                    // +-inf there (sucre.py:77), its residuals with it, the cost of the iteration is inf and the channel's
                    // three parameters are NaN from the next step on (seen with the reference itself on a scene of
                    // tools/parity_sweep.py).  The sums that factor J out would skip the pixel: hand them the infinity.
                    cs.sB[c] += dJ;
                    cs.sBeta[c] += dJ;
                    cs.sGZ[c] += dJ;
                    cs.cost += __builtin_inff();
                }
            }
        },
        chain);
}

template <bool kFused, int kFmt, bool kJOnly>
__global__ __launch_bounds__(256, kClosedWaves) void fit_closed_kernel(const uint8_t *__restrict__ ws,
                                                         const PlanItem *__restrict__ plan, const StripEntry *__restrict__ plan_strips, const uint32_t *__restrict__ plan_count,
                                                       uint32_t plan_stride, uint32_t plan_kmax,
                                                         float *pstate, const uint64_t *__restrict__ n_obs_total,
                                                         float *__restrict__ state, float *partials, const AdamCoef co,
                                                         unsigned *ticket, double *gpart, int n_groups, double *sums,
                                                         double *trace_row, const uint32_t *__restrict__ obs_format) {
    __shared__ FitLds lds;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const Water w = load_water_uniform(pstate);   // scalar registers: the 27 accumulators need the vector ones
    const RangeCodes rc = range_codes<kFmt>(obs_format);
    const bool fmt_ok = store_matches<kFmt>(obs_format);
    // update_J on a store of the other format poisons J instead of misreading it: every strip is still visited (the
    // plan was written for the store's format, so the items themselves are sound)
    const uint32_t wid = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t n_mine = (fmt_ok || kJOnly) ? plan_count[wid] : 0u;   // strips of this wave
    ClosedSums cs = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, fmt_ok ? 0.f : __builtin_nanf("")};
    if (kExpWaveTimes && !kJOnly && lane == 0 && wid < kExpWaveSlots) g_exp_wave_times[0][wid] = wall_clock64();
    closed_pass<kFmt, kJOnly>(lds, plan + (size_t)wid * plan_stride, plan_strips + (size_t)wid * plan_kmax, n_mine, ws, state, wave, lane, w, fmt_ok, cs, rc);
    if (kExpWaveTimes && !kJOnly && lane == 0 && wid < kExpWaveSlots) g_exp_wave_times[1][wid] = wall_clock64();
    if (kJOnly) return;
    float s[kNumSums] = {cs.sB[0], cs.sB[1], cs.sB[2], cs.sGZ[0], cs.sGZ[1], cs.sGZ[2], cs.sBeta[0], cs.sBeta[1], cs.sBeta[2], cs.cost};
    finish_launch<kFused>(lds, s, partials, co, ticket, gpart, n_groups, sums, pstate, n_obs_total, trace_row);
}

struct Params9 { float v[9]; };

// The group / batch kernels carry a few registers more than fit_grad_kernel: when an experiment build gives that one six waves per
// SIMD (SUCRE_FIT_WAVES=6) they keep five (their launches then run their last sixth of workgroups in a second round: timing
// experiments on fit_grad_kernel only).
constexpr int kGroupFitWaves = kFitWaves > 5 ? 5 : kFitWaves;

// ---------------------------------------------------------------------------------------------------------------
// Shared water parameters (north-star extension): every image of this rank -- and, through one
// all-reduce of the ten sums, of every other rank -- steps B, beta, gamma together.  One iteration is ONE launch and
// one collective: the launch walks all the rank's images (every wave streams its item list of image 0, then of
// image 1, ...; the lanes' shares of the sums simply keep accumulating), leaves the rank's reduced sums in the group
// buffer for the host's all-reduce, and the Adam step on the nine parameters that those all-reduced sums call for
// is taken in the PROLOGUE of the next launch, by every workgroup for itself from the same inputs (workgroup 0
// records it).  The water state is double-buffered so that late workgroups never read a stepped state.
// ---------------------------------------------------------------------------------------------------------------
struct GroupImage {
    uint8_t *ws;
    uint64_t off_plan[2], off_strips[2], off_count[2], off_state, off_format, off_params;
    uint32_t stride[2], kmax[2], n_waves[2];
};

struct GroupHeader {
    float pstate[2][32];                 // [k & 1] = B, beta, gamma + moments after k steps (27 floats used)
    double sums[kSumsPad];               // this rank's sums of the last pass; all-reduced in place by the host
    unsigned ticket[(1 + (kFitGrid + kGroup - 1) / kGroup) * kTicketStride];
    double gpart[kNumSums * ((kFitGrid + kGroup - 1) / kGroup)];
    float partials[kNumSums * kFitGrid];
};

__host__ __device__ __forceinline__ GroupImage *group_images(GroupHeader *g) {
    return reinterpret_cast<GroupImage *>(reinterpret_cast<uint8_t *>(g) + align_up(sizeof(GroupHeader), 256));
}

// The step that the all-reduced sums call for (water_step's arithmetic), from state `in`; lanes 0..8 of wave 0 return
// the stepped (p, m, v); all 64 lanes must call it.
// Agent-scope (sc1) accesses for everything the group launches hand to each other or receive from the host between
// launches: the water state, written early in a launch by workgroup 0 and read by every workgroup of the next one,
// and the sums, written by the last arriver and by the host's all-reduce.  (Measured with two processes sharing the
// GPU: with plain loads some workgroups of a launch stepped from a stale copy -- non-deterministic trajectories.)
__device__ __forceinline__ float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void group_water_step(const double *__restrict__ sums, const float *__restrict__ in,
                                                 uint64_t n_obs_total, const AdamCoef &co, bool apply, float &p, float &m, float &v) {
    const int q = threadIdx.x & 63;
    p = m = v = 0.f;
    if (q < 9) {
        p = ld_agent(in + q); m = ld_agent(in + 9 + q); v = ld_agent(in + 18 + q);
        if (apply) {
            const float scale = (1.0f / 3.0f) / (float)n_obs_total;
            const int c = q % 3;
            double g;
            if (q < 3) g = -2.0 * (double)scale * ld_agent(sums + c);
            else if (q < 6) g = 2.0 * (double)scale * ld_agent(sums + 6 + c);
            else g = -2.0 * (double)scale * (double)ld_agent(in + c) * ld_agent(sums + 3 + c);
            adam_update(p, m, v, (float)g, co);
        }
    }
}

template <int kMode, int kFmt>
__global__ __launch_bounds__(256, kMode ? kClosedWaves : kGroupFitWaves) void group_iter_kernel(GroupHeader *g, int n_images, uint64_t n_obs_total,
                                                                                           const AdamCoef co_prev, const AdamCoef co, int apply,
                                                                                           int in, int out, double *trace_prev, int n_groups,
                                                                                           const GroupImage *__restrict__ images) {
    __shared__ FitLds lds;
    __shared__ float wpar[9];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (t < 64) {
        float p, m, v;
        group_water_step(g->sums, g->pstate[in], n_obs_total, co_prev, apply != 0, p, m, v);
        if (lane < 9) wpar[lane] = p;
        if (blockIdx.x == 0 && apply) {   // every workgroup computes the same step; workgroup 0 records it
            if (lane < 9) {
                st_agent(&g->pstate[out][lane], p); st_agent(&g->pstate[out][9 + lane], m); st_agent(&g->pstate[out][18 + lane], v);
                if (trace_prev) trace_prev[1 + lane] = (double)p;
            }
            if (lane == 9 && trace_prev) trace_prev[0] = ld_agent(&g->sums[9]);
        }
    }
    __syncthreads();
    const Water w = load_water_uniform(wpar);
    const float gscale = -2.0f * ((1.0f / 3.0f) / (float)n_obs_total);
    const uint32_t wid = blockIdx.x * 4u + (uint32_t)wave;
    Acc acc;
    zero_acc(acc);
    float sBeta[3] = {0.f, 0.f, 0.f};
    ClosedSums cs = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, 0.f};
    // The lanes' ten float32 sums run over ONE image, as in the per-image kernels: at every image boundary the wave adds
    // them up (the shuffle tree of the launch's end) and lane 0 carries the wave's total on in float64 (LDS, the wave's own
    // slots).  Carried in float32 across a rank's 64 images (BASELINE config 4) the cost lost 1.2e-5 of its value to rounding
    // (tests/test_gpu_config4.py; 5e-7 at four images).  A group of one image keeps its bits: (float)(double)x = x.
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kNumSums; ++q) lds.wtotal[wave][q] = 0.0;
    }
    bool poisoned = false;
    for (int i = 0; i < n_images; ++i) {
        const GroupImage im = images[i];
        uint8_t *ws = uniform_ptr(im.ws);   // between images nothing is in flight: ordinary loads are harmless here
        const uint32_t *store_fmt = reinterpret_cast<const uint32_t *>(ws + im.off_format);
        const RangeCodes rc = range_codes<kFmt>(store_fmt);
        if (!store_matches<kFmt>(store_fmt)) { poisoned = true; continue; }
        if (wid >= im.n_waves[kMode]) continue;
        const uint32_t n_mine = __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(ws + im.off_count[kMode])[wid]);
        const PlanItem *plan = uniform_ptr(reinterpret_cast<const PlanItem *>(ws + im.off_plan[kMode]) + (size_t)wid * im.stride[kMode]);
        const StripEntry *strips = uniform_ptr(reinterpret_cast<const StripEntry *>(ws + im.off_strips[kMode]) + (size_t)wid * im.kmax[kMode]);
        float *state = reinterpret_cast<float *>(ws + im.off_state);
        float s[kNumSums];
        if (kMode == 0) {
            grad_pass<kFmt>(lds, plan, strips, n_mine, ws, state, wave, lane, w, gscale, co, acc, sBeta, rc);
            const float q[kNumSums] = {acc.sB[0], acc.sB[1], acc.sB[2], acc.sGZ[0], acc.sGZ[1], acc.sGZ[2], sBeta[0], sBeta[1], sBeta[2], acc.cost};
#pragma unroll
            for (int j = 0; j < kNumSums; ++j) s[j] = q[j];
            zero_acc(acc);
            sBeta[0] = sBeta[1] = sBeta[2] = 0.f;
        } else {
            closed_pass<kFmt, false>(lds, plan, strips, n_mine, ws, state, wave, lane, w, true, cs, rc);
            const float q[kNumSums] = {cs.sB[0], cs.sB[1], cs.sB[2], cs.sGZ[0], cs.sGZ[1], cs.sGZ[2], cs.sBeta[0], cs.sBeta[1], cs.sBeta[2], cs.cost};
#pragma unroll
            for (int j = 0; j < kNumSums; ++j) s[j] = q[j];
            cs = ClosedSums{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, 0.f};
        }
        wave_sums(s);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < kNumSums; ++j) lds.wtotal[wave][j] += (double)s[j];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < kNumSums; ++j) lds.wsum[wave][j] = (float)lds.wtotal[wave][j];
        if (poisoned) lds.wsum[wave][kNumSums - 1] = __builtin_nanf("");   // an image compacted in the other format: the logged cost turns NaN
    }
    finish_from_wave_sums<true, false>(lds, g->partials, co, g->ticket, g->gpart, n_groups, g->sums, nullptr, nullptr, nullptr);
}

// The last pending step; the final parameters also go to every image's own workspace (Restoration.params()).
__global__ __launch_bounds__(64) void group_finish_kernel(GroupHeader *g, int n_images, uint64_t n_obs_total, const AdamCoef co_prev,
                                                          int apply, int in, int out, double *trace_prev) {
    float p, m, v;
    const int lane = threadIdx.x;
    group_water_step(g->sums, g->pstate[in], n_obs_total, co_prev, apply != 0, p, m, v);
    if (lane < 9) {
        st_agent(&g->pstate[out][lane], p); st_agent(&g->pstate[out][9 + lane], m); st_agent(&g->pstate[out][18 + lane], v);
        if (apply && trace_prev) trace_prev[1 + lane] = (double)p;
        const GroupImage *images = group_images(g);
        for (int i = 0; i < n_images; ++i) {
            float *ps = reinterpret_cast<float *>(images[i].ws + images[i].off_params);
            ps[lane] = p; ps[9 + lane] = m; ps[18 + lane] = v;
        }
    }
    if (lane == 9 && apply && trace_prev) trace_prev[0] = ld_agent(&g->sums[9]);
}

__global__ void group_set_image_kernel(GroupHeader *g, int i, const GroupImage im) { group_images(g)[i] = im; }

__global__ __launch_bounds__(64) void group_init_kernel(GroupHeader *g, const Params9 p0) {
    const int t = threadIdx.x;
    if (t < 32) { g->pstate[0][t] = t < 9 ? p0.v[t] : 0.f; g->pstate[1][t] = 0.f; }
    if (t < kSumsPad) g->sums[t] = 0.0;
    for (int i = t; i < (int)(sizeof(g->ticket) / sizeof(unsigned)); i += 64) g->ticket[i] = 0u;
}

// ---------------------------------------------------------------------------------------------------------------
// Independent images in ONE launch per iteration (the reference's per-image mode, sucre.py:243-261: every image has its own
// B, beta, gamma, J and its own Adam state; nothing is shared).  A launch per image and iteration is what a small image pays
// for most: at 640x480 x 5 views (BASELINE config 1) a launch moves 31 MB and lasts 17 us, 0.24 of the HBM peak, each wave with
// less than one strip to work on.  Here every wave walks its item list of image 0, then of image 1, ... by itself (no barrier
// between images; the next image's first items are in flight while the wave adds up this image's sums), the workgroup leaves
// its partial of EVERY image at the launch's end, and a second small launch reduces and steps all images at once: the launch
// gap, the ramp-up and the reduction tail are paid once per batch instead of once per image.  Every image sees exactly the
// operations of its own fit_grad_kernel / fit_closed_kernel launch in the same order (same plan, same grid, same reduction
// tree), so its results are the bits of fitting it alone (tests/test_gpu_batch.py).
// All images of a batch have one size (H, W: one grid); their view counts -- hence their workspace layouts -- may differ.
// ---------------------------------------------------------------------------------------------------------------
struct BatchOffsets {
    uint64_t plan, strips, count, params, n_obs_total, state, partials, ticket, gpart, sums, format;
    uint32_t stride, kmax;
    int n_groups;
};
struct BatchEntry { uint8_t *ws; double *trace; BatchOffsets o; };   // trace: the image's (T, 10) log, or NULL

// The launch's end, once for ALL its images (what finish_from_wave_sums does per launch): the waves' sums of every image wait
// in LDS and the workgroup stores its partial of every image when it has walked them all.  A second, small launch
// (batch_tail_kernel) then reduces them in two levels and steps every image's parameters, ONE THREAD per (image, quantity):
// the single-thread forms below add the same numbers in the same association as reduce_group / reduce_total's shuffle
// trees, so an image's bits are those of its own launch.
// (Handing every image in by itself, its own arrival chain behind its own pass, cost 5.6 of an image's 14 us at 640x480 --
// the workgroup has nothing in flight while its stores drain and its arrival returns; tools/exp/batch_ablation.sh.)
constexpr int kBatchMax = 32;   // images per launch (their waves' sums wait in 5 KB of LDS)
static_assert((kFitGrid + kGroup - 1) / kGroup <= 64 && (kClosedGrid + kGroup - 1) / kGroup <= 64, "reduce_total_thread: one group per lane");

struct BatchLds {
    FitLds fit;
    float wsum[kBatchMax][4][kNumSums];
};

// wave_sum_fixed's tree as ONE thread adds it: lane 0 of the tree ends up with, level by level, v[l] + v[l + off] for off = 32,
// 16, ... 1 -- the same additions in the same association, so the same bits (zeros included: x + 0.0 is not always x).
template <int kOff, class F>
__device__ __forceinline__ double tree_node(F &v, int l) {
    if constexpr (kOff == 64) return v(l);
    else return tree_node<2 * kOff>(v, l) + tree_node<2 * kOff>(v, l + kOff);   // lane l at this level: its own sum + lane l + off's
}
template <class F>
__device__ __forceinline__ double tree_sum64(F v) { return tree_node<1>(v, 0); }

// reduce_group for one (image, quantity) by one thread.
__device__ __forceinline__ void reduce_group_thread(const float *partials, int n_blocks, int g, int q, double *gpart, int n_groups) {
    const float *row = partials + (size_t)q * n_blocks + (size_t)g * kGroup;
    const int have = min(kGroup, n_blocks - g * kGroup);
    static_assert(kGroup == 32, "the lanes beyond the group hold zeros in reduce_group");
    const double y = tree_sum64([&](int l) {
        return (l < kGroup && l < have) ? (double)__hip_atomic_load(row + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    });
    __hip_atomic_store(gpart + (size_t)q * n_groups + g, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// reduce_total for one (image, quantity) by one thread: lane l's running sum over the groups l, l + 64, ..., then the tree.
__device__ __forceinline__ double reduce_total_thread(const double *gpart, int n_groups, int q) {
    const double *row = gpart + (size_t)q * n_groups;   // (n_groups <= 64: kFitGrid / kGroup = 40; lane l's loop has one turn at most)
    return tree_sum64([&](int l) {
        double x = 0.0;
        if (l < n_groups) x += __hip_atomic_load(row + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return x;
    });
}

// What a wave needs to know of one image of the batch.  Everything is read through the constant address space (scalar loads
// the compiler tracks itself; nothing of it is written during the launch: the parameters are stepped by the tail launch), and
// the NEXT image's is asked for before this image's pass, so that no wave waits for a chain of dependent loads between two
// images (measured: 2.2 us per image with nothing else to do, tools/exp/batch_ablation.sh).
template <class T>
__device__ __forceinline__ T cload(const void *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

struct BatchView {
    uint8_t *ws;
    const PlanItem *plan;
    const StripEntry *strips;
    float *state;
    uint32_t n_mine;
    RangeCodes rc;
    bool fmt_ok;
    Water w;
    float gscale;
    ItemRegs head[kAhead + 1], se0;   // the wave's first plan items and strip entry (StreamChain)
};

template <int kFmt>
__device__ __forceinline__ BatchView batch_view(const BatchEntry *__restrict__ images, int i, uint32_t wid) {
    BatchView v;
    const BatchEntry *e = uniform_ptr(images + i);
    v.ws = cload<uint8_t *>(&e->ws);
    const uint64_t o_format = cload<uint64_t>(&e->o.format), o_count = cload<uint64_t>(&e->o.count), o_plan = cload<uint64_t>(&e->o.plan),
                   o_strips = cload<uint64_t>(&e->o.strips), o_params = cload<uint64_t>(&e->o.params), o_n = cload<uint64_t>(&e->o.n_obs_total),
                   o_state = cload<uint64_t>(&e->o.state);
    const uint32_t stride = cload<uint32_t>(&e->o.stride), kmax = cload<uint32_t>(&e->o.kmax);
    const uint32_t f0 = cload<uint32_t>(v.ws + o_format), f1 = cload<uint32_t>(v.ws + o_format + 4);
    v.rc = range_codes_of<kFmt>(f0, f1);
    v.fmt_ok = format_readable(kFmt, f0);
    v.n_mine = v.fmt_ok ? cload<uint32_t>(v.ws + o_count + 4ull * wid) : 0u;
    if (kExpBatch == 2) v.n_mine = 0u;
    v.plan = reinterpret_cast<const PlanItem *>(v.ws + o_plan) + (size_t)wid * stride;
    v.strips = reinterpret_cast<const StripEntry *>(v.ws + o_strips) + (size_t)wid * kmax;
    v.state = reinterpret_cast<float *>(v.ws + o_state);
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // load_water, from scalar loads
        v.w.B[c] = cload<float>(v.ws + o_params + 4 * c);
        v.w.nb[c] = -cload<float>(v.ws + o_params + 4 * (3 + c)) * kLog2e;
        v.w.ng[c] = -cload<float>(v.ws + o_params + 4 * (6 + c)) * kLog2e;
    }
    v.gscale = -2.0f * ((1.0f / 3.0f) / (float)cload<uint64_t>(v.ws + o_n));
    const ConstItems items = const_items(v.plan);   // (in bounds also for a wave without strips: every wave has its stride)
#pragma unroll
    for (int q = 0; q < kAhead + 1; ++q) v.head[q] = items[q];
    v.se0 = const_items(v.strips)[0];
    return v;
}

// The launch's workgroups take (image, workgroup-of-the-image) PAIRS in turn: pair q = image q / vblocks, the image's own
// workgroup q % vblocks (vblocks = Layout.fit_blocks: the grid of the image's own launches, for which its plan was written and
// whose partials its reduction trees add up), workgroup p of the launch takes the pairs p, p + gridDim.x, ...  An image at least as
// large as the persistent grid has vblocks = gridDim.x, and workgroup p then walks its own share of every image, as until round 5;
// a small image has fewer, fatter workgroups of its own (layout.h, kMinStripsPerWave), and the launch's workgroups share the
// batch's images out between them instead of every one of them taking 64 pixels of every image.
template <int kMode, int kFmt>
__global__ __launch_bounds__(256, kMode ? kClosedWaves : kGroupFitWaves) void batch_iter_kernel(const BatchEntry *__restrict__ images, int n_images,
                                                                                           const AdamCoef co, int row, uint32_t vblocks) {
    if (kExpNoBatchClosed && kMode == 1) return;   // (experiment.h: occupancy experiments on the closed-form kernel alone)
    __shared__ BatchLds blds;
    FitLds &lds = blds.fit;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t n_pairs = (uint32_t)n_images * vblocks, P = gridDim.x;
    auto view_of = [&](uint32_t q) { const uint32_t i = q / vblocks; return batch_view<kFmt>(images, (int)i, (q - i * vblocks) * 4u + (uint32_t)wave); };
    // Every wave walks the images by itself -- no barrier between them: its ten sums of an image go to LDS and the next
    // image's first items are already in flight.
    // The wave's streams of consecutive images are chained (StreamChain): image i + 1's first items take the places of image i's
    // trailing ones.  Only a wave without a strip in one of two neighbours drains its ring between them and starts over.
    auto prime = [&](const BatchView &x) {   // the ring is idle
        if (x.n_mine == 0u) return;
        const uint32_t ring0 = lds_addr(&lds.ring[wave][0][0]);
#pragma unroll
        for (int q = 0; q < kAhead; ++q) issue_item<kFmt>(x.head[q], ring0, (uint32_t)(q * kSlot), x.ws, lane, x.rc.fullB);
    };
    StreamChain ch;
    ch.cs = 0u;
    ch.behind = 0u;
    if (blockIdx.x >= n_pairs) return;   // (never: the launch has at most n_pairs workgroups)
    BatchView v = view_of(blockIdx.x);
    prime(v);
    int slot = 0;   // this workgroup's pairs in turn: where the waves' sums of pair blockIdx.x + slot P wait
    for (uint32_t pair = blockIdx.x; pair < n_pairs; pair += P, ++slot) {
        const bool more = pair + P < n_pairs;
        BatchView vn = v;
        if (more) vn = view_of(pair + P);   // on its way while this pair's strips are walked
        // Chained: this wave has strips in both pairs, and this one's stream is long enough to issue ALL of the next one's
        // first kAhead items in place of its trailing ones (a closed-form stream can be ONE item long: a strip of pixels nobody
        // observes -- its second item is already a trailing one).
        const bool chained = kExpBatchChain && more && v.n_mine != 0u && vn.n_mine != 0u && !(v.head[kAhead - 1].y & kShapeTrail);
#pragma unroll
        for (int q = 0; q < kAhead + 1; ++q) ch.head[q] = v.head[q];
        ch.se0 = v.se0;
        ch.next_on = chained;
#pragma unroll
        for (int q = 0; q < kAhead; ++q) ch.next_head[q] = vn.head[q];
        ch.next_ws = vn.ws;
        ch.next_fullB = vn.rc.fullB;
        float s[kNumSums];
        if (kMode == 0) {
            Acc acc;
            zero_acc(acc);
            if (!v.fmt_ok) acc.cost = __builtin_nanf("");
            float sBeta[3] = {0.f, 0.f, 0.f};
            grad_pass<kFmt, true, StreamChain>(lds, v.plan, v.strips, v.n_mine, v.ws, v.state, wave, lane, v.w, v.gscale, co, acc, sBeta, v.rc, &ch);
            const float q[kNumSums] = {acc.sB[0], acc.sB[1], acc.sB[2], acc.sGZ[0], acc.sGZ[1], acc.sGZ[2], sBeta[0], sBeta[1], sBeta[2], acc.cost};
#pragma unroll
            for (int j = 0; j < kNumSums; ++j) s[j] = q[j];
        } else {
            ClosedSums cs = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, v.fmt_ok ? 0.f : __builtin_nanf("")};
            closed_pass<kFmt, false, true, StreamChain>(lds, v.plan, v.strips, v.n_mine, v.ws, v.state, wave, lane, v.w, v.fmt_ok, cs, v.rc, &ch);
            const float q[kNumSums] = {cs.sB[0], cs.sB[1], cs.sB[2], cs.sGZ[0], cs.sGZ[1], cs.sGZ[2], cs.sBeta[0], cs.sBeta[1], cs.sBeta[2], cs.cost};
#pragma unroll
            for (int j = 0; j < kNumSums; ++j) s[j] = q[j];
        }
        if (more && !chained) {   // the wave's ring is idle: the pass has consumed its last item and drained its trailing ones
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            prime(vn);
        }
        wave_sums(s);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < kNumSums; ++j) blds.wsum[slot][wave][j] = s[j];
        }
        v = vn;
    }
    // ---- the launch's end: the partial of every (image, workgroup-of-the-image) pair this workgroup walked, the four waves in
    //      fixed order, where that image's own launch would have left it ----
    __syncthreads();
    for (int idx = t; idx < slot * kNumSums; idx += 256) {
        const int sl = idx / kNumSums, j = idx - sl * kNumSums;
        const uint32_t q = blockIdx.x + (uint32_t)sl * P, i = q / vblocks, vb = q - i * vblocks;
        float *partials = reinterpret_cast<float *>(images[i].ws + images[i].o.partials);
        __hip_atomic_store(partials + (size_t)j * vblocks + vb,
                           ((blds.wsum[sl][0][j] + blds.wsum[sl][1][j]) + blds.wsum[sl][2][j]) + blds.wsum[sl][3][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ... and the second launch of a batch iteration: one workgroup per reduction group.  Thread (image, quantity) -- sixteen
// threads to an image, sixteen images to a round -- adds its group's partials; the workgroup that arrives last takes every
// image's total, its step on B, beta, gamma and its log row.  (As the tail of the launch above these trees wanted more
// registers than its waves have: the kernel spilled.)
constexpr int kTailThreads = 16 * kBatchMax;   // sixteen threads to an image: all images of a launch in ONE round
__global__ __launch_bounds__(kTailThreads) void batch_tail_kernel(const BatchEntry *__restrict__ images, int n_images, int n_blocks, const AdamCoef co, int row) {
    __shared__ double stot[kBatchMax][kNumSums];
    __shared__ int is_last;
    const int t = threadIdx.x, g = blockIdx.x, n_groups = gridDim.x;
    const int i = t >> 4, q = t & 15;
    const bool mine = i < n_images && q < kNumSums;
    uint8_t *ws = mine ? images[i].ws : nullptr;
    if (mine)
        reduce_group_thread(reinterpret_cast<const float *>(ws + images[i].o.partials), n_blocks, g, q, reinterpret_cast<double *>(ws + images[i].o.gpart), n_groups);
    if (kExpBatch == 1) return;   // (timing experiment, experiment.h)
    // what the step needs besides the totals is asked for before the arrival (every workgroup: one of them will use it)
    float p = 0.f, m = 0.f, v = 0.f, Bc = 0.f, scale = 0.f;
    float *pstate = nullptr;
    if (mine && q < 9) {
        pstate = reinterpret_cast<float *>(ws + images[i].o.params);
        scale = (1.0f / 3.0f) / (float)(*reinterpret_cast<const uint64_t *>(ws + images[i].o.n_obs_total));
        p = pstate[q]; m = pstate[9 + q]; v = pstate[18 + q];
        Bc = pstate[q % 3];   // B before its step
    }
    unsigned *ticket = reinterpret_cast<unsigned *>(images[0].ws + images[0].o.ticket);   // the batch's arrival counter: the first image's
    if (!arrive_last(ticket, (unsigned)n_groups, &is_last)) return;
    if (mine) {
        const double y = reduce_total_thread(reinterpret_cast<const double *>(ws + images[i].o.gpart), n_groups, q);
        __hip_atomic_store(reinterpret_cast<double *>(ws + images[i].o.sums) + q, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stot[i][q] = y;
    }
    __syncthreads();
    if (!mine) return;
    // water_step (sucre.py:148-152), thread q of an image in the role of lane q
    double *trace = images[i].trace;
    if (q < 9) {
        const int c = q % 3;
        double grad;
        if (q < 3) grad = -2.0 * (double)scale * stot[i][c];
        else if (q < 6) grad = 2.0 * (double)scale * stot[i][6 + c];
        else grad = -2.0 * (double)scale * (double)Bc * stot[i][3 + c];
        adam_update(p, m, v, (float)grad, co);
        pstate[q] = p; pstate[9 + q] = m; pstate[18 + q] = v;
        if (trace) trace[(size_t)row * 10 + 1 + q] = (double)p;
    } else if (trace) {
        trace[(size_t)row * 10] = stot[i][9];
    }
}

constexpr int kBatchSet = 8;
struct BatchEntries { BatchEntry e[kBatchSet]; };
__global__ void batch_set_kernel(BatchEntry *dst, const BatchEntries src, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = src.e[threadIdx.x];
}

__global__ __launch_bounds__(256) void reduce_groups_kernel(const float *partials, int n_blocks, double *gpart,
                                                            int n_groups) {
    reduce_group(partials, n_blocks, blockIdx.x, gpart, n_groups);
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

// SUCRe.__init__ (sucre.py:36-50) in the sorted pixel order: one workgroup per sorted tile = four strips.
__global__ __launch_bounds__(256) void fit_init_kernel(const uint8_t *__restrict__ rgb1,
                                                       const float *__restrict__ depth1,
                                                       const float *__restrict__ J0, int H, int W, int tiles_x,
                                                       const uint32_t *__restrict__ perm, float *__restrict__ state,
                                                       float *__restrict__ pstate,
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
    float *st = state + ((size_t)tile * kStripsPerTile + (t >> 6)) * kStateFloats + (t & 63);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float Jc = __builtin_nanf("");
        if (valid) Jc = J0 ? J0[o * 3 + c] : unit_from_u8(rgb1[o * 3 + c]);
        st[c * kStripPx] = Jc;
        st[(3 + c) * kStripPx] = 0.f;
        st[(6 + c) * kStripPx] = 0.f;
    }
    if (tile == 0 && t < 27) pstate[t] = t < 9 ? p0.v[t] : 0.f;
    for (int i = tile * 256 + t; i < n_tickets; i += gridDim.x * 256) ticket[i] = 0u;
}

__global__ __launch_bounds__(256) void export_J_kernel(const float *__restrict__ state, int H, int W, int tiles_x,
                                                       const uint32_t *__restrict__ invperm, float *__restrict__ J) {
    const int tile = blockIdx.x, t = threadIdx.x;  // image tile / slot; invperm says where the pixel was sorted to
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int v = ty * kTile + (t >> 4), u = tx * kTile + (t & 15);
    if (v >= H || u >= W) return;
    const uint32_t dst = invperm[(size_t)tile * kTilePx + t];
    const size_t o = ((size_t)v * W + u) * 3;
    const float *st = state + (size_t)(dst / kStripPx) * kStateFloats + dst % kStripPx;
#pragma unroll
    for (int c = 0; c < 3; ++c) J[o + c] = st[c * kStripPx];
}

__global__ void set_n_obs_total_kernel(uint64_t *dst, uint64_t v) { *dst = v; }

hipError_t launch_fit_init(const Layout &L, uint8_t *ws, const uint8_t *rgb1, const float *depth1,
                           const float *params0, const float *J0, hipStream_t s) {
    Params9 p0;
    for (int i = 0; i < 9; ++i) p0.v[i] = params0[i];
    hipLaunchKernelGGL(fit_init_kernel, dim3(L.n_tiles), dim3(256), 0, s, rgb1, depth1, J0, L.H, L.W, L.tiles_x,
                       reinterpret_cast<const uint32_t *>(ws + L.off_perm), reinterpret_cast<float *>(ws + L.off_state),
                       reinterpret_cast<float *>(ws + L.off_params),
                       reinterpret_cast<unsigned *>(ws + L.off_ticket), (1 + L.n_groups) * kTicketStride, p0);  // n_groups = the larger grid's
    return hipGetLastError();
}

template <class Kernel>
static void launch_fit_kernel(Kernel kernel, int mode, const Layout &L, uint8_t *ws, const AdamCoef &co, double *trace_row,
                              hipStream_t s) {
    hipLaunchKernelGGL(kernel, dim3(L.fit_blocks[mode]), dim3(256), 0, s, ws,
                       reinterpret_cast<const PlanItem *>(ws + L.off_plan[mode]),
                       reinterpret_cast<const StripEntry *>(ws + L.off_plan_strips[mode]),
                       reinterpret_cast<const uint32_t *>(ws + L.off_plan_count[mode]), (uint32_t)L.plan_stride[mode], (uint32_t)L.plan_kmax[mode],
                       reinterpret_cast<float *>(ws + L.off_params),
                       reinterpret_cast<const uint64_t *>(ws + L.off_n_obs_total),
                       reinterpret_cast<float *>(ws + L.off_state), reinterpret_cast<float *>(ws + L.off_partials), co,
                       reinterpret_cast<unsigned *>(ws + L.off_ticket),
                       reinterpret_cast<double *>(ws + L.off_gpartials), L.fit_groups[mode],
                       reinterpret_cast<double *>(ws + L.off_sums), trace_row,
                       reinterpret_cast<const uint32_t *>(ws + L.off_total_chunks + sizeof(uint64_t)));
}

// J-parameter mode: gradient pass + Adam on J; closed-form mode: the one-pass kernel (J re-solved, then constant).
template <bool kFused>
static void launch_grad_variant(const Layout &L, uint8_t *ws, const AdamCoef &co, unsigned flags, double *trace_row,
                                hipStream_t s) {
    const bool u16 = (flags & SUCRE_FIT_OBS_U16MM) != 0;
    if (flags & SUCRE_FIT_CLOSED_FORM) {
        if (u16) launch_fit_kernel(fit_closed_kernel<kFused, 1, false>, 1, L, ws, co, trace_row, s);
        else launch_fit_kernel(fit_closed_kernel<kFused, 0, false>, 1, L, ws, co, trace_row, s);
    } else {
        if (u16) launch_fit_kernel(fit_grad_kernel<kFused, 1>, 0, L, ws, co, trace_row, s);
        else launch_fit_kernel(fit_grad_kernel<kFused, 0>, 0, L, ws, co, trace_row, s);
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
    const int mode = (flags & SUCRE_FIT_CLOSED_FORM) ? 1 : 0;
    hipLaunchKernelGGL(reduce_groups_kernel, dim3(L.fit_groups[mode]), dim3(256), 0, s,
                       reinterpret_cast<const float *>(ws + L.off_partials), L.fit_blocks[mode],
                       reinterpret_cast<double *>(ws + L.off_gpartials), L.fit_groups[mode]);
    hipLaunchKernelGGL(reduce_sums_kernel, dim3(1), dim3(256), 0, s,
                       reinterpret_cast<const double *>(ws + L.off_gpartials), L.fit_groups[mode],
                       reinterpret_cast<double *>(ws + L.off_sums));
    return hipGetLastError();
}

hipError_t launch_fit_step(const Layout &L, uint8_t *ws, const AdamCoef &co, double *trace_row, hipStream_t s) {
    hipLaunchKernelGGL(param_step_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const double *>(ws + L.off_sums),
                       reinterpret_cast<float *>(ws + L.off_params),
                       reinterpret_cast<const uint64_t *>(ws + L.off_n_obs_total), co, trace_row);
    return hipGetLastError();
}

hipError_t launch_update_J(const Layout &L, uint8_t *ws, int fmt, hipStream_t s) {
    if (fmt) launch_fit_kernel(fit_closed_kernel<false, 1, true>, 1, L, ws, AdamCoef{}, nullptr, s);
    else launch_fit_kernel(fit_closed_kernel<false, 0, true>, 1, L, ws, AdamCoef{}, nullptr, s);
    return hipGetLastError();
}

// Writes both plans (J-parameter and closed-form item streams) for the store the compaction has just built.
hipError_t launch_plan(const Layout &L, uint8_t *ws, hipStream_t s) {
    auto *meta = reinterpret_cast<const StripMeta *>(ws + L.off_strip_meta);
    const uint32_t W0 = (uint32_t)L.fit_blocks[0] * 4u, W1 = (uint32_t)L.fit_blocks[1] * 4u;
    const uint32_t Wmax = W0 < W1 ? W1 : W0;
    // Thread (wave, k) writes the items of the wave's k-th strip: ~17 scattered 8-byte stores.  A wave of the launch holds the
    // threads of 64 / Kmax fit waves; with every lane active each of its store instructions touches 64 cache lines, and the
    // kernel's time is those instructions (32 us per image at Kmax = 8 against 9 us at 32 slots per fit wave, most of them
    // idle): the threads are spread so that at most a quarter of a wave's lanes write.
    uint32_t Kmax = (uint32_t)(L.plan_kmax[0] > L.plan_kmax[1] ? L.plan_kmax[0] : L.plan_kmax[1]);   // strips of the busiest wave (of either mode)
    if (Kmax < 32u) Kmax = 32u;   // thread slots per fit wave (k >= the wave's strip count: nothing to do)
    hipLaunchKernelGGL(plan_kernel, dim3((Wmax * Kmax + 255u) / 256u, 2), dim3(256), 0, s, meta, L.n_strips,
                       reinterpret_cast<const uint32_t *>(ws + L.off_total_chunks) + 2, W0, W1, Kmax,
                       (uint32_t)L.plan_stride[0], (uint32_t)L.plan_stride[1], (uint32_t)L.plan_kmax[0], (uint32_t)L.plan_kmax[1],
                       reinterpret_cast<PlanItem *>(ws + L.off_plan[0]), reinterpret_cast<PlanItem *>(ws + L.off_plan[1]),
                       reinterpret_cast<StripEntry *>(ws + L.off_plan_strips[0]), reinterpret_cast<StripEntry *>(ws + L.off_plan_strips[1]),
                       reinterpret_cast<uint32_t *>(ws + L.off_plan_count[0]),
                       reinterpret_cast<uint32_t *>(ws + L.off_plan_count[1]), (uint64_t)L.off_comp, (uint64_t)L.off_state);
    return hipGetLastError();
}

hipError_t launch_export_J(const Layout &L, const uint8_t *ws, float *J, hipStream_t s) {
    hipLaunchKernelGGL(export_J_kernel, dim3(L.n_tiles), dim3(256), 0, s,
                       reinterpret_cast<const float *>(ws + L.off_state), L.H, L.W, L.tiles_x,
                       reinterpret_cast<const uint32_t *>(ws + L.off_invperm), J);
    return hipGetLastError();
}

size_t batch_bytes(int n_images) { return align_up((size_t)n_images * sizeof(BatchEntry), 256); }

// The image table of a batch for one J mode (the plans differ between the modes): entry i = image i's workspace, log and
// the offsets of its own layout.
hipError_t launch_batch_set(void *batch, int n_images, uint8_t *const *ws, double *const *trace, const Layout *layouts, unsigned flags,
                            hipStream_t s) {
    const int mode = (flags & SUCRE_FIT_CLOSED_FORM) ? 1 : 0;
    for (int i0 = 0; i0 < n_images; i0 += kBatchSet) {
        BatchEntries src;
        const int n = n_images - i0 < kBatchSet ? n_images - i0 : kBatchSet;
        for (int j = 0; j < kBatchSet; ++j) {
            const Layout &L = layouts[j < n ? i0 + j : i0];
            BatchOffsets o;
            o.plan = L.off_plan[mode]; o.strips = L.off_plan_strips[mode]; o.count = L.off_plan_count[mode];
            o.params = L.off_params; o.n_obs_total = L.off_n_obs_total; o.state = L.off_state; o.partials = L.off_partials;
            o.ticket = L.off_ticket; o.gpart = L.off_gpartials; o.sums = L.off_sums; o.format = L.off_total_chunks + sizeof(uint64_t);
            o.stride = (uint32_t)L.plan_stride[mode]; o.kmax = (uint32_t)L.plan_kmax[mode]; o.n_groups = L.fit_groups[mode];
            src.e[j] = BatchEntry{j < n ? ws[i0 + j] : nullptr, (j < n && trace) ? trace[i0 + j] : nullptr, o};
        }
        hipLaunchKernelGGL(batch_set_kernel, dim3(1), dim3(64), 0, s, static_cast<BatchEntry *>(batch) + i0, src, n);
    }
    return hipGetLastError();
}

// One iteration of every image of the batch: one launch (row = the iteration's row in every image's log).  L: any image's
// layout (the grid is a function of the image size).
hipError_t launch_batch_iter(const Layout &L, void *batch, int n_images, const AdamCoef &co, unsigned flags, int row, hipStream_t s) {
    const int mode = (flags & SUCRE_FIT_CLOSED_FORM) ? 1 : 0;
    const bool u16 = (flags & SUCRE_FIT_OBS_U16MM) != 0;
    const uint32_t vblocks = (uint32_t)L.fit_blocks[mode];   // the images' own grid (one size: one grid)
    const uint32_t full = (uint32_t)(mode ? kClosedGrid : 256 * kGroupFitWaves);   // what is resident at once
    const dim3 block(256);
    for (int i0 = 0; i0 < n_images; i0 += kBatchMax) {   // at most kBatchMax images per launch: a workgroup walks at most
        auto *b = static_cast<const BatchEntry *>(batch) + i0;   // kBatchMax pairs (vblocks <= full), whose sums wait in LDS
        const int n = n_images - i0 < kBatchMax ? n_images - i0 : kBatchMax;
        const uint32_t pairs = (uint32_t)n * vblocks;
        const dim3 grid(pairs < full ? pairs : full);
        if (mode) {
            if (u16) hipLaunchKernelGGL((batch_iter_kernel<1, 1>), grid, block, 0, s, b, n, co, row, vblocks);
            else hipLaunchKernelGGL((batch_iter_kernel<1, 0>), grid, block, 0, s, b, n, co, row, vblocks);
        } else {
            if (u16) hipLaunchKernelGGL((batch_iter_kernel<0, 1>), grid, block, 0, s, b, n, co, row, vblocks);
            else hipLaunchKernelGGL((batch_iter_kernel<0, 0>), grid, block, 0, s, b, n, co, row, vblocks);
        }
        hipLaunchKernelGGL(batch_tail_kernel, dim3(L.fit_groups[mode]), dim3(kTailThreads), 0, s, b, n, L.fit_blocks[mode], co, row);
    }
    return hipGetLastError();
}

size_t group_bytes(int n_images) { return align_up(sizeof(GroupHeader), 256) + (size_t)n_images * sizeof(GroupImage); }

int64_t group_sums_offset() { return (int64_t)offsetof(GroupHeader, sums); }

hipError_t launch_group_init(void *group, const float *params0, hipStream_t s) {
    Params9 p0;
    for (int i = 0; i < 9; ++i) p0.v[i] = params0[i];
    hipLaunchKernelGGL(group_init_kernel, dim3(1), dim3(64), 0, s, static_cast<GroupHeader *>(group), p0);
    return hipGetLastError();
}

hipError_t launch_group_set_image(void *group, int i, const Layout &L, uint8_t *ws, hipStream_t s) {
    GroupImage im;
    im.ws = ws;
    for (int m = 0; m < 2; ++m) {
        im.off_plan[m] = L.off_plan[m]; im.off_strips[m] = L.off_plan_strips[m]; im.off_count[m] = L.off_plan_count[m];
        im.stride[m] = (uint32_t)L.plan_stride[m]; im.kmax[m] = (uint32_t)L.plan_kmax[m]; im.n_waves[m] = (uint32_t)L.fit_blocks[m] * 4u;
    }
    im.off_state = L.off_state;
    im.off_format = L.off_total_chunks + sizeof(uint64_t);
    im.off_params = L.off_params;
    hipLaunchKernelGGL(group_set_image_kernel, dim3(1), dim3(1), 0, s, static_cast<GroupHeader *>(group), i, im);
    return hipGetLastError();
}

// step >= 1: the pass of iteration `step`; first applies the pending step of iteration step - 1 (if any).
hipError_t launch_group_iter(void *group, int n_images, int step, const AdamCoef &co_prev, const AdamCoef &co, unsigned flags,
                             uint64_t n_obs_total, double *trace_prev, hipStream_t s) {
    auto *g = static_cast<GroupHeader *>(group);
    const int apply = step > 1 ? 1 : 0;
    const int in = apply ? (step - 2) & 1 : 0, out = apply ? (step - 1) & 1 : 0;
    const bool u16 = (flags & SUCRE_FIT_OBS_U16MM) != 0;
    if (flags & SUCRE_FIT_CLOSED_FORM) {
        const int ng = (kClosedGrid + kGroup - 1) / kGroup;
        if (u16) hipLaunchKernelGGL((group_iter_kernel<1, 1>), dim3(kClosedGrid), dim3(256), 0, s, g, n_images, n_obs_total, co_prev, co, apply, in, out, trace_prev, ng, group_images(g));
        else hipLaunchKernelGGL((group_iter_kernel<1, 0>), dim3(kClosedGrid), dim3(256), 0, s, g, n_images, n_obs_total, co_prev, co, apply, in, out, trace_prev, ng, group_images(g));
    } else {
        const int ng = (kFitGrid + kGroup - 1) / kGroup;
        if (u16) hipLaunchKernelGGL((group_iter_kernel<0, 1>), dim3(kFitGrid), dim3(256), 0, s, g, n_images, n_obs_total, co_prev, co, apply, in, out, trace_prev, ng, group_images(g));
        else hipLaunchKernelGGL((group_iter_kernel<0, 0>), dim3(kFitGrid), dim3(256), 0, s, g, n_images, n_obs_total, co_prev, co, apply, in, out, trace_prev, ng, group_images(g));
    }
    return hipGetLastError();
}

// after the pass of iteration `step` (and its all-reduce): applies that last step
hipError_t launch_group_finish(void *group, int n_images, int step, const AdamCoef &co_prev, uint64_t n_obs_total,
                               double *trace_prev, hipStream_t s) {
    const int apply = step >= 1 ? 1 : 0;
    const int in = apply ? (step - 1) & 1 : 0, out = apply ? step & 1 : 0;
    hipLaunchKernelGGL(group_finish_kernel, dim3(1), dim3(64), 0, s, static_cast<GroupHeader *>(group), n_images, n_obs_total, co_prev,
                       apply, in, out, trace_prev);
    return hipGetLastError();
}

hipError_t launch_set_n_obs_total(const Layout &L, uint8_t *ws, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(set_n_obs_total_kernel, dim3(1), dim3(1), 0, s,
                       reinterpret_cast<uint64_t *>(ws + L.off_n_obs_total), n);
    return hipGetLastError();
}

}  // namespace sucre
