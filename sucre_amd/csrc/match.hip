// Dense two-way matching for gfx950: one wave owns one 16x16 tile of the target image, four pixels per lane.
//
// Replaces, per (target, view) pair, the reference's chain
//   unproject_depth_map (sfm.py:95-101) -> match_one_way x2 (sfm.py:115-119) -> Matches.map / __and__
//   (sfm.py:154-175) -> d = depth2[v2,u2] (sfm.py:137) -> I = rgb2[v2,u2] (loader.py:87)
//   -> cP = K2^-1 d [u2+.5, v2+.5, 1] (loader.py:113) -> z = ||cP|| (sucre.py:53)
// with one fused per-pixel test: p1 -> p2 = trunc(project(unproject(p1))) must be inside view 2, have
// depth2[p2] > 0, and p2 must project back onto exactly p1.  This is the map-free form of Matches.__and__ (every
// p2 occurs at most once in the backward list, so map[p2] == p1 iff the backward projection of p2 is p1).
//
// The arithmetic mirrors torch CPU float32 bit for bit: a (3x3)@(3xn) matmul is the FMA chain of dot3(), the
// divide and sqrt are IEEE (-fhip-fp32-correctly-rounded-divide-sqrt), and the file is built with
// -ffp-contract=off so no other operation is fused.  Match sets therefore equal the reference's exactly.
#include "experiment.h"
#include "launch.h"

namespace sucre {

__device__ __forceinline__ float dot3(const float *a, float x, float y, float z) {
    float acc = a[0] * x;
    acc = __builtin_fmaf(a[1], y, acc);
    acc = __builtin_fmaf(a[2], z, acc);
    return acc;
}

// A pinhole camera matrix and its inverse have the form [[a, 0, b], [0, c, d], [0, 0, 1]] (sfm.py:62-78; torch's LU
// inverse of an upper-triangular matrix keeps the zeros exact).  For such a matrix the FMA chain of dot3() loses its
// terms with a zero coefficient without changing a bit of any finite result: fma(0, y, acc) = acc and 0 * x = 0 exactly
// (only the sign of a zero can differ, which no comparison, truncation or product downstream sees), and a non-finite
// x, y or z still makes c0 or c1 non-finite, so the bound test rejects the point in both forms.  9 -> 4 operations per
// product; the flag is wave-uniform (one scalar branch).
__host__ __device__ __forceinline__ bool pinhole_form(const float *M) {
    return M[1] == 0.f && M[3] == 0.f && M[6] == 0.f && M[7] == 0.f && M[8] == 1.f;
}

__device__ __forceinline__ void mul3(const float *M, bool pin, float x, float y, float z, float out[3]) {
    if (pin) {
        out[0] = __builtin_fmaf(M[2], z, M[0] * x);
        out[1] = __builtin_fmaf(M[5], z, M[4] * y);
        out[2] = z;
    } else {
        out[0] = dot3(M + 0, x, y, z);
        out[1] = dot3(M + 3, x, y, z);
        out[2] = dot3(M + 6, x, y, z);
    }
}

// sfm.py:90-93
__device__ __forceinline__ void unproject(const float *Kinv, bool pin, float u, float v, float d, float out[3]) {
    const float x = d * (u + 0.5f), y = d * (v + 0.5f), z = d * 1.0f;
    mul3(Kinv, pin, x, y, z, out);
}

// sfm.py:49-55
__device__ __forceinline__ void rigid(const float *R, const float *t, const float p[3], float out[3]) {
    out[0] = dot3(R + 0, p[0], p[1], p[2]) + t[0];
    out[1] = dot3(R + 3, p[0], p[1], p[2]) + t[1];
    out[2] = dot3(R + 6, p[0], p[1], p[2]) + t[2];
}

// x / z and y / z as far as their users can tell.  Everything downstream of the two IEEE quotients of sfm.py:106 is a
// comparison with an integer (-1 < d, d < W) or Tensor.long() (truncation), so a quotient may be replaced by any value
// that lies strictly between the same two consecutive integers.  a = x * rcp(z) is within 1.5 * 2^-23 |a| of the real
// quotient (v_rcp_f32: 1 ulp; the product: half an ulp), and the correctly rounded quotient is monotone and reaches an
// integer n only from within half an ulp of n (<= 2^-24 (|a| + 1)): if a is farther than 2^-21 |a| + 2^-23 from every
// integer -- more than twice those bounds together -- truncation and bound tests of a ARE those of the IEEE quotient.
// Otherwise (a within that margin of an integer: ~0.1 % of the lanes at a = 1000; or a NaN / infinite / zero, which
// covers a denormal or zero z) the whole wave takes the two IEEE divisions.  12 instructions instead of 22 per pair of
// quotients, four pairs per matching pixel pair; the match sets keep equal to the reference's bit for bit
// (experiment.h: SUCRE_EXACT_DIV=1 builds the plain form; test_pixel_quotients_on_the_integer_boundaries).
__device__ __forceinline__ void pixel_quotients(float x, float y, float z, float *px, float *py) {
    if (kExactDiv) {
        *px = x / z;
        *py = y / z;
        return;
    }
    const float rc = __builtin_amdgcn_rcpf(z);
    float ax = x * rc, ay = y * rc;
    const float sx = __builtin_fmaf(-__builtin_fabsf(ax), 0x1p-21f, __builtin_fabsf(ax - __builtin_rintf(ax)));
    const float sy = __builtin_fmaf(-__builtin_fabsf(ay), 0x1p-21f, __builtin_fabsf(ay - __builtin_rintf(ay)));
    const bool unsure = !(sx > 0x1p-23f) || !(sy > 0x1p-23f);   // NaN lands here too
    if (__builtin_amdgcn_ballot_w64(unsure) != 0ull) {
        ax = x / z;
        ay = y / z;
    }
    *px = ax;
    *py = ay;
}

// sfm.py:103-107,116-117: world point -> continuous pixel; true when Tensor.long() of it lies inside WxH.
// trunc(x) in [0, W-1]  <=>  -1 < x < W ; NaN and +-inf fail both comparisons like INT64_MIN fails the bound test.
__device__ __forceinline__ bool project(const float *Rinv, const float *tinv, const float *K, bool pin, float Wf, float Hf,
                                        const float wP[3], float *px, float *py) {
    float cP[3], c[3];
    rigid(Rinv, tinv, wP, cP);
    mul3(K, pin, cP[0], cP[1], cP[2], c);
    pixel_quotients(c[0], c[1], c[2], px, py);
    return (*px > -1.0f) && (*px < Wf) && (*py > -1.0f) && (*py < Hf);
}

// The views' pixel pointers come out of a table in memory, so the compiler cannot know their address space and would
// gather through flat loads (which also count against the LDS counter); they are device-memory pointers by contract.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T *global_ptr(const T *p) {
    return (const __attribute__((address_space(1))) T *)p;
}

// The fused two-way test for one target pixel whose world point is wP: returns true and the matched pixel of
// view 2 (linear index q, camera-frame point c2) iff p1 -> p2 -> p1 closes (see the file header).
// `packed`: the view's `depth` pointer holds sucre_pack_view's 8-byte records {float32 depth, r, g, b, 0} (its `rgb`
// pointer is NULL): depth and colour of the landing pixel arrive with ONE gather -- *rgbw gets the colour word.
__device__ __forceinline__ bool match_pixel(const CamDev &c1, float W1f, float H1f, const sucre_view_t *vw, bool packed,
                                            float W2f, float H2f, bool pin, const float wP[3], int u1, int v1,
                                            size_t *q_out, float c2[3], uint32_t *rgbw) {
    float px, py;
    if (!project(vw->Rinv, vw->tinv, vw->K, pin, W2f, H2f, wP, &px, &py)) return false;
    const int u2 = (int)px, v2 = (int)py;
    const size_t q = (size_t)v2 * vw->W + u2;
    float d2;
    if (packed) {
        const unsigned long long rec = global_ptr(reinterpret_cast<const unsigned long long *>(vw->depth))[q];
        d2 = __uint_as_float((uint32_t)rec);
        *rgbw = (uint32_t)(rec >> 32);
    } else {
        d2 = global_ptr(vw->depth)[q];
    }
    if (!(d2 > 0.0f)) return false;
    float w2[3], qx, qy;
    unproject(vw->Kinv, pin, (float)u2, (float)v2, d2, c2);
    rigid(vw->R, vw->t, c2, w2);
    if (!project(c1.Rinv, c1.tinv, c1.K, pin, W1f, H1f, w2, &qx, &qy)) return false;
    if ((int)qx != u1 || (int)qy != v1) return false;
    *q_out = q;
    return true;
}

constexpr int kViewsPerGroup = 16;
// One 16x16 tile x up to 16 views per workgroup; wave w takes the views k = w (mod 4) of the group, four pixels per
// lane.  By-product: which pixels of the tile view k observes, as the four ballots of the lanes' pixels (`vbits`, 32 bytes
// per (tile, view): word j, bit l = slot 64 j + l): the compaction derives every pixel's view mask from them instead of re-reading every dense range.
//
// What did NOT pay here, each measured against the round-2 kernel on the same box (tools/exp/ab_match.sh; 1080p x 65
// views, 837-870 us): (a) a tile-level cull before matching -- a pre-pass projecting the 8 corners of every tile's
// frustum slab into every view and skipping the pairs that provably miss (34 % of all pairs, against 35 % that have no
// match): +-0 on this kernel plus 22 us for the pre-pass, because a missing pair already costs only its forward
// projection and the kernel's time is in the pairs that match; (b) the pinhole form of K and K^-1 (zeros dropped from
// the FMA chains, exact for every finite input; -9 % VALU instructions): +1 %; (c) the four pixels of a lane in phases
// (forward projections, the four depth gathers together, backward projections, colour gathers together) for
// memory-level parallelism: 112 VGPRs, 4 waves/SIMD instead of 7, +5 %; (d) global_load instead of flat_load gathers:
// +-0 (kept: it is what the pointers are); (e) per-pixel view masks combined in the kernel -- the four waves' bits
// through LDS behind a barrier: +6 % (the fast waves wait); one wave per 16 views, no barrier: +11 % (32 K long waves
// quantise badly over the SIMDs); (f) the dense chunks written with nontemporal stores, so that they do not push the
// views' records out of L2: -1 % (the run-to-run noise).  Bit parity with torch fixes the kernel's chains -- except the
// four quotients, which only feed truncations and bound tests (pixel_quotients): 306 M -> 274 M vector instructions per
// image (profiles/r03_jparam_summary.txt), -2.3 % on the same box.
//
// kExt: extension planes are written (`ext`: the camera points, or float32 colours with SUCRE_EXT_COLOUR).
// kBoth (SUCRE_EXT_POINTS_COLOUR): the views' colour images are float32 AND the camera points are kept -- cP goes to
// `ext`, the float32 colour to `ext2` (light model on resized images).
// Minimum / maximum over the wave by data-parallel-primitive moves (no LDS traffic; a shuffle tree is six ds_bpermute per
// value): row_shr 1, 2, 4, 8 inside the rows of 16, then the row broadcasts 15 and 31 -- lane 63 ends up with the result.
#define SUCRE_DPP_STEP(OP, X, CTRL, ROWS, FILL) X = OP(X, (uint32_t)__builtin_amdgcn_update_dpp((int)(FILL), (int)X, CTRL, ROWS, 0xf, false))
__device__ __forceinline__ uint32_t wave_umin_to_last(uint32_t x) {
    SUCRE_DPP_STEP(min, x, 0x111, 0xf, 0xffffffffu); SUCRE_DPP_STEP(min, x, 0x112, 0xf, 0xffffffffu);
    SUCRE_DPP_STEP(min, x, 0x114, 0xf, 0xffffffffu); SUCRE_DPP_STEP(min, x, 0x118, 0xf, 0xffffffffu);
    SUCRE_DPP_STEP(min, x, 0x142, 0xa, 0xffffffffu); SUCRE_DPP_STEP(min, x, 0x143, 0xc, 0xffffffffu);
    return x;
}
__device__ __forceinline__ uint32_t wave_umax_to_last(uint32_t x) {
    SUCRE_DPP_STEP(max, x, 0x111, 0xf, 0u); SUCRE_DPP_STEP(max, x, 0x112, 0xf, 0u);
    SUCRE_DPP_STEP(max, x, 0x114, 0xf, 0u); SUCRE_DPP_STEP(max, x, 0x118, 0xf, 0u);
    SUCRE_DPP_STEP(max, x, 0x142, 0xa, 0u); SUCRE_DPP_STEP(max, x, 0x143, 0xc, 0u);
    return x;
}
#undef SUCRE_DPP_STEP

template <bool kBoth, bool kExt>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void match_kernel(uint8_t *__restrict__ obs, uint16_t *__restrict__ cnt,
                                                    const float *__restrict__ depth1, const CamDev c1,
                                                    const sucre_view_t *__restrict__ views, int n_views, int k0,
                                                    int k1, int tiles_x, int n_tiles, int tiles_per_xcd,
                                                    size_t tile_stride, size_t view_stride, uint8_t *__restrict__ ext,
                                                    int ext_mode, uint8_t *__restrict__ ext2,
                                                    uint64_t *__restrict__ vbits, uint2 *__restrict__ zrange) {
    // A wave's results of one view pass through LDS on their way out: the lanes COMPUTE on rows of the tile (see below) but
    // the dense chunk is stored four adjacent slots per lane (one 16-byte + three 4-byte stores per lane, as the readers expect)
    __shared__ __attribute__((aligned(16))) float lz[4][kTilePx];
    __shared__ __attribute__((aligned(16))) uint8_t lc[4][3 * kTilePx];
    __shared__ __attribute__((aligned(16))) float le[kExt ? 4 : 1][kExt ? 3 : 1][kExt ? kTilePx : 1];
    __shared__ __attribute__((aligned(16))) float lf[kBoth ? 4 : 1][kBoth ? 3 : 1][kBoth ? kTilePx : 1];
    // Workgroups are dealt round-robin over the 8 XCDs: give every XCD one contiguous band of tiles so the
    // depth2 / rgb2 gathers of neighbouring tiles share that XCD's L2 (speed only, never correctness).
    const int tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    // Lane l works on the pixels (row 4 j + l / 16, column l % 16), j = 0..3, i.e. slot 64 j + l: sixteen adjacent lanes
    // hold sixteen adjacent pixels of a row, so a gather instruction touches 4 short row segments of the view instead of
    // 16 (the kernel's gathers -- 64 scattered lanes through the texture addresser -- are what it is bound by: with four
    // adjacent pixels per lane every gather spread over 16 rows; round 3, -x % on the kernel, tools/exp/ab_match.sh).
    const int u1 = tx * kTile + (lane & 15);
    const int v1b = ty * kTile + (lane >> 4);
    const float W1f = (float)c1.W, H1f = (float)c1.H;
    const bool pin1 = pinhole_form(c1.K) && pinhole_form(c1.Kinv);

    float wP[4][3];
    bool ok1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int v1 = v1b + 4 * j;
        float d = 0.0f;
        if (v1 < c1.H && u1 < c1.W) d = depth1[(size_t)v1 * c1.W + u1];
        ok1[j] = d > 0.0f;
        float cP[3];
        unproject(c1.Kinv, pin1, (float)u1, (float)v1, d, cP);
        rigid(c1.R, c1.t, cP, wP[j]);
    }

    const int kb = k0 + blockIdx.y * kViewsPerGroup;
    const int ke = min(kb + kViewsPerGroup, k1);
    // smallest / largest bit pattern of the wave's ranges (positive floats order like integers), over ALL the views the wave
    // walks: what the compaction needs to know to keep the ranges as 24-bit codes (layout.h, kStoreZ24) is the image's span,
    // so the wave reduces once, after its last view, and leaves the result in that view's entry (neutral pairs in the others)
    uint32_t zlo = 0xffffffffu, zhi = 0u;
    for (int k = kb + wave; k < ke; k += 4) {
        const sucre_view_t *vw = views + k;  // wave-uniform: scalar loads
        const uint8_t *__restrict__ rgb2 = vw->rgb;
        const bool packed = rgb2 == nullptr;   // wave-uniform: sucre_pack_view records behind vw->depth
        const float W2f = (float)vw->W, H2f = (float)vw->H;
        const bool pin = pin1 && pinhole_form(vw->K) && pinhole_form(vw->Kinv);   // the pair's four matrices: one scalar branch
        int total = 0;
        unsigned long long bal[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool m = ok1[j];
            float z = 0.0f;
            uint32_t r = 0, g = 0, b = 0;
            float c2[3] = {0.f, 0.f, 0.f};
            float f2[3] = {0.f, 0.f, 0.f};
            if (m) {
                size_t q;
                uint32_t rgbw = 0;
                m = match_pixel(c1, W1f, H1f, vw, packed, W2f, H2f, pin, wP[j], u1, v1b + 4 * j, &q, c2, &rgbw);
                if (m) {
                    z = sqrtf(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
                    zlo = min(zlo, __float_as_uint(z)); zhi = max(zhi, __float_as_uint(z));
                    if (kBoth) {                          // float32 colour image, camera point kept as well
                        const auto *pf = global_ptr(reinterpret_cast<const float *>(rgb2)) + q * 3;
                        f2[0] = pf[0]; f2[1] = pf[1]; f2[2] = pf[2];
                    } else if (kExt && ext_mode == SUCRE_EXT_COLOUR) {  // float32 colour image (resized inputs): the planes carry I
                        const auto *pf = global_ptr(reinterpret_cast<const float *>(rgb2)) + q * 3;
                        c2[0] = pf[0]; c2[1] = pf[1]; c2[2] = pf[2];
                    } else {
                        if (!packed) {
                            // one (unaligned) 4-byte gather instead of a 2-byte and a 1-byte one; the image's last pixel
                            // reads the dword that ENDS at its blue byte, so nothing past the buffer is touched
                            const bool last = q + 1 == (size_t)vw->H * vw->W;
                            typedef uint32_t __attribute__((aligned(1))) u32_any;
                            rgbw = *(const __attribute__((address_space(1))) u32_any *)(rgb2 + q * 3 - (last ? 1 : 0));
                            rgbw = last ? rgbw >> 8 : rgbw;
                        }
                        r = rgbw & 255u; g = (rgbw >> 8) & 255u; b = (rgbw >> 16) & 255u;
                    }
                }
            }
            const int slot = 64 * j + lane;
            lz[wave][slot] = z;
            lc[wave][slot] = (uint8_t)r; lc[wave][kTilePx + slot] = (uint8_t)g; lc[wave][2 * kTilePx + slot] = (uint8_t)b;
            if constexpr (kExt) { le[wave][0][slot] = m ? c2[0] : 0.f; le[wave][1][slot] = m ? c2[1] : 0.f; le[wave][2][slot] = m ? c2[2] : 0.f; }
            if constexpr (kBoth) { lf[wave][0][slot] = m ? f2[0] : 0.f; lf[wave][1][slot] = m ? f2[1] : 0.f; lf[wave][2][slot] = m ? f2[2] : 0.f; }
            bal[j] = __ballot(m);   // word j, bit l: slot 64 j + l
            total += __builtin_popcountll(bal[j]);
        }
        if (lane < 4)   // the four ballots in ONE store instruction (lane j holds word j)
            vbits[((size_t)tile * n_views + k) * 4 + lane] = lane == 0 ? bal[0] : lane == 1 ? bal[1] : lane == 2 ? bal[2] : bal[3];
        if (lane == 0) cnt[(size_t)tile * n_views + k] = (uint16_t)total;
        if (k + 4 >= ke) {   // the wave's last view (wave-uniform)
            const uint32_t lo = wave_umin_to_last(zlo), hi = wave_umax_to_last(zhi);
            if (lane == 63) zrange[(size_t)tile * n_views + k] = make_uint2(lo, hi);
        } else if (lane == 63) {
            zrange[(size_t)tile * n_views + k] = make_uint2(0xffffffffu, 0u);
        }
        if (total > 0 && !kExpMatchCountOnly) {  // wave-uniform; chunks of empty (tile, view) pairs are never read
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave reads back what its own lanes wrote: LDS
            __builtin_amdgcn_wave_barrier();                          // operations of one wave complete in order
            uint8_t *chunk = obs + (size_t)tile * tile_stride + (size_t)k * view_stride;
            *reinterpret_cast<float4 *>(chunk + lane * 16) = *reinterpret_cast<const float4 *>(&lz[wave][lane * 4]);
            uint32_t *c = reinterpret_cast<uint32_t *>(chunk + kChunkZ) + lane;  // planar R | G | B, 256 B each
            const uint32_t *cw = reinterpret_cast<const uint32_t *>(&lc[wave][0]) + lane;
            c[0] = cw[0]; c[64] = cw[64]; c[128] = cw[128];
            if constexpr (kExt) {  // extension planes: the camera-frame point cP of every observation (light model,
                                   // loader.py:113) or its float32 colour (SUCRE_EXT_COLOUR)
                uint8_t *e = ext + ((size_t)tile * n_views + k) * kExtChunk;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<float4 *>(e + pl * kChunkZ + lane * 16) = *reinterpret_cast<const float4 *>(&le[wave][pl][lane * 4]);
            }
            if constexpr (kBoth) {
                uint8_t *e = ext2 + ((size_t)tile * n_views + k) * kExtChunk;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<float4 *>(e + pl * kChunkZ + lane * 16) = *reinterpret_cast<const float4 *>(&lf[wave][pl][lane * 4]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // ... and the next view's writes come after these reads
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Explicit correspondences of one view: map[v1*W + u1] = v2*W2 + u2, or -1 (the (u1,v1,u2,v2) lists of
// sfm.Matches, sfm.py:145-152, in dense form).  One thread per target pixel; same per-pixel test as match_kernel.
__global__ __launch_bounds__(256) void match_map_kernel(const float *__restrict__ depth1, const CamDev c1,
                                                        const sucre_view_t *__restrict__ views, int k,
                                                        int32_t *__restrict__ map) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= c1.H * c1.W) return;
    const int v1 = idx / c1.W, u1 = idx - v1 * c1.W;
    const float d = depth1[idx];
    int32_t out = -1;
    if (d > 0.0f) {
        const sucre_view_t *vw = views + k;
        float cP[3], wP[3], c2[3];
        size_t q;
        const bool pin = pinhole_form(c1.K) && pinhole_form(c1.Kinv) && pinhole_form(vw->K) && pinhole_form(vw->Kinv);
        unproject(c1.Kinv, pin, (float)u1, (float)v1, d, cP);
        rigid(c1.R, c1.t, cP, wP);
        uint32_t rgbw;
        if (match_pixel(c1, (float)c1.W, (float)c1.H, vw, vw->rgb == nullptr, (float)vw->W, (float)vw->H, pin, wP, u1, v1, &q, c2, &rgbw))
            out = (int32_t)q;
    }
    map[idx] = out;
}

// Image.match_one_way's arithmetic for an explicit point list (sfm.py:103-107, 115-117): world point i -> linear pixel
// index in the view, -1 = outside.  wP is (3, n) row-major.
__global__ __launch_bounds__(256) void project_points_kernel(const CamDev cam, const float *__restrict__ wP, long long n,
                                                             int32_t *__restrict__ pix) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {wP[i], wP[n + i], wP[2 * n + i]};
    float px, py;
    const bool inside = project(cam.Rinv, cam.tinv, cam.K, pinhole_form(cam.K), (float)cam.W, (float)cam.H, p, &px, &py);
    pix[i] = inside ? (int32_t)py * cam.W + (int32_t)px : -1;
}

// len(matches) per view, the min_cover rule of sfm.py:136 (Python int/int true division, strict >) and
// n_obs = len(matches_data) (loader.py:52-53: matches of the kept views), in two small launches: every workgroup of the
// first adds up the counts of kStatTiles tiles for all views (lanes on views: the [tile][view] table is read in rows,
// 32 loads in flight per thread), the single workgroup of the second adds the partial rows up.  (One workgroup per view
// walking the table down a column -- 130-byte strides -- took 18 + 5 us per image.)
constexpr int kStatTiles = 32;

// Both also fold the pairs' range bits (match_kernel / count_view_kernel leave the smallest and the largest per (tile, view)
// pair with a match) into the image's: over ALL views -- a view the min_cover rule drops may widen the span, which only ever
// costs the 24-bit store (layout.h), never correctness.
__global__ __launch_bounds__(256) void view_partial_kernel(const uint16_t *__restrict__ cnt, int n_tiles, int n_views,
                                                           uint32_t *__restrict__ partial, const uint2 *__restrict__ zrange,
                                                           uint2 *__restrict__ zpart) {
    __shared__ uint32_t plo[4], phi[4];
    const int t0 = blockIdx.x * kStatTiles, n = min(kStatTiles, n_tiles - t0);
    uint32_t zlo = 0xffffffffu, zhi = 0u;
    for (int k = threadIdx.x; k < n_views; k += 256) {
        const uint16_t *p = cnt + (size_t)t0 * n_views + k;
        uint32_t s = 0;
        if (n == kStatTiles) {
#pragma unroll
            for (int i = 0; i < kStatTiles; ++i) s += p[(size_t)i * n_views];
        } else {
            for (int i = 0; i < n; ++i) s += p[(size_t)i * n_views];
        }
        partial[(size_t)blockIdx.x * n_views + k] = s;
    }
    {   // the rows of the workgroup's tiles are one contiguous run of pairs (every pair is written by whoever filled the view,
        // with or without a match: the neutral pair then): all 256 threads, coalesced
        const uint2 *base = zrange + (size_t)t0 * n_views;
        const int total = n * n_views;
        for (int i = threadIdx.x; i < total; i += 256) {
            const uint2 r = base[i];
            zlo = min(zlo, r.x); zhi = max(zhi, r.y);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { zlo = min(zlo, (uint32_t)__shfl_xor((int)zlo, off, 64)); zhi = max(zhi, (uint32_t)__shfl_xor((int)zhi, off, 64)); }
    if ((threadIdx.x & 63) == 0) { plo[threadIdx.x >> 6] = zlo; phi[threadIdx.x >> 6] = zhi; }
    __syncthreads();
    if (threadIdx.x == 0) zpart[blockIdx.x] = make_uint2(min(min(plo[0], plo[1]), min(plo[2], plo[3])), max(max(phi[0], phi[1]), max(phi[2], phi[3])));
}

__global__ __launch_bounds__(1024) void view_total_kernel(const uint32_t *__restrict__ partial, int n_rows, int n_views,
                                                          double min_cover, double hw, uint64_t *__restrict__ view_count,
                                                          uint32_t *__restrict__ view_keep, uint64_t *__restrict__ n_obs,
                                                          uint64_t *__restrict__ n_obs_total, const uint2 *__restrict__ zpart,
                                                          uint32_t *__restrict__ zspan) {
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t plo[16], phi[16];
    const int t = threadIdx.x;
    {   // the image's smallest / largest range bits
        uint32_t zlo = 0xffffffffu, zhi = 0u;
        for (int i = t; i < n_rows; i += 1024) { const uint2 r = zpart[i]; zlo = min(zlo, r.x); zhi = max(zhi, r.y); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { zlo = min(zlo, (uint32_t)__shfl_xor((int)zlo, off, 64)); zhi = max(zhi, (uint32_t)__shfl_xor((int)zhi, off, 64)); }
        if ((t & 63) == 0) { plo[t >> 6] = zlo; phi[t >> 6] = zhi; }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < 16; ++w) { zlo = min(zlo, plo[w]); zhi = max(zhi, phi[w]); }
            zspan[0] = zlo; zspan[1] = zhi;
        }
    }
    int kc = 64;   // views side by side (a power of two), 1024 / kc slices of the partial rows
    while (kc < n_views && kc < 1024) kc <<= 1;
    const int slices = 1024 / kc, k0 = t & (kc - 1), r = t / kc;
    unsigned long long kept = 0;
    for (int kb = 0; kb < n_views; kb += kc) {
        const int k = kb + k0;
        unsigned long long s = 0;
        if (k < n_views) {
            int i = r;
            for (; i + 7 * slices < n_rows; i += 8 * slices) {
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(i + j * slices) * n_views + k];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[j];
            }
            for (; i < n_rows; i += slices) s += partial[(size_t)i * n_views + k];
        }
        part[t] = s;
        __syncthreads();
        if (r == 0 && k < n_views) {
            for (int j = 1; j < slices; ++j) s += part[j * kc + k0];
            const bool keep = (double)s / hw > min_cover;
            view_count[k] = s;
            view_keep[k] = keep ? 1u : 0u;
            if (keep) kept += s;
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_down(kept, off, 64);
    if ((t & 63) == 0) wsum[t >> 6] = kept;
    __syncthreads();
    if (t == 0) {
        unsigned long long total = 0;
        for (int w = 0; w < 16; ++w) total += wsum[w];
        *n_obs = total;
        *n_obs_total = total;
    }
}

__global__ __launch_bounds__(256) void export_view_kernel(const uint8_t *__restrict__ obs,
                                                          const uint16_t *__restrict__ cnt, int n_views, int k,
                                                          int tiles_x, int H, int W, float *__restrict__ z_out,
                                                          uint8_t *__restrict__ rgb_out, size_t tile_stride,
                                                          size_t view_stride) {
    const int tile = blockIdx.x;
    const int slot = threadIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int v = ty * kTile + (slot >> 4), u = tx * kTile + (slot & 15);
    if (v >= H || u >= W) return;
    float z = 0.0f;
    uint8_t c[3] = {0, 0, 0};
    if (cnt[(size_t)tile * n_views + k] > 0) {
        const uint8_t *chunk = obs + (size_t)tile * tile_stride + (size_t)k * view_stride;
        z = reinterpret_cast<const float *>(chunk)[slot];
        const uint8_t *p = chunk + kChunkZ + slot;
        c[0] = p[0]; c[1] = p[kTilePx]; c[2] = p[2 * kTilePx];
    }
    const size_t o = (size_t)v * W + u;
    if (z_out) z_out[o] = z;
    if (rgb_out) {
        const bool has = z > 0.0f;
        rgb_out[o * 3 + 0] = has ? c[0] : 0;
        rgb_out[o * 3 + 1] = has ? c[1] : 0;
        rgb_out[o * 3 + 2] = has ? c[2] : 0;
    }
}

// MatchesFile.check_integrity (loader.py:89-101) for every view in one launch: per view, bit 0 = a stored range is not
// finite, bit 1 = a stored range is negative, and the number of ranges > 0 (compared with the view's match count by
// the second kernel -> bit 2).  Workgroup (k, j) walks view k's chunks of the tiles j, j + gridDim.y, ... and reports
// once.  (The first version, one workgroup per tile reporting per (tile, view, wave), queued 1.5 M atomics on 48
// addresses: 3.8 ms per 1080p image.)
__global__ __launch_bounds__(256) void integrity_scan_kernel(const uint8_t *__restrict__ obs,
                                                             const uint16_t *__restrict__ cnt, int n_views, int n_tiles,
                                                             size_t tile_stride, size_t view_stride,
                                                             uint32_t *__restrict__ verdict,
                                                             unsigned long long *__restrict__ positives) {
    const int k = blockIdx.x, slot = threadIdx.x;
    __shared__ unsigned long long s_pos;
    __shared__ uint32_t s_bad;
    if (slot == 0) { s_pos = 0ull; s_bad = 0u; }
    __syncthreads();
    uint32_t bad = 0u, pos = 0u;
    for (int tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        if (cnt[(size_t)tile * n_views + k] == 0) continue;   // workgroup-uniform: the chunk was never written
        const float z = reinterpret_cast<const float *>(obs + (size_t)tile * tile_stride + (size_t)k * view_stride)[slot];
        bad |= (__builtin_isfinite(z) ? 0u : 1u) | (z < 0.0f ? 2u : 0u);
        pos += z > 0.0f ? 1u : 0u;
    }
    if (pos != 0u) atomicAdd(&s_pos, (unsigned long long)pos);
    if (bad != 0u) atomicOr(&s_bad, bad);
    __syncthreads();
    if (slot == 0) {
        if (s_pos != 0ull) atomicAdd(positives + k, s_pos);
        if (s_bad != 0u) atomicOr(verdict + k, s_bad);
    }
}

__global__ void integrity_verdict_kernel(const unsigned long long *__restrict__ positives,
                                         const uint64_t *__restrict__ view_count, int n_views,
                                         uint32_t *__restrict__ verdict) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_views && positives[k] != view_count[k]) verdict[k] |= 4u;
}

// ---- import of an explicit match list (e.g. one group of a matches file written by the reference, loader.py:68-76)
// into the dense store: clear view k, scatter the observations, recount the tiles.
__global__ __launch_bounds__(256) void clear_view_kernel(uint8_t *__restrict__ obs, uint16_t *__restrict__ cnt,
                                                         int n_views, int k, size_t tile_stride, size_t view_stride) {
    const int tile = blockIdx.x;
    reinterpret_cast<float *>(obs + (size_t)tile * tile_stride + (size_t)k * view_stride)[threadIdx.x] = 0.0f;
    if (threadIdx.x == 0) cnt[(size_t)tile * n_views + k] = 0;
}

// rgb may be NULL (float-colour lists); ext (nullable): three planes [3][n] for the extension store of view k.
__global__ __launch_bounds__(256) void import_view_kernel(uint8_t *__restrict__ obs, int k, int tiles_x, int H, int W,
                                                          size_t tile_stride, size_t view_stride,
                                                          const int16_t *__restrict__ u1, const int16_t *__restrict__ v1,
                                                          const float *__restrict__ z, const uint8_t *__restrict__ rgb,
                                                          long long n, uint8_t *__restrict__ ext_dense, int n_views,
                                                          const float *__restrict__ ext, uint8_t *__restrict__ ext2_dense) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int u = u1[i], v = v1[i];
    if (u < 0 || v < 0 || u >= W || v >= H) return;  // check_integrity (loader.py:96-98) rejects these upstream
    const int tile = (v / kTile) * tiles_x + (u / kTile);
    const int slot = (v % kTile) * kTile + (u % kTile);
    uint8_t *chunk = obs + (size_t)tile * tile_stride + (size_t)k * view_stride;
    reinterpret_cast<float *>(chunk)[slot] = z[i];
    chunk[kChunkZ + slot] = rgb ? rgb[i * 3 + 0] : (uint8_t)0;
    chunk[kChunkZ + kTilePx + slot] = rgb ? rgb[i * 3 + 1] : (uint8_t)0;
    chunk[kChunkZ + 2 * kTilePx + slot] = rgb ? rgb[i * 3 + 2] : (uint8_t)0;
    if (ext) {
        float *e = reinterpret_cast<float *>(ext_dense + ((size_t)tile * n_views + k) * kExtChunk);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) e[pl * kTilePx + slot] = ext[(size_t)pl * n + i];
        if (ext2_dense) {   // SUCRE_EXT_POINTS_COLOUR: planes 3..5 of the list are the float32 colours
            float *e2 = reinterpret_cast<float *>(ext2_dense + ((size_t)tile * n_views + k) * kExtChunk);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) e2[pl * kTilePx + slot] = ext[(size_t)(3 + pl) * n + i];
        }
    }
}

// The extension planes of view k as dense planes (3, H, W): 0 where the view has no observation.
__global__ __launch_bounds__(256) void export_view_ext_kernel(const uint8_t *__restrict__ obs, const uint16_t *__restrict__ cnt,
                                                              const uint8_t *__restrict__ ext_dense, int n_views, int k,
                                                              int tiles_x, int H, int W, float *__restrict__ out,
                                                              size_t tile_stride, size_t view_stride) {
    const int tile = blockIdx.x, slot = threadIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int v = ty * kTile + (slot >> 4), u = tx * kTile + (slot & 15);
    if (v >= H || u >= W) return;
    float val[3] = {0.f, 0.f, 0.f};
    if (cnt[(size_t)tile * n_views + k] > 0) {
        const float z = reinterpret_cast<const float *>(obs + (size_t)tile * tile_stride + (size_t)k * view_stride)[slot];
        if (z > 0.0f) {
            const float *e = reinterpret_cast<const float *>(ext_dense + ((size_t)tile * n_views + k) * kExtChunk);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) val[pl] = e[pl * kTilePx + slot];
        }
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) out[((size_t)pl * H + v) * W + u] = val[pl];
}

// After an import: the tile's match count of view k and the view's pixel bits (what match_kernel leaves behind for the
// views it fills): thread t holds slot t, wave w's ballot is word w.
__global__ __launch_bounds__(256) void count_view_kernel(const uint8_t *__restrict__ obs, uint16_t *__restrict__ cnt,
                                                         int n_views, int k, size_t tile_stride, size_t view_stride,
                                                         uint64_t *__restrict__ vbits, uint2 *__restrict__ zrange) {
    __shared__ int part[4];
    __shared__ uint32_t plo[4], phi[4];
    const int tile = blockIdx.x, t = threadIdx.x;
    const float z = reinterpret_cast<const float *>(obs + (size_t)tile * tile_stride + (size_t)k * view_stride)[t];
    const unsigned long long bal = __ballot(z > 0.0f);
    uint32_t zlo = z > 0.0f ? __float_as_uint(z) : 0xffffffffu, zhi = z > 0.0f ? __float_as_uint(z) : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { zlo = min(zlo, (uint32_t)__shfl_xor((int)zlo, off, 64)); zhi = max(zhi, (uint32_t)__shfl_xor((int)zhi, off, 64)); }
    if ((t & 63) == 0) {
        vbits[((size_t)tile * n_views + k) * 4 + (t >> 6)] = bal;
        part[t >> 6] = __builtin_popcountll(bal);
        plo[t >> 6] = zlo; phi[t >> 6] = zhi;
    }
    __syncthreads();
    if (t == 0) {
        cnt[(size_t)tile * n_views + k] = (uint16_t)(part[0] + part[1] + part[2] + part[3]);
        zrange[(size_t)tile * n_views + k] = make_uint2(min(min(plo[0], plo[1]), min(plo[2], plo[3])), max(max(phi[0], phi[1]), max(phi[2], phi[3])));
    }
}

hipError_t launch_export_view_ext(const Layout &L, const uint8_t *ws, const uint8_t *ext_dense, int k, float *out, hipStream_t s) {
    hipLaunchKernelGGL(export_view_ext_kernel, dim3(L.n_tiles), dim3(256), 0, s, ws + L.off_obs,
                       reinterpret_cast<const uint16_t *>(ws + L.off_cnt), ext_dense, L.n_views, k, L.tiles_x, L.H, L.W, out,
                       L.obs_tile_stride, L.obs_view_stride);
    return hipGetLastError();
}

hipError_t launch_import_view(const Layout &L, uint8_t *ws, int k, const int16_t *u1, const int16_t *v1, const float *z,
                              const uint8_t *rgb, long long n, hipStream_t s, uint8_t *ext_dense, const float *ext,
                              uint8_t *ext2_dense) {
    auto *cnt = reinterpret_cast<uint16_t *>(ws + L.off_cnt);
    hipLaunchKernelGGL(clear_view_kernel, dim3(L.n_tiles), dim3(256), 0, s, ws + L.off_obs, cnt, L.n_views, k,
                       L.obs_tile_stride, L.obs_view_stride);
    if (n > 0)
        hipLaunchKernelGGL(import_view_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws + L.off_obs, k,
                           L.tiles_x, L.H, L.W, L.obs_tile_stride, L.obs_view_stride, u1, v1, z, rgb, n, ext_dense, L.n_views, ext, ext2_dense);
    hipLaunchKernelGGL(count_view_kernel, dim3(L.n_tiles), dim3(256), 0, s, ws + L.off_obs, cnt, L.n_views, k,
                       L.obs_tile_stride, L.obs_view_stride, reinterpret_cast<uint64_t *>(ws + L.off_vbits),
                       reinterpret_cast<uint2 *>(ws + L.off_zrange));
    return hipGetLastError();
}

static CamDev to_cam(const sucre_view_t &v) {
    CamDev c;
    for (int i = 0; i < 9; ++i) { c.K[i] = v.K[i]; c.Kinv[i] = v.Kinv[i]; c.R[i] = v.R[i]; c.Rinv[i] = v.Rinv[i]; }
    for (int i = 0; i < 3; ++i) { c.t[i] = v.t[i]; c.tinv[i] = v.tinv[i]; }
    c.H = v.H; c.W = v.W;
    return c;
}

// {float32 depth, r, g, b, 0} per pixel: what a packed view's `depth` pointer holds (sucre_pack_view).
// Four pixels per thread: one 16-byte load of depths, three dwords of colours (12 bytes: four pixels' r, g, b), two 16-byte
// stores.  (Until round 6 one pixel per thread with three byte loads: 9.5 us per 1080p view = 3.3 TB/s of its 15 bytes per
// pixel; a step of bench.py packs all 65 views of its image.)  kAligned: the three planes allow the wide accesses (torch's
// allocations do); the last n % 4 pixels, and everything of a view that does not, go one by one.
template <bool kAligned>
__global__ __launch_bounds__(256) void pack_view_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ rgb,
                                                        long long n, uint2 *__restrict__ out) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x, i = q * 4;
    if (i >= n) return;
    if (kAligned && i + 4 <= n) {
        const float4 d = *reinterpret_cast<const float4 *>(depth + i);
        const uint32_t *c = reinterpret_cast<const uint32_t *>(rgb + i * 3);
        const uint32_t c0 = c[0], c1 = c[1], c2 = c[2];   // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        uint4 *o = reinterpret_cast<uint4 *>(out + i);
        o[0] = make_uint4(__float_as_uint(d.x), c0 & 0xffffffu, __float_as_uint(d.y), (c0 >> 24) | ((c1 & 0xffffu) << 8));
        o[1] = make_uint4(__float_as_uint(d.z), (c1 >> 16) | ((c2 & 0xffu) << 16), __float_as_uint(d.w), c2 >> 8);
        return;
    }
    for (long long j = i; j < n && j < i + 4; ++j) {
        const uint8_t *p = rgb + j * 3;
        out[j] = make_uint2(__float_as_uint(depth[j]), (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16));
    }
}

// Sixteen views of one size per launch (blockIdx.y = the view): a target's 65 views in five launches.
constexpr int kPackBatch = 16;
struct PackBatch { const float *depth[kPackBatch]; const uint8_t *rgb[kPackBatch]; uint2 *out[kPackBatch]; };

__global__ __launch_bounds__(256) void pack_views_kernel(const PackBatch b, long long n) {
    const float *__restrict__ depth = b.depth[blockIdx.y];
    const uint8_t *__restrict__ rgb = b.rgb[blockIdx.y];
    uint2 *__restrict__ out = b.out[blockIdx.y];
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x, i = q * 4;
    if (i >= n) return;
    // (the wide accesses need the planes aligned: kernel-uniform per view)
    const bool al = reinterpret_cast<uintptr_t>(depth) % 16 == 0 && reinterpret_cast<uintptr_t>(rgb) % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
    if (al && i + 4 <= n) {
        const float4 d = *reinterpret_cast<const float4 *>(depth + i);
        const uint32_t *c = reinterpret_cast<const uint32_t *>(rgb + i * 3);
        const uint32_t c0 = c[0], c1 = c[1], c2 = c[2];
        uint4 *o = reinterpret_cast<uint4 *>(out + i);
        o[0] = make_uint4(__float_as_uint(d.x), c0 & 0xffffffu, __float_as_uint(d.y), (c0 >> 24) | ((c1 & 0xffffu) << 8));
        o[1] = make_uint4(__float_as_uint(d.z), (c1 >> 16) | ((c2 & 0xffu) << 16), __float_as_uint(d.w), c2 >> 8);
        return;
    }
    for (long long j = i; j < n && j < i + 4; ++j) {
        const uint8_t *p = rgb + j * 3;
        out[j] = make_uint2(__float_as_uint(depth[j]), (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16));
    }
}

hipError_t launch_pack_views(const float *const *depth, const uint8_t *const *rgb, void *const *packed, int n, int H, int W, hipStream_t s) {
    const long long px = (long long)H * W, threads = (px + 3) / 4;
    for (int k0 = 0; k0 < n; k0 += kPackBatch) {
        PackBatch b;
        const int m = n - k0 < kPackBatch ? n - k0 : kPackBatch;
        for (int j = 0; j < kPackBatch; ++j) {
            const int k = k0 + (j < m ? j : 0);
            b.depth[j] = depth[k]; b.rgb[j] = rgb[k]; b.out[j] = static_cast<uint2 *>(packed[k]);
        }
        hipLaunchKernelGGL(pack_views_kernel, dim3((unsigned)((threads + 255) / 256), (unsigned)m), dim3(256), 0, s, b, px);
    }
    return hipGetLastError();
}

hipError_t launch_pack_view(const float *depth, const uint8_t *rgb, int H, int W, void *packed, hipStream_t s) {
    const long long n = (long long)H * W, threads = (n + 3) / 4;
    const bool aligned = reinterpret_cast<uintptr_t>(depth) % 16 == 0 && reinterpret_cast<uintptr_t>(rgb) % 4 == 0 && reinterpret_cast<uintptr_t>(packed) % 16 == 0;
    if (aligned) hipLaunchKernelGGL(pack_view_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, depth, rgb, n, static_cast<uint2 *>(packed));
    else hipLaunchKernelGGL(pack_view_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, depth, rgb, n, static_cast<uint2 *>(packed));
    return hipGetLastError();
}

hipError_t launch_project_points(const sucre_view_t &view, const float *wP, long long n, int32_t *pix, hipStream_t s) {
    if (n > 0)
        hipLaunchKernelGGL(project_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, to_cam(view), wP, n, pix);
    return hipGetLastError();
}

hipError_t launch_match(const Layout &L, uint8_t *ws, const sucre_view_t &target, const sucre_view_t *views_dev,
                        int k0, int k1, hipStream_t s, uint8_t *ext, int ext_mode, uint8_t *ext2) {
    const int tiles_per_xcd = (L.n_tiles + 7) / 8;
    const dim3 grid(8 * tiles_per_xcd, (k1 - k0 + kViewsPerGroup - 1) / kViewsPerGroup);
    auto *cnt = reinterpret_cast<uint16_t *>(ws + L.off_cnt);
    auto *vbits = reinterpret_cast<uint64_t *>(ws + L.off_vbits);
    auto launch = [&](auto kernel, int mode) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, s, ws + L.off_obs, cnt, target.depth, to_cam(target), views_dev, L.n_views, k0,
                           k1, L.tiles_x, L.n_tiles, tiles_per_xcd, L.obs_tile_stride, L.obs_view_stride, ext, mode, ext2, vbits,
                           reinterpret_cast<uint2 *>(ws + L.off_zrange));
    };
    if (ext2) launch(match_kernel<true, true>, SUCRE_EXT_POINTS_COLOUR);
    else if (ext) launch(match_kernel<false, true>, ext_mode);
    else launch(match_kernel<false, false>, ext_mode);
    return hipGetLastError();
}

hipError_t launch_match_map(const Layout &L, const sucre_view_t &target, const sucre_view_t *views_dev, int k,
                            int32_t *map, hipStream_t s) {
    hipLaunchKernelGGL(match_map_kernel, dim3((L.H * L.W + 255) / 256), dim3(256), 0, s, target.depth, to_cam(target),
                       views_dev, k, map);
    return hipGetLastError();
}

hipError_t launch_finalize(const Layout &L, uint8_t *ws, double min_cover, hipStream_t s, const uint8_t *ext_dense,
                           uint8_t *ext_comp, int fmt, const uint8_t *ext2_dense, uint8_t *ext2_comp) {
    auto *cnt = reinterpret_cast<const uint16_t *>(ws + L.off_cnt);
    auto *vc = reinterpret_cast<uint64_t *>(ws + L.off_view_count);
    auto *vk = reinterpret_cast<uint32_t *>(ws + L.off_view_keep);
    auto *partial = reinterpret_cast<uint32_t *>(ws + L.off_view_partial);
    const int rows = (L.n_tiles + kStatTiles - 1) / kStatTiles;
    auto *zpart = reinterpret_cast<uint2 *>(ws + L.off_zpart);
    hipLaunchKernelGGL(view_partial_kernel, dim3(rows), dim3(256), 0, s, cnt, L.n_tiles, L.n_views, partial,
                       reinterpret_cast<const uint2 *>(ws + L.off_zrange), zpart);
    hipLaunchKernelGGL(view_total_kernel, dim3(1), dim3(1024), 0, s, partial, rows, L.n_views, min_cover,
                       (double)L.W * (double)L.H, vc, vk, reinterpret_cast<uint64_t *>(ws + L.off_n_obs),
                       reinterpret_cast<uint64_t *>(ws + L.off_n_obs_total), zpart,
                       reinterpret_cast<uint32_t *>(ws + L.off_total_chunks) + 4);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    return launch_compact(L, ws, s, ext_dense, ext_comp, fmt, ext2_dense, ext2_comp);
}

hipError_t launch_check_store(const Layout &L, const uint8_t *ws, uint32_t *verdict, uint64_t *scratch, hipStream_t s) {
    if (hipError_t e = hipMemsetAsync(verdict, 0, sizeof(uint32_t) * L.n_views, s); e != hipSuccess) return e;
    if (hipError_t e = hipMemsetAsync(scratch, 0, sizeof(uint64_t) * L.n_views, s); e != hipSuccess) return e;
    int per_view = 4096 / L.n_views;   // ~4096 workgroups in all
    per_view = per_view < 1 ? 1 : (per_view > L.n_tiles ? L.n_tiles : per_view);
    hipLaunchKernelGGL(integrity_scan_kernel, dim3(L.n_views, per_view), dim3(256), 0, s, ws + L.off_obs,
                       reinterpret_cast<const uint16_t *>(ws + L.off_cnt), L.n_views, L.n_tiles, L.obs_tile_stride,
                       L.obs_view_stride, verdict, reinterpret_cast<unsigned long long *>(scratch));
    hipLaunchKernelGGL(integrity_verdict_kernel, dim3((L.n_views + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const unsigned long long *>(scratch),
                       reinterpret_cast<const uint64_t *>(ws + L.off_view_count), L.n_views, verdict);
    return hipGetLastError();
}

hipError_t launch_export_view(const Layout &L, const uint8_t *ws, int k, float *z, uint8_t *rgb, hipStream_t s) {
    hipLaunchKernelGGL(export_view_kernel, dim3(L.n_tiles), dim3(256), 0, s, ws + L.off_obs,
                       reinterpret_cast<const uint16_t *>(ws + L.off_cnt), L.n_views, k, L.tiles_x, L.H, L.W, z, rgb,
                       L.obs_tile_stride, L.obs_view_stride);
    return hipGetLastError();
}

}  // namespace sucre
