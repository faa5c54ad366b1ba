// Artificial-light model (--light-model) for gfx950: sucre.py:54-61 with se3.exp (se3.py:22-27).
//
//   R, t = se3.exp(cam2light);  Sigma = sigma^T sigma;  lP = R cP + t;  lp = lP.xy / lP.z
//   l = exp(-(lp^T Sigma^-1 lp) / 2);   z = ||cP|| + ||lP||;   Ihat = l (J a + B (1 - g))
//
// The reference differentiates this with autograd; here the gradient is analytic (derivation in DESIGN.md).
// The model needs the camera-frame point cP of every observation, which
// the 7-byte store of the default path does not keep, so this mode works on an EXTENSION workspace `lws`:
// three float planes (cP.x, cP.y, cP.z) per chunk -- written by match_kernel, carried through the compaction --
// plus the 19 parameters (B, beta, gamma, cam2light[6], sigma[4]), their Adam moments, and the reduction buffers.
// Everything of the default path (J planes, compact store, perm, levels) is reused from the main workspace.
//
// One iteration = light_grad_kernel (persistent waves on strips like fit_grad_kernel; 26 sums) -> light_tail_kernel (one
// workgroup: fixed-order float64 reduction, chain rule through Sigma^-1 and through the matrix exponential -- six 8x8
// block exponentials evaluated in LDS -- Adam on the 19 parameters, next R, t, Sigma^-1, log row).
// Plain loads instead of the LDS-DMA ring: the kernel is instruction-limited (~135 instructions per observation).
#include <type_traits>
#include <mutex>
#include "fit_math.h"
#include "handoff.h"

namespace sucre {

// sB[3] sGZ[3] sBeta[3] cost | torque[3] = sum lP x dlP, force[3] = sum dlP | dM00, dM01, dM11.  (Until round 5: 26 sums, the
// 3x3 sum dlP cP^T and sum dlP in their place.  With T = exp(hat xi) and D exp(hat xi)[G_i] = hat(v_i) T, v_i = (omega_i, u_i):
// <sum dlP cP^T, D_R> + <sum dlP, D_t> = sum dlP . (omega_i x lP + u_i) = omega_i . torque + u_i . force -- six sums and six FMAs
// per observation instead of twelve and twelve; light_geometry leaves the six twists v_i behind instead of the derivatives.)
constexpr int kLightSums = 19;
constexpr int kLightParams = 19;

struct LightLayout {
    size_t off_ext_dense, off_ext_comp;   // dense: float [chunk][3][256]; compact: kExtLevelBytes per level of a strip
    size_t off_params;                    // float [19] params, [19] exp_avg, [19] exp_avg_sq
    size_t off_geom;                      // float [16]: R[9], t[3], M[4] = Sigma^-1 (row-major)
    size_t off_dexp;                      // double [6][12]: entries 0..5 of row i = the twist v_i = (omega_i, u_i) with D exp(hat xi)[G_i] = hat(v_i) exp(hat xi) (with the geometry)
    size_t off_partials;                  // float [26][n_blocks]
    size_t off_sums;                      // double [26]
    size_t off_deal;                      // uint32 [8192] strips per wave, then [8 n_strips] the waves' strip lists (light_deal_kernel)
    size_t off_ext2_dense, off_ext2_comp; // second set of planes (float32 colours next to camera points); only with ext_sets = 2
    size_t total;
};
constexpr uint32_t kLightMaxWaves = 8192;   // 8 waves per SIMD x 4 SIMDs x 256 CUs

// ext_sets = 2 appends the second set at the END, so every other offset is the same in both kinds of workspace
static bool make_light_layout(const Layout &L, LightLayout *X, int ext_sets = 1) {
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    const size_t chunks = (size_t)L.n_tiles * L.n_views;
    X->off_ext_dense = take(chunks * kExtChunk);
    X->off_ext_comp = take(chunks * kExtChunk);
    X->off_params = take(3 * kLightParams * sizeof(float));
    X->off_geom = take(16 * sizeof(float));
    X->off_dexp = take(72 * sizeof(double));
    X->off_partials = take((size_t)kLightSums * L.n_blocks * sizeof(float));
    X->off_sums = take(kLightSums * sizeof(double));
    X->off_deal = take(((size_t)kLightMaxWaves + (size_t)kMaxGen * L.n_strips + kLightMaxWaves) * sizeof(uint32_t));
    X->off_ext2_dense = X->off_ext2_comp = 0;
    if (ext_sets > 1) {
        X->off_ext2_dense = take(chunks * kExtChunk);
        X->off_ext2_comp = take(chunks * kExtChunk);
    }
    X->total = o;
    return true;
}

// ---- geometry from the parameters: exp(hat xi) and the twists of its derivative, in closed form ---------------------------
// xi = (omega, p) = cam2light (se3.py:22-27: the 4x4 twist [[hat omega, p], [0, 0]]), theta = |omega|, W = hat omega, P = hat p:
//   R = I + A W + B W^2,   J = I + B W + C W^2 (the left Jacobian of SO(3)),   t = J p
//   D exp(hat xi)[G_i] = hat(v_i) exp(hat xi),   v_i = (omega_i, u_i):   rotation generators i < 3:  omega_i = J e_i, u_i = Q e_i;
//   translation generators: omega_i = 0, u_i = J e_(i-3);   Q = P/2 + C (W P + P W + W P W) + D (W^2 P + P W^2 - 3 W P W)
//   + E (W P W^2 + W^2 P W)   [Barfoot, State Estimation for Robotics, the left Jacobian of SE(3)]
//   A = sin th / th, B = (1 - cos th) / th^2, C = (th - sin th) / th^3, D = (th^2 + 2 cos th - 2) / (2 th^4),
//   E = (2 th - 3 sin th + th cos th) / (2 th^5): their power series below th^2 = 1/4 (cam2light starts at zero), float64.
// Checked against the block exponentials exp([[hat xi, G_i], [0, hat xi]]) this replaced: 1e-14 (tools/exp/se3_closed_check.py).
// (Until round 5 six 8x8 exponentials -- scaling, degree-12 Taylor, six squarings, 40 barriers -- ran in LDS in every
// iteration's tail: 25 of light_tail_kernel's 40 us.)
struct Mat3 { double m[9]; };
__device__ __forceinline__ Mat3 mul3(const Mat3 &a, const Mat3 &b) {
    Mat3 c;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) c.m[r * 3 + k] = a.m[r * 3] * b.m[k] + a.m[r * 3 + 1] * b.m[3 + k] + a.m[r * 3 + 2] * b.m[6 + k];
    return c;
}
__device__ __forceinline__ Mat3 hat3(double x, double y, double z) { return Mat3{{0.0, -z, y, z, 0.0, -x, -y, x, 0.0}}; }

__device__ __forceinline__ void se3_coefficients(double th2, double &A, double &B, double &C, double &D, double &E) {
    if (th2 < 0.25) {
        const double x = th2;
        A = 1.0 - x / 6.0 * (1.0 - x / 20.0 * (1.0 - x / 42.0 * (1.0 - x / 72.0 * (1.0 - x / 110.0 * (1.0 - x / 156.0)))));
        B = 0.5 * (1.0 - x / 12.0 * (1.0 - x / 30.0 * (1.0 - x / 56.0 * (1.0 - x / 90.0 * (1.0 - x / 132.0 * (1.0 - x / 182.0))))));
        C = (1.0 / 6.0) * (1.0 - x / 20.0 * (1.0 - x / 42.0 * (1.0 - x / 72.0 * (1.0 - x / 110.0 * (1.0 - x / 156.0 * (1.0 - x / 210.0))))));
        // D = sum_(n >= 2) (-1)^n x^(n-2) / (2n)!,  E = sum_(n >= 2) (-1)^n x^(n-2) (1/(2n)! - 3/(2n+1)!) / 2
        double d = 0.0, e = 0.0, xp = 1.0, f = 24.0;   // f = (2n)!, starting at n = 2
        for (int n = 2; n < 10; ++n) {
            const double sgn = (n & 1) ? -1.0 : 1.0;
            d += sgn * xp / f;
            e += sgn * xp * (1.0 / f - 3.0 / (f * (2 * n + 1)));
            xp *= x;
            f *= (double)((2 * n + 1) * (2 * n + 2));
        }
        D = d; E = 0.5 * e;
        return;
    }
    const double th = sqrt(th2), sn = sin(th), cs = cos(th);
    A = sn / th; B = (1.0 - cs) / th2; C = (th - sn) / (th2 * th);
    D = (th2 + 2.0 * cs - 2.0) / (2.0 * th2 * th2); E = (2.0 * th - 3.0 * sn + th * cs) / (2.0 * th2 * th2 * th);
}

// geom (R, t, M) and the six twists from the parameters (the twists are what the NEXT step's chain rule needs: the step that
// follows a gradient pass differentiates at the parameters the pass ran with, i.e. the ones this call sees).  Thread 0 does the
// pose, thread 1 M; all threads of the workgroup must call it.
__device__ __forceinline__ void light_geometry(const float *params, float *geom, double *dexp) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        const double w0 = params[9], w1 = params[10], w2 = params[11], p0 = params[12], p1 = params[13], p2 = params[14];
        double A, B, C, D, E;
        se3_coefficients(w0 * w0 + w1 * w1 + w2 * w2, A, B, C, D, E);
        const Mat3 W = hat3(w0, w1, w2), P = hat3(p0, p1, p2), W2 = mul3(W, W);
        const Mat3 WP = mul3(W, P), PW = mul3(P, W), WPW = mul3(WP, W), W2P = mul3(W2, P), PW2 = mul3(P, W2), WPW2 = mul3(WPW, W), W2PW = mul3(W2P, W);
        double J[9], Q[9];
        for (int i = 0; i < 9; ++i) {
            const double id = (i % 4 == 0) ? 1.0 : 0.0;
            geom[i] = (float)(id + A * W.m[i] + B * W2.m[i]);
            J[i] = id + B * W.m[i] + C * W2.m[i];
            Q[i] = 0.5 * P.m[i] + C * (WP.m[i] + PW.m[i] + WPW.m[i]) + D * (W2P.m[i] + PW2.m[i] - 3.0 * WPW.m[i]) + E * (WPW2.m[i] + W2PW.m[i]);
        }
        for (int r = 0; r < 3; ++r) geom[9 + r] = (float)(J[r * 3] * p0 + J[r * 3 + 1] * p1 + J[r * 3 + 2] * p2);
        for (int i = 0; i < 6; ++i) {
            double *v = dexp + i * 12;
            for (int r = 0; r < 3; ++r) {
                v[r] = i < 3 ? J[r * 3 + i] : 0.0;
                v[3 + r] = i < 3 ? Q[r * 3 + i] : J[r * 3 + (i - 3)];
            }
            for (int k = 6; k < 12; ++k) v[k] = 0.0;
        }
    }
    if (tid == (blockDim.x > 64 ? 64 : 1)) {  // M = (sigma^T sigma)^-1, float32 like the reference's Sigma.inverse() (another wave, when there is one)
        const float *sg = params + 15;
        const float S00 = sg[0] * sg[0] + sg[2] * sg[2], S01 = sg[0] * sg[1] + sg[2] * sg[3];
        const float S11 = sg[1] * sg[1] + sg[3] * sg[3];
        const float det = S00 * S11 - S01 * S01;
        geom[12] = S11 / det; geom[13] = -S01 / det; geom[14] = -S01 / det; geom[15] = S00 / det;
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void light_init_kernel(float *pstate, float *geom, double *dexp, const float *p0) {
    if (threadIdx.x < 3 * kLightParams) pstate[threadIdx.x] = threadIdx.x < kLightParams ? p0[threadIdx.x] : 0.f;
    __syncthreads();
    light_geometry(pstate, geom, dexp);
}

// geometry of whatever parameters are stored now (the caller may have written them since the last step)
__global__ __launch_bounds__(64) void light_geometry_kernel(const float *pstate, float *geom, double *dexp) {
    light_geometry(pstate, geom, dexp);
}

// ---- gradient pass ---------------------------------------------------------------------------------------------------
struct LightAcc {
    float s[kLightSums];           // lane-level global sums (slots 6..8 = sBeta are filled at the end of every strip)
};

// l, total range z and the light-frame quantities of one observation (sucre.py:55-63)
struct LightObs { float l, z, nl, inl, iz, lp0, lp1, w0, w1, q, lP[3]; };   // q = lp^T M lp = lp . (w0, w1)

template <bool kGradual = false>
__device__ __forceinline__ LightObs light_obs(const float cP[3], float zc, const float (&R)[9], const float (&tl)[3],
                                              const float (&M)[4]) {
    LightObs o;
#pragma unroll
    for (int a = 0; a < 3; ++a)   // (every a*b+c of this kernel is an explicit FMA: the library is built with -ffp-contract=off)
        o.lP[a] = __builtin_fmaf(R[a * 3 + 2], cP[2], __builtin_fmaf(R[a * 3 + 1], cP[1], __builtin_fmaf(R[a * 3], cP[0], tl[a])));
    // hardware reciprocal / reciprocal square root (1 ulp): this path is held to a tolerance, not to bit parity,
    // and the IEEE sequences were a quarter of the kernel's instructions
    o.iz = __builtin_amdgcn_rcpf(o.lP[2]);
    o.lp0 = o.lP[0] * o.iz;
    o.lp1 = o.lP[1] * o.iz;
    o.w0 = __builtin_fmaf(M[0], o.lp0, M[1] * o.lp1);   // M = (sigma^T sigma)^-1 is symmetric: M[1] == M[2]
    o.w1 = __builtin_fmaf(M[2], o.lp0, M[3] * o.lp1);
    o.q = __builtin_fmaf(o.lp0, o.w0, o.lp1 * o.w1);
    o.l = exp2_as<kGradual>(o.q * (-0.5f * kLog2e));
    const float n2 = __builtin_fmaf(o.lP[2], o.lP[2], __builtin_fmaf(o.lP[1], o.lP[1], o.lP[0] * o.lP[0]));
    o.inl = __builtin_amdgcn_rsqf(n2);
    o.nl = n2 * o.inl;
    o.z = zc + o.nl;
    return o;
}

struct LightChunk { float zz[4], xx[4], yy[4], ww[4]; uint32_t cc[3]; };
struct ColourChunk { float c[3][4]; };   // float32 colours of the same four levels (second extension set)

// The three planes of chunk g of a strip's SECOND extension set -> this lane's pixel (same arrangement as the first).
__device__ __forceinline__ ColourChunk load_colour_chunk(const uint8_t *sext2, uint32_t g, uint32_t r, int lane) {
    const float *ex = reinterpret_cast<const float *>(sext2 + (size_t)g * (kGroupLv * kExtLevelBytes));
    ColourChunk q;
    if (r == kGroupLv) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const float4 v = *reinterpret_cast<const float4 *>(ex + pl * kStripPx * kGroupLv + lane * 4);
            q.c[pl][0] = v.x; q.c[pl][1] = v.y; q.c[pl][2] = v.z; q.c[pl][3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < kGroupLv; ++j) {
            const bool has = (uint32_t)j < r;
            const uint32_t i = lane * r + (has ? j : 0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) q.c[pl][j] = has ? ex[pl * kStripPx * r + i] : 0.0f;
        }
    }
    return q;
}

// Chunk g of a strip (r of its four levels exist, layout.h / compact.hip) -> this lane's pixel: ranges, extension
// planes (camera point or float colour) and colour bytes of up to four levels; levels that do not exist read z = 0.
// kLastUse: this is the launch's last read of the chunk -> non-temporal loads, like the default path's DMAs (measured,
// J as a parameter: 356 -> 341 us per iteration at config 2).  The closed-form kernel reads every chunk twice, a strip
// apart: its first read stays a plain load (with both reads non-temporal it was 4 % slower: what the caches still hold
// of the strip serves the second read).
// kNoZ (round 6, the J-parameter light kernel): the chunk's ranges are NOT loaded -- 16 of a lane's 76 bytes per chunk; the kernel
// forms ||cP|| from the camera point it loads anyway (zz then carries the point's z coordinate, whose being non-zero marks a real
// observation: it is the depth, and an empty slot's planes are zeros).
template <bool kLastUse = true, bool kNoZ = false>
__device__ __forceinline__ LightChunk load_light_chunk(const uint8_t *sobs, const uint8_t *sext, uint32_t g, uint32_t r, int lane) {
    const uint8_t *ch = sobs + (size_t)g * (kGroupLv * level_bytes(0));
    const float *ex = reinterpret_cast<const float *>(sext + (size_t)g * (kGroupLv * kExtLevelBytes));
    LightChunk c;
    if (kExpNoLoad) {  // ablation build only (experiment.h): no memory instruction, plausible values
        const float f = 1.0f + 0.001f * (float)lane;
        c = LightChunk{{3.0f * f, 3.1f * f, 3.2f * f, r == kGroupLv ? 3.3f * f : 0.0f}, {0.1f * f, 0.2f, -0.1f, 0.3f},
                       {0.2f, -0.1f * f, 0.1f, 0.2f}, {3.0f, 3.1f, 3.2f * f, 3.3f}, {0x40506070u, 0x30405060u + lane, 0x20304050u}};
        return c;
    }
    if (r == kGroupLv) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u3 __attribute__((ext_vector_type(3)));
        auto ld4 = [](const void *p) { return kLastUse ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p)) : *reinterpret_cast<const f4 *>(p); };
        f4 z4;
        if (!kNoZ) z4 = ld4(ch + lane * 16);
        const u3 cw = kLastUse ? __builtin_nontemporal_load(reinterpret_cast<const u3 *>(ch + 4 * kStripPx * kGroupLv + 12 * lane))
                               : *reinterpret_cast<const u3 *>(ch + 4 * kStripPx * kGroupLv + 12 * lane);
        const f4 x4 = ld4(ex + lane * 4);
        const f4 y4 = ld4(ex + kStripPx * kGroupLv + lane * 4);
        const f4 w4 = ld4(ex + 2 * kStripPx * kGroupLv + lane * 4);
        if (kNoZ) z4 = w4;
        c = LightChunk{{z4.x, z4.y, z4.z, z4.w}, {x4.x, x4.y, x4.z, x4.w}, {y4.x, y4.y, y4.z, y4.w}, {w4.x, w4.y, w4.z, w4.w},
                       {cw.x, cw.y, cw.z}};
    } else {
        c.cc[0] = c.cc[1] = c.cc[2] = 0u;
        const uint8_t *cb = ch + 4 * kStripPx * r;
#pragma unroll
        for (int j = 0; j < kGroupLv; ++j) {
            const bool has = (uint32_t)j < r;
            const uint32_t i = lane * r + (has ? j : 0);
            c.xx[j] = has ? ex[i] : 0.0f;
            c.yy[j] = has ? ex[kStripPx * r + i] : 0.0f;
            c.ww[j] = has ? ex[2 * kStripPx * r + i] : 0.0f;
            c.zz[j] = kNoZ ? c.ww[j] : has ? reinterpret_cast<const float *>(ch)[i] : 0.0f;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) c.cc[pl] |= (has ? (uint32_t)cb[pl * kStripPx * r + i] : 0u) << (8 * j);
        }
    }
    return c;
}

// every loaded word of a chunk enters one float (the arithmetic-free ablation build's only use of the data)
__device__ __forceinline__ float chunk_checksum(const LightChunk &k) {
    float s = (float)(k.cc[0] ^ k.cc[1] ^ k.cc[2]);
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (k.zz[j] + k.xx[j]) + (k.yy[j] + k.ww[j]);
    return s;
}

// The 26 sums over the workgroups' partials, float64, fixed order (thread t adds the workgroups t, t + 256, ..., then a
// fixed-shape shuffle tree, then the four waves in order); all loads are issued before any is waited for.  256 threads.
__device__ __forceinline__ void light_reduce(const float *partials, int n_blocks, double *sums, double (*w4)[4]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double x[kLightSums];
#pragma unroll
    for (int q = 0; q < kLightSums; ++q) x[q] = 0.0;
    for (int b = t; b < n_blocks; b += 256) {
#pragma unroll
        for (int q = 0; q < kLightSums; ++q)
            x[q] += (double)partials[(size_t)q * n_blocks + b];
    }
#pragma unroll
    for (int q = 0; q < kLightSums; ++q) {
        const double y = wave_sum_lane0(x[q]);
        if (lane == 0) w4[q][wave] = y;
    }
    __syncthreads();
    if (t < kLightSums) sums[t] = ((w4[t][0] + w4[t][1]) + w4[t][2]) + w4[t][3];
    __syncthreads();
}

// Gradients of the 19 parameters from the sums (LDS), Adam, next geometry, log row.  Any multiple of 64 threads >= 64.
struct LightLds {
    double grad[kLightParams];
    double sums[kLightSums];
    double w4[kLightSums][4];
};

__device__ __forceinline__ void light_step(LightLds &lds, float *pstate, float *geom, double *dexp,
                                           const uint64_t *__restrict__ n_obs_total, const AdamCoef &co,
                                           double *__restrict__ trace_row) {
    const double *sums = lds.sums;
    double *grad = lds.grad;
    const int tid = threadIdx.x;
    const double f = -2.0 * (double)((1.0f / 3.0f) / (float)(*n_obs_total));   // dL/dIhat = f * r
    if (tid < 6) {  // cam2light: omega_i . torque + u_i . force, the twist left behind with the geometry
        double s = 0.0;
        for (int r = 0; r < 3; ++r) s += sums[10 + r] * dexp[tid * 12 + r] + sums[13 + r] * dexp[tid * 12 + 3 + r];
        grad[9 + tid] = f * s;
    }
    if (tid == 6) {  // water parameters (same combinations as water_step of the default path)
        for (int c = 0; c < 3; ++c) {
            grad[c] = f * sums[c];
            grad[3 + c] = -f * sums[6 + c];
            grad[6 + c] = f * (double)pstate[c] * sums[3 + c];
        }
    }
    if (tid == 7) {  // sigma: M = Sigma^-1, Sigma = sigma^T sigma;  dSigma = -M^T dM M^T;  dsigma = sigma (dSigma + dSigma^T)
        const double M[4] = {geom[12], geom[13], geom[14], geom[15]};
        const double dM[4] = {f * sums[16], f * sums[17], f * sums[17], f * sums[18]};  // (lp lp^T is symmetric)
        const double Mt[4] = {M[0], M[2], M[1], M[3]};
        double T1[4], dS[4];
        T1[0] = Mt[0] * dM[0] + Mt[1] * dM[2]; T1[1] = Mt[0] * dM[1] + Mt[1] * dM[3];
        T1[2] = Mt[2] * dM[0] + Mt[3] * dM[2]; T1[3] = Mt[2] * dM[1] + Mt[3] * dM[3];
        dS[0] = -(T1[0] * Mt[0] + T1[1] * Mt[2]); dS[1] = -(T1[0] * Mt[1] + T1[1] * Mt[3]);
        dS[2] = -(T1[2] * Mt[0] + T1[3] * Mt[2]); dS[3] = -(T1[2] * Mt[1] + T1[3] * Mt[3]);
        const double sym[4] = {2 * dS[0], dS[1] + dS[2], dS[1] + dS[2], 2 * dS[3]};
        const double sg[4] = {pstate[15], pstate[16], pstate[17], pstate[18]};
        grad[15] = sg[0] * sym[0] + sg[1] * sym[2]; grad[16] = sg[0] * sym[1] + sg[1] * sym[3];
        grad[17] = sg[2] * sym[0] + sg[3] * sym[2]; grad[18] = sg[2] * sym[1] + sg[3] * sym[3];
    }
    __syncthreads();
    if (tid < kLightParams) {
        float p = pstate[tid], m = pstate[kLightParams + tid], v = pstate[2 * kLightParams + tid];
        adam_update(p, m, v, (float)grad[tid], co);
        pstate[tid] = p; pstate[kLightParams + tid] = m; pstate[2 * kLightParams + tid] = v;
        if (trace_row) trace_row[1 + tid] = (double)p;
    }
    if (tid == 32 && trace_row) trace_row[0] = sums[9];
    __syncthreads();   // (workgroup scope is enough for the read-back below; an agent-scope fence here wrote the L2 back: +50 us)
    light_geometry(pstate, geom, dexp);
}

// kClosed: J is re-solved in closed form at the top of the iteration (sucre.py:141, 66-77 with absorption = l a,
// backscatter = l B (1 - g)) and is a constant of the gradient; kJOnly: only that closed-form J (final update_J).
// kColour (SUCRE_EXT_COLOUR): the extension planes carry the observation's float32 colour instead of its camera
// point; there is no light then (l = 1, z = the stored range) and the light sums stay zero.
// kBoth (SUCRE_EXT_POINTS_COLOUR): the light model with float32 colours -- camera points in `ext`, colours in `ext2`.
// Every wave works alone on strips of 64 sorted pixels, one pixel per lane (the deal of fit.hip).
// (Built and not kept, round 6: five workgroups per CU for the J-parameter instantiation -- 96 registers instead of 108 -- spill, 48
// bytes per lane with the two register sets of the prefetch, 76 with one; chunks without an empty slot run without the per-level
// validity tests: the scheduler then interleaves the four levels, 130 registers, three workgroups per CU, or 8 bytes of scratch at 128;
// two chunks loaded ahead in three register sets (123 registers): 302 against 298 us -- it does not wait for latency.)
template <bool kClosed, bool kJOnly, bool kColour, bool kBoth>
__global__ __launch_bounds__(256) void light_grad_kernel(const uint8_t *__restrict__ comp, const uint8_t *__restrict__ ext,
                                                         const uint8_t *__restrict__ ext2,
                                                         const StripMeta *__restrict__ meta, int n_strips,
                                                         const float *__restrict__ pstate, const float *__restrict__ geom,
                                                         const uint64_t *__restrict__ n_obs_total,
                                                         float *__restrict__ state, float *__restrict__ partials,
                                                         const AdamCoef co, const uint32_t *__restrict__ deal_count,
                                                         const uint32_t *__restrict__ deal_strips, uint32_t deal_kmax) {
    __shared__ float wsum[4][kLightSums];
    constexpr bool kPingPong = !kColour && !kBoth;   // two named register sets for the chunk prefetch (the light model on uint8 colours)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n_blocks = gridDim.x;
    float B[3], nb[3], ng[3], beta[3], gamma[3], R[9], tl[3], M[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        B[c] = pstate[c]; beta[c] = pstate[3 + c]; gamma[c] = pstate[6 + c];
        nb[c] = -beta[c] * kLog2e; ng[c] = -gamma[c] * kLog2e;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = geom[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) tl[i] = geom[9 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) M[i] = geom[12 + i];
    const float gscale = -2.0f * ((1.0f / 3.0f) / (float)(*n_obs_total));
    const float gB[3] = {gamma[0] * B[0], gamma[1] * B[1], gamma[2] * B[2]};
    constexpr float kInv255L = (float)(1.0 / 255.0);

    LightAcc acc;
#pragma unroll
    for (int q = 0; q < kLightSums; ++q) acc.s[q] = 0.f;

    // The wave's strips: from the deal light_deal_kernel wrote for THIS grid (layout.h: on a full grid the waves of the
    // workgroups that arrive first on their CU get larger shares -- the SIMD issues its oldest wave first), or the plain
    // boustrophedon deal when there is no table (update_J).
    const uint32_t W = (uint32_t)n_blocks * 4u, wid = blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t n_mine = deal_count ? __builtin_amdgcn_readfirstlane(deal_count[wid]) : 0xffffffffu;
    for (uint32_t k = 0; k < n_mine; ++k) {
        uint32_t strip;
        if (deal_count) {
            strip = __builtin_amdgcn_readfirstlane(deal_strips[(size_t)wid * deal_kmax + k]);
        } else {
            strip = k * W + ((k & 1u) ? W - 1u - wid : wid);
            if (k * W >= (uint32_t)n_strips) break;
            if (strip >= (uint32_t)n_strips) continue;
        }
        const StripMeta sm = meta[strip];
        const uint32_t n = kExpLightVectorBases ? sm.levels : __builtin_amdgcn_readfirstlane(sm.levels), nch = (n + 3u) >> 2;
        // wave-uniform bases, held in scalar registers: the loads below are base + 32-bit lane offset (with per-lane 64-bit
        // pointers the address arithmetic was 7 vector instructions per observation)
        const uint64_t lvoff = kExpLightVectorBases ? sm.lvoff : ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(sm.lvoff >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)sm.lvoff);
        const uint8_t *sobs = comp + lvoff * (uint64_t)level_bytes(0);
        const uint8_t *sext = ext + lvoff * (uint64_t)kExtLevelBytes;
        const uint8_t *sext2 = kBoth ? ext2 + lvoff * (uint64_t)kExtLevelBytes : nullptr;
        float *st = state + (size_t)strip * kStateFloats + lane;
        float J[3];
        if (kClosed) {
            // closed-form J of this pixel: numerator / denominator per channel over all levels
            float num[3], den[3];
            auto solve = [&](auto gradual) {
                constexpr bool kGradual = decltype(gradual)::value;
#pragma unroll
                for (int c = 0; c < 3; ++c) num[c] = den[c] = 0.f;
                auto load1 = [&](uint32_t g) { return load_light_chunk<kJOnly>(sobs, sext, g, min((uint32_t)kGroupLv, n - g * kGroupLv), lane); };
                auto solve_chunk = [&](const LightChunk &kk, uint32_t g) {
                    ColourChunk fc;
                    if (kBoth) fc = load_colour_chunk(sext2, g, min((uint32_t)kGroupLv, n - g * kGroupLv), lane);
                    if (kExpNoCompute) {  // ablation build only: touch the data, skip the model
                        num[0] += chunk_checksum(kk);
                        den[0] = den[1] = den[2] = 1.0f;
                        return;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!(kk.zz[j] > 0.0f)) continue;
                        const float cP[3] = {kk.xx[j], kk.yy[j], kk.ww[j]};
                        LightObs o;
                        if (kColour) { o.l = 1.0f; o.z = kk.zz[j]; }
                        else o = light_obs<kGradual>(cP, kk.zz[j], R, tl, M);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float a = o.l * exp2_as<kGradual>(o.z * nb[c]);
                            const float b = o.l * B[c] * (1.0f - exp2_as<kGradual>(o.z * ng[c]));
                            // uint8 colours: I = k/255 folded into I - b (one rounding instead of two, two instructions fewer), as
                            // fit.hip's closed_terms does; float32 colours are what they are
                            const float y = kBoth ? fc.c[c][j] - b : kColour ? cP[c] - b
                                                  : kExpLightFold1 ? __builtin_fmaf((float)((kk.cc[c] >> (8 * j)) & 255u), kInv255L, -b)
                                                                   : unit_from_u8((kk.cc[c] >> (8 * j)) & 255u) - b;
                            num[c] = __builtin_fmaf(y, a, num[c]);
                            den[c] = __builtin_fmaf(a, a, den[c]);
                        }
                    }
                };
                // the next chunk's loads before this chunk's arithmetic, two named register sets (as in the gradient pass below;
                // until round 6 this pass loaded, waited, computed)
                if constexpr (kPingPong) {
                    LightChunk ka = nch ? load1(0) : LightChunk{}, kb2 = LightChunk{};
                    for (uint32_t g = 0; g < nch; g += 2u) {
                        if (g + 1u < nch) kb2 = load1(g + 1u);
                        solve_chunk(ka, g);
                        if (g + 1u < nch) {
                            if (g + 2u < nch) ka = load1(g + 2u);
                            solve_chunk(kb2, g + 1u);
                        }
                    }
                } else {
                    for (uint32_t g = 0; g < nch; ++g) solve_chunk(load1(g), g);
                }
            };
            solve(std::false_type{});
            // an observed pixel whose denominator is zero: v_exp_f32's flushed results decide between +-inf and NaN here
            // (fit_math.h) -- the strip is solved again with gradual underflow (same bits for every other pixel)
            const bool zero = den[0] == 0.f || den[1] == 0.f || den[2] == 0.f;
            if (n > 0 && __any(zero)) {   // rare; an unobserved pixel (the strip where the counts reach 0) has no level 0 either
                const LightChunk k0 = load_light_chunk<false>(sobs, sext, 0, min((uint32_t)kGroupLv, n), lane);
                if (__any(zero && k0.zz[0] > 0.0f)) solve(std::true_type{});
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                J[c] = num[c] / den[c];  // 0/0 = NaN where nothing was observed
                st[c * kStripPx] = J[c];
            }
            if (kJOnly) continue;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) J[c] = st[c * kStripPx];
        }
        float pa[3] = {0.f, 0.f, 0.f}, pb[3] = {0.f, 0.f, 0.f};
        const float bJ[3] = {beta[0] * J[0], beta[1] * J[1], beta[2] * J[2]};   // constant over the pixel's observations
        // The next chunk's loads are issued before this chunk's arithmetic (one chunk = 19 registers ahead): the loop used to
        // load, wait, compute -- with four waves per SIMD neither the memory system nor the VALU stayed busy (both ablations
        // near the full kernel's time, profiles/r04_ablation_table.txt).
        // Closed-form mode has just read the strip front to back for J: the gradient pass walks it BACK to front, so that what
        // it reads first is what the caches saw last (a second front-to-back walk over a strip larger than a wave's share of
        // the caches would find none of it).
        auto chunk_at = [&](uint32_t i) { return kClosed ? nch - 1u - i : i; };
        auto levels_of = [&](uint32_t g) { return min((uint32_t)kGroupLv, n - g * kGroupLv); };
        // The J-parameter kernel on uint8 colours leaves the ranges in HBM and forms ||cP|| itself (v_sqrt_f32, 1 ulp: this mode is held
        // to a tolerance; the closed-form trajectories are not touched -- they amplify a last-bit change of one range, see the
        // k/255 fold in experiment.h): 21 % fewer bytes for a kernel that waits for its loads, three FMAs and a square root more.
        constexpr bool kNoZ = !kClosed && !kColour && !kBoth && !kExpLightLoadZ;
        auto load_at = [&](uint32_t i) { return load_light_chunk<true, kNoZ>(sobs, sext, chunk_at(i), levels_of(chunk_at(i)), lane); };
        // one chunk: four levels of this lane's pixel
        auto grad_chunk = [&](const LightChunk &kk, uint32_t gi) {
            ColourChunk fc;
            if (kBoth) fc = load_colour_chunk(sext2, chunk_at(gi), levels_of(chunk_at(gi)), lane);
            if (kExpNoCompute) {
                acc.s[9] += chunk_checksum(kk);
                return;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (kNoZ ? kk.zz[j] == 0.0f : !(kk.zz[j] > 0.0f)) continue;  // padding slot
                const float cP[3] = {kk.xx[j], kk.yy[j], kk.ww[j]};
                LightObs o;
                if (kColour) { o.l = 1.0f; o.z = kk.zz[j]; }
                else o = light_obs(cP, kNoZ ? __builtin_amdgcn_sqrtf(__builtin_fmaf(cP[2], cP[2], __builtin_fmaf(cP[1], cP[1], cP[0] * cP[0]))) : kk.zz[j], R, tl, M);
                const float l = o.l, z = o.z;
                float dl = 0.f, dz = 0.f;   // dl = l dL/dl (the factor l rides in the modelled colour l E), dz = dL/dz
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float a = fast_exp2(z * nb[c]), g2 = fast_exp2(z * ng[c]);
                    const float omg = 1.0f - g2;
                    const float lE = l * __builtin_fmaf(J[c], a, B[c] * omg);
                    // I = k/255 folded into the residual (one rounding instead of two), as in fit_grad_kernel (round 6: in
                    // closed-form mode too, like fit.hip's closed_terms -- held to the reference's own spread by the knee tests)
                    const uint32_t kb = (kk.cc[c] >> (8 * j)) & 255u;
                    const float r = kBoth ? fc.c[c][j] - lE
                                  : kColour ? cP[c] - lE
                                  : (kClosed && !kExpLightFold2) ? unit_from_u8(kb) - lE : __builtin_fmaf((float)kb, kInv255L, -lE);
                    const float rl = r * l;
                    const float rlz = rl * z;
                    acc.s[9] = __builtin_fmaf(r, r, acc.s[9]);
                    pa[c] = __builtin_fmaf(rl, a, pa[c]);
                    pb[c] = __builtin_fmaf(rlz, a, pb[c]);
                    acc.s[c] = __builtin_fmaf(rl, omg, acc.s[c]);
                    acc.s[3 + c] = __builtin_fmaf(rlz, g2, acc.s[3 + c]);
                    dl = __builtin_fmaf(r, lE, dl);
                    dz = __builtin_fmaf(rl, __builtin_fmaf(gB[c], g2, -(bJ[c] * a)), dz);
                }
                if (kColour) continue;  // no light: nothing flows into cam2light / sigma
                // chain rule into lP (common factor -2 s applied in the step kernel).  l = exp(-q / 2), q = lp^T M lp, M symmetric:
                // dL/dlp = dl (-1/2) 2 M lp = -dl (w0, w1); lp = lP.xy / lP.z: dL/dlP.xy = dL/dlp iz, dL/dlP.z = -(dL/dlp . lp) iz
                // = dl q iz (w . lp = q); z = ||cP|| + ||lP||: dL/dlP += dz lP / ||lP||.
                const float kz = dl * o.iz;          // (-kz) (w0, w1, -q) is the light cone's share of dL/dlP
                const float dzi = dz * o.inl;
                float dlP[3];
                dlP[0] = __builtin_fmaf(dzi, o.lP[0], -(kz * o.w0));
                dlP[1] = __builtin_fmaf(dzi, o.lP[1], -(kz * o.w1));
                dlP[2] = __builtin_fmaf(dzi, o.lP[2], kz * o.q);
                // sum lP x dlP and sum dlP: what the six cam2light gradients factor through (kLightSums)
                acc.s[10] = __builtin_fmaf(o.lP[1], dlP[2], __builtin_fmaf(-o.lP[2], dlP[1], acc.s[10]));
                acc.s[11] = __builtin_fmaf(o.lP[2], dlP[0], __builtin_fmaf(-o.lP[0], dlP[2], acc.s[11]));
                acc.s[12] = __builtin_fmaf(o.lP[0], dlP[1], __builtin_fmaf(-o.lP[1], dlP[0], acc.s[12]));
#pragma unroll
                for (int a = 0; a < 3; ++a) acc.s[13 + a] += dlP[a];
                // dL/dM = sum dl (-1/2) lp lp^T
                const float kf = -0.5f * dl;
                const float k0 = kf * o.lp0;
                acc.s[16] = __builtin_fmaf(k0, o.lp0, acc.s[16]);
                acc.s[17] = __builtin_fmaf(k0, o.lp1, acc.s[17]);   // (the other off-diagonal entry is the same: lp lp^T is symmetric)
                acc.s[18] = __builtin_fmaf(kf * o.lp1, o.lp1, acc.s[18]);
            }
        };
        // The next chunk's loads are issued before this chunk's arithmetic (one chunk = 19 registers ahead): the loop used to
        // load, wait, compute -- with four waves per SIMD neither the memory system nor the VALU stayed busy (both ablations
        // near the full kernel's time, profiles/r04_ablation_table.txt).  Two chunks per turn, in two named sets of registers
        // (round 6: handing the prefetched chunk over -- `kk = kn` -- was 19 v_mov_b32 per chunk, one instruction in twenty-seven).
        // Closed-form mode has just read the strip front to back for J: the gradient pass walks it BACK to front, so that what
        // it reads first is what the caches saw last (a second front-to-back walk over a strip larger than a wave's share of
        // the caches would find none of it).
        if constexpr (kPingPong) {
            LightChunk ka = nch ? load_at(0) : LightChunk{}, kb2 = LightChunk{};
            for (uint32_t gi = 0; gi < nch; gi += 2u) {
                if (gi + 1u < nch) kb2 = load_at(gi + 1u);
                grad_chunk(ka, gi);
                if (gi + 1u < nch) {
                    if (gi + 2u < nch) ka = load_at(gi + 2u);
                    grad_chunk(kb2, gi + 1u);
                }
            }
        } else {   // (the float-colour instantiations hold more per observation: the second register set would cost them a wave per SIMD)
            LightChunk kn = nch ? load_at(0) : LightChunk{};
            for (uint32_t gi = 0; gi < nch; ++gi) {
                const LightChunk kk = kn;
                if (gi + 1u < nch) kn = load_at(gi + 1u);
                grad_chunk(kk, gi);
            }
        }
        // the pixel's tail, in the lane that owns it
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (kClosed) {
                acc.s[6 + c] += (pb[c] == 0.0f) ? 0.0f : J[c] * pb[c];
            } else {
                float Jc = J[c], m = st[(3 + c) * kStripPx], v = st[(6 + c) * kStripPx];
                acc.s[6 + c] += (pb[c] == 0.0f) ? 0.0f : Jc * pb[c];
                adam_update_J(Jc, m, v, gscale * pa[c], co);
                st[c * kStripPx] = Jc; st[(3 + c) * kStripPx] = m; st[(6 + c) * kStripPx] = v;
            }
        }
    }
    if (kJOnly) return;
    // workgroup sums: shuffle tree, then the four waves in fixed order
#pragma unroll
    for (int q = 0; q < kLightSums; ++q) {
        const float x = wave_sum_lane0(acc.s[q]);   // (handoff.h: the __shfl_down tree's additions, register to register)
        if (lane == 0) wsum[wave][q] = x;
    }
    __syncthreads();
    if (t < kLightSums) partials[(size_t)t * n_blocks + blockIdx.x] = ((wsum[0][t] + wsum[1][t]) + wsum[2][t]) + wsum[3][t];
}

// The iteration's tail, one workgroup: the 26 sums, the step on the 19 parameters, the next geometry, the log row.
// (Folding it into the gradient launch as a last-arriver tail, like fit.hip does, was measured: -29 us per iteration with
// J as a parameter but +9 us in closed-form mode, whose two-pass kernel came out 5 % slower with the tail's code and
// 14 KB of LDS in it; as its own launch it costs ~20 us + one launch gap in both modes.  Round 2 had two launches here,
// a reduction that waited for every load before issuing the next (49 us) and the step (14 us).)
__global__ __launch_bounds__(256) void light_tail_kernel(const float *partials, int n_blocks, double *sums, float *pstate,
                                                         float *geom, double *dexp, const uint64_t *__restrict__ n_obs_total,
                                                         const AdamCoef co, double *trace_row) {
    __shared__ LightLds lds;
    light_reduce(partials, n_blocks, lds.sums, lds.w4);
    if (threadIdx.x < kLightSums) sums[threadIdx.x] = lds.sums[threadIdx.x];
    light_step(lds, pstate, geom, dexp, n_obs_total, co, trace_row);
}

struct LightParams19 { float v[kLightParams]; };

__global__ void light_params_upload_kernel(float *dst, const LightParams19 p) {
    if (threadIdx.x < kLightParams) dst[threadIdx.x] = p.v[threadIdx.x];
}

// The deal of a gradient launch of `n_blocks` workgroups, written once per sucre_fit_run_light call: one thread per wave.
__global__ __launch_bounds__(256) void light_deal_kernel(const StripMeta *__restrict__ meta, int n_strips, uint32_t W, uint32_t kmax,
                                                         const DealShares sh, uint32_t *__restrict__ count, uint32_t *__restrict__ strips) {
    const uint32_t wid = blockIdx.x * 256u + threadIdx.x;
    if (wid >= W) return;
    uint32_t *mine = strips + (size_t)wid * kmax;
    const uint32_t K = deal_walk(wid, W, (uint32_t)n_strips, sh, [&](uint32_t s) { return meta[s].levels; },
                                 [&](uint32_t k, uint32_t strip) { if (k < kmax) mine[k] = strip; });
    // never more than the table holds (tests/native/deal_check.cpp: K <= deal_rounds for these grids; were it ever not so, the wave
    // would walk its neighbour's strips -- stepped and counted twice, silently: ADVICE round 5): the surplus strips stay unvisited
    // and their pixels keep their J, which the parity tests see
    count[wid] = K < kmax ? K : kmax;
}

size_t light_workspace_bytes(const Layout &L, int ext_sets) {
    LightLayout X;
    make_light_layout(L, &X, ext_sets);
    return X.total;
}

int64_t light_params_offset(const Layout &L) {
    LightLayout X;
    make_light_layout(L, &X);
    return (int64_t)X.off_params;
}

uint8_t *light_ext_dense(const Layout &L, uint8_t *lws) {
    LightLayout X;
    make_light_layout(L, &X);
    return lws + X.off_ext_dense;
}

uint8_t *light_ext_comp(const Layout &L, uint8_t *lws) {
    LightLayout X;
    make_light_layout(L, &X);
    return lws + X.off_ext_comp;
}

uint8_t *light_ext2_dense(const Layout &L, uint8_t *lws) {
    LightLayout X;
    make_light_layout(L, &X, 2);
    return lws + X.off_ext2_dense;
}

uint8_t *light_ext2_comp(const Layout &L, uint8_t *lws) {
    LightLayout X;
    make_light_layout(L, &X, 2);
    return lws + X.off_ext2_comp;
}

hipError_t launch_light_init(const Layout &L, uint8_t *lws, const float *params19, hipStream_t s) {
    LightLayout X;
    make_light_layout(L, &X);
    LightParams19 p;
    for (int i = 0; i < kLightParams; ++i) p.v[i] = params19[i];
    float *pstate = reinterpret_cast<float *>(lws + X.off_params);
    float *scratch = reinterpret_cast<float *>(lws + X.off_sums);  // 19 floats staged in the (still unused) sums area
    hipLaunchKernelGGL(light_params_upload_kernel, dim3(1), dim3(64), 0, s, scratch, p);
    hipLaunchKernelGGL(light_init_kernel, dim3(1), dim3(64), 0, s, pstate, reinterpret_cast<float *>(lws + X.off_geom),
                       reinterpret_cast<double *>(lws + X.off_dexp), scratch);
    return hipGetLastError();
}

// Workgroups of a gradient launch: as many as are resident at once.  The strips are dealt statically over the launch's
// waves, so a workgroup that has to wait for a slot runs its whole share after everybody else; the instantiations differ
// in registers (4 or 5 waves per SIMD; forcing 5 spills), and a grid of kFitGrid = 5 per CU left a fifth of the
// J-parameter kernel's work to a second round: 0.413 -> 0.372 ms per iteration.  Asked of the runtime once per instantiation.
// Cached per (instantiation, device) under a mutex: the command line submits from several threads, and a process may drive
// GPUs with different CU counts (the grid fixes the summation order, so it must be the CURRENT device's, every time).
template <class K>
static int resident_grid(K kernel, const Layout &L) {
    constexpr int kMaxDevices = 64;
    static int per_device[kMaxDevices] = {};
    static std::mutex lock;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return L.n_blocks < kFitGrid ? L.n_blocks : kFitGrid;
    int grid;
    {
        std::lock_guard<std::mutex> hold(lock);
        if (per_device[dev] == 0) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) == hipSuccess &&
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus > 0)
                per_device[dev] = per_cu * cus;
            else
                per_device[dev] = kFitGrid;
        }
        grid = per_device[dev];
    }
    if (SUCRE_EXP_LIGHT_GRID > 0) grid = SUCRE_EXP_LIGHT_GRID;   // (experiment.h: fewer resident workgroups -- a smaller working set)
    return L.n_blocks < grid ? L.n_blocks : grid;
}

// What a gradient launch's deal is: its grid (the resident workgroups of ITS instantiation), the shares, the strips reserved
// per wave.
struct LightDeal { int grid; DealShares sh; uint32_t kmax; };

template <class K>
static LightDeal light_deal_of(K kernel, const Layout &L) {
    LightDeal d;
    d.grid = resident_grid(kernel, L);
    d.sh = deal_shares_resident((uint32_t)d.grid);
    d.kmax = deal_rounds((uint32_t)d.grid * 4u, (uint32_t)L.n_strips, d.sh);
    return d;
}

// kDeal 0: launch with the table light_deal_kernel wrote; 1: write that table; 2: launch without one (plain deal).
template <bool kClosed, bool kJOnly, bool kColour, bool kBoth = false>
static int launch_light_grad_c(const Layout &L, const LightLayout &X, uint8_t *ws, uint8_t *lws, const AdamCoef &co,
                               hipStream_t s, int deal_mode) {
    const LightDeal d = light_deal_of(light_grad_kernel<kClosed, kJOnly, kColour, kBoth>, L);
    uint32_t *count = reinterpret_cast<uint32_t *>(lws + X.off_deal), *strips = count + kLightMaxWaves;
    const bool fits = (uint32_t)d.grid * 4u <= kLightMaxWaves && (size_t)d.grid * 4u * d.kmax <= (size_t)kMaxGen * L.n_strips + kLightMaxWaves;
    if (deal_mode == 1) {
        if (fits)
            hipLaunchKernelGGL(light_deal_kernel, dim3(((uint32_t)d.grid * 4u + 255u) / 256u), dim3(256), 0, s,
                               reinterpret_cast<const StripMeta *>(ws + L.off_strip_meta), L.n_strips, (uint32_t)d.grid * 4u, d.kmax, d.sh, count, strips);
        return d.grid;
    }
    const bool table = deal_mode == 0 && fits;
    hipLaunchKernelGGL((light_grad_kernel<kClosed, kJOnly, kColour, kBoth>), dim3(d.grid), dim3(256), 0, s, ws + L.off_comp,
                       lws + X.off_ext_comp, kBoth ? light_ext2_comp(L, lws) : nullptr,
                       reinterpret_cast<const StripMeta *>(ws + L.off_strip_meta), L.n_strips,
                       reinterpret_cast<const float *>(lws + X.off_params), reinterpret_cast<const float *>(lws + X.off_geom),
                       reinterpret_cast<const uint64_t *>(ws + L.off_n_obs_total),
                       reinterpret_cast<float *>(ws + L.off_state), reinterpret_cast<float *>(lws + X.off_partials), co,
                       table ? count : nullptr, table ? strips : nullptr, d.kmax);
    return d.grid;
}

template <bool kClosed, bool kJOnly>
static int launch_light_grad(const Layout &L, const LightLayout &X, uint8_t *ws, uint8_t *lws, const AdamCoef &co,
                             unsigned flags, hipStream_t s, int deal_mode) {
    if (flags & SUCRE_FIT_EXT_BOTH) return launch_light_grad_c<kClosed, kJOnly, false, true>(L, X, ws, lws, co, s, deal_mode);
    if (flags & SUCRE_FIT_EXT_COLOUR) return launch_light_grad_c<kClosed, kJOnly, true>(L, X, ws, lws, co, s, deal_mode);
    return launch_light_grad_c<kClosed, kJOnly, false>(L, X, ws, lws, co, s, deal_mode);
}

// Once per sucre_fit_run_light call, before its iterations: the deal of the gradient launches those iterations will make.
hipError_t launch_light_deal(const Layout &L, uint8_t *ws, uint8_t *lws, unsigned flags, hipStream_t s) {
    LightLayout X;
    make_light_layout(L, &X);
    if (flags & SUCRE_FIT_CLOSED_FORM) launch_light_grad<true, false>(L, X, ws, lws, AdamCoef{}, flags, s, 1);
    else launch_light_grad<false, false>(L, X, ws, lws, AdamCoef{}, flags, s, 1);
    return hipGetLastError();
}

hipError_t launch_light_update_J(const Layout &L, uint8_t *ws, uint8_t *lws, unsigned flags, hipStream_t s) {
    LightLayout X;
    make_light_layout(L, &X);
    hipLaunchKernelGGL(light_geometry_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const float *>(lws + X.off_params),
                       reinterpret_cast<float *>(lws + X.off_geom), reinterpret_cast<double *>(lws + X.off_dexp));
    launch_light_grad<true, true>(L, X, ws, lws, AdamCoef{}, flags, s, 2);
    return hipGetLastError();
}

hipError_t launch_light_iter(const Layout &L, uint8_t *ws, uint8_t *lws, const AdamCoef &co, unsigned flags,
                             double *trace_row, hipStream_t s) {
    LightLayout X;
    make_light_layout(L, &X);
    const int grid = (flags & SUCRE_FIT_CLOSED_FORM) ? launch_light_grad<true, false>(L, X, ws, lws, co, flags, s, 0)
                                                     : launch_light_grad<false, false>(L, X, ws, lws, co, flags, s, 0);
    hipLaunchKernelGGL(light_tail_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<const float *>(lws + X.off_partials), grid,
                       reinterpret_cast<double *>(lws + X.off_sums), reinterpret_cast<float *>(lws + X.off_params),
                       reinterpret_cast<float *>(lws + X.off_geom), reinterpret_cast<double *>(lws + X.off_dexp),
                       reinterpret_cast<const uint64_t *>(ws + L.off_n_obs_total), co, trace_row);
    return hipGetLastError();
}

}  // namespace sucre
