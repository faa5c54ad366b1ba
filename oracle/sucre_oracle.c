/*
 * sucre_oracle.c -- CPU restatement of the SUCRe hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP engine in sucre_amd/csrc.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; nothing under sucre_amd/ may import, link or call it.
 *
 * It restates, in plain C and in the reference's own ragged data format (per-view lists of
 * (u1, v1, cP, I), loader.py:33-53), the algorithm of clementinboittiaux/sucre:
 *   - dense two-way matching           sfm.py:90-125, 154-175   (oracle_match_view)
 *   - observation preparation          sfm.py:137, loader.py:87,113 (oracle_unproject)
 *   - image-formation model + Adam fit sucre.py:52-82, 124-157  (oracle_fit)
 *   - closed-form J                    sucre.py:66-77           (oracle_update_J)
 *
 * Pinning: the reference has no tests of its own, so the oracle is pinned against
 * golden vectors produced by running the reference itself in the dev container
 * (tests/golden/gen_golden.py -> tests/golden/<name>.npz; checked by tests/test_oracle_golden.py).
 *
 * Arithmetic notes (all verified against torch 2.10 CPU):
 *   - a float32 (3x3)@(3xn) torch matmul is, per output element, the FMA chain
 *     fma(a2,b2, fma(a1,b1, a0*b0)); dot3() below reproduces it bit for bit, so match sets are bit-exact.
 *   - Tensor.long() of a float truncates toward zero; NaN / out-of-range give INT64_MIN (x86 cvttss2si).
 *   - torch.optim.Adam (single-tensor, non-capturable branch) is restated in adam_step().
 * Build with -ffp-contract=off: every fused multiply-add here is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t H, W;
    float K[9];     /* camera matrix                      sfm.py:62-73  */
    float Kinv[9];  /* K.inverse() as torch computes it   sfm.py:92     */
    float R[9];     /* world-from-camera rotation         sfm.py:32-40  */
    float t[3];
    float Rinv[9];  /* Pose.inverse(): R.T                sfm.py:42-47  */
    float tinv[3];  /*                 -R.T @ t                          */
} oracle_cam_t;

int oracle_version(void) { return 1; }

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* one row of a float32 torch matmul with inner dimension 3 */
static inline float dot3(const float *a, float x, float y, float z) {
    float acc = a[0] * x;
    acc = fmaf(a[1], y, acc);
    acc = fmaf(a[2], z, acc);
    return acc;
}

/* sfm.py:90-93  cP = K^-1 @ (d * [u+0.5, v+0.5, 1]) */
static inline void unproject(const oracle_cam_t *c, float u, float v, float d, float out[3]) {
    const float x = d * (u + 0.5f), y = d * (v + 0.5f), z = d * 1.0f;
    out[0] = dot3(c->Kinv + 0, x, y, z);
    out[1] = dot3(c->Kinv + 3, x, y, z);
    out[2] = dot3(c->Kinv + 6, x, y, z);
}

/* sfm.py:49-55  R @ P + t */
static inline void rigid(const float *R, const float *t, const float p[3], float out[3]) {
    out[0] = dot3(R + 0, p[0], p[1], p[2]) + t[0];
    out[1] = dot3(R + 3, p[0], p[1], p[2]) + t[1];
    out[2] = dot3(R + 6, p[0], p[1], p[2]) + t[2];
}

/* Tensor.long() on x86 */
static inline int64_t to_long(float x) {
    if (!(x > -9.2233720368547758e18f && x < 9.2233720368547758e18f)) return INT64_MIN;
    return (int64_t)x;
}

/* sfm.py:103-107 + 116-117: world point -> integer pixel of camera c, 1 if inside the image */
static inline int project_px(const oracle_cam_t *c, const float wP[3], int64_t *u, int64_t *v) {
    float cP[3], cp[3];
    rigid(c->Rinv, c->tinv, wP, cP);
    cp[0] = dot3(c->K + 0, cP[0], cP[1], cP[2]);
    cp[1] = dot3(c->K + 3, cP[0], cP[1], cP[2]);
    cp[2] = dot3(c->K + 6, cP[0], cP[1], cP[2]);
    *u = to_long(cp[0] / cp[2]);
    *v = to_long(cp[1] / cp[2]);
    return (0 <= *u) && (*u < c->W) && (0 <= *v) && (*v < c->H);
}

/*
 * Image.match_two_way (sfm.py:121-125) of image 1 against image 2, followed by d = depth2[v2,u2] (sfm.py:137).
 * Outputs are in torch.where order of image 1 (row-major), sized for H1*W1 entries; any may be NULL.
 * Returns the number of matches, or -1 on allocation failure.
 */
int64_t oracle_match_view(const float *depth1, const oracle_cam_t *c1, const float *depth2, const oracle_cam_t *c2,
                          int16_t *u1o, int16_t *v1o, int16_t *u2o, int16_t *v2o, float *d2o) {
    const int64_t n2 = (int64_t)c2->H * c2->W;
    /* matches2.map(): where every pixel of image 2 lands in image 1, -1 = nowhere (sfm.py:154-159) */
    int64_t *map = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)n2);
    if (!map) return -1;
    for (int64_t i = 0; i < 2 * n2; ++i) map[i] = -1;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n2; ++p) {
        const float d = depth2[p];
        if (!(d > 0)) continue;
        const int64_t v = p / c2->W, u = p % c2->W;
        float cP[3], wP[3];
        int64_t ub, vb;
        unproject(c2, (float)u, (float)v, d, cP);
        rigid(c2->R, c2->t, cP, wP);
        if (project_px(c1, wP, &ub, &vb)) {
            map[2 * p + 0] = vb;
            map[2 * p + 1] = ub;
        }
    }
    /* matches1 and the intersection (sfm.py:115-119, 171-175) */
    int64_t n = 0;
    for (int64_t v = 0; v < c1->H; ++v) {
        for (int64_t u = 0; u < c1->W; ++u) {
            const float d = depth1[v * c1->W + u];
            if (!(d > 0)) continue;
            float cP[3], wP[3];
            int64_t uf, vf;
            unproject(c1, (float)u, (float)v, d, cP);
            rigid(c1->R, c1->t, cP, wP);
            if (!project_px(c2, wP, &uf, &vf)) continue;
            const int64_t q = vf * c2->W + uf;
            if (map[2 * q + 0] != v || map[2 * q + 1] != u) continue;
            if (u1o) u1o[n] = (int16_t)u;
            if (v1o) v1o[n] = (int16_t)v;
            if (u2o) u2o[n] = (int16_t)uf;
            if (v2o) v2o[n] = (int16_t)vf;
            if (d2o) d2o[n] = depth2[q];
            ++n;
        }
    }
    free(map);
    return n;
}

/* loader.py:113  cP = image.unproject_depth(u2, v2, d), u2/v2 int16; cP is (3,n) row-major */
void oracle_unproject(const oracle_cam_t *c, const int16_t *u, const int16_t *v, const float *d, int64_t n, float *cP) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float p[3];
        unproject(c, (float)u[i], (float)v[i], d[i], p);
        cP[i] = p[0];
        cP[n + i] = p[1];
        cP[2 * n + i] = p[2];
    }
}

/* loader.py:87  I = rgb[v2,u2].T with rgb = uint8/255 (loader.py:157,163); rgb is (H,W,3) uint8, I is (3,n) */
void oracle_gather_rgb(const uint8_t *rgb, int W, const int16_t *u, const int16_t *v, int64_t n, float *I) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c)
            I[c * n + i] = (float)((double)rgb[((int64_t)v[i] * W + u[i]) * 3 + c] / 255.0);
}

/* sucre.py:53 */
static inline float norm3(const float *cP, int64_t n, int64_t i) {
    const float x = cP[i], y = cP[n + i], z = cP[2 * n + i];
    return sqrtf(x * x + y * y + z * z);
}

/*
 * SUCRe.update_J (sucre.py:66-77): J = sum_k (I-b)*a / sum_k a^2, one view at a time, float32.
 * J is (H,W,3); pixels without observation become 0/0 = NaN.  Returns -1 on allocation failure.
 */
int oracle_update_J(int H, int W, int n_samples, const int64_t *counts, const int16_t *const *us,
                    const int16_t *const *vs, const float *const *cPs, const float *const *Is,
                    const float *params, float *J) {
    const size_t npx = (size_t)H * W * 3;
    float *num = (float *)calloc(npx, sizeof(float)), *den = (float *)calloc(npx, sizeof(float));
    if (!num || !den) { free(num); free(den); return -1; }
    const float *B = params, *beta = params + 3, *gamma = params + 6;
    for (int s = 0; s < n_samples; ++s) {
        const int64_t n = counts[s];
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            const float z = norm3(cPs[s], n, i);
            const size_t px = ((size_t)vs[s][i] * W + us[s][i]) * 3;
            for (int c = 0; c < 3; ++c) {
                const float a = expf(-beta[c] * z);
                const float b = B[c] * (1.0f - expf(-gamma[c] * z));
                num[px + c] += (Is[s][c * n + i] - b) * a;
                den[px + c] += a * a;
            }
        }
    }
    for (size_t i = 0; i < npx; ++i) J[i] = num[i] / den[i];
    free(num);
    free(den);
    return 0;
}

/* torch/optim/adam.py::_single_tensor_adam, non-capturable, no amsgrad / weight decay */
typedef struct { float w1, beta2, w2, step_size_neg, bc2_sqrt, eps; } adam_coef_t;

static adam_coef_t adam_coef(int step, double lr, double beta1, double beta2, double eps) {
    adam_coef_t c;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    c.w1 = (float)(1.0 - beta1);
    c.beta2 = (float)beta2;
    c.w2 = (float)(1.0 - beta2);
    c.step_size_neg = (float)(-(lr / bc1));
    c.bc2_sqrt = (float)sqrt(bc2);
    c.eps = (float)eps;
    return c;
}

static inline void adam_step(float *p, float *m, float *v, float g, const adam_coef_t *c) {
    *m = fmaf(c->w1, g - *m, *m);                 /* exp_avg.lerp_(grad, 1-beta1)                    */
    *v = (*v * c->beta2) + (c->w2 * g) * g;       /* exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)  */
    const float denom = sqrtf(*v) / c->bc2_sqrt + c->eps;
    *p = *p + (c->step_size_neg * *m) / denom;    /* param.addcdiv_(exp_avg, denom, value=-step_size) */
}

/*
 * sucre.adam (sucre.py:124-157) on the reference's observation lists.
 *   J       (H,W,3) float32; J-parameter mode: in = rgb1 with NaN where depth1<=0 (sucre.py:46-49), out = fitted.
 *           closed-form mode: output only (final update_J, sucre.py:156).
 *   params  B[3], beta[3], gamma[3] in/out (init 0.1, sucre.py:41-43).
 *   trace   num_iter x 10 doubles: cost (sum of squared residuals before the step, sucre.py:144-146) and the
 *           nine parameters after the step.
 * Gradient of L = sum r^2 / (3 n_obs) is analytic; the nine global sums and the cost are
 * accumulated in double, per-pixel J gradients in float32.
 */
int oracle_fit(int H, int W, int n_samples, const int64_t *counts, const int16_t *const *us,
               const int16_t *const *vs, const float *const *cPs, const float *const *Is,
               float *J, float *params, int num_iter, double lr, int use_closed_form, double *trace) {
    const size_t npx = (size_t)H * W * 3;
    int64_t n_obs = 0;
    for (int s = 0; s < n_samples; ++s) n_obs += counts[s];
    float *gJ = (float *)calloc(npx, sizeof(float));
    float *mJ = (float *)calloc(npx, sizeof(float)), *vJ = (float *)calloc(npx, sizeof(float));
    float **zs = (float **)calloc((size_t)n_samples, sizeof(float *));
    if (!gJ || !mJ || !vJ || !zs) return -1;
    for (int s = 0; s < n_samples; ++s) {
        zs[s] = (float *)malloc(sizeof(float) * (size_t)(counts[s] > 0 ? counts[s] : 1));
        if (!zs[s]) return -1;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < counts[s]; ++i) zs[s][i] = norm3(cPs[s], counts[s], i);
    }
    float mP[9] = {0}, vP[9] = {0};
    /* (loss / n_obs / 3).backward(): d/dloss = (1/3)/n_obs in float32 (sucre.py:145) */
    const float scale = (1.0f / 3.0f) / (float)n_obs;
    float *B = params, *beta = params + 3, *gamma = params + 6;

    for (int it = 0; it < num_iter; ++it) {
        if (use_closed_form &&
            oracle_update_J(H, W, n_samples, counts, us, vs, cPs, Is, params, J) != 0) return -1;
        memset(gJ, 0, sizeof(float) * npx);
        double gB[3] = {0, 0, 0}, gbeta[3] = {0, 0, 0}, ggamma[3] = {0, 0, 0}, cost = 0.0;
        for (int s = 0; s < n_samples; ++s) {
            const int64_t n = counts[s];
            double tB[3] = {0, 0, 0}, tb[3] = {0, 0, 0}, tg[3] = {0, 0, 0}, tc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : tB[:3], tb[:3], tg[:3], tc)
            for (int64_t i = 0; i < n; ++i) {
                const float z = zs[s][i];
                const size_t px = ((size_t)vs[s][i] * W + us[s][i]) * 3;
                for (int c = 0; c < 3; ++c) {
                    const float a = expf(-beta[c] * z);
                    const float g = expf(-gamma[c] * z);
                    const float Jc = J[px + c];
                    const float Ihat = Jc * a + B[c] * (1.0f - g);      /* sucre.py:79-82, l = 1 */
                    const float r = Is[s][c * n + i] - Ihat;
                    const float dr = -2.0f * r * scale;                 /* dL/dIhat */
                    tc += (double)r * (double)r;
                    if (!use_closed_form) gJ[px + c] += dr * a;          /* unique (v,u) within one view */
                    tB[c] += (double)(dr * (1.0f - g));
                    tb[c] += (double)(dr * Jc * a * -z);
                    tg[c] += (double)(dr * B[c] * g * z);
                }
            }
            for (int c = 0; c < 3; ++c) { gB[c] += tB[c]; gbeta[c] += tb[c]; ggamma[c] += tg[c]; }
            cost += tc;
        }
        const adam_coef_t co = adam_coef(it + 1, lr, 0.9, 0.999, 1e-8);
        for (int c = 0; c < 3; ++c) {
            adam_step(&B[c], &mP[c], &vP[c], (float)gB[c], &co);
            adam_step(&beta[c], &mP[3 + c], &vP[3 + c], (float)gbeta[c], &co);
            adam_step(&gamma[c], &mP[6 + c], &vP[6 + c], (float)ggamma[c], &co);
        }
        if (!use_closed_form) {
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < (int64_t)npx; ++i) adam_step(&J[i], &mJ[i], &vJ[i], gJ[i], &co);
        }
        if (trace) {
            trace[it * 10] = cost;
            for (int k = 0; k < 9; ++k) trace[it * 10 + 1 + k] = (double)params[k];
        }
    }
    int rc = 0;
    if (use_closed_form) rc = oracle_update_J(H, W, n_samples, counts, us, vs, cPs, Is, params, J);
    for (int s = 0; s < n_samples; ++s) free(zs[s]);
    free(zs); free(gJ); free(mJ); free(vJ);
    return rc;
}

/* SUCRe.__init__ (sucre.py:46-49): J0 = rgb (uint8/255), NaN where depth <= 0 */
void oracle_init_J(const uint8_t *rgb, const float *depth, int H, int W, float *J) {
    for (int64_t p = 0; p < (int64_t)H * W; ++p)
        for (int c = 0; c < 3; ++c)
            J[p * 3 + c] = (depth[p] <= 0) ? NAN : (float)((double)rgb[p * 3 + c] / 255.0);
}

/*
 * Shared-water extension (north star; NOT reference behaviour): several images step in lock-step and share
 * B, beta, gamma; objective  sum_images sum_obs r^2 / (3 * n_obs_total).  Oracle = the reference's per-image
 * arithmetic (oracle_fit above) with the gradient scale taken from the total observation count and the nine
 * water gradients summed over images before one Adam step (a tied-parameter composition of reference modules,
 * SURVEY.md section 8e).  One image's half-iteration:
 *   sums[0..2] = dS/dB, sums[3..5] = dS/dbeta, sums[6..8] = dS/dgamma for S = sum r^2 (unscaled, float64),
 *   sums[9] = S;  J takes its Adam step with gradient (dS/dJ) * (1/3)/n_obs_total (J-parameter mode), or is re-solved
 *   by update_J from the shared parameters before the pass and held constant (closed-form mode, sucre.py:141).
 */
int oracle_shared_grad_mode(int H, int W, int n_samples, const int64_t *counts, const int16_t *const *us,
                            const int16_t *const *vs, const float *const *cPs, const float *const *Is, float *J,
                            float *mJ, float *vJ, const float *params, int step, double lr, int64_t n_obs_total,
                            int use_closed_form, double *sums) {
    const size_t npx = (size_t)H * W * 3;
    /* closed-form mode (sucre.py:141): J is re-solved from the shared parameters at the top of the iteration and is
     * a constant of the gradient pass; it takes no Adam step. */
    if (use_closed_form && oracle_update_J(H, W, n_samples, counts, us, vs, cPs, Is, params, J) != 0) return -1;
    float *gJ = use_closed_form ? NULL : (float *)calloc(npx, sizeof(float));
    if (!use_closed_form && !gJ) return -1;
    const float scale = (1.0f / 3.0f) / (float)n_obs_total;
    const float *B = params, *beta = params + 3, *gamma = params + 6;
    for (int k = 0; k < 10; ++k) sums[k] = 0.0;
    for (int s = 0; s < n_samples; ++s) {
        const int64_t n = counts[s];
        double tB[3] = {0, 0, 0}, tb[3] = {0, 0, 0}, tg[3] = {0, 0, 0}, tc = 0.0;
        /* (v,u) are unique within one view, so gJ has no race inside this loop */
#pragma omp parallel for schedule(static) reduction(+ : tB[:3], tb[:3], tg[:3], tc)
        for (int64_t i = 0; i < n; ++i) {
            const float z = norm3(cPs[s], n, i);
            const size_t px = ((size_t)vs[s][i] * W + us[s][i]) * 3;
            for (int c = 0; c < 3; ++c) {
                const float a = expf(-beta[c] * z), g = expf(-gamma[c] * z);
                const float Jc = J[px + c];
                const float r = Is[s][c * n + i] - (Jc * a + B[c] * (1.0f - g));
                tc += (double)r * (double)r;
                if (gJ) gJ[px + c] += (-2.0f * r * scale) * a;
                tB[c] += (double)(-2.0f * r * (1.0f - g));
                tb[c] += (double)(-2.0f * r * Jc * a * -z);
                tg[c] += (double)(-2.0f * r * B[c] * g * z);
            }
        }
        for (int c = 0; c < 3; ++c) { sums[c] += tB[c]; sums[3 + c] += tb[c]; sums[6 + c] += tg[c]; }
        sums[9] += tc;
    }
    if (gJ) {
        const adam_coef_t co = adam_coef(step, lr, 0.9, 0.999, 1e-8);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)npx; ++i) adam_step(&J[i], &mJ[i], &vJ[i], gJ[i], &co);
        free(gJ);
    }
    return 0;
}

int oracle_shared_grad(int H, int W, int n_samples, const int64_t *counts, const int16_t *const *us,
                       const int16_t *const *vs, const float *const *cPs, const float *const *Is, float *J,
                       float *mJ, float *vJ, const float *params, int step, double lr, int64_t n_obs_total,
                       double *sums) {
    return oracle_shared_grad_mode(H, W, n_samples, counts, us, vs, cPs, Is, J, mJ, vJ, params, step, lr, n_obs_total, 0, sums);
}

/* pstate = params[9], exp_avg[9], exp_avg_sq[9]; sums = all-reduced output of oracle_shared_grad */
void oracle_shared_step(float *pstate, const double *sums, int step, double lr, int64_t n_obs_total) {
    const float scale = (1.0f / 3.0f) / (float)n_obs_total;
    const adam_coef_t co = adam_coef(step, lr, 0.9, 0.999, 1e-8);
    for (int k = 0; k < 9; ++k)
        adam_step(&pstate[k], &pstate[9 + k], &pstate[18 + k], (float)(sums[k] * (double)scale), &co);
}

/* =====================================================================================================================
 * Artificial-light model (--light-model; sucre.py:54-61 with se3.exp, se3.py:22-27).  TEST INFRASTRUCTURE ONLY.
 *
 *   R, t = se3.exp(cam2light);  Sigma = sigma^T sigma;  lP = R cP + t;  lp = lP.xy / lP.z
 *   l = exp(-(lp^T Sigma^-1 lp) / 2);   z = ||cP|| + ||lP||;   Ihat = l (J a + B (1 - g))
 *
 * The reference differentiates this with autograd; the oracle uses the analytic gradient:
 *   E = J a + B(1-g),  dI = dL/dIhat = -2 r s,   dl = sum_c dI E,   dz = sum_c dI l (-beta J a + gamma B g)
 *   dlP = dz lP/||lP|| + (dlp/dlP)^T (dl * (-l M lp)),  M = Sigma^-1;   dt = sum dlP;   dR = sum dlP cP^T
 *   dM = sum dl (-l/2) lp lp^T;   dSigma = -M^T dM M^T;   dsigma = sigma (dSigma + dSigma^T)
 *   dxi_i = <[dR dt; 0 0], D exp(hat(xi))[G_i]>   (directional derivative via the 8x8 block exponential)
 * Parameter vector (19): B[3], beta[3], gamma[3], cam2light[6], sigma[4] (row-major 2x2).
 * ===================================================================================================================*/

static void mat_mul(const double *A, const double *B, double *C, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j];
            C[i * n + j] = s;
        }
}

/* exp(A) for a small n x n matrix (n <= 8): scaling and squaring with a degree-18 Taylor series, float64 */
static void mat_exp(const double *A, double *E, int n) {
    double nrm = 0.0, S[64], T[64], P[64];
    for (int i = 0; i < n * n; ++i) nrm = fmax(nrm, fabs(A[i]));
    int sq = 0;
    while (nrm > 0.25) { nrm *= 0.5; ++sq; }
    const double sc = ldexp(1.0, -sq);
    for (int i = 0; i < n * n; ++i) S[i] = A[i] * sc;
    for (int i = 0; i < n * n; ++i) { E[i] = (i % (n + 1) == 0) ? 1.0 : 0.0; T[i] = E[i]; }
    for (int k = 1; k <= 18; ++k) {
        mat_mul(T, S, P, n);
        for (int i = 0; i < n * n; ++i) { T[i] = P[i] / k; E[i] += T[i]; }
    }
    for (int s = 0; s < sq; ++s) { mat_mul(E, E, P, n); memcpy(E, P, sizeof(double) * n * n); }
}

static void se3_hat(const double xi[6], double A[16]) {
    const double w1 = xi[0], w2 = xi[1], w3 = xi[2];
    const double M[16] = {0, -w3, w2, xi[3], w3, 0, -w1, xi[4], -w2, w1, 0, xi[5], 0, 0, 0, 0};
    memcpy(A, M, sizeof(M));
}

/* se3.exp (se3.py:22-27): twist -> R (row-major 3x3), t */
void oracle_se3_exp(const float *xi, float *R, float *t) {
    double x[6], A[16], E[16];
    for (int i = 0; i < 6; ++i) x[i] = xi[i];
    se3_hat(x, A);
    mat_exp(A, E, 4);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = (float)E[i * 4 + j]; t[i] = (float)E[i * 4 + 3]; }
}

/* dxi[i] = sum_jk dT[j][k] * (D exp(hat(xi))[G_i])[j][k], dT = [dR dt; 0 0] */
static void se3_exp_backward(const double xi[6], const double dR[9], const double dt[3], double dxi[6]) {
    double A[16];
    se3_hat(xi, A);
    for (int i = 0; i < 6; ++i) {
        double e[6] = {0, 0, 0, 0, 0, 0}, G[16], BM[64], EX[64];
        e[i] = 1.0;
        se3_hat(e, G);
        memset(BM, 0, sizeof(BM));
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                BM[r * 8 + c] = A[r * 4 + c];
                BM[(r + 4) * 8 + c + 4] = A[r * 4 + c];
                BM[r * 8 + c + 4] = G[r * 4 + c];
            }
        mat_exp(BM, EX, 8);  /* upper-right 4x4 block = directional derivative */
        double s = 0.0;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) s += dR[r * 3 + c] * EX[r * 8 + c + 4];
            s += dt[r] * EX[r * 8 + 7];
        }
        dxi[i] = s;
    }
}

typedef struct { float R[9], t[3], M[4]; } light_t;

static void light_setup(const float *p, light_t *L) {
    oracle_se3_exp(p + 9, L->R, L->t);
    const float *sg = p + 15;  /* sigma row-major: [s00 s01; s10 s11]; Sigma = sigma^T sigma */
    const float S00 = sg[0] * sg[0] + sg[2] * sg[2], S01 = sg[0] * sg[1] + sg[2] * sg[3];
    const float S10 = S01, S11 = sg[1] * sg[1] + sg[3] * sg[3];
    const float det = S00 * S11 - S01 * S10;
    L->M[0] = S11 / det; L->M[1] = -S01 / det; L->M[2] = -S10 / det; L->M[3] = S00 / det;
}

/* l and total range of one observation (sucre.py:52-64); also returns lP, lp, ||lP|| for the gradient */
static inline void light_eval(const light_t *L, const float cP[3], float *l, float *z, float lP[3], float lp[2], float *nl) {
    rigid(L->R, L->t, cP, lP);
    lp[0] = lP[0] / lP[2];
    lp[1] = lP[1] / lP[2];
    const float q = lp[0] * (L->M[0] * lp[0] + L->M[1] * lp[1]) + lp[1] * (L->M[2] * lp[0] + L->M[3] * lp[1]);
    *l = expf(-q / 2.0f);
    *nl = sqrtf(lP[0] * lP[0] + lP[1] * lP[1] + lP[2] * lP[2]);
    *z = sqrtf(cP[0] * cP[0] + cP[1] * cP[1] + cP[2] * cP[2]) + *nl;
}

/* SUCRe.update_J with the light model (sucre.py:66-77): absorption = l a, backscatter = l B (1-g) */
int oracle_update_J_light(int H, int W, int n_samples, const int64_t *counts, const int16_t *const *us,
                          const int16_t *const *vs, const float *const *cPs, const float *const *Is,
                          const float *params, float *J) {
    const size_t npx = (size_t)H * W * 3;
    float *num = (float *)calloc(npx, sizeof(float)), *den = (float *)calloc(npx, sizeof(float));
    if (!num || !den) { free(num); free(den); return -1; }
    light_t L;
    light_setup(params, &L);
    const float *B = params, *beta = params + 3, *gamma = params + 6;
    for (int s = 0; s < n_samples; ++s) {
        const int64_t n = counts[s];
        for (int64_t i = 0; i < n; ++i) {
            const float cP[3] = {cPs[s][i], cPs[s][n + i], cPs[s][2 * n + i]};
            float l, z, lP[3], lp[2], nl;
            light_eval(&L, cP, &l, &z, lP, lp, &nl);
            const size_t px = ((size_t)vs[s][i] * W + us[s][i]) * 3;
            for (int c = 0; c < 3; ++c) {
                const float a = l * expf(-beta[c] * z);
                const float b = l * B[c] * (1.0f - expf(-gamma[c] * z));
                num[px + c] += (Is[s][c * n + i] - b) * a;
                den[px + c] += a * a;
            }
        }
    }
    for (size_t i = 0; i < npx; ++i) J[i] = num[i] / den[i];
    free(num); free(den);
    return 0;
}

/*
 * sucre.adam with light_model=True (sucre.py:124-157); use_closed_form as in oracle_fit.  params: 19 floats in/out.
 * trace: num_iter x 20 doubles (cost, 19 parameters after the step).
 */
int oracle_fit_light(int H, int W, int n_samples, const int64_t *counts, const int16_t *const *us,
                     const int16_t *const *vs, const float *const *cPs, const float *const *Is, float *J,
                     float *params, int num_iter, double lr, int use_closed_form, double *trace) {
    const size_t npx = (size_t)H * W * 3;
    int64_t n_obs = 0;
    for (int s = 0; s < n_samples; ++s) n_obs += counts[s];
    float *gJ = (float *)calloc(npx, sizeof(float));
    float *mJ = (float *)calloc(npx, sizeof(float)), *vJ = (float *)calloc(npx, sizeof(float));
    if (!gJ || !mJ || !vJ) return -1;
    float mP[19] = {0}, vP[19] = {0};
    const float scale = (1.0f / 3.0f) / (float)n_obs;
    float *B = params, *beta = params + 3, *gamma = params + 6;
    for (int it = 0; it < num_iter; ++it) {
        if (use_closed_form && oracle_update_J_light(H, W, n_samples, counts, us, vs, cPs, Is, params, J) != 0) return -1;
        light_t L;
        light_setup(params, &L);
        memset(gJ, 0, sizeof(float) * npx);
        double g[19] = {0}, dR[9] = {0}, dt[3] = {0}, dM[4] = {0}, cost = 0.0;
        for (int s = 0; s < n_samples; ++s) {
            const int64_t n = counts[s];
            for (int64_t i = 0; i < n; ++i) {
                const float cP[3] = {cPs[s][i], cPs[s][n + i], cPs[s][2 * n + i]};
                float l, z, lP[3], lp[2], nl;
                light_eval(&L, cP, &l, &z, lP, lp, &nl);
                const size_t px = ((size_t)vs[s][i] * W + us[s][i]) * 3;
                double dl = 0.0, dz = 0.0;
                for (int c = 0; c < 3; ++c) {
                    const float a = expf(-beta[c] * z), gg = expf(-gamma[c] * z);
                    const float Jc = J[px + c];
                    const float E = Jc * a + B[c] * (1.0f - gg);
                    const float r = Is[s][c * n + i] - l * E;
                    const float dI = -2.0f * r * scale;
                    cost += (double)r * (double)r;
                    gJ[px + c] += dI * l * a;
                    g[c] += (double)(dI * l * (1.0f - gg));
                    g[3 + c] += (double)(dI * l * Jc * a * -z);
                    g[6 + c] += (double)(dI * l * B[c] * gg * z);
                    dl += (double)dI * (double)E;
                    dz += (double)dI * (double)l * ((double)(-beta[c] * Jc * a) + (double)(gamma[c] * B[c] * gg));
                }
                /* through l = exp(-lp^T M lp / 2) and z = ... + ||lP|| */
                const double Mlp0 = (double)L.M[0] * lp[0] + (double)L.M[1] * lp[1];
                const double Mtlp0 = (double)L.M[0] * lp[0] + (double)L.M[2] * lp[1];   /* (M + M^T) lp / 2 halves */
                const double Mlp1 = (double)L.M[2] * lp[0] + (double)L.M[3] * lp[1];
                const double Mtlp1 = (double)L.M[1] * lp[0] + (double)L.M[3] * lp[1];
                const double dlp0 = dl * (-(double)l / 2.0) * (Mlp0 + Mtlp0);
                const double dlp1 = dl * (-(double)l / 2.0) * (Mlp1 + Mtlp1);
                const double Z = lP[2];
                double dlP[3];
                dlP[0] = dz * lP[0] / nl + dlp0 / Z;
                dlP[1] = dz * lP[1] / nl + dlp1 / Z;
                dlP[2] = dz * lP[2] / nl - (dlp0 * lP[0] + dlp1 * lP[1]) / (Z * Z);
                for (int a = 0; a < 3; ++a) {
                    dt[a] += dlP[a];
                    for (int b = 0; b < 3; ++b) dR[a * 3 + b] += dlP[a] * (double)cP[b];
                }
                const double k = dl * (-(double)l / 2.0);
                dM[0] += k * lp[0] * lp[0]; dM[1] += k * lp[0] * lp[1];
                dM[2] += k * lp[1] * lp[0]; dM[3] += k * lp[1] * lp[1];
            }
        }
        /* cam2light */
        double xi[6];
        for (int i = 0; i < 6; ++i) xi[i] = params[9 + i];
        se3_exp_backward(xi, dR, dt, g + 9);
        /* sigma: M = Sigma^-1, Sigma = sigma^T sigma */
        {
            const double M[4] = {L.M[0], L.M[1], L.M[2], L.M[3]};
            /* dSigma = -M^T dM M^T */
            double Mt[4] = {M[0], M[2], M[1], M[3]}, T1[4], dS[4];
            mat_mul(Mt, dM, T1, 2);
            mat_mul(T1, Mt, dS, 2);
            for (int i = 0; i < 4; ++i) dS[i] = -dS[i];
            const double sg[4] = {params[15], params[16], params[17], params[18]};
            const double sym[4] = {2 * dS[0], dS[1] + dS[2], dS[1] + dS[2], 2 * dS[3]};
            mat_mul(sg, sym, g + 15, 2);   /* dsigma = sigma (dSigma + dSigma^T) */
        }
        const adam_coef_t co = adam_coef(it + 1, lr, 0.9, 0.999, 1e-8);
        for (int k = 0; k < 19; ++k) adam_step(&params[k], &mP[k], &vP[k], (float)g[k], &co);
        if (!use_closed_form)
            for (size_t i = 0; i < npx; ++i) adam_step(&J[i], &mJ[i], &vJ[i], gJ[i], &co);
        if (trace) {
            trace[it * 20] = cost;
            for (int k = 0; k < 19; ++k) trace[it * 20 + 1 + k] = (double)params[k];
        }
    }
    free(gJ); free(mJ); free(vJ);
    if (use_closed_form) return oracle_update_J_light(H, W, n_samples, counts, us, vs, cPs, Is, params, J);
    return 0;
}
