// extern "C" surface of libsucre_hip.so (include/sucre_hip.h): argument validation, error strings, launch order.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "launch.h"

namespace sucre {

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return SUCRE_OK;
    return fail(SUCRE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

static bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

static int check_ws(const void *ws, int H, int W, int n_views, Layout *L) {
    if (!make_layout(H, W, n_views, L))
        return fail(SUCRE_ERR_ARG, "invalid geometry H=%d W=%d n_views=%d (need 1..32767, 1..%d views)", H, W, n_views,
                    kMaxViews);
    if (!ws) return fail(SUCRE_ERR_ARG, "workspace is NULL");
    if (!aligned(ws, 256)) return fail(SUCRE_ERR_ARG, "workspace must be 256-byte aligned");
    return SUCRE_OK;
}

// torch/optim/adam.py: bias_correction1 = 1 - beta1 ** step; step_size = lr / bias_correction1;
// bias_correction2_sqrt = (1 - beta2 ** step) ** 0.5  -- Python floats (double), cast to float32 at use.
AdamCoef adam_coef(int step, double lr, double beta1, double beta2, double eps) {
    AdamCoef c;
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    c.w1 = (float)(1.0 - beta1);
    c.beta2 = (float)beta2;
    c.w2 = (float)(1.0 - beta2);
    c.step_size_neg = (float)(-(lr / bc1));
    c.bc2_sqrt = (float)std::sqrt(bc2);
    c.eps = (float)eps;
    return c;
}

static int check_adam(int step, double lr, double beta1, double beta2, double eps) {
    if (step < 1) return fail(SUCRE_ERR_RANGE, "Adam step must be >= 1 (got %d)", step);
    if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0))
        return fail(SUCRE_ERR_ARG, "invalid Adam hyper-parameters lr=%g betas=(%g,%g) eps=%g", lr, beta1, beta2, eps);
    return SUCRE_OK;
}

}  // namespace sucre

using namespace sucre;

extern "C" {

int sucre_version(void) { return SUCRE_ABI_VERSION; }

const char *sucre_last_error(void) { return g_err; }

size_t sucre_workspace_bytes(int H, int W, int n_views) {
    Layout L;
    if (!make_layout(H, W, n_views, &L)) {
        fail(SUCRE_ERR_ARG, "invalid geometry H=%d W=%d n_views=%d", H, W, n_views);
        return 0;
    }
    return L.total;
}

int64_t sucre_ws_offset(int H, int W, int n_views, int region) {
    Layout L;
    if (!make_layout(H, W, n_views, &L)) return fail(SUCRE_ERR_ARG, "invalid geometry H=%d W=%d n_views=%d", H, W, n_views);
    switch (region) {
        case SUCRE_WS_VIEW_COUNT: return (int64_t)L.off_view_count;
        case SUCRE_WS_VIEW_KEEP: return (int64_t)L.off_view_keep;
        case SUCRE_WS_N_OBS: return (int64_t)L.off_n_obs;
        case SUCRE_WS_PARAMS: return (int64_t)L.off_params;
        case SUCRE_WS_SUMS: return (int64_t)L.off_sums;
        case SUCRE_WS_N_OBS_TOTAL: return (int64_t)L.off_n_obs_total;
        case SUCRE_WS_STORE_FORMAT: return (int64_t)L.off_total_chunks + 8;
        default: return fail(SUCRE_ERR_RANGE, "unknown workspace region %d", region);
    }
}

int sucre_match_views(void *ws, int H, int W, int n_views, const sucre_view_t *target,
                      const sucre_view_t *views_dev, int k0, int k1, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (!target || !views_dev) return fail(SUCRE_ERR_ARG, "target / views_dev is NULL");
    if (!target->depth) return fail(SUCRE_ERR_ARG, "target depth map is NULL");
    if (target->H != H || target->W != W)
        return fail(SUCRE_ERR_ARG, "target is %dx%d but the workspace is laid out for %dx%d", target->W, target->H, W, H);
    if (!aligned(views_dev, 8)) return fail(SUCRE_ERR_ARG, "views_dev must be 8-byte aligned");
    if (k0 < 0 || k1 > n_views || k0 >= k1) return fail(SUCRE_ERR_RANGE, "view range [%d,%d) outside [0,%d)", k0, k1, n_views);
    return check_hip(launch_match(L, static_cast<uint8_t *>(ws), *target, views_dev, k0, k1,
                                  static_cast<hipStream_t>(stream)), "sucre_match_views");
}

int sucre_match_map(int H, int W, int n_views, const sucre_view_t *target, const sucre_view_t *views_dev, int k,
                    int32_t *map_dev, void *stream) {
    Layout L;
    if (!make_layout(H, W, n_views, &L)) return fail(SUCRE_ERR_ARG, "invalid geometry H=%d W=%d n_views=%d", H, W, n_views);
    if (!target || !views_dev || !map_dev) return fail(SUCRE_ERR_ARG, "target / views_dev / map_dev is NULL");
    if (!target->depth) return fail(SUCRE_ERR_ARG, "target depth map is NULL");
    if (target->H != H || target->W != W) return fail(SUCRE_ERR_ARG, "target is %dx%d, expected %dx%d", target->W, target->H, W, H);
    if (k < 0 || k >= n_views) return fail(SUCRE_ERR_RANGE, "view %d outside [0,%d)", k, n_views);
    return check_hip(launch_match_map(L, *target, views_dev, k, map_dev, static_cast<hipStream_t>(stream)), "sucre_match_map");
}

int sucre_pack_view(const float *depth_dev, const uint8_t *rgb_dev, int H, int W, void *packed_dev, void *stream) {
    if (H <= 0 || W <= 0 || H > 32767 || W > 32767) return fail(SUCRE_ERR_ARG, "invalid image size %dx%d", W, H);
    if (!depth_dev || !rgb_dev || !packed_dev) return fail(SUCRE_ERR_ARG, "depth_dev / rgb_dev / packed_dev is NULL");
    if (!aligned(packed_dev, 8)) return fail(SUCRE_ERR_ARG, "packed_dev must be 8-byte aligned");
    return check_hip(launch_pack_view(depth_dev, rgb_dev, H, W, packed_dev, static_cast<hipStream_t>(stream)), "sucre_pack_view");
}

int sucre_pack_views(const float *const *depth_dev, const uint8_t *const *rgb_dev, void *const *packed_dev, int n, int H, int W,
                     void *stream) {
    if (H <= 0 || W <= 0 || H > 32767 || W > 32767) return fail(SUCRE_ERR_ARG, "invalid image size %dx%d", W, H);
    if (n < 0) return fail(SUCRE_ERR_RANGE, "negative view count %d", n);
    if (n == 0) return SUCRE_OK;
    if (!depth_dev || !rgb_dev || !packed_dev) return fail(SUCRE_ERR_ARG, "depth_dev / rgb_dev / packed_dev is NULL");
    for (int k = 0; k < n; ++k) {
        if (!depth_dev[k] || !rgb_dev[k] || !packed_dev[k]) return fail(SUCRE_ERR_ARG, "view %d: depth / rgb / packed pointer is NULL", k);
        if (!aligned(packed_dev[k], 8)) return fail(SUCRE_ERR_ARG, "view %d: packed_dev must be 8-byte aligned", k);
    }
    return check_hip(launch_pack_views(depth_dev, rgb_dev, packed_dev, n, H, W, static_cast<hipStream_t>(stream)), "sucre_pack_views");
}

int sucre_project_points(const sucre_view_t *view, const float *wP_dev, int64_t n, int32_t *pix_dev, void *stream) {
    if (!view) return fail(SUCRE_ERR_ARG, "view is NULL");
    if (view->H <= 0 || view->W <= 0 || view->H > 32767 || view->W > 32767)
        return fail(SUCRE_ERR_ARG, "invalid sensor size %dx%d", view->W, view->H);
    if (n < 0) return fail(SUCRE_ERR_RANGE, "negative point count %lld", (long long)n);
    if (n == 0) return SUCRE_OK;   // nothing to launch
    if (!wP_dev || !pix_dev) return fail(SUCRE_ERR_ARG, "wP_dev / pix_dev is NULL");
    return check_hip(launch_project_points(*view, wP_dev, (long long)n, pix_dev, static_cast<hipStream_t>(stream)),
                     "sucre_project_points");
}

int sucre_import_view(void *ws, int H, int W, int n_views, int k, const int16_t *u1_dev, const int16_t *v1_dev,
                      const float *z_dev, const uint8_t *rgb_dev, int64_t n, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (k < 0 || k >= n_views) return fail(SUCRE_ERR_RANGE, "view %d outside [0,%d)", k, n_views);
    if (n < 0) return fail(SUCRE_ERR_RANGE, "negative observation count %lld", (long long)n);
    if (n > 0 && (!u1_dev || !v1_dev || !z_dev || !rgb_dev)) return fail(SUCRE_ERR_ARG, "NULL match list");
    return check_hip(launch_import_view(L, static_cast<uint8_t *>(ws), k, u1_dev, v1_dev, z_dev, rgb_dev, (long long)n,
                                        static_cast<hipStream_t>(stream)), "sucre_import_view");
}

int sucre_finalize_matches(void *ws, int H, int W, int n_views, double min_cover, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (std::isnan(min_cover)) return fail(SUCRE_ERR_ARG, "min_cover is NaN");
    return check_hip(launch_finalize(L, static_cast<uint8_t *>(ws), min_cover, static_cast<hipStream_t>(stream)),
                     "sucre_finalize_matches");
}

int sucre_finalize_matches_fmt(void *ws, int H, int W, int n_views, double min_cover, int obs_format, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (std::isnan(min_cover)) return fail(SUCRE_ERR_ARG, "min_cover is NaN");
    if (obs_format != SUCRE_OBS_F32 && obs_format != SUCRE_OBS_U16MM && obs_format != SUCRE_OBS_F32_PLAIN && obs_format != SUCRE_OBS_F32_Z26)
        return fail(SUCRE_ERR_ARG, "unknown observation format %d", obs_format);
    return check_hip(launch_finalize(L, static_cast<uint8_t *>(ws), min_cover, static_cast<hipStream_t>(stream),
                                     nullptr, nullptr, obs_format), "sucre_finalize_matches_fmt");
}

int sucre_fit_init(void *ws, int H, int W, int n_views, const uint8_t *rgb1_dev, const float *depth1_dev,
                   const float *params0, const float *J0_dev, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (!depth1_dev || !params0) return fail(SUCRE_ERR_ARG, "depth1_dev / params0 is NULL");
    if (!rgb1_dev && !J0_dev) return fail(SUCRE_ERR_ARG, "need rgb1_dev or J0_dev");
    return check_hip(launch_fit_init(L, static_cast<uint8_t *>(ws), rgb1_dev, depth1_dev, params0, J0_dev,
                                     static_cast<hipStream_t>(stream)), "sucre_fit_init");
}

int sucre_fit_grad(void *ws, int H, int W, int n_views, int step, double lr, double beta1, double beta2,
                   double eps, unsigned flags, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_adam(step, lr, beta1, beta2, eps)) return rc;
    if (flags & ~(SUCRE_FIT_CLOSED_FORM | SUCRE_FIT_OBS_U16MM)) return fail(SUCRE_ERR_ARG, "unknown fit flags 0x%x", flags);
    if ((flags & SUCRE_FIT_CLOSED_FORM) && step == 1)  // see sucre_fit_run: start the one-pass kernel from a solved J
        if (int rc = check_hip(launch_update_J(L, static_cast<uint8_t *>(ws), (flags & SUCRE_FIT_OBS_U16MM) ? SUCRE_OBS_U16MM : SUCRE_OBS_F32,
                                               static_cast<hipStream_t>(stream)), "sucre_fit_grad/update_J")) return rc;
    return check_hip(launch_fit_grad(L, static_cast<uint8_t *>(ws), adam_coef(step, lr, beta1, beta2, eps), flags,
                                     static_cast<hipStream_t>(stream)), "sucre_fit_grad");
}

int sucre_fit_step(void *ws, int H, int W, int n_views, int step, double lr, double beta1, double beta2,
                   double eps, double *trace_row_dev, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_adam(step, lr, beta1, beta2, eps)) return rc;
    if (trace_row_dev && !aligned(trace_row_dev, 8)) return fail(SUCRE_ERR_ARG, "trace must be 8-byte aligned");
    return check_hip(launch_fit_step(L, static_cast<uint8_t *>(ws), adam_coef(step, lr, beta1, beta2, eps),
                                     trace_row_dev, static_cast<hipStream_t>(stream)), "sucre_fit_step");
}

int sucre_fit_run(void *ws, int H, int W, int n_views, int t0, int T, double lr, double beta1, double beta2,
                  double eps, unsigned flags, double *trace_dev, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (t0 < 0 || T < 0) return fail(SUCRE_ERR_RANGE, "t0=%d T=%d must be >= 0", t0, T);
    if (int rc = check_adam(t0 + 1, lr, beta1, beta2, eps)) return rc;
    if (flags & ~(SUCRE_FIT_CLOSED_FORM | SUCRE_FIT_OBS_U16MM | SUCRE_FIT_KEEP_J)) return fail(SUCRE_ERR_ARG, "unknown fit flags 0x%x", flags);
    if (trace_dev && !aligned(trace_dev, 8)) return fail(SUCRE_ERR_ARG, "trace must be 8-byte aligned");
    auto *w = static_cast<uint8_t *>(ws);
    auto s = static_cast<hipStream_t>(stream);
    const int fmt = (flags & SUCRE_FIT_OBS_U16MM) ? SUCRE_OBS_U16MM : SUCRE_OBS_F32;
    // The one-pass closed-form kernel forms its sums relative to the previous J (fit.hip, AccOne): at t0 = 0 that is
    // the initial image, far from the solved J, and the differences S - dJ S' cancel digits in iteration 0 (seen as
    // 2.6e-5 relative on the logged cost at 4K x 25 views).  One plain update_J first makes dJ small from the start.
    if ((flags & SUCRE_FIT_CLOSED_FORM) && t0 == 0 && T > 0)
        if (int rc = check_hip(launch_update_J(L, w, fmt, s), "sucre_fit_run/initial update_J")) return rc;
    for (int it = 0; it < T; ++it) {
        const AdamCoef co = adam_coef(t0 + it + 1, lr, beta1, beta2, eps);
        if (int rc = check_hip(launch_fit_iter_fused(L, w, co, flags, trace_dev ? trace_dev + (size_t)it * 10 : nullptr, s),
                               "sucre_fit_run")) return rc;
    }
    if ((flags & SUCRE_FIT_CLOSED_FORM) && !(flags & SUCRE_FIT_KEEP_J))
        return check_hip(launch_update_J(L, w, fmt, s), "sucre_fit_run/update_J");
    return SUCRE_OK;
}

int sucre_set_n_obs_total(void *ws, int H, int W, int n_views, uint64_t n_obs_total, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (n_obs_total == 0) return fail(SUCRE_ERR_RANGE, "n_obs_total must be > 0");
    return check_hip(launch_set_n_obs_total(L, static_cast<uint8_t *>(ws), n_obs_total,
                                            static_cast<hipStream_t>(stream)), "sucre_set_n_obs_total");
}

int sucre_update_J(void *ws, int H, int W, int n_views, void *stream) {
    return sucre_update_J_fmt(ws, H, W, n_views, SUCRE_OBS_F32, stream);
}

int sucre_update_J_fmt(void *ws, int H, int W, int n_views, int obs_format, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (obs_format != SUCRE_OBS_F32 && obs_format != SUCRE_OBS_U16MM && obs_format != SUCRE_OBS_F32_PLAIN && obs_format != SUCRE_OBS_F32_Z26)
        return fail(SUCRE_ERR_ARG, "unknown observation format %d", obs_format);
    return check_hip(launch_update_J(L, static_cast<uint8_t *>(ws), obs_format == SUCRE_OBS_U16MM ? SUCRE_OBS_U16MM : SUCRE_OBS_F32,
                                     static_cast<hipStream_t>(stream)),
                     "sucre_update_J");
}

int sucre_export_J(const void *ws, int H, int W, int n_views, float *J_dev, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (!J_dev) return fail(SUCRE_ERR_ARG, "J_dev is NULL");
    return check_hip(launch_export_J(L, static_cast<const uint8_t *>(ws), J_dev, static_cast<hipStream_t>(stream)),
                     "sucre_export_J");
}

int sucre_export_view(const void *ws, int H, int W, int n_views, int k, float *z_dev, uint8_t *rgb_dev,
                      void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (k < 0 || k >= n_views) return fail(SUCRE_ERR_RANGE, "view %d outside [0,%d)", k, n_views);
    if (!z_dev && !rgb_dev) return fail(SUCRE_ERR_ARG, "both outputs are NULL");
    return check_hip(launch_export_view(L, static_cast<const uint8_t *>(ws), k, z_dev, rgb_dev,
                                        static_cast<hipStream_t>(stream)), "sucre_export_view");
}

int sucre_check_store(const void *ws, int H, int W, int n_views, uint32_t *verdict_dev, uint64_t *scratch_dev,
                      void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (!verdict_dev || !scratch_dev) return fail(SUCRE_ERR_ARG, "verdict_dev / scratch_dev is NULL");
    if (!aligned(verdict_dev, 4) || !aligned(scratch_dev, 8)) return fail(SUCRE_ERR_ARG, "outputs must be 4 / 8-byte aligned");
    return check_hip(launch_check_store(L, static_cast<const uint8_t *>(ws), verdict_dev, scratch_dev,
                                        static_cast<hipStream_t>(stream)), "sucre_check_store");
}

/* ---- shared water parameters over the images of a rank: one launch + one collective per iteration -------------- */

size_t sucre_group_bytes(int n_images) { return n_images > 0 ? group_bytes(n_images) : 0; }

int64_t sucre_group_sums_offset(void) { return group_sums_offset(); }

int sucre_group_init(void *group_dev, int n_images, const sucre_group_image_t *images, const float *params0, void *stream) {
    if (!group_dev || !aligned(group_dev, 256)) return fail(SUCRE_ERR_ARG, "group buffer is NULL or not 256-byte aligned");
    if (n_images < 1 || !images || !params0) return fail(SUCRE_ERR_ARG, "need at least one image and params0");
    auto s = static_cast<hipStream_t>(stream);
    if (int rc = check_hip(launch_group_init(group_dev, params0, s), "sucre_group_init")) return rc;
    for (int i = 0; i < n_images; ++i) {
        Layout L;
        if (int rc = check_ws(images[i].ws, images[i].H, images[i].W, images[i].n_views, &L)) return rc;
        if (int rc = check_hip(launch_group_set_image(group_dev, i, L, static_cast<uint8_t *>(images[i].ws), s), "sucre_group_init")) return rc;
    }
    return SUCRE_OK;
}

int sucre_group_iter(void *group_dev, int n_images, int step, double lr, double beta1, double beta2, double eps, unsigned flags,
                     uint64_t n_obs_total, double *trace_dev, void *stream) {
    if (!group_dev || n_images < 1) return fail(SUCRE_ERR_ARG, "group buffer is NULL / no image");
    if (int rc = check_adam(step, lr, beta1, beta2, eps)) return rc;
    if (flags & ~(SUCRE_FIT_CLOSED_FORM | SUCRE_FIT_OBS_U16MM)) return fail(SUCRE_ERR_ARG, "unknown fit flags 0x%x", flags);
    if (n_obs_total == 0) return fail(SUCRE_ERR_RANGE, "n_obs_total must be > 0");
    if (trace_dev && !aligned(trace_dev, 8)) return fail(SUCRE_ERR_ARG, "trace must be 8-byte aligned");
    const AdamCoef co_prev = adam_coef(step > 1 ? step - 1 : 1, lr, beta1, beta2, eps);
    double *row = (trace_dev && step > 1) ? trace_dev + (size_t)(step - 2) * 10 : nullptr;
    return check_hip(launch_group_iter(group_dev, n_images, step, co_prev, adam_coef(step, lr, beta1, beta2, eps), flags, n_obs_total,
                                       row, static_cast<hipStream_t>(stream)), "sucre_group_iter");
}

int sucre_group_finish(void *group_dev, int n_images, int step, double lr, double beta1, double beta2, double eps,
                       uint64_t n_obs_total, double *trace_dev, void *stream) {
    if (!group_dev || n_images < 1) return fail(SUCRE_ERR_ARG, "group buffer is NULL / no image");
    if (step < 0) return fail(SUCRE_ERR_RANGE, "step=%d must be >= 0", step);
    if (step >= 1) if (int rc = check_adam(step, lr, beta1, beta2, eps)) return rc;
    if (n_obs_total == 0) return fail(SUCRE_ERR_RANGE, "n_obs_total must be > 0");
    double *row = (trace_dev && step >= 1) ? trace_dev + (size_t)(step - 1) * 10 : nullptr;
    return check_hip(launch_group_finish(group_dev, n_images, step, adam_coef(step >= 1 ? step : 1, lr, beta1, beta2, eps), n_obs_total,
                                         row, static_cast<hipStream_t>(stream)), "sucre_group_finish");
}

/* ---- independent images, one launch per iteration ---------------------------------------------------------------- */

size_t sucre_batch_bytes(int n_images) { return n_images > 0 ? batch_bytes(n_images) : 0; }

int sucre_fit_run_batch(void *batch_dev, int n_images, void *const *ws, double *const *trace_dev, int H, int W, const int *n_views,
                        int t0, int T, double lr, double beta1, double beta2, double eps, unsigned flags, void *stream) {
    if (!batch_dev || !aligned(batch_dev, 256)) return fail(SUCRE_ERR_ARG, "batch buffer is NULL or not 256-byte aligned");
    if (n_images < 1 || !ws || !n_views) return fail(SUCRE_ERR_ARG, "need at least one image (ws and n_views arrays)");
    if (n_images > 4096) return fail(SUCRE_ERR_RANGE, "n_images=%d: at most 4096 images per batch", n_images);
    std::vector<Layout> layouts((size_t)n_images);
    for (int i = 0; i < n_images; ++i) {
        if (int rc = check_ws(ws[i], H, W, n_views[i], &layouts[(size_t)i])) return rc;
        for (int j = 0; j < i; ++j)
            if (ws[j] == ws[i]) return fail(SUCRE_ERR_ARG, "images %d and %d share a workspace", j, i);
        if (trace_dev && trace_dev[i] && !aligned(trace_dev[i], 8)) return fail(SUCRE_ERR_ARG, "trace of image %d must be 8-byte aligned", i);
    }
    if (t0 < 0 || T < 0) return fail(SUCRE_ERR_RANGE, "t0=%d T=%d must be >= 0", t0, T);
    if (int rc = check_adam(t0 + 1, lr, beta1, beta2, eps)) return rc;
    if (flags & ~(SUCRE_FIT_CLOSED_FORM | SUCRE_FIT_OBS_U16MM | SUCRE_FIT_KEEP_J)) return fail(SUCRE_ERR_ARG, "unknown fit flags 0x%x", flags);
    auto s = static_cast<hipStream_t>(stream);
    const int fmt = (flags & SUCRE_FIT_OBS_U16MM) ? SUCRE_OBS_U16MM : SUCRE_OBS_F32;
    if (int rc = check_hip(launch_batch_set(batch_dev, n_images, reinterpret_cast<uint8_t *const *>(ws), trace_dev, layouts.data(), flags, s),
                           "sucre_fit_run_batch/table")) return rc;
    if ((flags & SUCRE_FIT_CLOSED_FORM) && t0 == 0 && T > 0)   // as sucre_fit_run: the one-pass kernel starts from a solved J
        for (int i = 0; i < n_images; ++i)
            if (int rc = check_hip(launch_update_J(layouts[(size_t)i], static_cast<uint8_t *>(ws[i]), fmt, s), "sucre_fit_run_batch/initial update_J")) return rc;
    for (int it = 0; it < T; ++it)
        if (int rc = check_hip(launch_batch_iter(layouts[0], batch_dev, n_images, adam_coef(t0 + it + 1, lr, beta1, beta2, eps), flags, it, s),
                               "sucre_fit_run_batch")) return rc;
    if ((flags & SUCRE_FIT_CLOSED_FORM) && !(flags & SUCRE_FIT_KEEP_J))
        for (int i = 0; i < n_images; ++i)
            if (int rc = check_hip(launch_update_J(layouts[(size_t)i], static_cast<uint8_t *>(ws[i]), fmt, s), "sucre_fit_run_batch/update_J")) return rc;
    return SUCRE_OK;
}

size_t sucre_select_scratch_bytes(void) { return select_scratch_bytes(); }

int sucre_select_ranks(const float *J_dev, int H, int W, int n_ranks, const uint64_t *ranks, float *out_dev, void *scratch_dev,
                       void *stream) {
    if (H <= 0 || W <= 0) return fail(SUCRE_ERR_ARG, "invalid image size %dx%d", W, H);
    if (!J_dev || !ranks || !out_dev || !scratch_dev) return fail(SUCRE_ERR_ARG, "J_dev / ranks / out_dev / scratch_dev is NULL");
    if (n_ranks < 1 || n_ranks > 8) return fail(SUCRE_ERR_RANGE, "n_ranks=%d outside [1,8]", n_ranks);
    if (!aligned(scratch_dev, 8) || !aligned(J_dev, 4) || !aligned(out_dev, 4)) return fail(SUCRE_ERR_ARG, "misaligned pointer");
    return check_hip(launch_select_ranks(J_dev, H, W, n_ranks, ranks, out_dev, scratch_dev, static_cast<hipStream_t>(stream)),
                     "sucre_select_ranks");
}

int sucre_count_valid(const float *J_dev, int H, int W, uint64_t *count_dev, void *stream) {
    if (H <= 0 || W <= 0) return fail(SUCRE_ERR_ARG, "invalid image size %dx%d", W, H);
    if (!J_dev || !count_dev) return fail(SUCRE_ERR_ARG, "J_dev / count_dev is NULL");
    if (!aligned(J_dev, 4) || !aligned(count_dev, 8)) return fail(SUCRE_ERR_ARG, "misaligned pointer");
    return check_hip(launch_count_valid(J_dev, H, W, count_dev, static_cast<hipStream_t>(stream)), "sucre_count_valid");
}

int sucre_plot_stretch(const float *J_dev, int H, int W, const float *lo, const float *hi, uint8_t *out_dev, void *stream) {
    if (H <= 0 || W <= 0) return fail(SUCRE_ERR_ARG, "invalid image size %dx%d", W, H);
    if (!J_dev || !lo || !hi || !out_dev) return fail(SUCRE_ERR_ARG, "J_dev / lo / hi / out_dev is NULL");
    if (!aligned(J_dev, 4)) return fail(SUCRE_ERR_ARG, "misaligned pointer");
    return check_hip(launch_plot_stretch(J_dev, H, W, lo, hi, out_dev, static_cast<hipStream_t>(stream)), "sucre_plot_stretch");
}

/* ---- artificial-light model (--light-model) ------------------------------------------------------------------ */

size_t sucre_light_workspace_bytes(int H, int W, int n_views) {
    Layout L;
    if (!make_layout(H, W, n_views, &L)) { fail(SUCRE_ERR_ARG, "invalid geometry H=%d W=%d n_views=%d", H, W, n_views); return 0; }
    return light_workspace_bytes(L);
}

size_t sucre_light_workspace_bytes_ext(int H, int W, int n_views, int ext_mode) {
    Layout L;
    if (!make_layout(H, W, n_views, &L)) { fail(SUCRE_ERR_ARG, "invalid geometry H=%d W=%d n_views=%d", H, W, n_views); return 0; }
    if (ext_mode < SUCRE_EXT_POINTS || ext_mode > SUCRE_EXT_POINTS_COLOUR) { fail(SUCRE_ERR_ARG, "unknown extension mode %d", ext_mode); return 0; }
    return light_workspace_bytes(L, ext_mode == SUCRE_EXT_POINTS_COLOUR ? 2 : 1);
}

int64_t sucre_light_params_offset(int H, int W, int n_views) {
    Layout L;
    if (!make_layout(H, W, n_views, &L)) return fail(SUCRE_ERR_ARG, "invalid geometry H=%d W=%d n_views=%d", H, W, n_views);
    return light_params_offset(L);
}

static int check_lws(const void *lws) {
    if (!lws) return fail(SUCRE_ERR_ARG, "light workspace is NULL");
    if (!aligned(lws, 256)) return fail(SUCRE_ERR_ARG, "light workspace must be 256-byte aligned");
    return SUCRE_OK;
}

int sucre_match_views_light(void *ws, void *lws, int H, int W, int n_views, const sucre_view_t *target,
                            const sucre_view_t *views_dev, int k0, int k1, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (!target || !views_dev || !target->depth) return fail(SUCRE_ERR_ARG, "target / views_dev / target depth is NULL");
    if (target->H != H || target->W != W) return fail(SUCRE_ERR_ARG, "target is %dx%d, expected %dx%d", target->W, target->H, W, H);
    if (k0 < 0 || k1 > n_views || k0 >= k1) return fail(SUCRE_ERR_RANGE, "view range [%d,%d) outside [0,%d)", k0, k1, n_views);
    return check_hip(launch_match(L, static_cast<uint8_t *>(ws), *target, views_dev, k0, k1, static_cast<hipStream_t>(stream),
                                  light_ext_dense(L, static_cast<uint8_t *>(lws))), "sucre_match_views_light");
}

int sucre_match_views_fcolour(void *ws, void *lws, int H, int W, int n_views, const sucre_view_t *target,
                              const sucre_view_t *views_dev, int k0, int k1, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (!target || !views_dev || !target->depth) return fail(SUCRE_ERR_ARG, "target / views_dev / target depth is NULL");
    if (target->H != H || target->W != W) return fail(SUCRE_ERR_ARG, "target is %dx%d, expected %dx%d", target->W, target->H, W, H);
    if (k0 < 0 || k1 > n_views || k0 >= k1) return fail(SUCRE_ERR_RANGE, "view range [%d,%d) outside [0,%d)", k0, k1, n_views);
    return check_hip(launch_match(L, static_cast<uint8_t *>(ws), *target, views_dev, k0, k1, static_cast<hipStream_t>(stream),
                                  light_ext_dense(L, static_cast<uint8_t *>(lws)), SUCRE_EXT_COLOUR), "sucre_match_views_fcolour");
}

int sucre_match_views_light_fcolour(void *ws, void *lws, int H, int W, int n_views, const sucre_view_t *target,
                                    const sucre_view_t *views_dev, int k0, int k1, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (!target || !views_dev || !target->depth) return fail(SUCRE_ERR_ARG, "target / views_dev / target depth is NULL");
    if (target->H != H || target->W != W) return fail(SUCRE_ERR_ARG, "target is %dx%d, expected %dx%d", target->W, target->H, W, H);
    if (k0 < 0 || k1 > n_views || k0 >= k1) return fail(SUCRE_ERR_RANGE, "view range [%d,%d) outside [0,%d)", k0, k1, n_views);
    auto *l = static_cast<uint8_t *>(lws);
    return check_hip(launch_match(L, static_cast<uint8_t *>(ws), *target, views_dev, k0, k1, static_cast<hipStream_t>(stream),
                                  light_ext_dense(L, l), SUCRE_EXT_POINTS_COLOUR, light_ext2_dense(L, l)),
                     "sucre_match_views_light_fcolour");
}

int sucre_finalize_matches_ext(void *ws, void *lws, int H, int W, int n_views, double min_cover, int ext_mode, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (std::isnan(min_cover)) return fail(SUCRE_ERR_ARG, "min_cover is NaN");
    if (ext_mode < SUCRE_EXT_POINTS || ext_mode > SUCRE_EXT_POINTS_COLOUR) return fail(SUCRE_ERR_ARG, "unknown extension mode %d", ext_mode);
    auto *l = static_cast<uint8_t *>(lws);
    const bool both = ext_mode == SUCRE_EXT_POINTS_COLOUR;
    return check_hip(launch_finalize(L, static_cast<uint8_t *>(ws), min_cover, static_cast<hipStream_t>(stream),
                                     light_ext_dense(L, l), light_ext_comp(L, l), SUCRE_OBS_F32,
                                     both ? light_ext2_dense(L, l) : nullptr, both ? light_ext2_comp(L, l) : nullptr),
                     "sucre_finalize_matches_ext");
}

int sucre_import_view_ext(void *ws, void *lws, int H, int W, int n_views, int k, const int16_t *u1_dev, const int16_t *v1_dev,
                          const float *z_dev, const uint8_t *rgb_dev, const float *ext_dev, int64_t n, int ext_mode, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (k < 0 || k >= n_views) return fail(SUCRE_ERR_RANGE, "view %d outside [0,%d)", k, n_views);
    if (n < 0) return fail(SUCRE_ERR_RANGE, "negative observation count %lld", (long long)n);
    if (ext_mode < SUCRE_EXT_POINTS || ext_mode > SUCRE_EXT_POINTS_COLOUR) return fail(SUCRE_ERR_ARG, "unknown extension mode %d", ext_mode);
    if (n > 0 && (!u1_dev || !v1_dev || !z_dev || !ext_dev)) return fail(SUCRE_ERR_ARG, "NULL match list");
    if (n > 0 && ext_mode == SUCRE_EXT_POINTS && !rgb_dev) return fail(SUCRE_ERR_ARG, "camera-point lists need uint8 colours too");
    auto *l = static_cast<uint8_t *>(lws);
    return check_hip(launch_import_view(L, static_cast<uint8_t *>(ws), k, u1_dev, v1_dev, z_dev, rgb_dev, (long long)n,
                                        static_cast<hipStream_t>(stream), light_ext_dense(L, l), ext_dev,
                                        ext_mode == SUCRE_EXT_POINTS_COLOUR ? light_ext2_dense(L, l) : nullptr),
                     "sucre_import_view_ext");
}

int sucre_export_view_ext(const void *ws, const void *lws, int H, int W, int n_views, int k, float *planes_dev, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (k < 0 || k >= n_views) return fail(SUCRE_ERR_RANGE, "view %d outside [0,%d)", k, n_views);
    if (!planes_dev) return fail(SUCRE_ERR_ARG, "planes_dev is NULL");
    return check_hip(launch_export_view_ext(L, static_cast<const uint8_t *>(ws),
                                            light_ext_dense(L, const_cast<uint8_t *>(static_cast<const uint8_t *>(lws))), k,
                                            planes_dev, static_cast<hipStream_t>(stream)), "sucre_export_view_ext");
}

int sucre_finalize_matches_light(void *ws, void *lws, int H, int W, int n_views, double min_cover, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (std::isnan(min_cover)) return fail(SUCRE_ERR_ARG, "min_cover is NaN");
    auto *l = static_cast<uint8_t *>(lws);
    return check_hip(launch_finalize(L, static_cast<uint8_t *>(ws), min_cover, static_cast<hipStream_t>(stream),
                                     light_ext_dense(L, l), light_ext_comp(L, l)), "sucre_finalize_matches_light");
}

int sucre_fit_init_light(void *ws, void *lws, int H, int W, int n_views, const uint8_t *rgb1_dev, const float *depth1_dev,
                         const float *params0, const float *J0_dev, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (!depth1_dev || !params0) return fail(SUCRE_ERR_ARG, "depth1_dev / params0 is NULL");
    if (!rgb1_dev && !J0_dev) return fail(SUCRE_ERR_ARG, "need rgb1_dev or J0_dev");
    auto s = static_cast<hipStream_t>(stream);
    if (int rc = check_hip(launch_fit_init(L, static_cast<uint8_t *>(ws), rgb1_dev, depth1_dev, params0, J0_dev, s),
                           "sucre_fit_init_light/J")) return rc;
    return check_hip(launch_light_init(L, static_cast<uint8_t *>(lws), params0, s), "sucre_fit_init_light");
}

int sucre_update_J_light(void *ws, void *lws, int H, int W, int n_views, void *stream) {
    return sucre_update_J_ext(ws, lws, H, W, n_views, 0u, stream);
}

int sucre_update_J_ext(void *ws, void *lws, int H, int W, int n_views, unsigned flags, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (flags & ~(SUCRE_FIT_EXT_COLOUR | SUCRE_FIT_EXT_BOTH)) return fail(SUCRE_ERR_ARG, "unknown flags 0x%x", flags);
    if ((flags & SUCRE_FIT_EXT_COLOUR) && (flags & SUCRE_FIT_EXT_BOTH)) return fail(SUCRE_ERR_ARG, "SUCRE_FIT_EXT_COLOUR and SUCRE_FIT_EXT_BOTH exclude each other");
    return check_hip(launch_light_update_J(L, static_cast<uint8_t *>(ws), static_cast<uint8_t *>(lws), flags,
                                           static_cast<hipStream_t>(stream)), "sucre_update_J_ext");
}

int sucre_fit_run_light(void *ws, void *lws, int H, int W, int n_views, int t0, int T, double lr, double beta1,
                        double beta2, double eps, unsigned flags, double *trace_dev, void *stream) {
    Layout L;
    if (int rc = check_ws(ws, H, W, n_views, &L)) return rc;
    if (int rc = check_lws(lws)) return rc;
    if (t0 < 0 || T < 0) return fail(SUCRE_ERR_RANGE, "t0=%d T=%d must be >= 0", t0, T);
    if (int rc = check_adam(t0 + 1, lr, beta1, beta2, eps)) return rc;
    if (trace_dev && !aligned(trace_dev, 8)) return fail(SUCRE_ERR_ARG, "trace must be 8-byte aligned");
    if (flags & ~(SUCRE_FIT_CLOSED_FORM | SUCRE_FIT_EXT_COLOUR | SUCRE_FIT_KEEP_J | SUCRE_FIT_EXT_BOTH)) return fail(SUCRE_ERR_ARG, "unknown fit flags 0x%x", flags);
    if ((flags & SUCRE_FIT_EXT_COLOUR) && (flags & SUCRE_FIT_EXT_BOTH)) return fail(SUCRE_ERR_ARG, "SUCRE_FIT_EXT_COLOUR and SUCRE_FIT_EXT_BOTH exclude each other");
    if (T > 0)   // which strips every wave of the gradient launches works on (layout.h, the deal): written once per call
        if (int rc = check_hip(launch_light_deal(L, static_cast<uint8_t *>(ws), static_cast<uint8_t *>(lws), flags,
                                                 static_cast<hipStream_t>(stream)), "sucre_fit_run_light/deal")) return rc;
    for (int it = 0; it < T; ++it) {
        const AdamCoef co = adam_coef(t0 + it + 1, lr, beta1, beta2, eps);
        if (int rc = check_hip(launch_light_iter(L, static_cast<uint8_t *>(ws), static_cast<uint8_t *>(lws), co, flags,
                                                 trace_dev ? trace_dev + (size_t)it * 20 : nullptr,
                                                 static_cast<hipStream_t>(stream)), "sucre_fit_run_light")) return rc;
    }
    if ((flags & SUCRE_FIT_CLOSED_FORM) && !(flags & SUCRE_FIT_KEEP_J))  // the final update_J of sucre.py:156
        return check_hip(launch_light_update_J(L, static_cast<uint8_t *>(ws), static_cast<uint8_t *>(lws),
                                               flags & (SUCRE_FIT_EXT_COLOUR | SUCRE_FIT_EXT_BOTH), static_cast<hipStream_t>(stream)),
                         "sucre_fit_run_light/update_J");
    return SUCRE_OK;
}

}  // extern "C"
