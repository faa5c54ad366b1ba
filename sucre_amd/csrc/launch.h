// Internal launcher interface between abi.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/sucre_hip.h"
#include "layout.h"

namespace sucre {

// Camera of the target image, passed by value as a kernel argument (SGPR-resident).
struct CamDev {
    float K[9], Kinv[9], R[9], t[3], Rinv[9], tinv[3];
    int H, W;
};

// torch.optim.Adam scalars of one step, computed on the host in double like torch does
// (torch/optim/adam.py::_single_tensor_adam, non-capturable branch).
struct AdamCoef {
    float w1;             // 1 - beta1            exp_avg.lerp_(grad, 1 - beta1)
    float beta2;          // beta2                exp_avg_sq.mul_(beta2)
    float w2;             // 1 - beta2            .addcmul_(grad, grad, value=1 - beta2)
    float step_size_neg;  // -(lr / (1 - beta1^t))
    float bc2_sqrt;       // sqrt(1 - beta2^t)
    float eps;
};

AdamCoef adam_coef(int step, double lr, double beta1, double beta2, double eps);

hipError_t launch_match(const Layout &L, uint8_t *ws, const sucre_view_t &target, const sucre_view_t *views_dev,
                        int k0, int k1, hipStream_t s, uint8_t *ext = nullptr, int ext_mode = SUCRE_EXT_POINTS,
                        uint8_t *ext2 = nullptr);
hipError_t launch_match_map(const Layout &L, const sucre_view_t &target, const sucre_view_t *views_dev, int k,
                            int32_t *map, hipStream_t s);
hipError_t launch_pack_view(const float *depth, const uint8_t *rgb, int H, int W, void *packed, hipStream_t s);
hipError_t launch_pack_views(const float *const *depth, const uint8_t *const *rgb, void *const *packed, int n, int H, int W, hipStream_t s);
hipError_t launch_project_points(const sucre_view_t &view, const float *wP, long long n, int32_t *pix, hipStream_t s);
hipError_t launch_import_view(const Layout &L, uint8_t *ws, int k, const int16_t *u1, const int16_t *v1, const float *z,
                              const uint8_t *rgb, long long n, hipStream_t s, uint8_t *ext_dense = nullptr,
                              const float *ext = nullptr, uint8_t *ext2_dense = nullptr);
hipError_t launch_export_view_ext(const Layout &L, const uint8_t *ws, const uint8_t *ext_dense, int k, float *out, hipStream_t s);
hipError_t launch_finalize(const Layout &L, uint8_t *ws, double min_cover, hipStream_t s,
                           const uint8_t *ext_dense = nullptr, uint8_t *ext_comp = nullptr, int fmt = SUCRE_OBS_F32,
                           const uint8_t *ext2_dense = nullptr, uint8_t *ext2_comp = nullptr);
hipError_t launch_compact(const Layout &L, uint8_t *ws, hipStream_t s, const uint8_t *ext_dense = nullptr,
                          uint8_t *ext_comp = nullptr, int fmt = SUCRE_OBS_F32, const uint8_t *ext2_dense = nullptr,
                          uint8_t *ext2_comp = nullptr);
hipError_t launch_check_store(const Layout &L, const uint8_t *ws, uint32_t *verdict, uint64_t *scratch, hipStream_t s);
hipError_t launch_export_view(const Layout &L, const uint8_t *ws, int k, float *z, uint8_t *rgb, hipStream_t s);

hipError_t launch_fit_init(const Layout &L, uint8_t *ws, const uint8_t *rgb1, const float *depth1,
                           const float *params0, const float *J0, hipStream_t s);
hipError_t launch_fit_iter_fused(const Layout &L, uint8_t *ws, const AdamCoef &co, unsigned flags, double *trace_row,
                                 hipStream_t s);
hipError_t launch_fit_grad(const Layout &L, uint8_t *ws, const AdamCoef &co, unsigned flags, hipStream_t s);
hipError_t launch_fit_step(const Layout &L, uint8_t *ws, const AdamCoef &co, double *trace_row, hipStream_t s);
hipError_t launch_update_J(const Layout &L, uint8_t *ws, int fmt, hipStream_t s);
hipError_t launch_plan(const Layout &L, uint8_t *ws, hipStream_t s);
hipError_t launch_export_J(const Layout &L, const uint8_t *ws, float *J, hipStream_t s);
hipError_t launch_set_n_obs_total(const Layout &L, uint8_t *ws, uint64_t n, hipStream_t s);
// shared water parameters over several images: one launch + one collective per iteration (fit.hip)
size_t group_bytes(int n_images);
int64_t group_sums_offset();
hipError_t launch_group_init(void *group, const float *params0, hipStream_t s);
hipError_t launch_group_set_image(void *group, int i, const Layout &L, uint8_t *ws, hipStream_t s);
hipError_t launch_group_iter(void *group, int n_images, int step, const AdamCoef &co_prev, const AdamCoef &co, unsigned flags,
                             uint64_t n_obs_total, double *trace_prev, hipStream_t s);
hipError_t launch_group_finish(void *group, int n_images, int step, const AdamCoef &co_prev, uint64_t n_obs_total,
                               double *trace_prev, hipStream_t s);

// independent images, one launch per iteration (fit.hip)
size_t batch_bytes(int n_images);
hipError_t launch_batch_set(void *batch, int n_images, uint8_t *const *ws, double *const *trace, const Layout *layouts, unsigned flags,
                            hipStream_t s);
hipError_t launch_batch_iter(const Layout &L, void *batch, int n_images, const AdamCoef &co, unsigned flags, int row, hipStream_t s);

// output stage (plot.hip)
size_t select_scratch_bytes();
hipError_t launch_select_ranks(const float *J, int H, int W, int n_ranks, const uint64_t *ranks, float *out, void *scratch,
                               hipStream_t s);
hipError_t launch_plot_stretch(const float *J, int H, int W, const float *lo, const float *hi, uint8_t *out, hipStream_t s);
hipError_t launch_count_valid(const float *J, int H, int W, uint64_t *count, hipStream_t s);

// artificial-light model (light.hip)
size_t light_workspace_bytes(const Layout &L, int ext_sets = 1);
uint8_t *light_ext2_dense(const Layout &L, uint8_t *lws);   // second extension set (float32 colours next to camera points);
uint8_t *light_ext2_comp(const Layout &L, uint8_t *lws);    // only in a workspace of light_workspace_bytes(L, 2) bytes
int64_t light_params_offset(const Layout &L);
uint8_t *light_ext_dense(const Layout &L, uint8_t *lws);
uint8_t *light_ext_comp(const Layout &L, uint8_t *lws);
hipError_t launch_light_init(const Layout &L, uint8_t *lws, const float *params19, hipStream_t s);
hipError_t launch_light_deal(const Layout &L, uint8_t *ws, uint8_t *lws, unsigned flags, hipStream_t s);   // once per fit call, before launch_light_iter
hipError_t launch_light_iter(const Layout &L, uint8_t *ws, uint8_t *lws, const AdamCoef &co, unsigned flags,
                             double *trace_row, hipStream_t s);
hipError_t launch_light_update_J(const Layout &L, uint8_t *ws, uint8_t *lws, unsigned flags, hipStream_t s);

}  // namespace sucre
