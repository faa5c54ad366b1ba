// Small device helpers shared by fit.hip and light.hip.
#pragma once
#include "experiment.h"
#include "launch.h"

namespace sucre {

constexpr float kLog2e = 1.44269504088896340736f;
// float32(float64(k)/255) for every uint8 k (loader.py:157,163) as fma(k, hi, k*lo); checked for all 256 values
// in tests/test_host_logic.py and on the GPU by tests/test_gpu_parity.py.
constexpr float kInv255Hi = (float)(1.0 / 255.0);
constexpr float kInv255Lo = (float)(1.0 / 255.0 - (double)kInv255Hi);

__device__ __forceinline__ float unit_from_u8(uint32_t k) {
    const float kf = (float)k;
    return __builtin_fmaf(kf, kInv255Hi, kf * kInv255Lo);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// v_exp_f32 flushes results below 2^-126 to zero; torch.exp underflows gradually.  Nothing in a gradient can tell the
// two apart, but the closed-form J = sum(y a) / sum(a^2) of a pixel whose every a^2 underflows is +-inf with a denormal
// a and 0/0 = NaN with a flushed one: the solves redo such a pixel with this form (3 more instructions, never on the
// hot path).
__device__ __forceinline__ float gradual_exp2(float x) {
    const bool tiny = x < -126.0f;
    return fast_exp2(tiny ? x + 64.0f : x) * (tiny ? 0x1p-64f : 1.0f);
}
template <bool kGradual>
__device__ __forceinline__ float exp2_as(float x) { return kGradual ? gradual_exp2(x) : fast_exp2(x); }

// torch/optim/adam.py::_single_tensor_adam (non-capturable, no amsgrad, no weight decay), float32
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, const AdamCoef &co) {
    m = __builtin_fmaf(co.w1, g - m, m);
    v = (v * co.beta2) + (co.w2 * g) * g;
    const float denom = sqrtf(v) / co.bc2_sqrt + co.eps;
    p = p + (co.step_size_neg * m) / denom;
}

// The same step for a pixel's J (three per pixel and iteration: 158 of the J-parameter kernel's ~2000 instructions per
// pixel were the two IEEE divisions and the IEEE square root of these three steps): hardware square root and reciprocals
// (1 ulp each) instead of the IEEE sequences, 11 instead of ~42 instructions per channel.  The fit is held to a tolerance,
// not to bit parity -- its sums already run in another order than torch's -- and the difference sits below that noise:
// RMS(J) against the reference's own 200-step runs 5.8e-8 / 5.3e-8 with this form, 5.3e-8 / 5.5e-8 with the IEEE one, and
// the oracle itself is 5.7e-8 / 5.5e-8 from the reference (tools/exp/adam_accuracy.py).  +2.9 % on the headline, same box
// (tools/exp/ab_bench.sh exactadam).  Special values behave as in adam_update: v = 0 -> denom = eps; v = inf -> no step;
// a NaN J stays NaN.  SUCRE_EXACT_J_ADAM=1 (experiment.h) builds the IEEE form.
__device__ __forceinline__ void adam_update_J(float &p, float &m, float &v, float g, const AdamCoef &co) {
    if (kExactJAdam) {
        adam_update(p, m, v, g, co);
        return;
    }
    m = __builtin_fmaf(co.w1, g - m, m);
    v = (v * co.beta2) + (co.w2 * g) * g;
    const float denom = __builtin_fmaf(__builtin_amdgcn_sqrtf(v), __builtin_amdgcn_rcpf(co.bc2_sqrt), co.eps);
    p = __builtin_fmaf(co.step_size_neg * m, __builtin_amdgcn_rcpf(denom), p);
}

}  // namespace sucre
