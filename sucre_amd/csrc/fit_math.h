// Small device helpers shared by fit.hip and light.hip.
#pragma once
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

// torch/optim/adam.py::_single_tensor_adam (non-capturable, no amsgrad, no weight decay), float32
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, const AdamCoef &co) {
    m = __builtin_fmaf(co.w1, g - m, m);
    v = (v * co.beta2) + (co.w2 * g) * g;
    const float denom = sqrtf(v) / co.bc2_sqrt + co.eps;
    p = p + (co.step_size_neg * m) / denom;
}

}  // namespace sucre
