// In-launch hand-off between workgroups (fit.hip): the last workgroup to arrive on a counter continues with
// everybody's results, so a reduction + parameter step needs no second launch.
#pragma once
#include <hip/hip_runtime.h>

namespace sucre {

// Arrival on a counter: EVERY wave of the workgroup drains its own hand-off stores (vmcnt is per wave; reduce_group's
// results are stored by all four waves), the workgroup meets, and only then one lane signals with a relaxed
// agent-scope fetch_add.  Returns true in the workgroup that arrived last, which has then done its agent acquire and
// re-armed the counter.  (Until round 2 only the signalling wave drained: the other waves' group sums could still be
// in flight when the top-level last arriver read them -- seen as run-to-run differences of the beta gradient, the
// last values each wave stores, once two processes shared the GPU.)
__device__ __forceinline__ bool arrive_last(unsigned *counter, unsigned expected, int *flag) {
    const int t = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
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
    __syncthreads();
    return *flag != 0;
}

// lane l <- x of lane l + kOff, for the lanes whose partner sits in the same half (32), quarter (16) or row of the wave: what
// lane 0's tree needs at every level.  Register-to-register (v_permlane32_swap / v_permlane16_swap of gfx950, DPP row shifts);
// __shfl_down goes through the LDS crossbar (ds_bpermute: sixty of them and as many waits per image in a batch launch).
template <int kOff>
__device__ __forceinline__ uint32_t lane_down_bits(uint32_t x) {
    if constexpr (kOff == 32) return __builtin_amdgcn_permlane32_swap(x, x, false, false)[1];
    else if constexpr (kOff == 16) return __builtin_amdgcn_permlane16_swap(x, x, false, false)[1];
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x100 + kOff, 0xf, 0xf, true);   // row_shl:kOff
}
template <int kOff>
__device__ __forceinline__ float lane_down(float x) { return __uint_as_float(lane_down_bits<kOff>(__float_as_uint(x))); }
template <int kOff>
__device__ __forceinline__ double lane_down(double x) {
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = lane_down_bits<kOff>((uint32_t)u), hi = lane_down_bits<kOff>((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// lane 0 <- the wave's sum in the association of  for (off = 32; off; off >>= 1) x += __shfl_down(x, off)  (the other lanes end
// with values nobody reads).
template <class T>
__device__ __forceinline__ T wave_sum_lane0(T x) {
    x += lane_down<32>(x);
    x += lane_down<16>(x);
    x += lane_down<8>(x);
    x += lane_down<4>(x);
    x += lane_down<2>(x);
    x += lane_down<1>(x);
    return x;
}

}  // namespace sucre
