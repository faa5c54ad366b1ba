// In-launch hand-off between workgroups (fit.hip, light.hip): the last workgroup to arrive on a counter continues with
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

}  // namespace sucre
