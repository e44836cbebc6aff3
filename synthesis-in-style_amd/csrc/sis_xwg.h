// Hand-over of partial results between the workgroups of ONE launch ("the workgroup that completes a group finishes it"),
// used by group_norm.hip to drop the tiny merge launches behind its statistics kernels.  (Measured on column_sum.hip as
// well: 1 024 workgroups each waiting for their write-through stores cost the TransUNet step 1 % more than the second
// launch they saved; not used there.)
//
// Partial results travel between workgroups (possibly on different XCDs = different L2s) as device-scope relaxed atomics:
// xwg_publish writes through to the device's coherence point, the counter increment is issued once those stores have
// completed (s_waitcnt vmcnt(0)), xwg_peek reads past the non-coherent caches.  No device-scope FENCE anywhere: a
// release / acquire fence writes back / invalidates the whole L2 of the XCD, and with one per workgroup TransUNet's norms
// ran 15 ms per step slower than with the merge as a launch of its own.
// Counters: ints, ZERO before the launch; the completing workgroup puts its counter back to zero, so a buffer private to the
// stream can serve every launch on it without being cleared in between.
//
// Hardware assumption, stated because the HIP / LLVM memory model does not promise it: on gfx942 / gfx950 a relaxed AGENT-scope
// atomic store is an `sc1` write-through store and a relaxed agent-scope atomic load an `sc1` load that bypasses the vector L1
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": the `sc1` store / `sc1` load /
// counter form, one lane signalling for its workgroup behind a barrier and a drained vmcnt).  Every handed-off word is
// written by xwg_publish and read by xwg_peek -- no plain access touches it -- which is that table's condition.  This
// library is built for gfx950 only (Makefile ARCH); the static_assert below keeps the file from compiling for anything else.
#pragma once
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "sis_xwg.h relies on gfx942 / gfx950 sc1 write-through semantics of agent-scope relaxed atomics"
#endif

__device__ __forceinline__ void xwg_publish(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float xwg_peek(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// True in every thread of the workgroup that finds `counter` at count - 1, i.e. the last of `count` workgroups to arrive.
// ALL_PUBLISH: every thread of the workgroup published something (false: only thread 0 did).
template <bool ALL_PUBLISH>
__device__ __forceinline__ bool xwg_complete(int* counter, int count) {
    __shared__ int last;
    if (ALL_PUBLISH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's xwg_publish stores have completed
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == count - 1;
    }
    __syncthreads();
    if (!last) return false;
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}
