// K1: fused bias + activation for gfx950.  HBM-bound streaming kernel: 8 B/element (fp32).
//
// Behaviour follows the element formula of the reference kernel
// (networks/stylegan2/op/fused_bias_act_kernel.cu:25-47) and its bias indexing (:62-71);
// the launch shape does not: one lane owns 16 bytes (4 x f32 / 8 x f16) per trip of a
// grid-stride loop, so every wave instruction moves 1 KiB, and the grid is capped at 8 waves
// per SIMD-set (2048 blocks of 256) as the guide's Guideline 11 prescribes for memory-bound ops.
#include "sis_common.h"

namespace {

template <typename A>
__device__ __forceinline__ A act_apply(A v, A r, int mode, A alpha, A scale) {
    A y;
    switch (mode) {
        case 12: case 32: y = (A)0; break;
        case 30: y = (v > (A)0) ? v : v * alpha; break;
        case 31: y = (r > (A)0) ? v : v * alpha; break;
        default: y = v; break;  // 10, 11 and unknown modes are linear
    }
    return y * scale;
}

// Generic path: one element per lane per trip; any dtype, any shape.
template <typename T>
__global__ __launch_bounds__(256) void fba_scalar_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                         const T* __restrict__ b, const T* __restrict__ ref,
                                                         int64_t n, int64_t step_b, int64_t size_b, int mode,
                                                         float alpha_f, float scale_f) {
    typedef typename sis_acc<T>::type A;
    const A alpha = (A)alpha_f, scale = (A)scale_f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        A v = sis_ld(x, i);
        if (b) v += sis_ld(b, (i / step_b) % size_b);
        const A r = ref ? sis_ld(ref, i) : (A)0;
        sis_st(out, i, act_apply<A>(v, r, mode, alpha, scale));
    }
}

// fp32 fast path: float4 per lane; requires n % 4 == 0, step_b % 4 == 0 (so the 4 elements share
// one bias entry) and 16-byte aligned pointers.  32-bit index math (n < 2^31 checked by the host).
__global__ __launch_bounds__(256) void fba_f32x4_kernel(float4* __restrict__ out, const float4* __restrict__ x,
                                                        const float* __restrict__ b, const float4* __restrict__ ref,
                                                        unsigned n4, unsigned step_b4, unsigned size_b, int mode,
                                                        float alpha, float scale) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x[i];
        if (b) {
            const float bb = b[(i / step_b4) % size_b];
            v.x += bb; v.y += bb; v.z += bb; v.w += bb;
        }
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ref) r = ref[i];
        float4 y;
        y.x = act_apply<float>(v.x, r.x, mode, alpha, scale);
        y.y = act_apply<float>(v.y, r.y, mode, alpha, scale);
        y.z = act_apply<float>(v.z, r.z, mode, alpha, scale);
        y.w = act_apply<float>(v.w, r.w, mode, alpha, scale);
        out[i] = y;
    }
}

template <typename T>
int launch_scalar(void* out, const void* x, const void* bias, const void* ref, int64_t n, int64_t step_b,
                  int64_t size_b, int mode, float alpha, float scale, hipStream_t st) {
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(fba_scalar_kernel<T>, dim3(blocks), dim3(256), 0, st, (T*)out, (const T*)x, (const T*)bias,
                       (const T*)ref, n, step_b, size_b, mode, alpha, scale);
    SIS_CHECK_LAUNCH("sis_fused_bias_act");
    return 0;
}

}  // namespace

extern "C" int sis_fused_bias_act(void* out, const void* x, const void* bias, const void* ref, int dtype,
                                  int64_t numel, int64_t step_b, int64_t size_b, int act, int grad, float alpha,
                                  float scale, void* stream) {
    SIS_REQUIRE(numel >= 0, "sis_fused_bias_act: negative numel");
    if (numel == 0) return 0;
    SIS_REQUIRE(out && x, "sis_fused_bias_act: null input/output pointer");
    if (size_b <= 0) bias = nullptr;
    if (bias) SIS_REQUIRE(step_b > 0, "sis_fused_bias_act: step_b must be positive when a bias is given");
    hipStream_t st = (hipStream_t)stream;
    const int mode = act * 10 + grad;
    switch (dtype) {
        case SIS_F32: {
            const bool aligned = (((uintptr_t)out | (uintptr_t)x | (uintptr_t)ref) & 15) == 0;
            if (aligned && numel % 4 == 0 && (!bias || step_b % 4 == 0) && numel < (int64_t)1 << 31) {
                const unsigned n4 = (unsigned)(numel / 4);
                const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
                hipLaunchKernelGGL(fba_f32x4_kernel, dim3(blocks), dim3(256), 0, st, (float4*)out, (const float4*)x,
                                   (const float*)bias, (const float4*)ref, n4, (unsigned)(bias ? step_b / 4 : 1),
                                   (unsigned)(bias ? size_b : 1), mode, alpha, scale);
                SIS_CHECK_LAUNCH("sis_fused_bias_act");
                return 0;
            }
            return launch_scalar<float>(out, x, bias, ref, numel, step_b, size_b, mode, alpha, scale, st);
        }
        case SIS_F64: return launch_scalar<double>(out, x, bias, ref, numel, step_b, size_b, mode, alpha, scale, st);
        case SIS_F16: return launch_scalar<__half>(out, x, bias, ref, numel, step_b, size_b, mode, alpha, scale, st);
        case SIS_BF16: return launch_scalar<__hip_bfloat16>(out, x, bias, ref, numel, step_b, size_b, mode, alpha, scale, st);
        default: return sis_fail("sis_fused_bias_act: unsupported dtype code %d", dtype);
    }
}
