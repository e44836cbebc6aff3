// Weight standardisation of TransUNet's ResNetV2 stem / blocks (networks/trans_u_net/vit_seg_modeling_resnet_skip.py:
// 20-27: every StdConv2d forward computes w_hat = (w - mean) / sqrt(var + 1e-5) over (Cin, kh, kw) per output channel).
// ATen runs this as var_mean + sub + add + sqrt + div (and ~10 more kernels in backward) for each of the 53
// convolutions; here it is one launch per direction, one workgroup per output channel, two-pass statistics in fp32,
// and the result can be written directly in the autocast dtype (bf16 / f16) so the convolution needs no cast kernel.
//   forward   w_hat[r, i] = (w[r, i] - mean_r) / sqrt(var_r + eps);  invstd_r kept for the backward
//   backward  dw[r, i] = invstd_r * (g[r, i] - mean_i(g) - w_hat[r, i] * mean_i(g * w_hat))
#include "sis_common.h"

namespace {

__device__ __forceinline__ float ws_block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename TO>
__global__ __launch_bounds__(256) void weight_std_fwd_kernel(TO* __restrict__ what, float* __restrict__ invstd,
                                                             const float* __restrict__ w, int n, float eps) {
    __shared__ float red[4];
    const float* row = w + (int64_t)blockIdx.x * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += row[i];
    const float mean = ws_block_sum(s, red) / (float)n;
    float m2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const float d = row[i] - mean; m2 += d * d; }
    const float var = ws_block_sum(m2, red) / (float)n;
    const float sd = sqrtf(var + eps);
    if (threadIdx.x == 0) invstd[blockIdx.x] = 1.f / sd;
    TO* o = what + (int64_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += 256) sis_st(o, i, (row[i] - mean) / sd);
}

template <typename TG>
__global__ __launch_bounds__(256) void weight_std_bwd_kernel(float* __restrict__ dw, const TG* __restrict__ g,
                                                             const float* __restrict__ w, const float* __restrict__ invstd,
                                                             int n, float eps) {
    __shared__ float red[4];
    const float* row = w + (int64_t)blockIdx.x * n;
    const TG* grow = g + (int64_t)blockIdx.x * n;
    // w_hat is recomputed from w (fp32) instead of being read back in a 16-bit type
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += row[i];
    const float mean = ws_block_sum(s, red) / (float)n;
    const float is = invstd[blockIdx.x];
    float sg = 0.f, sgw = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = sis_ld(grow, i), wh = (row[i] - mean) * is;
        sg += gi; sgw += gi * wh;
    }
    const float mg = ws_block_sum(sg, red) / (float)n;
    const float mgw = ws_block_sum(sgw, red) / (float)n;
    float* o = dw + (int64_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = sis_ld(grow, i), wh = (row[i] - mean) * is;
        o[i] = is * (gi - mg - wh * mgw);
    }
}

}  // namespace

extern "C" int sis_weight_std_fwd(void* w_hat, float* invstd, const float* w, int out_dtype, int rows, int n, float eps,
                                  void* stream) {
    if (rows == 0) return 0;
    SIS_REQUIRE(w_hat && invstd && w, "sis_weight_std_fwd: null pointer");
    SIS_REQUIRE(rows > 0 && n > 0, "sis_weight_std_fwd: non-positive size");
    hipStream_t st = (hipStream_t)stream;
    switch (out_dtype) {
        case SIS_F32: hipLaunchKernelGGL(weight_std_fwd_kernel<float>, dim3(rows), dim3(256), 0, st, (float*)w_hat, invstd, w, n, eps); break;
        case SIS_F16: hipLaunchKernelGGL(weight_std_fwd_kernel<__half>, dim3(rows), dim3(256), 0, st, (__half*)w_hat, invstd, w, n, eps); break;
        case SIS_BF16: hipLaunchKernelGGL(weight_std_fwd_kernel<__hip_bfloat16>, dim3(rows), dim3(256), 0, st, (__hip_bfloat16*)w_hat, invstd, w, n, eps); break;
        default: return sis_fail("sis_weight_std_fwd: dtype code %d not supported (f32, f16, bf16)", out_dtype);
    }
    SIS_CHECK_LAUNCH("weight_std_fwd_kernel");
    return 0;
}

extern "C" int sis_weight_std_bwd(float* dw, const void* grad_w_hat, const float* w, const float* invstd, int grad_dtype,
                                  int rows, int n, float eps, void* stream) {
    if (rows == 0) return 0;
    SIS_REQUIRE(dw && grad_w_hat && w && invstd, "sis_weight_std_bwd: null pointer");
    SIS_REQUIRE(rows > 0 && n > 0, "sis_weight_std_bwd: non-positive size");
    hipStream_t st = (hipStream_t)stream;
    switch (grad_dtype) {
        case SIS_F32: hipLaunchKernelGGL(weight_std_bwd_kernel<float>, dim3(rows), dim3(256), 0, st, dw, (const float*)grad_w_hat, w, invstd, n, eps); break;
        case SIS_F16: hipLaunchKernelGGL(weight_std_bwd_kernel<__half>, dim3(rows), dim3(256), 0, st, dw, (const __half*)grad_w_hat, w, invstd, n, eps); break;
        case SIS_BF16: hipLaunchKernelGGL(weight_std_bwd_kernel<__hip_bfloat16>, dim3(rows), dim3(256), 0, st, dw, (const __hip_bfloat16*)grad_w_hat, w, invstd, n, eps); break;
        default: return sis_fail("sis_weight_std_bwd: dtype code %d not supported (f32, f16, bf16)", grad_dtype);
    }
    SIS_CHECK_LAUNCH("weight_std_bwd_kernel");
    return 0;
}
