// Bias gradient of the encoder's Linear layers (networks/trans_u_net/vit_seg_modeling.py:53-110: query / key / value /
// out, fc1 / fc2): d(bias)[c] = sum over the rows of dL/dy [rows][n], 16-bit or fp32 in, fp32 out.
// Stage 1: workgroup (column group of 256, row slice): lane l of every wave owns columns 4l..4l+3 (one 8 / 16-byte load
// per row), the 4 waves take rows r, r+4, ...; the four per-wave sums are combined through LDS and written as one partial
// row per slice.  Stage 2 adds the slices in fixed order (deterministic, no atomics).  ATen's generic reduction moves
// ~2 TB/s on these shapes ([8192][768..3072] bf16) and rounds the result to bf16 first.
#include "sis_common.h"

namespace {

constexpr int CS_COLS = 256;  // columns per workgroup (64 lanes x 4)

template <typename T>
__device__ __forceinline__ void cs_load4(const T* p, float* v) {
    if constexpr (sizeof(T) == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        T t[4];
        __builtin_memcpy(t, &q, 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sis_ld(t, e);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void column_sum_partial_kernel(float* __restrict__ part, const T* __restrict__ x, int rows,
                                                                 int n, int rows_per_slice) {
    __shared__ float red[4][CS_COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * CS_COLS + 4 * lane;
    const int r_lo = blockIdx.y * rows_per_slice, r_hi = min(rows, r_lo + rows_per_slice);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < n) {  // n % 4 == 0 (host-checked): a lane's four columns are all inside or all outside
        int r = r_lo + wave;
        for (; r + 4 < r_hi; r += 8) {  // two independent loads in flight
            float a[4], b[4];
            cs_load4(x + (int64_t)r * n + c, a);
            cs_load4(x + (int64_t)(r + 4) * n + c, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += a[e] + b[e];
        }
        for (; r < r_hi; r += 4) {
            float a[4];
            cs_load4(x + (int64_t)r * n + c, a);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += a[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][4 * lane + e] = acc[e];
    __syncthreads();
    const int cc = blockIdx.x * CS_COLS + threadIdx.x;
    if (cc < n)
        part[(int64_t)blockIdx.y * n + cc] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// 64 columns per workgroup, the slices dealt to the four waves (wave q adds slices q, q + 4, ... in order, 8 loads in
// flight), then (s0 + s1) + (s2 + s3): a fixed order, and ~2 us instead of one 64-deep chain of dependent-latency loads.
__global__ __launch_bounds__(256) void column_sum_finish_kernel(float* __restrict__ out, const float* __restrict__ part, int n,
                                                                int slices) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.f;
    if (c < n) {
#pragma unroll 8
        for (int k = q; k < slices; k += 4) s += part[(int64_t)k * n + c];
    }
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < n) out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

constexpr int CS_MAX_SLICES = 64;

}  // namespace

extern "C" int64_t sis_column_sum_workspace_floats(int n) { return (int64_t)CS_MAX_SLICES * n; }

extern "C" int sis_column_sum(float* out, float* workspace, const void* x, int x_dtype, int rows, int n, void* stream) {
    SIS_REQUIRE(out && workspace && x, "sis_column_sum: null pointer");
    SIS_REQUIRE(rows > 0 && n > 0 && n % 4 == 0, "sis_column_sum: rows %d, n %d (n must be a positive multiple of 4)", rows, n);
    SIS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "sis_column_sum: input must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int groups = sis_cdiv(n, CS_COLS);
    int slices = sis_cdiv(1024, groups);  // ~1024 workgroups (4 per CU)
    if (slices > CS_MAX_SLICES) slices = CS_MAX_SLICES;
    if (slices > sis_cdiv(rows, 8)) slices = sis_cdiv(rows, 8);
    if (slices < 1) slices = 1;
    const int rps = sis_cdiv(rows, slices);
    slices = sis_cdiv(rows, rps);
    const dim3 grid(groups, slices);
    switch (x_dtype) {
        case SIS_F32: hipLaunchKernelGGL(column_sum_partial_kernel<float>, grid, dim3(256), 0, st, workspace, (const float*)x, rows, n, rps); break;
        case SIS_F16: hipLaunchKernelGGL(column_sum_partial_kernel<__half>, grid, dim3(256), 0, st, workspace, (const __half*)x, rows, n, rps); break;
        case SIS_BF16: hipLaunchKernelGGL(column_sum_partial_kernel<__hip_bfloat16>, grid, dim3(256), 0, st, workspace, (const __hip_bfloat16*)x, rows, n, rps); break;
        default: return sis_fail("sis_column_sum: dtype code %d not supported (f32, f16, bf16)", x_dtype);
    }
    SIS_CHECK_LAUNCH("column_sum_partial_kernel");
    hipLaunchKernelGGL(column_sum_finish_kernel, dim3(sis_cdiv(n, 64)), dim3(256), 0, st, out, workspace, n, slices);
    SIS_CHECK_LAUNCH("column_sum_finish_kernel");
    return 0;
}
