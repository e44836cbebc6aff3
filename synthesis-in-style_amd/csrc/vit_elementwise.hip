// Element-wise companions of the fused ViT-encoder GEMMs (csrc/gemm_bf16.hip), TransUNet under bf16 autocast.
//
//   sis_dropout_advance   steps the 64-bit seed word every dropout site of one training iteration reads.  It lives in device
//                         memory so that a captured hipGraph of the iteration draws fresh masks on every replay.
//   sis_dropout_bwd_cast  gradient of `resid + dropout(linear)` (vit_seg_modeling.py:181-189 with :96 / :121-122) w.r.t. the
//                         Linear output: bf16( g * dropout_factor ), g = the fp32 gradient of the residual stream.  The
//                         factor is recomputed from (seed, site, element index): no mask tensor exists.
#include "vit_common.h"

namespace {

__global__ void dropout_advance_kernel(unsigned long long* seed) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *seed = *seed * 6364136223846793005ull + 1442695040888963407ull;  // LCG step (Knuth)
}

__global__ __launch_bounds__(256) void dropout_bwd_cast_kernel(unsigned short* __restrict__ out, const float* __restrict__ g,
                                                               long long quads, const unsigned long long* __restrict__ seed,
                                                               unsigned site, unsigned thr, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    const float4 v = *reinterpret_cast<const float4*>(g + 4 * i);
    float f[4] = {v.x, v.y, v.z, v.w};
    if (thr) {
        float keep[4];
        sis_drop_quad(sis_drop_key(seed, site), (unsigned)i, thr, scale, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] *= keep[e];
    }
    *reinterpret_cast<uint2*>(out + 4 * i) = make_uint2(sis_pack_bf16x2(f[0], f[1]), sis_pack_bf16x2(f[2], f[3]));
}

}  // namespace

extern "C" int sis_dropout_advance(void* seed, void* stream) {
    SIS_REQUIRE(seed, "sis_dropout_advance: null pointer");
    hipLaunchKernelGGL(dropout_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)seed);
    SIS_CHECK_LAUNCH("dropout_advance_kernel");
    return 0;
}

extern "C" int sis_dropout_bwd_cast(void* out, const float* grad, int64_t numel, const void* seed, int site, float drop_p, void* stream) {
    if (numel == 0) return 0;
    SIS_REQUIRE(out && grad, "sis_dropout_bwd_cast: null pointer");
    SIS_REQUIRE(numel % 4 == 0 && numel < (1LL << 32), "sis_dropout_bwd_cast: element count %lld must be a multiple of 4 below 2^32", (long long)numel);
    SIS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sis_dropout_bwd_cast: dropout probability %f / seed word", drop_p);
    const unsigned thr = sis_drop_thr16(drop_p);
    const float scale = sis_drop_scale(thr);
    hipLaunchKernelGGL(dropout_bwd_cast_kernel, dim3(sis_cdiv(numel / 4, 256)), dim3(256), 0, (hipStream_t)stream, (unsigned short*)out,
                       grad, (long long)(numel / 4), (const unsigned long long*)seed, (unsigned)site, thr, scale);
    SIS_CHECK_LAUNCH("dropout_bwd_cast_kernel");
    return 0;
}

// ---- transposed bf16 copies of the encoder's Linear weights, all of them by ONE launch per training step.
// The data-gradient GEMM of y = x W^T is dx = g W (W [out][in] is the K-major operand); with W^T [in][out] at hand it is an
// NT product like the forward and runs on the same 256-row tiles (gemm256_bf16.hip).  The bf16 shadows of the weights are
// rewritten by the optimizer launch every step (csrc/seg_ops.hip), so the transposes are re-derived once per forward:
// 2 x 170 MB of traffic for ViT-B/16, ~0.1 ms, against ~1 ms of data-gradient GEMM time it saves.
// table rows (int64): {src, dst, rows, cols, first_tile}; tiles are 64 x 64, numbered row-major inside a matrix.
namespace {
__global__ __launch_bounds__(256) void transpose_bf16_multi_kernel(const long long* __restrict__ table, int n) {
    __shared__ unsigned short tile[64][66];
    int which = 0;
    for (int i = 1; i < n; ++i)
        if ((long long)blockIdx.x >= table[5 * i + 4]) which = i;
    const long long* row = table + 5 * which;
    const unsigned short* src = reinterpret_cast<const unsigned short*>(row[0]);
    unsigned short* dst = reinterpret_cast<unsigned short*>(row[1]);
    const int rows = (int)row[2], cols = (int)row[3];
    const int t = (int)(blockIdx.x - row[4]), tiles_c = (cols + 63) / 64;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // 64 consecutive columns of a source row per wave: 128-byte reads
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // 64 consecutive source rows = 64 consecutive columns of a destination row
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long long)c * rows + r] = tile[tx][ty + 4 * i];
    }
}
}  // namespace

// [batch][rows][cols] -> [batch][cols][rows] for 2- and 4-byte elements: the patch embedding's NCHW feature map -> tokens
// (networks/trans_u_net/vit_seg_modeling.py:151-153 `x.flatten(2).transpose(-1, -2)`), the decoder's tokens -> NCHW
// (:341-344 `hidden_states.permute(0, 2, 1).contiguous().view(...)`) and their gradients.  64 x 64 tiles through LDS, 128- /
// 256-byte runs per wave on both sides (the ATen strided copy these replace ran the 12.6 MB tensors at 0.8 TB/s).
namespace {
template <typename U>
__global__ __launch_bounds__(256) void transpose_batched_kernel(U* __restrict__ dst, const U* __restrict__ src, int rows, int cols) {
    __shared__ U tile[64][65 + (sizeof(U) == 2 ? 1 : 0)];
    const int tiles_c = (cols + 63) / 64;
    const int r0 = ((int)blockIdx.x / tiles_c) * 64, c0 = ((int)blockIdx.x % tiles_c) * 64;
    const long long base = (long long)blockIdx.y * rows * cols;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 4 * i][tx] = src[base + (long long)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[base + (long long)c * rows + r] = tile[tx][ty + 4 * i];
    }
}
}  // namespace

extern "C" int sis_transpose_batched(void* dst, const void* src, int elem_bytes, int batch, int rows, int cols, void* stream) {
    if (batch <= 0 || rows <= 0 || cols <= 0) return 0;
    SIS_REQUIRE(dst && src && dst != src, "sis_transpose_batched: null or aliased pointers");
    SIS_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "sis_transpose_batched: 2- or 4-byte elements, got %d", elem_bytes);
    SIS_REQUIRE(batch <= 65535, "sis_transpose_batched: batch %d exceeds the grid's second dimension", batch);
    const dim3 grid((unsigned)(((rows + 63) / 64) * ((cols + 63) / 64)), (unsigned)batch);
    if (elem_bytes == 2)
        hipLaunchKernelGGL(transpose_batched_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned short*)dst,
                           (const unsigned short*)src, rows, cols);
    else
        hipLaunchKernelGGL(transpose_batched_kernel<unsigned int>, grid, dim3(256), 0, (hipStream_t)stream, (unsigned int*)dst,
                           (const unsigned int*)src, rows, cols);
    SIS_CHECK_LAUNCH("transpose_batched_kernel");
    return 0;
}

extern "C" int sis_transpose_bf16_multi(const void* table, int n_tensors, int total_tiles, void* stream) {
    if (n_tensors <= 0 || total_tiles <= 0) return 0;
    SIS_REQUIRE(table, "sis_transpose_bf16_multi: null table");
    hipLaunchKernelGGL(transpose_bf16_multi_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const long long*)table, n_tensors);
    SIS_CHECK_LAUNCH("transpose_bf16_multi_kernel");
    return 0;
}
