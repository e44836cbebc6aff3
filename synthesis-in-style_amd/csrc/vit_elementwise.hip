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
