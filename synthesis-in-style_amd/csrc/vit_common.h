// Device helpers shared by the ViT-encoder kernels of TransUNet (gemm_bf16.hip, attention_bf16.hip, vit_elementwise.hip).
#pragma once
#include "sis_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 sis_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 sis_bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 sis_bf16x2;
typedef __attribute__((ext_vector_type(4))) float sis_f32x4;
typedef __attribute__((ext_vector_type(16))) float sis_f32x16;

// ---- dropout as a counter-based stream.  The elements of a dropout site are numbered row-major; every consumer handles
// aligned QUADS (4 consecutive elements, index % 4 == 0: a lane's 4 accumulator columns / one float4).  Quad `quad` of site
// `site` in the step whose seed word is `seed` draws four 16-bit uniforms from two chained 32-bit mixes of (seed, site, quad);
// element e is dropped iff u_e < thr16 (thr16 = round(p * 65536): p is realised to 2^-16), survivors are scaled by
// 65536 / (65536 - thr16).  Forward and backward evaluate the same function, so no mask is stored; the seed word lives in
// device memory (advanced once per training step by sis_dropout_advance), which keeps a captured hipGraph of the step
// drawing fresh masks on every replay.
// Reference: nn.Dropout(config.transformer["dropout_rate"]) in networks/trans_u_net/vit_seg_modeling.py:70-71,108,138
// (torch's Philox stream there; another stream of independent uniforms is the same operator in distribution).
struct SisDropKey { unsigned s0, s1; };

__device__ __forceinline__ SisDropKey sis_drop_key(const unsigned long long* seed, unsigned site) {
    const unsigned long long s = seed ? *seed : 0ull;
    SisDropKey k;
    k.s0 = (unsigned)s ^ (site * 0x632BE5ABu);
    k.s1 = (unsigned)(s >> 32) + site * 0x9E3779B9u + 0x7F4A7C15u;
    return k;
}

__host__ __device__ __forceinline__ unsigned sis_drop_thr16(float p) { return p > 0.f ? (unsigned)(p * 65536.f + 0.5f) : 0u; }
__host__ __device__ __forceinline__ float sis_drop_scale(unsigned thr16) { return 65536.f / (float)(65536u - thr16); }

// f[e] = keep ? scale : 0 for the four elements of quad `quad`
__device__ __forceinline__ void sis_drop_quad(SisDropKey k, unsigned quad, unsigned thr16, float scale, float* f) {
    unsigned h = quad * 0x9E3779B1u + k.s0;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;   // murmur3 finaliser: a bijection of quad
    // the second pair of uniforms comes from an INDEPENDENT mix of (quad, s1) -- a second finaliser with other constants on
    // another affine image of quad -- not from h: elements 2, 3 of a quad must not be functions of elements 0, 1 (ADVICE r3;
    // tests/test_gemm_bf16_gpu.py::test_dropout_quads_are_pairwise_independent checks the joint drop frequencies)
    unsigned h2 = quad * 0xC2B2AE3Du + k.s1;
    h2 ^= h2 >> 15; h2 *= 0x2C1B3C6Du; h2 ^= h2 >> 12; h2 *= 0x297A2D39u; h2 ^= h2 >> 15;
    f[0] = (h & 0xFFFFu) < thr16 ? 0.f : scale;
    f[1] = (h >> 16) < thr16 ? 0.f : scale;
    f[2] = (h2 & 0xFFFFu) < thr16 ? 0.f : scale;
    f[3] = (h2 >> 16) < thr16 ? 0.f : scale;
}

// ---- erf GELU (F.gelu's default, vit_seg_modeling.py:34 ACT2FN["gelu"]) and its derivative from ONE exponential:
// Phi(x) = 0.5 erfc(-x / sqrt 2) with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 round-off of the sums
// that feed it), phi(x) = exp(-x^2 / 2) / sqrt(2 pi) from the same exp(-x^2 / 2).
__device__ __forceinline__ void sis_gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.f));
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);   // exp(-x^2 / 2)
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * e;            // 0.5 erfc(|x| / sqrt 2)
    cdf = x >= 0.f ? 1.f - half_erfc : half_erfc;
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float sis_gelu(float x) {
    float cdf, pdf;
    sis_gelu_parts(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float sis_gelu_grad(float x) {
    float cdf, pdf;
    sis_gelu_parts(x, cdf, pdf);
    return __builtin_fmaf(x, pdf, cdf);
}

__device__ __forceinline__ unsigned sis_pack_bf16x2(float a, float b) {
    sis_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float sis_bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float sis_bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
