// Device helpers shared by the ViT-encoder kernels of TransUNet (gemm_bf16.hip, attention_bf16.hip, vit_elementwise.hip).
#pragma once
#include "sis_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 sis_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 sis_bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 sis_bf16x2;
typedef __attribute__((ext_vector_type(4))) float sis_f32x4;
typedef __attribute__((ext_vector_type(16))) float sis_f32x16;

// ---- dropout as a counter-based stream: element `idx` of dropout site `site` in the step whose seed word is `seed` is
// dropped iff hash < thr (thr = round(p * 2^32)).  Forward and backward evaluate the same function, so no mask is stored;
// the seed word lives in device memory (advanced once per training step by sis_dropout_advance), which keeps a captured
// hipGraph of the step drawing fresh masks on every replay.
// Reference: nn.Dropout(config.transformer["dropout_rate"]) in networks/trans_u_net/vit_seg_modeling.py:70-71,108,138
// (torch's Philox stream there; any independent uniform stream is the same operator).
struct SisDropKey { unsigned s0, s1; };

__device__ __forceinline__ SisDropKey sis_drop_key(const unsigned long long* seed, unsigned site) {
    const unsigned long long s = seed ? *seed : 0ull;
    SisDropKey k;
    k.s0 = (unsigned)s ^ (site * 0x632BE5ABu);
    k.s1 = (unsigned)(s >> 32) + site * 0x9E3779B9u + 0x7F4A7C15u;
    return k;
}

__device__ __forceinline__ unsigned sis_drop_hash(SisDropKey k, unsigned idx) {
    unsigned h = idx * 0x9E3779B1u + k.s0;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;   // murmur3 finaliser: a bijection
    h ^= k.s1; h *= 0x27D4EB2Fu; h ^= h >> 15;
    return h;
}

// keep ? scale : 0
__device__ __forceinline__ float sis_drop_factor(SisDropKey k, unsigned idx, unsigned thr, float scale) {
    return sis_drop_hash(k, idx) < thr ? 0.f : scale;
}

__device__ __forceinline__ float sis_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float sis_gelu_grad(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    return cdf + x * (0.3989422804014327f * __expf(-0.5f * x * x));
}

__device__ __forceinline__ unsigned sis_pack_bf16x2(float a, float b) {
    sis_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float sis_bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float sis_bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
