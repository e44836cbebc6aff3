// TransUNet's training objective in one forward and one backward pass over the logits:
//     loss = 0.5 * CrossEntropy(logits, labels) + 0.5 * Dice(softmax(logits), one_hot(labels))
// Reference: updater/segmentation_updater.py:95-102 (TransUNetUpdater: nn.CrossEntropyLoss + DiceLoss(softmax=True), 0.5 / 0.5)
// and networks/trans_u_net/utils.py:7-42 (DiceLoss: per class 1 - (2 sum(p t) + 1e-5) / (sum(p^2) + sum(t^2) + 1e-5), sums over
// the whole batch, mean over the classes).  torch runs this as ~25 launches (float cast, log_softmax, nll_loss2d, softmax,
// one-hot compare, three reductions, their backward kernels) over a [B, C, H, W] tensor; here the logits are read once per
// direction, in the dtype the segmentation head wrote them (bf16 under autocast, fp32 otherwise).
//
// Forward: every thread walks pixel quads (4 consecutive pixels of a plane, 16 / 8 bytes per class), keeps per-class sums
// in registers; per-workgroup partials are added in workgroup order by the finish kernel (deterministic).  The finish kernel
// also derives the two per-class constants the backward needs, so the backward is a pure element-wise pass:
//     dL/dz_k = g * [ 0.5 (p_k - t_k) / n_valid + 0.5 p_k (G_k - sum_c G_c p_c) ],  G_c = a_c t_c + b_c p_c,
//     a_c = -2 / (C (D_c + s)),  b_c = 2 (2 I_c + s) / (C (D_c + s)^2),  I = sum p t, D = sum p^2 + sum t.
// Labels outside [0, C) count as "ignore" for the cross entropy (nn.CrossEntropyLoss's ignore_index) and as an all-zero
// one-hot row for the Dice term (what the reference's comparison against the class ids produces).
#include "sis_common.h"

namespace {

constexpr int LOSS_MAXC = 8;
constexpr int LOSS_BLOCKS = 512;
constexpr float DICE_SMOOTH = 1e-5f;

template <typename T>
__device__ __forceinline__ void load_quad(const T* p, float* v) {
    if constexpr (sizeof(T) == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        v[0] = __builtin_bit_cast(float, q.x << 16); v[1] = __builtin_bit_cast(float, q.x & 0xFFFF0000u);
        v[2] = __builtin_bit_cast(float, q.y << 16); v[3] = __builtin_bit_cast(float, q.y & 0xFFFF0000u);
    }
}
template <typename T>
__device__ __forceinline__ void store_quad(T* p, const float* v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        const b2 lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<uint2*>(p) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
    }
}

// partial[block][0] = sum of -log p[label]; [1] = valid labels; [2 + 3c ..] = I_c, P_c, T_c
template <typename T, int C>
__global__ __launch_bounds__(256) void ce_dice_fwd_kernel(float* __restrict__ partial, const T* __restrict__ logits,
                                                          const long* __restrict__ labels, int batch, int hw) {
    constexpr int NV = 2 + 3 * C;
    __shared__ float red[4][NV];
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const int quads_per_img = hw >> 2;
    const long long quads = (long long)batch * quads_per_img;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
        const int b = (int)(q / quads_per_img), pix = (int)(q - (long long)b * quads_per_img) * 4;
        const T* zb = logits + ((long long)b * C) * hw + pix;
        float z[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c) load_quad(zb + (long long)c * hw, z[c]);
        long lab[4];
        {
            const long* lp = labels + (long long)b * hw + pix;
            const longlong2 l01 = *reinterpret_cast<const longlong2*>(lp), l23 = *reinterpret_cast<const longlong2*>(lp + 2);
            lab[0] = l01.x; lab[1] = l01.y; lab[2] = l23.x; lab[3] = l23.y;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m = z[0][e];
#pragma unroll
            for (int c = 1; c < C; ++c) m = fmaxf(m, z[c][e]);
            float p[C], s = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) { p[c] = __expf(z[c][e] - m); s += p[c]; }
            const float inv = 1.f / s;
            const bool valid = lab[e] >= 0 && lab[e] < C;
            float zl = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                p[c] *= inv;
                const bool hit = lab[e] == c;
                if (hit) zl = z[c][e];
                acc[2 + 3 * c] += hit ? p[c] : 0.f;
                acc[3 + 3 * c] += p[c] * p[c];
                acc[4 + 3 * c] += hit ? 1.f : 0.f;
            }
            if (valid) { acc[0] += (m + __logf(s)) - zl; acc[1] += 1.f; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NV) partial[(long long)blockIdx.x * NV + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[0] = 0.5 ce + 0.5 dice, out[1] = ce, out[2] = dice;  stats[0] = 1 / n_valid, stats[1 + 2c] = a_c, stats[2 + 2c] = b_c
__global__ __launch_bounds__(1024) void ce_dice_finish_kernel(float* __restrict__ out, float* __restrict__ stats, const float* __restrict__ partial,
                                                              int blocks, int C) {
    // 32 lanes = the (up to 32) sums, 32 groups of lanes = 32 contiguous ranges of the workgroups' partial rows: every lane adds its
    // range in order, lane (0, v) then adds the 32 range sums in order -- a fixed association, and 1 / 32 of the dependent-load chain
    // one lane per sum walked over all rows (61 us for the 2 048 rows of a 512 x 512 batch of 8, on the step's critical path)
    __shared__ float tot[2 + 3 * LOSS_MAXC];
    __shared__ float part[32][33];
    const int nv = 2 + 3 * C;
    const int v = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int per = (blocks + 31) / 32, k0 = r * per, k1 = min(blocks, k0 + per);
    float s = 0.f;
    if (v < nv)
        for (int k = k0; k < k1; ++k) s += partial[(long long)k * nv + v];
    part[r][v] = s;
    __syncthreads();
    if (threadIdx.x < nv) {
        float t = 0.f;
        for (int j = 0; j < 32; ++j) t += part[j][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv_valid = tot[1] > 0.f ? 1.f / tot[1] : 0.f;
        const float ce = tot[0] * inv_valid;
        float dice = 0.f;
        for (int c = 0; c < C; ++c) {
            const float I = tot[2 + 3 * c], D = tot[3 + 3 * c] + tot[4 + 3 * c];
            dice += 1.f - (2.f * I + DICE_SMOOTH) / (D + DICE_SMOOTH);
            stats[1 + 2 * c] = -2.f / ((float)C * (D + DICE_SMOOTH));
            stats[2 + 2 * c] = 2.f * (2.f * I + DICE_SMOOTH) / ((float)C * (D + DICE_SMOOTH) * (D + DICE_SMOOTH));
        }
        dice /= (float)C;
        stats[0] = inv_valid;
        out[0] = 0.5f * ce + 0.5f * dice; out[1] = ce; out[2] = dice;
    }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void ce_dice_bwd_kernel(T* __restrict__ grad, const T* __restrict__ logits, const long* __restrict__ labels,
                                                          const float* __restrict__ stats, const float* __restrict__ grad_loss, int batch, int hw) {
    const int quads_per_img = hw >> 2;
    const long long quads = (long long)batch * quads_per_img;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const float g = grad_loss ? *grad_loss : 1.f;
    const float wce = 0.5f * g * stats[0], wd = 0.5f * g;
    float a[C], bb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { a[c] = stats[1 + 2 * c]; bb[c] = stats[2 + 2 * c]; }
    const int b = (int)(q / quads_per_img), pix = (int)(q - (long long)b * quads_per_img) * 4;
    const long long base = ((long long)b * C) * hw + pix;
    float z[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c) load_quad(logits + base + (long long)c * hw, z[c]);
    long lab[4];
    {
        const long* lp = labels + (long long)b * hw + pix;
        const longlong2 l01 = *reinterpret_cast<const longlong2*>(lp), l23 = *reinterpret_cast<const longlong2*>(lp + 2);
        lab[0] = l01.x; lab[1] = l01.y; lab[2] = l23.x; lab[3] = l23.y;
    }
    float out[C][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float m = z[0][e];
#pragma unroll
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c][e]);
        float p[C], s = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { p[c] = __expf(z[c][e] - m); s += p[c]; }
        const float inv = 1.f / s;
        const bool valid = lab[e] >= 0 && lab[e] < C;
        float G[C], dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            p[c] *= inv;
            G[c] = (lab[e] == c ? a[c] : 0.f) + bb[c] * p[c];
            dot += G[c] * p[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float ce = valid ? wce * (p[c] - (lab[e] == c ? 1.f : 0.f)) : 0.f;
            out[c][e] = ce + wd * p[c] * (G[c] - dot);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) store_quad(grad + base + (long long)c * hw, out[c]);
}

}  // namespace

extern "C" int sis_ce_dice_workspace_floats(int classes) { return LOSS_BLOCKS * (2 + 3 * classes); }

#define LOSS_SWITCH_C(CV, CALL)                                                                                          \
    switch (CV) {                                                                                                        \
        case 2: { constexpr int C = 2; CALL; } break;                                                                    \
        case 3: { constexpr int C = 3; CALL; } break;                                                                    \
        case 4: { constexpr int C = 4; CALL; } break;                                                                    \
        case 5: { constexpr int C = 5; CALL; } break;                                                                    \
        case 6: { constexpr int C = 6; CALL; } break;                                                                    \
        case 7: { constexpr int C = 7; CALL; } break;                                                                    \
        case 8: { constexpr int C = 8; CALL; } break;                                                                    \
        default: return sis_fail("ce + dice loss: %d classes (2..8 are built)", (CV));                                   \
    }

extern "C" int sis_ce_dice_fwd(float* out3, float* stats, float* workspace, const void* logits, int dtype, const int64_t* labels,
                               int batch, int classes, int hw, void* stream) {
    SIS_REQUIRE(out3 && stats && workspace && logits && labels, "sis_ce_dice_fwd: null pointer");
    SIS_REQUIRE(batch > 0 && hw > 0 && hw % 4 == 0, "sis_ce_dice_fwd: plane size %d must be a positive multiple of 4", hw);
    SIS_REQUIRE(dtype == SIS_F32 || dtype == SIS_BF16, "sis_ce_dice_fwd: logits must be float32 or bfloat16");
    hipStream_t st = (hipStream_t)stream;
    const long long quads = (long long)batch * (hw / 4);
    const int blocks = (int)(quads < (long long)LOSS_BLOCKS * 256 ? (quads + 255) / 256 : LOSS_BLOCKS);
    if (dtype == SIS_F32) {
        LOSS_SWITCH_C(classes, hipLaunchKernelGGL((ce_dice_fwd_kernel<float, C>), dim3(blocks), dim3(256), 0, st, workspace,
                                                  (const float*)logits, (const long*)labels, batch, hw))
    } else {
        LOSS_SWITCH_C(classes, hipLaunchKernelGGL((ce_dice_fwd_kernel<unsigned short, C>), dim3(blocks), dim3(256), 0, st, workspace,
                                                  (const unsigned short*)logits, (const long*)labels, batch, hw))
    }
    SIS_CHECK_LAUNCH("ce_dice_fwd_kernel");
    hipLaunchKernelGGL(ce_dice_finish_kernel, dim3(1), dim3(1024), 0, st, out3, stats, workspace, blocks, classes);
    SIS_CHECK_LAUNCH("ce_dice_finish_kernel");
    return 0;
}

extern "C" int sis_ce_dice_bwd(void* grad_logits, const void* logits, int dtype, const int64_t* labels, const float* stats,
                               const float* grad_loss, int batch, int classes, int hw, void* stream) {
    SIS_REQUIRE(grad_logits && logits && labels && stats, "sis_ce_dice_bwd: null pointer");
    SIS_REQUIRE(batch > 0 && hw > 0 && hw % 4 == 0, "sis_ce_dice_bwd: plane size %d must be a positive multiple of 4", hw);
    SIS_REQUIRE(dtype == SIS_F32 || dtype == SIS_BF16, "sis_ce_dice_bwd: logits must be float32 or bfloat16");
    hipStream_t st = (hipStream_t)stream;
    const long long quads = (long long)batch * (hw / 4);
    const int blocks = (int)((quads + 255) / 256);
    if (dtype == SIS_F32) {
        LOSS_SWITCH_C(classes, hipLaunchKernelGGL((ce_dice_bwd_kernel<float, C>), dim3(blocks), dim3(256), 0, st, (float*)grad_logits,
                                                  (const float*)logits, (const long*)labels, stats, grad_loss, batch, hw))
    } else {
        LOSS_SWITCH_C(classes, hipLaunchKernelGGL((ce_dice_bwd_kernel<unsigned short, C>), dim3(blocks), dim3(256), 0, st,
                                                  (unsigned short*)grad_logits, (const unsigned short*)logits, (const long*)labels, stats,
                                                  grad_loss, batch, hw))
    }
    SIS_CHECK_LAUNCH("ce_dice_bwd_kernel");
    return 0;
}
