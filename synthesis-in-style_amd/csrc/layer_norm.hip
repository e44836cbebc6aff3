// LayerNorm of the TransUNet encoder (networks/trans_u_net/vit_seg_modeling.py:171-190,233-250: two per block, one
// final), forward and backward, rows of n = hidden elements, n % 256 == 0 (ViT-B: 768).
// One wave per row: lane l holds elements [4 (l + 64 j), +4) for j < n / 256 in registers (one 16/8-byte load each),
// so the two-pass mean / variance and the normalisation never re-read memory.  The output can be written in a 16-bit
// type (what the following Linear consumes under autocast: no separate cast kernel).
// Backward: dx per row as usual; d(gamma) / d(beta) are column sums over all rows -- every workgroup walks a strip
// of rows, keeps its column sums in registers, writes one partial row per workgroup; ln_param_reduce adds the partials
// in fixed order (deterministic, no atomics).
#include <algorithm>
#include "vit_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void ln_load4(const T* p, float* v) {
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
__device__ __forceinline__ void ln_store4(T* p, const float* v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        T t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) sis_st(t, e, v[e]);
        uint2 q;
        __builtin_memcpy(&q, t, 8);
        *reinterpret_cast<uint2*>(p) = q;
    }
}
__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename TI, typename TO, int NJ>
__global__ __launch_bounds__(256) void ln_fwd_kernel(TO* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, const TI* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int rows, float eps) {
    constexpr int N = NJ * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const TI* xr = x + (int64_t)row * N;
    float v[NJ][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        ln_load4(xr + 4 * (lane + 64 * j), v[j]);
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = ln_wave_sum(s) / (float)N;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; m2 += d * d; }
    const float rstd = rsqrtf(ln_wave_sum(m2) / (float)N + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    TO* yr = y + (int64_t)row * N;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 4 * (lane + 64 * j);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
        float o[4] = {(v[j][0] - mean) * rstd * g.x + b.x, (v[j][1] - mean) * rstd * g.y + b.y,
                      (v[j][2] - mean) * rstd * g.z + b.z, (v[j][3] - mean) * rstd * g.w + b.w};
        ln_store4(yr + c, o);
    }
}

// Optional fusions for the pre-norm residual blocks of the ViT encoder (x_out = x + f(LN(x))): `radd` (fp32, x's shape) is
// the gradient arriving over the skip connection, added to dx; `cast_out` receives bf16(dx_total * dropout factor) -- the
// gradient w.r.t. the Linear output that the PREVIOUS residual add dropped out and added (site / threshold / scale of that
// dropout, stream position = element index), i.e. what sis_dropout_bwd_cast would compute in a pass of its own.
struct LnBwdExtra {
    const float* radd; unsigned short* cast_out; const unsigned long long* seed; unsigned site, thr; float scale;
};

template <typename TI, typename TG, int NJ>
__global__ __launch_bounds__(256) void ln_bwd_kernel(TI* __restrict__ dx, float* __restrict__ part, const TG* __restrict__ g,
                                                     const TI* __restrict__ x, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                     int rows, LnBwdExtra ex) {
    constexpr int N = NJ * 256;
    __shared__ float red[2][4][N];  // per-wave column sums, merged by wave 0 at the end
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[NJ][4], db[NJ][4], gm[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 q = *reinterpret_cast<const float4*>(gamma + 4 * (lane + 64 * j));
        gm[j][0] = q.x; gm[j][1] = q.y; gm[j][2] = q.z; gm[j][3] = q.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dg[j][e] = 0.f; db[j][e] = 0.f; }
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const TI* xr = x + (int64_t)row * N;
        const TG* gr = g + (int64_t)row * N;
        const float mean = mean_in[row], rstd = rstd_in[row];
        float xh[NJ][4], gv[NJ][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ln_load4(xr + 4 * (lane + 64 * j), xh[j]);
            ln_load4(gr + 4 * (lane + 64 * j), gv[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[j][e] = (xh[j][e] - mean) * rstd;
                dg[j][e] += gv[j][e] * xh[j][e];
                db[j][e] += gv[j][e];
                gv[j][e] *= gm[j][e];
                s1 += gv[j][e]; s2 += gv[j][e] * xh[j][e];
            }
        }
        const float m1 = ln_wave_sum(s1) / (float)N, m2 = ln_wave_sum(s2) / (float)N;
        TI* dr = dx + (int64_t)row * N;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (gv[j][e] - m1 - xh[j][e] * m2);
            const int col = 4 * (lane + 64 * j);
            if (ex.radd) {
                const float4 r = *reinterpret_cast<const float4*>(ex.radd + (int64_t)row * N + col);
                o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
            }
            ln_store4(dr + col, o);
            if (ex.cast_out) {
                float f[4] = {o[0], o[1], o[2], o[3]};
                if (ex.thr) {
                    float keep[4];
                    sis_drop_quad(sis_drop_key(ex.seed, ex.site), ((unsigned)row * (unsigned)N + (unsigned)col) >> 2, ex.thr, ex.scale, keep);
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] *= keep[e];
                }
                *reinterpret_cast<uint2*>(ex.cast_out + (int64_t)row * N + col) =
                    make_uint2(sis_pack_bf16x2(f[0], f[1]), sis_pack_bf16x2(f[2], f[3]));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[0][wave][4 * (lane + 64 * j) + e] = dg[j][e];
            red[1][wave][4 * (lane + 64 * j) + e] = db[j][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < N; c += 256) {
        part[((int64_t)blockIdx.x * 2) * N + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        part[((int64_t)blockIdx.x * 2 + 1) * N + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// 64 columns per workgroup of 16 waves; wave w adds the partials k = w, w + 16, ... (fixed order, 8 loads requested per trip),
// then the sixteen are combined in a fixed tree: deterministic, and ~3 us for 512 partial rows instead of one chain of
// dependent-latency loads per wave
__device__ __forceinline__ void ln_param_reduce_body(float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     const float* __restrict__ part, int n_part, int n, int block) {
    __shared__ float red[2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = block * 64 + lane;
    float a = 0.f, b = 0.f;
    if (c < n) {
        int k = wave;
        for (; k + 7 * 16 < n_part; k += 8 * 16) {
            float pa[8], pb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pa[u] = part[((int64_t)(k + 16 * u) * 2) * n + c];
                pb[u] = part[((int64_t)(k + 16 * u) * 2 + 1) * n + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += pa[u]; b += pb[u]; }
        }
        for (; k < n_part; k += 16) { a += part[((int64_t)k * 2) * n + c]; b += part[((int64_t)k * 2 + 1) * n + c]; }
    }
    red[0][wave][lane] = a; red[1][wave][lane] = b;
    __syncthreads();
    if (wave < 2 && c < n) {   // wave 0: d(gamma), wave 1: d(beta)
        float s[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) s[w] = red[wave][w][lane];
#pragma unroll
        for (int st = 1; st < 16; st <<= 1)
#pragma unroll
            for (int w = 0; w < 16; w += 2 * st) s[w] += s[w + st];
        (wave == 0 ? dgamma : dbeta)[c] = s[0];
    }
}

__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               const float* __restrict__ part, int n_part, int n) {
    ln_param_reduce_body(dgamma, dbeta, part, n_part, n, blockIdx.x);
}

// The same reduction for SEVERAL LayerNorm backwards in one launch (their partial rows wait in their workspaces: nothing reads
// d(gamma) / d(beta) before the optimizer or the gradient exchange).  A ViT-B/16 encoder has 25 norms: 25 launches of ~3 us of
// work between ~2 us kernel boundaries become one.  Job descriptors travel as kernel arguments (hipGraph-capturable as they are).
constexpr int LN_MULTI_MAX = 32;
struct LnReduceJobs {
    float* dgamma[LN_MULTI_MAX]; float* dbeta[LN_MULTI_MAX]; const float* part[LN_MULTI_MAX];
    int n_part[LN_MULTI_MAX]; int n[LN_MULTI_MAX]; int first_block[LN_MULTI_MAX + 1]; int count;
};
__global__ __launch_bounds__(1024) void ln_param_reduce_multi_kernel(LnReduceJobs j) {
    int job = 0;
    while (job + 1 < j.count && (int)blockIdx.x >= j.first_block[job + 1]) ++job;
    ln_param_reduce_body(j.dgamma[job], j.dbeta[job], j.part[job], j.n_part[job], j.n[job], (int)blockIdx.x - j.first_block[job]);
}

constexpr int LN_BWD_BLOCKS = 512;  // workgroups of the backward pass = partial rows to add afterwards

}  // namespace

extern "C" int sis_layer_norm_workspace_floats(int n) { return 2 * LN_BWD_BLOCKS * n; }

#define LN_SWITCH_NJ(NJV, CALL)                  \
    switch (NJV) {                               \
        case 1: { constexpr int NJ = 1; CALL; } break; \
        case 2: { constexpr int NJ = 2; CALL; } break; \
        case 3: { constexpr int NJ = 3; CALL; } break; \
        case 4: { constexpr int NJ = 4; CALL; } break; \
        default: return sis_fail("layer norm: row length %d not supported (256, 512, 768 or 1024)", (NJV) * 256); \
    }

extern "C" int sis_layer_norm_fwd(void* y, float* mean, float* rstd, const void* x, const float* gamma, const float* beta,
                                  int x_dtype, int y_dtype, int rows, int n, float eps, void* stream) {
    if (rows == 0) return 0;
    SIS_REQUIRE(y && mean && rstd && x && gamma && beta, "sis_layer_norm_fwd: null pointer");
    SIS_REQUIRE(rows > 0 && n > 0 && n % 256 == 0, "sis_layer_norm_fwd: row length %d must be a multiple of 256", n);
    SIS_REQUIRE((x_dtype == SIS_F32 || x_dtype == SIS_BF16) && (y_dtype == SIS_F32 || y_dtype == SIS_BF16),
                "sis_layer_norm_fwd: dtypes must be f32 or bf16");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(sis_cdiv(rows, 4));
#define LN_FWD(TI, TO) hipLaunchKernelGGL((ln_fwd_kernel<TI, TO, NJ>), grid, dim3(256), 0, st, (TO*)y, mean, rstd, (const TI*)x, gamma, beta, rows, eps)
    if (x_dtype == SIS_F32 && y_dtype == SIS_F32) { LN_SWITCH_NJ(n / 256, LN_FWD(float, float)) }
    else if (x_dtype == SIS_F32) { LN_SWITCH_NJ(n / 256, LN_FWD(float, __hip_bfloat16)) }
    else if (y_dtype == SIS_F32) { LN_SWITCH_NJ(n / 256, LN_FWD(__hip_bfloat16, float)) }
    else { LN_SWITCH_NJ(n / 256, LN_FWD(__hip_bfloat16, __hip_bfloat16)) }
#undef LN_FWD
    SIS_CHECK_LAUNCH("ln_fwd_kernel");
    return 0;
}

static int ln_bwd_impl(void* dx, float* dgamma, float* dbeta, float* workspace, const void* grad_y, const void* x,
                       const float* mean, const float* rstd, const float* gamma, int x_dtype, int g_dtype, int rows,
                       int n, LnBwdExtra ex, void* stream, bool reduce = true) {
    if (rows == 0) return 0;
    SIS_REQUIRE(dx && dgamma && dbeta && workspace && grad_y && x && mean && rstd && gamma, "sis_layer_norm_bwd: null pointer");
    SIS_REQUIRE(rows > 0 && n > 0 && n % 256 == 0, "sis_layer_norm_bwd: row length %d must be a multiple of 256", n);
    SIS_REQUIRE((x_dtype == SIS_F32 || x_dtype == SIS_BF16) && (g_dtype == SIS_F32 || g_dtype == SIS_BF16),
                "sis_layer_norm_bwd: dtypes must be f32 or bf16");
    hipStream_t st = (hipStream_t)stream;
    int blocks = sis_cdiv(rows, 4);
    if (blocks > LN_BWD_BLOCKS) blocks = LN_BWD_BLOCKS;
    const dim3 grid(blocks);
#define LN_BWD(TI, TG) hipLaunchKernelGGL((ln_bwd_kernel<TI, TG, NJ>), grid, dim3(256), 0, st, (TI*)dx, workspace, (const TG*)grad_y, (const TI*)x, mean, rstd, gamma, rows, ex)
    if (x_dtype == SIS_F32 && g_dtype == SIS_F32) { LN_SWITCH_NJ(n / 256, LN_BWD(float, float)) }
    else if (x_dtype == SIS_F32) { LN_SWITCH_NJ(n / 256, LN_BWD(float, __hip_bfloat16)) }
    else if (g_dtype == SIS_F32) { LN_SWITCH_NJ(n / 256, LN_BWD(__hip_bfloat16, float)) }
    else { LN_SWITCH_NJ(n / 256, LN_BWD(__hip_bfloat16, __hip_bfloat16)) }
#undef LN_BWD
    SIS_CHECK_LAUNCH("ln_bwd_kernel");
    if (!reduce) return 0;   // the partial rows stay in `workspace` for sis_layer_norm_param_reduce_multi
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(sis_cdiv(n, 64)), dim3(1024), 0, st, dgamma, dbeta, workspace, blocks, n);
    SIS_CHECK_LAUNCH("ln_param_reduce_kernel");
    return 0;
}

extern "C" int sis_layer_norm_bwd(void* dx, float* dgamma, float* dbeta, float* workspace, const void* grad_y, const void* x,
                                  const float* mean, const float* rstd, const float* gamma, int x_dtype, int g_dtype, int rows,
                                  int n, void* stream) {
    return ln_bwd_impl(dx, dgamma, dbeta, workspace, grad_y, x, mean, rstd, gamma, x_dtype, g_dtype, rows, n,
                       LnBwdExtra{nullptr, nullptr, nullptr, 0u, 0u, 1.f}, stream);
}

extern "C" int sis_layer_norm_bwd_fused(void* dx, float* dgamma, float* dbeta, float* workspace, const void* grad_y, const void* x,
                                        const float* mean, const float* rstd, const float* gamma, int x_dtype, int g_dtype,
                                        int rows, int n, const float* residual_grad, void* cast_out, const void* seed, int site,
                                        float drop_p, void* stream) {
    SIS_REQUIRE(!(residual_grad || cast_out) || x_dtype == SIS_F32, "sis_layer_norm_bwd_fused: the fusions are for an fp32 residual stream");
    SIS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sis_layer_norm_bwd_fused: dropout probability %f / seed word", drop_p);
    SIS_REQUIRE((int64_t)rows * n < (1LL << 32), "sis_layer_norm_bwd_fused: more than 2^32 elements");
    LnBwdExtra ex;
    ex.radd = residual_grad; ex.cast_out = (unsigned short*)cast_out; ex.seed = (const unsigned long long*)seed; ex.site = (unsigned)site;
    ex.thr = sis_drop_thr16(drop_p);
    ex.scale = sis_drop_scale(ex.thr);
    return ln_bwd_impl(dx, dgamma, dbeta, workspace, grad_y, x, mean, rstd, gamma, x_dtype, g_dtype, rows, n, ex, stream);
}

/* sis_layer_norm_bwd_fused without its second launch: dx (and the cast) are complete, d(gamma) / d(beta) are NOT written -- the
 * sis_layer_norm_bwd_parts(rows) partial rows wait in `workspace` for sis_layer_norm_param_reduce_multi. */
extern "C" int sis_layer_norm_bwd_fused_partial(void* dx, float* workspace, const void* grad_y, const void* x, const float* mean,
                                                const float* rstd, const float* gamma, int x_dtype, int g_dtype, int rows, int n,
                                                const float* residual_grad, void* cast_out, const void* seed, int site,
                                                float drop_p, void* stream) {
    SIS_REQUIRE(!(residual_grad || cast_out) || x_dtype == SIS_F32, "sis_layer_norm_bwd_fused_partial: the fusions are for an fp32 residual stream");
    SIS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed), "sis_layer_norm_bwd_fused_partial: dropout probability %f / seed word", drop_p);
    SIS_REQUIRE((int64_t)rows * n < (1LL << 32), "sis_layer_norm_bwd_fused_partial: more than 2^32 elements");
    LnBwdExtra ex;
    ex.radd = residual_grad; ex.cast_out = (unsigned short*)cast_out; ex.seed = (const unsigned long long*)seed; ex.site = (unsigned)site;
    ex.thr = sis_drop_thr16(drop_p);
    ex.scale = sis_drop_scale(ex.thr);
    float dummy = 0.f;   // (ln_bwd_impl checks the pointers it will not use)
    return ln_bwd_impl(dx, &dummy, &dummy, workspace, grad_y, x, mean, rstd, gamma, x_dtype, g_dtype, rows, n, ex, stream, false);
}

extern "C" int sis_layer_norm_bwd_parts(int rows) {
    const int blocks = sis_cdiv(rows, 4);
    return blocks > LN_BWD_BLOCKS ? LN_BWD_BLOCKS : blocks;
}

/* d(gamma) / d(beta) of `count` LayerNorm backwards from their partial rows: dgamma[i], dbeta[i] [n[i]] float32, part[i] the
 * workspace of job i holding n_part[i] = sis_layer_norm_bwd_parts(rows_i) partial rows; HOST arrays of device pointers / ints.
 * Same summation order as sis_layer_norm_bwd_fused's own reduction (bitwise the same results). */
extern "C" int sis_layer_norm_param_reduce_multi(void* const* dgamma, void* const* dbeta, const void* const* part, const int* n_part,
                                                 const int* n, int count, void* stream) {
    if (count <= 0) return 0;
    SIS_REQUIRE(dgamma && dbeta && part && n_part && n, "sis_layer_norm_param_reduce_multi: null pointer");
    for (int j0 = 0; j0 < count; j0 += LN_MULTI_MAX) {
        LnReduceJobs jobs;
        jobs.count = std::min(LN_MULTI_MAX, count - j0);
        int blocks = 0;
        for (int i = 0; i < jobs.count; ++i) {
            SIS_REQUIRE(dgamma[j0 + i] && dbeta[j0 + i] && part[j0 + i] && n_part[j0 + i] > 0 && n[j0 + i] > 0,
                        "sis_layer_norm_param_reduce_multi: job %d is incomplete", j0 + i);
            jobs.dgamma[i] = (float*)dgamma[j0 + i]; jobs.dbeta[i] = (float*)dbeta[j0 + i]; jobs.part[i] = (const float*)part[j0 + i];
            jobs.n_part[i] = n_part[j0 + i]; jobs.n[i] = n[j0 + i];
            jobs.first_block[i] = blocks;
            blocks += sis_cdiv(n[j0 + i], 64);
        }
        jobs.first_block[jobs.count] = blocks;
        hipLaunchKernelGGL(ln_param_reduce_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, jobs);
        SIS_CHECK_LAUNCH("ln_param_reduce_multi_kernel");
    }
    return 0;
}
