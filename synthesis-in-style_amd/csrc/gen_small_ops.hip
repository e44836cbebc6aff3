// Small generator-side kernels (fp32): mapping network pieces, weight prepack, demodulation
// coefficients and the ToRGB / skip accumulation.  None of these is on the MFMA roofline; they
// are latency- or HBM-bound and written so that each reads its operands exactly once.
#include <type_traits>
#include "sis_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// PixelNorm (model.py:19-20): one wave per row.
__global__ __launch_bounds__(256) void pixel_norm_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                         int batch, int dim) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= batch) return;
    const float* xr = x + (int64_t)row * dim;
    float ss = 0.f;
    for (int i = lane; i < dim; i += 64) { const float v = xr[i]; ss += v * v; }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)dim + 1e-8f);
    for (int i = lane; i < dim; i += 64) out[(int64_t)row * dim + i] = xr[i] * r;
}

// EqualLinear (model.py:152-162).  One wave per output feature; the wave keeps its weight row in
// registers (in_dim <= 64*ELW floats) and walks the batch, so the [out,in] matrix is read once.
constexpr int ELW = 16;  // supports in_dim up to 1024
constexpr int EL_ROWS = 4;  // batch rows per workgroup (blockIdx.y walks the batch)
__global__ __launch_bounds__(256) void equal_linear_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                           int64_t x_row_stride, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int batch, int in_dim,
                                                           int out_dim, float scale, float lr_mul, int activation) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= out_dim) return;
    float wr[ELW];
#pragma unroll
    for (int j = 0; j < ELW; ++j) {
        const int i = lane + 64 * j;
        wr[j] = (i < in_dim) ? w[(int64_t)o * in_dim + i] : 0.f;
    }
    const float b = bias ? bias[o] * lr_mul : 0.f;
    const int r0 = blockIdx.y * EL_ROWS;
    float acc[EL_ROWS];
#pragma unroll
    for (int rr = 0; rr < EL_ROWS; ++rr) {
        acc[rr] = 0.f;
        const int r = r0 + rr;
        if (r < batch) {
            const float* xr = x + (int64_t)r * x_row_stride;
#pragma unroll
            for (int j = 0; j < ELW; ++j) {
                const int i = lane + 64 * j;
                if (i < in_dim) acc[rr] += xr[i] * wr[j];
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < EL_ROWS; ++rr) acc[rr] = wave_sum(acc[rr]);
    if (lane == 0) {
#pragma unroll
        for (int rr = 0; rr < EL_ROWS; ++rr) {
            const int r = r0 + rr;
            if (r < batch) {
                float v = acc[rr] * scale + b;
                if (activation) v = (v > 0.f ? v : v * 0.2f) * 1.4142135623730951f;
                out[(int64_t)r * out_dim + o] = v;
            }
        }
    }
}

// The same layer when in_dim is exactly NJ * 64 (the 512-wide mapping network): every load of the wave -- its weight row and
// EL_ROWS input rows -- is requested before the first one is used.  In the general kernel above the predicated loads became
// load - wait - multiply per element, 32 dependent L2 round trips per wave: 14.5 us for a 32 x 512 x 512 layer.
template <int NJ>
__global__ __launch_bounds__(256) void equal_linear_full_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                                int64_t x_row_stride, const float* __restrict__ w,
                                                                const float* __restrict__ bias, int batch, int out_dim,
                                                                float scale, float lr_mul, int activation) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= out_dim) return;
    const int r0 = blockIdx.y * EL_ROWS;
    float wr[NJ], xv[EL_ROWS][NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wr[j] = w[(int64_t)o * (NJ * 64) + lane + 64 * j];
#pragma unroll
    for (int rr = 0; rr < EL_ROWS; ++rr) {
        const float* xr = x + (int64_t)min(r0 + rr, batch - 1) * x_row_stride;  // (rows past the batch: recomputed, not stored)
#pragma unroll
        for (int j = 0; j < NJ; ++j) xv[rr][j] = xr[lane + 64 * j];
    }
    const float b = bias ? bias[o] * lr_mul : 0.f;
    float acc[EL_ROWS];
#pragma unroll
    for (int rr = 0; rr < EL_ROWS; ++rr) {
        acc[rr] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[rr] += xv[rr][j] * wr[j];  // (same order as the general kernel: j ascending)
    }
#pragma unroll
    for (int rr = 0; rr < EL_ROWS; ++rr) acc[rr] = wave_sum(acc[rr]);
    if (lane == 0) {
#pragma unroll
        for (int rr = 0; rr < EL_ROWS; ++rr) {
            const int r = r0 + rr;
            if (r < batch) {
                float v = acc[rr] * scale + b;
                if (activation) v = (v > 0.f ? v : v * 0.2f) * 1.4142135623730951f;
                out[(int64_t)r * out_dim + o] = v;
            }
        }
    }
}

// ---- The batched "head" GEMMs of a forward (modulation, demodulation; MODE 0 = a single EqualLinear) on the fp32 matrix cores.
// out[r][o] = epilogue(sum_k A[r][k] * W[o][k]) with r = batch row (32 per tile = one MFMA M block), W rows k-contiguous
// as torch stores nn.Linear weights.  Workgroup = 4 waves = 32 rows x 128 outputs (wave w: outputs 32 w .. 32 w + 31),
// K in 64-wide chunks staged through LDS with coalesced 16-byte loads (row stride 66 floats: 8-byte aligned, at most
// 2-way bank conflicts on the b32 operand reads).  These layers are tiny (<= 0.3 GFLOP together); as one-wave-per-output
// dot-product kernels the two batched launches took 104 us + 83 us (profiles/r02_z_step_timeline.txt); here 41 + 39 us.
// (The 8 mapping layers stay on equal_linear_kernel: 32 x 512 x 512 gives this tile shape 4 workgroups and a serial chain
// of 8 load latencies -- 24 us against 14.)
constexpr int HG_ROWS = 32, HG_COLS = 128, HG_KC = 64, HG_LD = 66;
typedef __attribute__((ext_vector_type(16))) float hg_f32x16;

struct HeadArgs {
    // MODE 0 (EqualLinear): a/w/bias/out direct.  MODE 1 (modulation) / 2 (demodulation): table rows, see the launchers.
    float* out; const float* a; const float* w; const float* bias; const int64_t* table;
    int64_t a_row_stride;
    int batch, k_dim, out_dim, n_layers, n_latent, activation;
    float scale, lr_mul;
};

template <int MODE>
__global__ __launch_bounds__(256) void head_gemm_kernel(const HeadArgs h) {
    __shared__ __attribute__((aligned(16))) float Al[HG_ROWS * HG_LD];
    __shared__ __attribute__((aligned(16))) float Wl[HG_COLS * HG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const float* a = h.a; const float* w = h.w; const float* bias = h.bias; float* out = h.out;
    int k_dim = h.k_dim, out_dim = h.out_dim, tile = blockIdx.x, demodulate = 1;
    int64_t a_stride = h.a_row_stride;
    float scale = h.scale;
    if (MODE != 0) {
        int layer = 0;
        for (int l = 1; l < h.n_layers; ++l)
            if ((int)blockIdx.x >= (int)h.table[l * 8 + 5]) layer = l;
        const int64_t* row = h.table + layer * 8;
        w = reinterpret_cast<const float*>(row[0]);
        tile = (int)blockIdx.x - (int)row[5];
        out_dim = (int)row[4];
        if (MODE == 1) {
            bias = reinterpret_cast<const float*>(row[1]);
            out = h.out + row[2];
            a = h.a + row[3] * h.k_dim;                 // latent[:, index, :]
            a_stride = (int64_t)h.n_latent * h.k_dim;
        } else {
            scale = __int_as_float((int)row[1]);
            k_dim = (int)row[6]; demodulate = (int)row[7];
            a = h.a + row[2]; a_stride = k_dim;         // s[b, :] of this layer
            out = h.out + row[3];
        }
    }
    const int r0 = blockIdx.y * HG_ROWS, o0 = tile * HG_COLS;
    if (MODE == 2 && !demodulate) {
        for (int e = tid; e < HG_ROWS * HG_COLS; e += 256) {
            const int r = r0 + e / HG_COLS, o = o0 + e % HG_COLS;
            if (r < h.batch && o < out_dim) out[(int64_t)r * out_dim + o] = scale;
        }
        return;
    }
    hg_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int srow = tid >> 3, sk = (tid & 7) * 8;  // staging: 8 consecutive k of one row per thread and pass
    // Two register sets: chunk c + 2 is requested while chunk c multiplies (one chunk ahead, the 1 us MFMA phase of a chunk did
    // not cover the loads' latency: ~4.5 us per 64-wide chunk, 41 us for the modulation launch).
    float av[2][8], wv[2][4][8];
    // Interior tiles (whole rows, whole output block, whole aligned chunks -- wave-uniform) load without predicates: a branch
    // around a load makes the next wait a wait for every load in flight, the set requested two chunks ahead included.
    const bool interior = r0 + HG_ROWS <= h.batch && o0 + HG_COLS <= out_dim && k_dim % HG_KC == 0 && a_stride % 4 == 0 &&
                          ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
    auto load_chunk = [&](auto setc, int k0) {  // global -> registers
        constexpr int S = decltype(setc)::value;
        if (interior) {
            const float* src = a + (int64_t)(r0 + srow) * a_stride + k0 + sk;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            av[S][0] = v0.x; av[S][1] = v0.y; av[S][2] = v0.z; av[S][3] = v0.w; av[S][4] = v1.x; av[S][5] = v1.y; av[S][6] = v1.z; av[S][7] = v1.w;
#pragma unroll
            for (int pss = 0; pss < 4; ++pss) {
                const float* ws = w + (int64_t)(o0 + pss * 32 + srow) * k_dim + k0 + sk;
                const float4 w0 = *reinterpret_cast<const float4*>(ws), w1 = *reinterpret_cast<const float4*>(ws + 4);
                wv[S][pss][0] = w0.x; wv[S][pss][1] = w0.y; wv[S][pss][2] = w0.z; wv[S][pss][3] = w0.w;
                wv[S][pss][4] = w1.x; wv[S][pss][5] = w1.y; wv[S][pss][6] = w1.z; wv[S][pss][7] = w1.w;
            }
            return;
        }
        {
            const int r = r0 + srow;
            const float* src = a + (int64_t)r * a_stride + k0 + sk;
            const bool full = r < h.batch && k0 + sk + 8 <= k_dim && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
            if (full) {
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                av[S][0] = v0.x; av[S][1] = v0.y; av[S][2] = v0.z; av[S][3] = v0.w; av[S][4] = v1.x; av[S][5] = v1.y; av[S][6] = v1.z; av[S][7] = v1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) av[S][j] = (r < h.batch && k0 + sk + j < k_dim) ? src[j] : 0.f;
            }
        }
#pragma unroll
        for (int pss = 0; pss < 4; ++pss) {
            const int o = o0 + pss * 32 + srow;
            const float* src = w + (int64_t)o * k_dim + k0 + sk;
            const bool full = o < out_dim && k0 + sk + 8 <= k_dim && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
            if (full) {
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                wv[S][pss][0] = v0.x; wv[S][pss][1] = v0.y; wv[S][pss][2] = v0.z; wv[S][pss][3] = v0.w;
                wv[S][pss][4] = v1.x; wv[S][pss][5] = v1.y; wv[S][pss][6] = v1.z; wv[S][pss][7] = v1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[S][pss][j] = (o < out_dim && k0 + sk + j < k_dim) ? src[j] : 0.f;
            }
        }
    };
    auto chunk = [&](auto setc, int k0) {
        constexpr int S = decltype(setc)::value;
        if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float t = av[S][j] * scale; av[S][j] = t * t; }
        }
        __syncthreads();  // the previous chunk's operand reads are done
#pragma unroll
        for (int j = 0; j < 8; j += 2) *reinterpret_cast<float2*>(Al + srow * HG_LD + sk + j) = make_float2(av[S][j], av[S][j + 1]);
#pragma unroll
        for (int pss = 0; pss < 4; ++pss)
#pragma unroll
            for (int j = 0; j < 8; j += 2)
                *reinterpret_cast<float2*>(Wl + (pss * 32 + srow) * HG_LD + sk + j) = make_float2(wv[S][pss][j], wv[S][pss][j + 1]);
        __syncthreads();
        if (k0 + 2 * HG_KC < k_dim) load_chunk(setc, k0 + 2 * HG_KC);  // this set is free again
        const float* ap = Al + l31 * HG_LD + half;
        const float* wp = Wl + (wave * 32 + l31) * HG_LD + half;
#pragma unroll
        for (int st = 0; st < HG_KC / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], wp[2 * st], acc, 0, 0, 0);
    };
    const std::integral_constant<int, 0> set0;
    const std::integral_constant<int, 1> set1;
    load_chunk(set0, 0);
    if (HG_KC < k_dim) load_chunk(set1, HG_KC);
    for (int k0 = 0; k0 < k_dim; k0 += 2 * HG_KC) {
        chunk(set0, k0);
        if (k0 + HG_KC < k_dim) chunk(set1, k0 + HG_KC);
    }
    const int o = o0 + wave * 32 + l31;
    if (o >= out_dim) return;
    float b = 0.f;
    if (MODE == 0) b = bias ? bias[o] * h.lr_mul : 0.f;
    if (MODE == 1) b = bias ? bias[o] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + (i & 3) + 8 * (i >> 2) + 4 * half;
        if (r >= h.batch) continue;
        float v;
        if (MODE == 2) v = scale * rsqrtf(acc[i] + 1e-8f);
        else v = acc[i] * scale + b;
        if (MODE == 0 && h.activation) v = (v > 0.f ? v : v * 0.2f) * 1.4142135623730951f;
        out[(int64_t)r * out_dim + o] = v;
    }
}

__global__ __launch_bounds__(256) void truncate_kernel(float* __restrict__ out, const float* __restrict__ w,
                                                       const float* __restrict__ mean, float psi, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float m = mean[i % dim]; out[i] = m + psi * (w[i] - m); }
}

// wpk[ci][tap][co] = w[co][ci][tap];  wsq[co][ci] = sum_tap w^2.  One lane per (co, ci).
__global__ __launch_bounds__(256) void prepack_kernel(float* __restrict__ wpk, float* __restrict__ wsq,
                                                      const float* __restrict__ w, int cout, int cin, int taps) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // i = ci * cout + co  (co fastest -> coalesced writes)
    if (i >= (int64_t)cout * cin) return;
    const int co = (int)(i % cout), ci = (int)(i / cout);
    const float* src = w + ((int64_t)co * cin + ci) * taps;
    float ss = 0.f;
    for (int t = 0; t < taps; ++t) {
        const float v = src[t];
        ss += v * v;
        wpk[((int64_t)ci * taps + t) * cout + co] = v;
    }
    wsq[(int64_t)co * cin + ci] = ss;
}

// dscale[b,co] = scale * rsqrt(scale^2 * sum_ci s^2 * wsq + 1e-8): one wave per (b, co).
__global__ __launch_bounds__(256) void demod_kernel(float* __restrict__ dscale, const float* __restrict__ s,
                                                    const float* __restrict__ wsq, int batch, int cin, int cout,
                                                    float scale, int demodulate) {
    const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= (int64_t)batch * cout) return;
    const int b = (int)(idx / cout), co = (int)(idx % cout);
    if (!demodulate) { if (lane == 0) dscale[idx] = scale; return; }
    const float* sr = s + (int64_t)b * cin;
    const float* wr = wsq + (int64_t)co * cin;
    float acc = 0.f;
    for (int i = lane; i < cin; i += 64) { const float sv = sr[i] * scale; acc += sv * sv * wr[i]; }
    acc = wave_sum(acc);
    if (lane == 0) dscale[idx] = scale * rsqrtf(acc + 1e-8f);
}

// ToRGB (model.py:355-364): 1x1 modulated conv to <=4 channels + bias + upsampled skip.
// HBM-bound: reads x once (Cin planes), 4 pixels per lane (float4 when W % 4 == 0).
struct RgbParams {
    int batch, cin, cout, h, w, kh, kw, pad0, sh, sw;
    float scale;
};
constexpr int RGB_MAXC = 4;

__device__ __forceinline__ float skip_up2(const float* __restrict__ sp, const float* __restrict__ taps,
                                          const RgbParams& p, int y, int x) {
    // upfirdn2d with up=2, down=1 at output (y,x): polyphase walk of upfirdn2d_kernel.cu:112-129
    const int mid_x = x + 1 - p.pad0, mid_y = y + 1 - p.pad0;
    const int ix0 = (mid_x >= 0) ? mid_x / 2 : -((1 - mid_x) / 2);
    const int iy0 = (mid_y >= 0) ? mid_y / 2 : -((1 - mid_y) / 2);
    const int kx0 = (ix0 + 1) * 2 - mid_x - 1, ky0 = (iy0 + 1) * 2 - mid_y - 1;
    float v = 0.f;
    for (int fy = ky0, iy = iy0; fy < p.kh; fy += 2, ++iy) {
        if (iy < 0 || iy >= p.sh) continue;
        for (int fx = kx0, ix = ix0; fx < p.kw; fx += 2, ++ix) {
            if (ix < 0 || ix >= p.sw) continue;
            v += sp[iy * p.sw + ix] * taps[(p.kh - 1 - fy) * p.kw + (p.kw - 1 - fx)];
        }
    }
    return v;
}

// One workgroup = one sample x one group of 64*VEC pixels; its 4 waves each take every 4th input channel
// (4x the loads in flight: the low-resolution layers have only B x 1 pixel groups and were latency-bound with
// one wave walking all 512 channels), partial sums meet in LDS, wave 0 adds bias + upsampled skip and stores.
template <int VEC>
__global__ __launch_bounds__(256) void to_rgb_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ s,
                                                     const float* __restrict__ bias, const float* __restrict__ skip,
                                                     const float* __restrict__ taps, RgbParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* weff = smem;                       // [RGB_MAXC][cin]: scale * w[c,ci] * s[b,ci], zero rows past cout
    float* part = smem + RGB_MAXC * p.cin;    // [3 waves][RGB_MAXC][VEC][64]
    const int hw = p.h * p.w;
    const int groups = (hw + 64 * VEC - 1) / (64 * VEC);
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < RGB_MAXC * p.cin; e += 256) {
        const int ci = e % p.cin;
        weff[e] = e < p.cout * p.cin ? p.scale * w[e] * s[(int64_t)b * p.cin + ci] : 0.f;
    }
    __syncthreads();
    const int pix = (g * 64 + lane) * VEC;
    const bool live = pix < hw;
    float acc[RGB_MAXC][VEC];
#pragma unroll
    for (int c = 0; c < RGB_MAXC; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[c][v] = 0.f;
    if (live) {
        // Eight channels of the wave per trip, their loads requested before the first multiply (with `if (c < cout)` inside the
        // loop every channel was load - wait - multiply: one 16-byte load in flight per lane).  All RGB_MAXC output rows are formed
        // (zero weights past cout); a trip's channels past cin re-read the last one with weight 0.
        constexpr int U = 8;
        const float* xb = x + (int64_t)b * p.cin * hw + pix;
        for (int ci0 = wave; ci0 < p.cin; ci0 += 4 * U) {
            float xv[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ci = min(ci0 + 4 * u, p.cin - 1);
                if (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(xb + (int64_t)ci * hw);
                    xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
                } else {
                    xv[u][0] = xb[(int64_t)ci * hw];
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // all eight requests before the first multiply
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ci = ci0 + 4 * u;
                const int cc = min(ci, p.cin - 1);
#pragma unroll
                for (int c = 0; c < RGB_MAXC; ++c) {
                    const float wl = weff[c * p.cin + cc];
                    const float wv = ci < p.cin ? wl : 0.f;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[c][v] += wv * xv[u][v];
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int c = 0; c < RGB_MAXC; ++c)
#pragma unroll
            for (int v = 0; v < VEC; ++v) part[(((wave - 1) * RGB_MAXC + c) * VEC + v) * 64 + lane] = acc[c][v];
    }
    __syncthreads();
    if (wave != 0 || !live) return;
    // Upsampled skip image, taps <= 4 x 4 (two per dimension after the polyphase split): per output channel the 4 * VEC skip
    // values and tap weights are requested together with clamped indices, then masked and added in skip_up2's order (the
    // per-tap `continue`s of skip_up2 made every tap a dependent round trip: 24 per lane, as long as the channel loop itself).
    const bool skip_fast = skip && p.kh <= 4 && p.kw <= 4;
    int s_off[VEC][4], t_off[VEC][4];
    unsigned s_ok = 0;
    if (skip_fast) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int y = (pix + v) / p.w, xx = (pix + v) % p.w;
            const int mid_x = xx + 1 - p.pad0, mid_y = y + 1 - p.pad0;
            const int ix0 = (mid_x >= 0) ? mid_x / 2 : -((1 - mid_x) / 2);
            const int iy0 = (mid_y >= 0) ? mid_y / 2 : -((1 - mid_y) / 2);
            const int kx0 = (ix0 + 1) * 2 - mid_x - 1, ky0 = (iy0 + 1) * 2 - mid_y - 1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int iy = iy0 + (t >> 1), ix = ix0 + (t & 1), fy = ky0 + 2 * (t >> 1), fx = kx0 + 2 * (t & 1);
                const bool ok = iy >= 0 && iy < p.sh && ix >= 0 && ix < p.sw && fy < p.kh && fx < p.kw;
                s_ok |= (ok ? 1u : 0u) << (v * 4 + t);
                s_off[v][t] = min(max(iy, 0), p.sh - 1) * p.sw + min(max(ix, 0), p.sw - 1);
                t_off[v][t] = (p.kh - 1 - min(fy, p.kh - 1)) * p.kw + (p.kw - 1 - min(fx, p.kw - 1));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < RGB_MAXC; ++c) {
        if (c >= p.cout) continue;
        const float bb = bias ? bias[c] : 0.f;
        float* o = out + ((int64_t)b * p.cout + c) * hw + pix;
        float r[VEC];
        float sv[VEC][4], tv[VEC][4];
        if (skip_fast) {
            const float* sp = skip + ((int64_t)b * p.cout + c) * p.sh * p.sw;
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int t = 0; t < 4; ++t) { sv[v][t] = sp[s_off[v][t]]; tv[v][t] = taps[t_off[v][t]]; }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float a = acc[c][v];
#pragma unroll
            for (int ww = 0; ww < 3; ++ww) a += part[((ww * RGB_MAXC + c) * VEC + v) * 64 + lane];
            a += bb;
            if (skip_fast) {
                float u = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if ((s_ok >> (v * 4 + t)) & 1u) u += sv[v][t] * tv[v][t];
                a += u;
            } else if (skip) {
                const int y = (pix + v) / p.w, xx = (pix + v) % p.w;
                a += skip_up2(skip + ((int64_t)b * p.cout + c) * p.sh * p.sw, taps, p, y, xx);
            }
            r[v] = a;
        }
        if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
        else o[0] = r[0];
    }
}

}  // namespace

extern "C" int sis_pixel_norm(float* out, const float* x, int batch, int dim, void* stream) {
    if (batch <= 0 || dim <= 0) return 0;
    SIS_REQUIRE(out && x, "sis_pixel_norm: null pointer");
    hipLaunchKernelGGL(pixel_norm_kernel, dim3(sis_cdiv(batch, 4)), dim3(256), 0, (hipStream_t)stream, out, x, batch, dim);
    SIS_CHECK_LAUNCH("sis_pixel_norm");
    return 0;
}

extern "C" int sis_head_gemm_tile() { return HG_COLS; }

extern "C" int sis_equal_linear(float* out, const float* x, int64_t x_row_stride, const float* w, const float* bias,
                                int batch, int in_dim, int out_dim, float scale, float lr_mul, int activation,
                                void* stream) {
    if (batch <= 0 || out_dim <= 0) return 0;
    SIS_REQUIRE(out && x && w, "sis_equal_linear: null pointer");
    SIS_REQUIRE(in_dim > 0, "sis_equal_linear: in_dim %d", in_dim);
    const dim3 grid(sis_cdiv(out_dim, 4), sis_cdiv(batch, EL_ROWS));
    if (in_dim == 512) {
        hipLaunchKernelGGL(equal_linear_full_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, out, x, x_row_stride, w, bias, batch, out_dim,
                           scale, lr_mul, activation);
    } else if (in_dim == 256) {
        hipLaunchKernelGGL(equal_linear_full_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, out, x, x_row_stride, w, bias, batch, out_dim,
                           scale, lr_mul, activation);
    } else if (in_dim <= 64 * ELW) {
        hipLaunchKernelGGL(equal_linear_kernel, grid, dim3(256), 0, (hipStream_t)stream, out, x,
                           x_row_stride, w, bias, batch, in_dim, out_dim, scale, lr_mul, activation);
    } else {  // wider than a wave's register row: the matrix-core tile kernel takes any in_dim
        HeadArgs h = {};
        h.out = out; h.a = x; h.w = w; h.bias = bias; h.a_row_stride = x_row_stride; h.batch = batch; h.k_dim = in_dim;
        h.out_dim = out_dim; h.scale = scale; h.lr_mul = lr_mul; h.activation = activation;
        hipLaunchKernelGGL(head_gemm_kernel<0>, dim3(sis_cdiv(out_dim, HG_COLS), sis_cdiv(batch, HG_ROWS)), dim3(256), 0, (hipStream_t)stream, h);
    }
    SIS_CHECK_LAUNCH("sis_equal_linear");
    return 0;
}

extern "C" int sis_modulation_batch(float* out_base, const float* latent, const int64_t* table, int n_layers,
                                    int total_blocks, int batch, int n_latent, int dim, float scale, void* stream) {
    if (n_layers <= 0 || batch <= 0) return 0;
    SIS_REQUIRE(out_base && latent && table, "sis_modulation_batch: null pointer");
    SIS_REQUIRE(dim > 0, "sis_modulation_batch: style dim %d", dim);
    HeadArgs h = {};
    h.out = out_base; h.a = latent; h.table = table; h.batch = batch; h.k_dim = dim; h.n_layers = n_layers; h.n_latent = n_latent;
    h.scale = scale;
    hipLaunchKernelGGL(head_gemm_kernel<1>, dim3(total_blocks, sis_cdiv(batch, HG_ROWS)), dim3(256), 0, (hipStream_t)stream, h);
    SIS_CHECK_LAUNCH("sis_modulation_batch");
    return 0;
}

extern "C" int sis_demod_batch(float* dscale_base, const float* s_base, const int64_t* table, int n_layers,
                               int total_blocks, int batch, void* stream) {
    if (n_layers <= 0 || batch <= 0) return 0;
    SIS_REQUIRE(dscale_base && s_base && table, "sis_demod_batch: null pointer");
    HeadArgs h = {};
    h.out = dscale_base; h.a = s_base; h.table = table; h.batch = batch; h.n_layers = n_layers;
    hipLaunchKernelGGL(head_gemm_kernel<2>, dim3(total_blocks, sis_cdiv(batch, HG_ROWS)), dim3(256), 0, (hipStream_t)stream, h);
    SIS_CHECK_LAUNCH("sis_demod_batch");
    return 0;
}

extern "C" int sis_truncate(float* out, const float* w, const float* mean, float psi, int batch, int dim,
                            void* stream) {
    const int64_t n = (int64_t)batch * dim;
    if (n <= 0) return 0;
    SIS_REQUIRE(out && w && mean, "sis_truncate: null pointer");
    hipLaunchKernelGGL(truncate_kernel, dim3(sis_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, w, mean, psi, n, dim);
    SIS_CHECK_LAUNCH("sis_truncate");
    return 0;
}

extern "C" int sis_modconv_prepack(float* wpk, float* wsq, const float* w, int cout, int cin, int ksize,
                                   void* stream) {
    SIS_REQUIRE(wpk && wsq && w, "sis_modconv_prepack: null pointer");
    SIS_REQUIRE(cout > 0 && cin > 0 && ksize > 0, "sis_modconv_prepack: bad sizes");
    const int64_t n = (int64_t)cout * cin;
    hipLaunchKernelGGL(prepack_kernel, dim3(sis_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, wpk, wsq, w, cout, cin,
                       ksize * ksize);
    SIS_CHECK_LAUNCH("sis_modconv_prepack");
    return 0;
}

extern "C" int sis_modconv_demod(float* dscale, const float* s, const float* wsq, int batch, int cin, int cout,
                                 float scale, int demodulate, void* stream) {
    if (batch <= 0 || cout <= 0) return 0;
    SIS_REQUIRE(dscale && (!demodulate || (s && wsq)), "sis_modconv_demod: null pointer");
    hipLaunchKernelGGL(demod_kernel, dim3(sis_cdiv((int64_t)batch * cout, 4)), dim3(256), 0, (hipStream_t)stream, dscale, s,
                       wsq, batch, cin, cout, scale, demodulate);
    SIS_CHECK_LAUNCH("sis_modconv_demod");
    return 0;
}

extern "C" int sis_to_rgb(float* out, const float* x, const float* w, const float* s, const float* bias,
                          const float* skip, const float* taps, int batch, int cin, int cout, int h, int wd, int kh,
                          int kw, int pad0, int pad1, float scale, void* stream) {
    if (batch <= 0 || h <= 0 || wd <= 0) return 0;
    SIS_REQUIRE(out && x && w && s, "sis_to_rgb: null pointer");
    SIS_REQUIRE(cout >= 1 && cout <= RGB_MAXC, "sis_to_rgb: cout %d outside 1..%d", cout, RGB_MAXC);
    SIS_REQUIRE(cin >= 1 && (size_t)cin * RGB_MAXC * 4 <= 48 * 1024, "sis_to_rgb: cin too large");
    RgbParams p;
    p.batch = batch; p.cin = cin; p.cout = cout; p.h = h; p.w = wd; p.kh = kh; p.kw = kw; p.pad0 = pad0;
    p.sh = 0; p.sw = 0; p.scale = scale;
    if (skip) {
        SIS_REQUIRE(taps && kh >= 1 && kw >= 1, "sis_to_rgb: skip given without taps");
        SIS_REQUIRE(h % 2 == 0 && wd % 2 == 0, "sis_to_rgb: skip path needs even output size");
        p.sh = h / 2; p.sw = wd / 2;
        SIS_REQUIRE(sis_upfirdn2d_out_size(p.sh, 2, 1, pad0, pad1, kh) == h &&
                        sis_upfirdn2d_out_size(p.sw, 2, 1, pad0, pad1, kw) == wd,
                    "sis_to_rgb: upsampled skip would not match the %dx%d output", h, wd);
    }
    const int hw = h * wd;
    hipStream_t st = (hipStream_t)stream;
    if (hw % 4 == 0 && hw >= 256 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0) {
        const size_t lds = ((size_t)cin * RGB_MAXC + 3 * RGB_MAXC * 4 * 64) * sizeof(float);
        hipLaunchKernelGGL(to_rgb_kernel<4>, dim3(batch * sis_cdiv(hw, 256)), dim3(256), lds, st, out, x, w, s, bias, skip,
                           taps, p);
    } else {
        const size_t lds = ((size_t)cin * RGB_MAXC + 3 * RGB_MAXC * 64) * sizeof(float);
        hipLaunchKernelGGL(to_rgb_kernel<1>, dim3(batch * sis_cdiv(hw, 64)), dim3(256), lds, st, out, x, w, s, bias, skip,
                           taps, p);
    }
    SIS_CHECK_LAUNCH("sis_to_rgb");
    return 0;
}
