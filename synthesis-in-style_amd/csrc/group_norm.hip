// GroupNorm (+ optional ReLU) of TransUNet's ResNetV2 trunk (networks/trans_u_net/vit_seg_modeling_resnet_skip.py:
// 40-75,114-126: a GroupNorm after every StdConv2d, ReLU after most), forward and backward, with 16-bit or fp32
// tensors in and out and fp32 arithmetic.  Under bf16 autocast ATen runs each of the 52 norms as: cast to fp32, row
// moments, normalise, (ReLU), cast back -- and twice that in backward; here one launch per direction reads the
// convolution's bf16 output and writes what the next convolution consumes.
//
// Two launches per direction, each fully parallel: (1) one workgroup per (sample, channel) plane reduces it -- mean /
// M2 forward, sum(g') / sum(g' * xhat) backward (g' = g masked by the recomputed ReLU: nothing but x is saved); the
// workgroup that completes a group (a device-scope counter per group, left at zero again: sis_xwg.h) merges the group's planes
// (Chan's formula, in channel order whichever workgroup does it: deterministic) into per-plane scale / shift
// coefficients; (2) a grid-strided element-wise kernel applies them (4 elements per lane when HW % 4 == 0), and its first
// workgroup adds the plane sums over the batch in fixed order: d(gamma) / d(beta).  (Batch-norm mode and callers without a
// counter buffer keep the merge as a launch of its own.)
#include <cstdlib>
#include "sis_common.h"
#include "sis_xwg.h"

namespace {

__device__ __forceinline__ float gn_block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// VEC consecutive elements as floats: one 16-byte (f32) or 8-byte (f16 / bf16) load when VEC == 4
template <int VEC, typename T>
__device__ __forceinline__ void gn_load(const T* p, float* v) {
    if constexpr (VEC == 4 && sizeof(T) == 4) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else if constexpr (VEC == 4 && sizeof(T) == 2) {
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        T t[4];
        __builtin_memcpy(t, &q, 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sis_ld(t, e);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = sis_ld(p, e);
    }
}

template <int VEC, typename T>
__device__ __forceinline__ void gn_store(T* p, const float* v) {
    if constexpr (VEC == 4 && sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (VEC == 4 && sizeof(T) == 2) {
        T t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) sis_st(t, e, v[e]);
        uint2 q;
        __builtin_memcpy(&q, t, 8);
        *reinterpret_cast<uint2*>(p) = q;
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) sis_st(p, e, v[e]);
    }
}

// Elements [lo, hi) of a plane whose first element sits at flat offset `base` of a vector-aligned tensor: scalar head up
// to the next multiple of VEC, vector body, scalar tail -- planes of odd size (127 x 127 after the un-padded max-pool)
// still move 8 / 16 bytes per lane.  VEC == 1: everything through `scalar`.
template <int VEC, typename FS, typename FV>
__device__ __forceinline__ void gn_span(int64_t base, int lo, int hi, FS scalar, FV vec) {
    int a0 = lo, a1 = lo;
    if constexpr (VEC > 1) {
        a0 = lo + (int)((VEC - ((base + lo) & (VEC - 1))) & (VEC - 1));
        if (a0 > hi) a0 = hi;
        a1 = a0 + ((hi - a0) & ~(VEC - 1));
        // four vectors per trip: their loads are requested together (these workgroups are short and latency-bound, a dependent
        // load-use chain per vector cost the statistics / reduction passes half their bandwidth)
        int i = a0 + threadIdx.x * VEC;
        for (; i + 3 * 256 * VEC < a1; i += 4 * 256 * VEC) { vec(i); vec(i + 256 * VEC); vec(i + 2 * 256 * VEC); vec(i + 3 * 256 * VEC); }
        for (; i < a1; i += 256 * VEC) vec(i);
    }
    const int nh = a0 - lo, nt = hi - a1;
    for (int j = threadIdx.x; j < nh + nt; j += 256) scalar(j < nh ? lo + j : a1 + (j - nh));
}

// ---- statistics: one workgroup per (sample, channel) plane -> (mean, M2) of the plane; gn_row_finish merges the
// planes of a group with Chan's formula in channel order (no E[x^2] - E[x]^2 cancellation, deterministic).
// Planes are cut into S slices of `sl` elements (blockIdx.y) so that few-channel, high-resolution tensors (the decoder's
// 16 x 512^2 maps) still fill the chip; part[plane][slice] = (count, mean, M2).
__device__ __forceinline__ void gn_row_finish(int row, int lane, int lanes, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                              float* __restrict__ ab, const float* __restrict__ part, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, int groups, int cpg, int S, float eps);
__device__ __forceinline__ void gn_bwd_row(int row, int lane, int lanes, float* __restrict__ coef, float* __restrict__ psum,
                                           const float* __restrict__ part, const float* __restrict__ rstd_in,
                                           const float* __restrict__ gamma, int groups, int cpg, int hw, int S);

struct GnFinish {   // counters == nullptr: no merge in the statistics kernel
    int* counters; float* mean; float* rstd; float* ab; const float* gamma; const float* beta; int groups, cpg; float eps;
    // batch-norm mode (bn_batch > 0, groups = C): one counter per CHANNEL, bn_batch * slices arrivals
    int bn_batch = 0; float* running_mean = nullptr; float* running_var = nullptr; float momentum = 0.f;
};
__device__ __forceinline__ void bn_chan_finish_wave(int c, int lane, const GnFinish& f, const float* __restrict__ part, int S);

template <typename TI, int VEC>
__global__ __launch_bounds__(256) void gn_plane_stats_kernel(float* __restrict__ part, const TI* __restrict__ x, int hw,
                                                             int sl, GnFinish fin) {
    __shared__ float red[4];
    const int lo = blockIdx.y * sl, hi = min(hw, lo + sl);
    const int64_t base = (int64_t)blockIdx.x * hw;
    const TI* pl = x + base;
    float s = 0.f;
    gn_span<VEC>(base, lo, hi, [&](int i) { s += sis_ld(pl, i); }, [&](int i) {
        float v[VEC];
        gn_load<VEC>(pl + i, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += v[e];
    });
    const float cnt = (float)(hi - lo);
    const float mean = gn_block_sum(s, red) / cnt;
    float m2 = 0.f;
    gn_span<VEC>(base, lo, hi, [&](int i) { const float d = sis_ld(pl, i) - mean; m2 += d * d; }, [&](int i) {
        float v[VEC];
        gn_load<VEC>(pl + i, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float d = v[e] - mean; m2 += d * d; }
    });
    m2 = gn_block_sum(m2, red);
    if (threadIdx.x == 0) {
        float* o = part + 3 * ((int64_t)blockIdx.x * gridDim.y + blockIdx.y);
        xwg_publish(o, cnt); xwg_publish(o + 1, mean); xwg_publish(o + 2, m2);
    }
    if (fin.counters && fin.bn_batch) {   // batch norm: plane = sample * C + channel
        const int c = (int)(blockIdx.x % fin.groups);
        if (xwg_complete<false>(fin.counters + c, fin.bn_batch * gridDim.y) && threadIdx.x < 64)
            bn_chan_finish_wave(c, threadIdx.x, fin, part, gridDim.y);
    } else if (fin.counters) {
        const int row = blockIdx.x / fin.cpg;
        if (xwg_complete<false>(fin.counters + row, fin.cpg * gridDim.y) && threadIdx.x < 64)
            gn_row_finish(row, threadIdx.x, 64, fin.mean, fin.rstd, fin.ab, part, fin.gamma, fin.beta, fin.groups, fin.cpg, gridDim.y, fin.eps);
    }
}

// bn_chan_finish_kernel's work for ONE channel by the wave that completed it: 64 partial results per round trip to memory,
// merged in the same (sample, slice) order as the kernel below.
__device__ __forceinline__ void bn_chan_finish_wave(int c, int lane, const GnFinish& f, const float* __restrict__ part, int S) {
    const int C = f.groups, total = f.bn_batch * S;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int j0 = 0; j0 < total; j0 += 64) {
        const int j = j0 + lane, cnt = min(64, total - j0);
        float pn = 0.f, pm = 0.f, p2 = 0.f;
        if (j < total) {
            const int b = j / S, k = j - b * S;
            const float* p = part + 3 * (((int64_t)b * C + c) * S + k);
            pn = xwg_peek(p); pm = xwg_peek(p + 1); p2 = xwg_peek(p + 2);
        }
        for (int t = 0; t < cnt; ++t) {
            const float nb = __shfl(pn, t, 64), mb = __shfl(pm, t, 64), m2b = __shfl(p2, t, 64);
            const float nt = n + nb, delta = mb - mean;
            mean += delta * (nb / nt);
            m2 += m2b + delta * delta * (n * nb / nt);
            n = nt;
        }
    }
    const float rstd = rsqrtf(m2 / n + f.eps);
    if (lane == 0) {
        f.mean[c] = mean; f.rstd[c] = rstd;
        if (f.running_mean) {
            f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (n > 1.f ? m2 / (n - 1.f) : m2 / n);
        }
    }
    const float a = rstd * f.gamma[c], sh = f.beta[c] - mean * a;
    for (int b = lane; b < f.bn_batch; b += 64) { f.ab[2 * ((int64_t)b * C + c)] = a; f.ab[2 * ((int64_t)b * C + c) + 1] = sh; }
}

// per (sample, channel): a = rstd_row * gamma_c, b = beta_c - mean_row * a  (the apply kernel's scale / shift).
// `lane` / `lanes`: the channels of the group are spread over the calling lanes, the merge itself is done by each of them.
__device__ __forceinline__ void gn_row_finish(int row, int lane, int lanes, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                              float* __restrict__ ab, const float* __restrict__ part, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, int groups, int cpg, int S, float eps) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
    auto merge = [&](float nb, float mb, float m2b) {
        const float nt = n + nb, delta = mb - mean;
        mean += delta * (nb / nt);
        m2 += m2b + delta * delta * (n * nb / nt);
        n = nt;
    };
    // the group's planes are adjacent: (row * cpg + c) * S + slice
    if (lanes == 64) {   // one wave: 64 partial results per round trip to memory, merged in the same (index) order
        for (int j0 = 0; j0 < cpg * S; j0 += 64) {
            const int j = j0 + lane, cnt = min(64, cpg * S - j0);
            float pn = 0.f, pm = 0.f, p2 = 0.f;
            if (j < cpg * S) {
                const float* p = part + 3 * ((int64_t)row * cpg * S + j);
                pn = xwg_peek(p); pm = xwg_peek(p + 1); p2 = xwg_peek(p + 2);
            }
            for (int t = 0; t < cnt; ++t) merge(__shfl(pn, t, 64), __shfl(pm, t, 64), __shfl(p2, t, 64));
        }
    } else {
        for (int j = 0; j < cpg * S; ++j) {
            const float* p = part + 3 * ((int64_t)row * cpg * S + j);
            merge(xwg_peek(p), xwg_peek(p + 1), xwg_peek(p + 2));
        }
    }
    const float rstd = rsqrtf(m2 / n + eps);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const int c0 = (row % groups) * cpg;
    for (int c = lane; c < cpg; c += lanes) {
        const float a = rstd * gamma[c0 + c];
        ab[2 * ((int64_t)row * cpg + c)] = a;
        ab[2 * ((int64_t)row * cpg + c) + 1] = beta[c0 + c] - mean * a;
    }
}

__global__ __launch_bounds__(64) void gn_row_finish_kernel(float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           float* __restrict__ ab, const float* __restrict__ part,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int rows, int groups, int cpg, int S, float eps) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    gn_row_finish(row, 0, 1, mean_out, rstd_out, ab, part, gamma, beta, groups, cpg, S, eps);
}

// ReLU gate of the residual form as ONE BIT per element (bit i of a flat bit array, element i of the flat tensor): the forward's
// apply pass writes it, both backward passes read 1/8 byte instead of the 4-byte output per element (the residual norms' outputs
// are the trunk's fp32 stream: block 1's are 132 MB each).
__device__ __forceinline__ float gn_gate(const unsigned char* __restrict__ bits, int64_t idx) {   // 1 = open, 0 = closed
    return (float)((bits[idx >> 3] >> (idx & 7)) & 1);
}

// y = relu?(x * a[plane] + b[plane] (+ residual)); VEC elements per lane.  FLAT: hw is not a multiple of VEC, a lane's
// vector may straddle two planes (VEC <= hw: at most one boundary) and picks its coefficients per element.
template <typename TI, typename TO, int VEC, bool FLAT>
__global__ __launch_bounds__(256) void gn_apply_kernel(TO* __restrict__ y, TI* __restrict__ y_lp, const TI* __restrict__ x,
                                                       const float* __restrict__ ab, const float* __restrict__ res, int hw,
                                                       int64_t total, int relu, unsigned char* __restrict__ bits) {
    static_assert(VEC == 4 || VEC == 1, "the gate bits are assembled from wave ballots for these two widths");
    const int64_t stride = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < total; i += stride) {
        const int64_t plane = i / hw;
        const int rem = FLAT ? (int)(i - plane * hw) : 0;
        const float a0 = ab[2 * plane], b0 = ab[2 * plane + 1];
        float a1 = a0, b1 = b0;
        if (FLAT && rem + VEC > hw) { a1 = ab[2 * plane + 2]; b1 = ab[2 * plane + 3]; }
        float xv[VEC], rv[VEC], out[VEC];
        gn_load<VEC>(x + i, xv);
        if (res) gn_load<VEC>(res + i, rv);  // residual sum of a bottleneck (fp32), before the ReLU
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool nx = FLAT && rem + e >= hw;
            float v = xv[e] * (nx ? a1 : a0) + (nx ? b1 : b0);
            if (res) v += rv[e];
            if (relu) v = fmaxf(v, 0.f);
            out[e] = v;
        }
        gn_store<VEC>(y + i, out);
        if (y_lp) gn_store<VEC>(y_lp + i, out);  // the same values rounded to the convolutions' dtype (what autocast would cast to)
        if (bits) {   // (a wave's lanes hold consecutive vectors; lanes past the end are not in the ballots)
            const int ln = threadIdx.x & 63;
            if constexpr (VEC == 4) {
                unsigned long long bal[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) bal[e] = __ballot(out[e] > 0.f);
                if ((ln & 1) == 0) {   // an even lane and its neighbour share a byte (i % 8 == 0 here)
                    unsigned byte = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) byte |= (unsigned)((bal[e] >> ln) & 1ull) << e | (unsigned)((bal[e] >> (ln + 1)) & 1ull) << (4 + e);
                    bits[i >> 3] = (unsigned char)byte;
                }
            } else {
                const unsigned long long bal = __ballot(out[0] > 0.f);
                if (ln == 0) *reinterpret_cast<unsigned long long*>(bits + (i >> 3)) = bal;   // 64 elements = 8 bytes (i % 64 == 0)
            }
        }
    }
}

// bn_bwd_chan_kernel's work for ONE channel by the wave that completed it (same (sample, slice) order of the sums)
__device__ __forceinline__ void bn_bwd_chan_wave(int c, int lane, float* __restrict__ coef, float* __restrict__ dgamma,
                                                 float* __restrict__ dbeta, const float* __restrict__ part,
                                                 const float* __restrict__ rstd_in, const float* __restrict__ gamma, int batch, int C,
                                                 int hw, int S) {
    const int total = batch * S;
    float s1 = 0.f, s2 = 0.f;
    for (int j0 = 0; j0 < total; j0 += 64) {
        const int j = j0 + lane, cnt = min(64, total - j0);
        float a = 0.f, b2 = 0.f;
        if (j < total) {
            const int b = j / S, k = j - b * S;
            const float* p = part + 2 * (((int64_t)b * C + c) * S + k);
            a = xwg_peek(p); b2 = xwg_peek(p + 1);
        }
        for (int t = 0; t < cnt; ++t) { s1 += __shfl(a, t, 64); s2 += __shfl(b2, t, 64); }
    }
    if (lane == 0) { dbeta[c] = s1; dgamma[c] = s2; }
    const float n = (float)batch * (float)hw, k1 = rstd_in[c] * gamma[c];
    for (int b = lane; b < batch; b += 64) {
        float* k = coef + 3 * ((int64_t)b * C + c);
        k[0] = k1; k[1] = k1 * s1 / n; k[2] = k1 * s2 / n;
    }
}

// ---- backward.  g' = g * [y > 0] (mask recomputed: y = x*a + b).  Per plane: sum(g'), sum(g' * xhat).
template <typename TI, typename TG, int VEC>
__global__ __launch_bounds__(256) void gn_bwd_plane_kernel(float* __restrict__ part, const TG* __restrict__ g,
                                                           const TI* __restrict__ g_lp, const TI* __restrict__ x, const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ ymask, int C,
                                                           int cpg, int hw, int sl, int relu, int* __restrict__ counters,
                                                           float* __restrict__ coef, float* __restrict__ psum,
                                                           const unsigned char* __restrict__ bits, int bn_batch = 0,
                                                           float* __restrict__ bn_dgamma = nullptr, float* __restrict__ bn_dbeta = nullptr) {
    __shared__ float red[4];
    const int64_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const int64_t row = cpg > 0 ? plane / cpg : c;  // group norm: sample * groups + group; batch norm (cpg = 0): channel
    const float mean = mean_in[row], rstd = rstd_in[row], gm = gamma[c], bt = beta[c];
    const TI* px = x + plane * hw;
    const TG* pg = g + plane * hw;
    const TI* pg2 = g_lp ? g_lp + plane * hw : nullptr;  // gradient that arrived through the 16-bit copy of the output
    const float* pm = ymask ? ymask + plane * hw : nullptr;
    const int lo = blockIdx.y * sl, hi = min(hw, lo + sl);
    float sg = 0.f, sgx = 0.f;
    const bool gated = pm || bits;   // the gate comes from the saved output (its sign) or from the forward's bit per element
    const int64_t flat0 = plane * hw;
    auto one = [&](float xe, float ge, float me) {
        const float xh = (xe - mean) * rstd;
        if (relu && (gated ? me <= 0.f : xh * gm + bt <= 0.f)) ge = 0.f;
        sg += ge; sgx += ge * xh;
    };
    gn_span<VEC>(plane * hw, lo, hi, [&](int i) {
        one(sis_ld(px, i), sis_ld(pg, i) + (pg2 ? sis_ld(pg2, i) : 0.f), pm ? pm[i] : bits ? gn_gate(bits, flat0 + i) : 1.f);
    }, [&](int i) {
        float xv[VEC], gv[VEC], g2[VEC], mv[VEC];
        gn_load<VEC>(px + i, xv);
        gn_load<VEC>(pg + i, gv);
        if (pg2) gn_load<VEC>(pg2 + i, g2);
        if (pm) gn_load<VEC>(pm + i, mv);
        else if (bits) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) mv[e] = gn_gate(bits, flat0 + i + e);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) one(xv[e], gv[e] + (pg2 ? g2[e] : 0.f), gated ? mv[e] : 1.f);
    });
    sg = gn_block_sum(sg, red);
    sgx = gn_block_sum(sgx, red);
    if (threadIdx.x == 0) {
        float* o = part + 2 * (plane * gridDim.y + blockIdx.y);
        xwg_publish(o, sg); xwg_publish(o + 1, sgx);
    }
    if (counters && cpg == 0) {   // batch norm: the channel's last workgroup does bn_bwd_chan_kernel's work
        if (xwg_complete<false>(counters + c, bn_batch * gridDim.y) && threadIdx.x < 64)
            bn_bwd_chan_wave(c, threadIdx.x, coef, bn_dgamma, bn_dbeta, part, rstd_in, gamma, bn_batch, C, hw, gridDim.y);
    } else if (counters) {
        if (xwg_complete<false>(counters + row, cpg * gridDim.y) && threadIdx.x < 64)
            gn_bwd_row((int)row, threadIdx.x, 64, coef, psum, part, rstd_in, gamma, C / cpg, cpg, hw, gridDim.y);
    }
}

// per row: m1 = mean(g'*gamma), m2 = mean(g'*gamma*xhat) -> per plane coefficients (k1, k2, k3) with
// dx = k1 * g' - k2 - k3 * xhat;  k1 = rstd*gamma_c, k2 = rstd*m1, k3 = rstd*m2
__device__ __forceinline__ void gn_bwd_row(int row, int lane, int lanes, float* __restrict__ coef, float* __restrict__ psum,
                                           const float* __restrict__ part, const float* __restrict__ rstd_in,
                                           const float* __restrict__ gamma, int groups, int cpg, int hw, int S) {
    const int c0 = (row % groups) * cpg;
    float s1 = 0.f, s2 = 0.f;
    if (lanes == 64) {   // lane = channel: its S slices summed in slice order, then the channels in channel order
        for (int cb = 0; cb < cpg; cb += 64) {
            const int c = cb + lane, cnt = min(64, cpg - cb);
            float a = 0.f, b = 0.f, gm = 0.f;
            if (c < cpg) {
                for (int k = 0; k < S; ++k) {
                    const float* p = part + 2 * (((int64_t)row * cpg + c) * S + k);
                    a += xwg_peek(p); b += xwg_peek(p + 1);
                }
                psum[2 * ((int64_t)row * cpg + c)] = a; psum[2 * ((int64_t)row * cpg + c) + 1] = b;  // plane sums (for d gamma / beta)
                gm = gamma[c0 + c];
            }
            for (int t = 0; t < cnt; ++t) { s1 += __shfl(gm, t, 64) * __shfl(a, t, 64); s2 += __shfl(gm, t, 64) * __shfl(b, t, 64); }
        }
    } else {
        for (int c = 0; c < cpg; ++c) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < S; ++k) {
                const float* p = part + 2 * (((int64_t)row * cpg + c) * S + k);
                a += xwg_peek(p); b += xwg_peek(p + 1);
            }
            if (c % lanes == lane) {  // plane sums (for d gamma / beta)
                psum[2 * ((int64_t)row * cpg + c)] = a; psum[2 * ((int64_t)row * cpg + c) + 1] = b;
            }
            s1 += gamma[c0 + c] * a; s2 += gamma[c0 + c] * b;
        }
    }
    const float n = (float)cpg * (float)hw, rstd = rstd_in[row];
    for (int c = lane; c < cpg; c += lanes) {
        float* k = coef + 3 * ((int64_t)row * cpg + c);
        k[0] = rstd * gamma[c0 + c]; k[1] = rstd * s1 / n; k[2] = rstd * s2 / n;
    }
}

__global__ __launch_bounds__(64) void gn_bwd_row_kernel(float* __restrict__ coef, float* __restrict__ psum,
                                                        const float* __restrict__ part, const float* __restrict__ rstd_in,
                                                        const float* __restrict__ gamma, int rows, int groups, int cpg, int hw,
                                                        int S) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    gn_bwd_row(row, 0, 1, coef, psum, part, rstd_in, gamma, groups, cpg, hw, S);
}

// d(gamma) / d(beta) of channel c = its plane sums over the batch, in sample order
__device__ __forceinline__ void gn_param_reduce(int c, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                const float* __restrict__ psum, int batch, int C) {
    float a = 0.f, b = 0.f;
    for (int n = 0; n < batch; ++n) { b += psum[2 * ((int64_t)n * C + c)]; a += psum[2 * ((int64_t)n * C + c) + 1]; }
    dgamma[c] = a; dbeta[c] = b;
}

template <typename TI, typename TG, int VEC, bool FLAT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(TI* __restrict__ dx, const TG* __restrict__ g,
                                                           const TI* __restrict__ g_lp, const TI* __restrict__ x,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ ymask, float* __restrict__ dres, int C,
                                                           int cpg, int hw, int64_t total, int relu, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, const float* __restrict__ psum, int batch,
                                                           const unsigned char* __restrict__ bits) {
    if (dgamma && blockIdx.x == 0)   // the plane sums are complete since the previous launch
        for (int c = threadIdx.x; c < C; c += 256) gn_param_reduce(c, dgamma, dbeta, psum, batch, C);
    struct PlaneCoef { float mean, rstd, gm, bt, k1, k2, k3; };
    auto coefs = [&](int64_t plane) {
        const int c = (int)(plane % C);
        const int64_t row = cpg > 0 ? plane / cpg : c;
        return PlaneCoef{mean_in[row], rstd_in[row], gamma[c], beta[c], coef[3 * plane], coef[3 * plane + 1], coef[3 * plane + 2]};
    };
    const int64_t stride = (int64_t)gridDim.x * 256 * VEC;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < total; i += stride) {
        const int64_t plane = i / hw;
        const int rem = FLAT ? (int)(i - plane * hw) : 0;
        const PlaneCoef p0 = coefs(plane);
        PlaneCoef p1 = p0;
        if (FLAT && rem + VEC > hw) p1 = coefs(plane + 1);
        float xv[VEC], gv[VEC], g2[VEC], mv[VEC], dxv[VEC], drv[VEC];
        gn_load<VEC>(x + i, xv);
        gn_load<VEC>(g + i, gv);
        if (g_lp) gn_load<VEC>(g_lp + i, g2);
        if (ymask) gn_load<VEC>(ymask + i, mv);
        else if (bits) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) mv[e] = gn_gate(bits, i + e);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const PlaneCoef& p = (FLAT && rem + e >= hw) ? p1 : p0;
            const float xh = (xv[e] - p.mean) * p.rstd;
            float gi = gv[e] + (g_lp ? g2[e] : 0.f);
            if (relu && ((ymask || bits) ? mv[e] <= 0.f : xh * p.gm + p.bt <= 0.f)) gi = 0.f;
            drv[e] = gi;  // gradient of the residual branch = masked incoming gradient
            dxv[e] = p.k1 * gi - p.k2 - p.k3 * xh;
        }
        if (dres) gn_store<VEC>(dres + i, drv);
        gn_store<VEC>(dx + i, dxv);
    }
}

__global__ __launch_bounds__(256) void gn_param_reduce_kernel(float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              const float* __restrict__ part, int batch, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) gn_param_reduce(c, dgamma, dbeta, part, batch, C);
}

// ---- batch-norm mode (statistics per channel over the batch): the planes of channel c are n * C + c.
__global__ __launch_bounds__(64) void bn_chan_finish_kernel(float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            float* __restrict__ ab, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, const float* __restrict__ part,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int batch, int C, int S, float eps, float momentum) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < S; ++k) {
            const float* p = part + 3 * (((int64_t)b * C + c) * S + k);
            const float nb = p[0], nt = n + nb, delta = p[1] - mean;
            mean += delta * (nb / nt);
            m2 += p[2] + delta * delta * (n * nb / nt);
            n = nt;
        }
    const float rstd = rsqrtf(m2 / n + eps);
    mean_out[c] = mean; rstd_out[c] = rstd;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : m2 / n);
    }
    const float a = rstd * gamma[c], sh = beta[c] - mean * a;
    for (int b = 0; b < batch; ++b) { ab[2 * ((int64_t)b * C + c)] = a; ab[2 * ((int64_t)b * C + c) + 1] = sh; }
}

// dx = rstd*gamma * (g' - sum(g')/N - xhat * sum(g'*xhat)/N),  d(gamma) = sum(g'*xhat),  d(beta) = sum(g')
__global__ __launch_bounds__(64) void bn_bwd_chan_kernel(float* __restrict__ coef, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, const float* __restrict__ part,
                                                         const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                         int batch, int C, int hw, int S) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < S; ++k) {
            const float* p = part + 2 * (((int64_t)b * C + c) * S + k);
            s1 += p[0]; s2 += p[1];
        }
    dbeta[c] = s1; dgamma[c] = s2;
    const float n = (float)batch * (float)hw, k1 = rstd_in[c] * gamma[c];
    for (int b = 0; b < batch; ++b) {
        float* k = coef + 3 * ((int64_t)b * C + c);
        k[0] = k1; k[1] = k1 * s1 / n; k[2] = k1 * s2 / n;
    }
}

// ---- single-pass kernels for groups that fit the registers of one workgroup (16-bit x, hw % 8 == 0, <= 32 768 elements per
// group: every norm of the trunk's blocks 2 and 3 at 512^2 input -- tensors of 2-17 MB whose four launches per step were
// pure latency).  One workgroup per (sample, group) = one contiguous chunk of cpg * hw elements; wave w owns the vectors
// (8 elements) [w * VW, (w + 1) * VW), 64 consecutive ones per iteration (NIT iterations, all in registers), so a wave's
// vectors of one iteration lie in ONE channel (hw / 8 is a multiple of 64).  Statistics are the exact two-pass mean /
// variance of the group (fp32).  Reductions across waves go through LDS and are summed by every thread in wave order:
// deterministic.
struct GnGroupArgs {
    int hw, cpg, groups, relu; float eps;
};

template <typename TI>
__device__ __forceinline__ void gn_load8(const TI* p, float* v) {   // 8 consecutive 16-bit elements: one 16-byte load
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    TI t[8];
    __builtin_memcpy(t, &q, 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = sis_ld(t, e);
}
template <typename TI>
__device__ __forceinline__ void gn_store8(TI* p, const float* v) {
    TI t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sis_st(t, e, v[e]);
    uint4 q;
    __builtin_memcpy(&q, t, 16);
    *reinterpret_cast<uint4*>(p) = q;
}
__device__ __forceinline__ void gn_load8(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void gn_store8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// sum over the workgroup (<= 16 waves), the same value in every thread
__device__ __forceinline__ float gn_group_sum(float v, float* red, int nw) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[w];
    return s;
}

template <typename TI, typename TO, int NIT>
__global__ __launch_bounds__(1024) void gn_group_fwd_kernel(TO* __restrict__ y, TI* __restrict__ y_lp, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, const TI* __restrict__ x,
                                                            const float* __restrict__ res, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, GnGroupArgs a, unsigned char* __restrict__ bits) {
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int row = blockIdx.x;
    const int64_t base = (int64_t)row * a.cpg * a.hw;
    const int v0 = wave * (NIT * 64) + lane;       // this lane's vector of iteration 0 (+ 64 per iteration)
    float xv[NIT][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        gn_load8(x + base + 8 * (int64_t)(v0 + it * 64), xv[it]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xv[it][e];
    }
    const float n = (float)a.cpg * (float)a.hw;
    const float mean = gn_group_sum(s, red, nw) / n;
    float m2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xv[it][e] - mean; m2 += d * d; }
    const float rstd = rsqrtf(gn_group_sum(m2, red, nw) / n + a.eps);
    if (threadIdx.x == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const int c0 = (row % a.groups) * a.cpg;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int64_t off = 8 * (int64_t)(v0 + it * 64);
        const int c = c0 + (int)(off / a.hw);
        const float ga = rstd * gamma[c], gb = beta[c] - mean * ga;
        float rv[8], out[8];
        if (res) gn_load8(res + base + off, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = xv[it][e] * ga + gb;
            if (res) v += rv[e];
            if (a.relu) v = fmaxf(v, 0.f);
            out[e] = v;
        }
        gn_store8(y + base + off, out);
        if (y_lp) gn_store8(y_lp + base + off, out);
        if (bits) {   // this lane's 8 elements are one byte of the gate bits
            unsigned byte = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) byte |= (out[e] > 0.f ? 1u : 0u) << e;
            bits[(base + off) >> 3] = (unsigned char)byte;
        }
    }
}

// backward: g' = masked gradient, dx = k1 g' - k2 - k3 xhat (see gn_bwd_row); plane sums for d gamma / d beta go to
// psum[plane] (published: the last workgroup of the launch adds them over the batch in sample order).
template <typename TI, typename TG, int NIT, bool HAS_LP, bool HAS_MASK>
__global__ __launch_bounds__(1024) void gn_group_bwd_kernel(TI* __restrict__ dx, float* __restrict__ dres, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ psum,
                                                            int* __restrict__ counter, const TG* __restrict__ g,
                                                            const TI* __restrict__ g_lp, const TI* __restrict__ x,
                                                            const float* __restrict__ ymask, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, GnGroupArgs a, int batch,
                                                            const unsigned char* __restrict__ bits) {
    __shared__ float part[16 * NIT][2];   // per (wave, iteration): sum g', sum g' xhat -- all of one channel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int row = blockIdx.x;
    const int64_t base = (int64_t)row * a.cpg * a.hw;
    const int v0 = wave * (NIT * 64) + lane;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const int c0 = (row % a.groups) * a.cpg;
    float xh[NIT][8], gi[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int64_t off = 8 * (int64_t)(v0 + it * 64);
        const int c = c0 + (int)(off / a.hw);
        float xv[8], g2[8], mv[8];
        gn_load8(x + base + off, xv);
        gn_load8(g + base + off, gi[it]);
        if (HAS_LP) gn_load8(g_lp + base + off, g2);
        if (HAS_MASK) {
            if (bits) {
                const unsigned byte = bits[(base + off) >> 3];
#pragma unroll
                for (int e = 0; e < 8; ++e) mv[e] = (float)((byte >> e) & 1u);
            } else {
                gn_load8(ymask + base + off, mv);
            }
        }
        const float gm = gamma[c], bt = beta[c];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float h = (xv[e] - mean) * rstd;
            float ge = gi[it][e] + (HAS_LP ? g2[e] : 0.f);
            if (a.relu && (HAS_MASK ? mv[e] <= 0.f : h * gm + bt <= 0.f)) ge = 0.f;
            xh[it][e] = h; gi[it][e] = ge;
            sg += ge; sgx += ge * h;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
        if (lane == 0) { part[wave * NIT + it][0] = sg; part[wave * NIT + it][1] = sgx; }
    }
    __syncthreads();
    // entries (wave, it) in vector order; entry j covers vectors [64 j, 64 j + 64): channel 512 j / hw of the group
    const int entries = nw * NIT, per_chan = a.hw / 512;
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < entries; ++j) {
        const float gm = gamma[c0 + j / per_chan];
        s1 += gm * part[j][0]; s2 += gm * part[j][1];
    }
    if ((int)threadIdx.x < a.cpg) {   // plane sums of channel threadIdx.x of the group
        float pa = 0.f, pb = 0.f;
        for (int j = threadIdx.x * per_chan; j < (threadIdx.x + 1) * per_chan; ++j) { pa += part[j][0]; pb += part[j][1]; }
        float* o = psum + 2 * ((int64_t)row * a.cpg + threadIdx.x);
        xwg_publish(o, pa); xwg_publish(o + 1, pb);
    }
    const float n = (float)a.cpg * (float)a.hw;
    const float k2 = rstd * s1 / n, k3 = rstd * s2 / n;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int64_t off = 8 * (int64_t)(v0 + it * 64);
        const float k1 = rstd * gamma[c0 + (int)(off / a.hw)];
        float out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = k1 * gi[it][e] - k2 - k3 * xh[it][e];
        gn_store8(dx + base + off, out);
        if (HAS_MASK && dres) gn_store8(dres + base + off, gi[it]);
    }
    // d gamma / d beta: the last workgroup of the launch adds the plane sums over the batch (counter == nullptr: launches of
    // thousands of workgroups leave that to gn_param_reduce_kernel -- see gn_bwd_run)
    if (counter && xwg_complete<true>(counter, gridDim.x)) {
        const int C = a.groups * a.cpg;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float da = 0.f, db = 0.f;
            for (int b = 0; b < batch; ++b) { db += xwg_peek(psum + 2 * ((int64_t)b * C + c)); da += xwg_peek(psum + 2 * ((int64_t)b * C + c) + 1); }
            dgamma[c] = da; dbeta[c] = db;
        }
    }
}

// launch geometry of the single-pass kernels: vectors per group V = cpg * hw / 8, NIT iterations of 64 vectors per wave
inline bool gn_group_plan(int cpg, int hw, int* nit, int* nw) {
    static const bool enabled = [] { const char* e = getenv("SIS_GN_SINGLE_PASS"); return !(e && e[0] == '0'); }();   // 0: A/B runs
    if (!enabled || hw % 512) return false;           // a (wave, iteration) = 512 elements lies in one channel
    const int64_t E = (int64_t)cpg * hw;
    if (E > 32768) return false;
    const int V = (int)(E / 8);
    *nit = V <= 1024 ? 1 : V <= 2048 ? 2 : 4;
    *nw = V / (64 * *nit);
    return *nw >= 1 && *nw <= 16 && *nw * 64 * *nit == V;
}

constexpr int GN_SLICE = 16384;  // elements of a plane per statistics workgroup
inline int gn_slices(int hw) { return (hw + GN_SLICE - 1) / GN_SLICE; }
inline int gn_slice_len(int hw) { const int S = gn_slices(hw); return (((hw + S - 1) / S) + 3) & ~3; }

inline unsigned gn_grid(int64_t total, int vec) {
    const int64_t blocks = (total / vec + 255) / 256;
    return (unsigned)(blocks < 16384 ? (blocks > 0 ? blocks : 1) : 16384);
}

inline bool gn_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// element-wise kernels: 4 per lane when planes are vector-aligned (hw % 4 == 0) or, failing that, when the flat tensor is
// (FLAT: per-element plane lookup); plane kernels: 4 per lane with head / tail peeling whenever the base pointers are aligned
#define GN_DISPATCH_APPLY(KERNEL, T1, T2, ok4, ...)                                                                            \
    do {                                                                                                                       \
        if ((ok4) && hw % 4 == 0)                                                                                              \
            hipLaunchKernelGGL((KERNEL<T1, T2, 4, false>), dim3(gn_grid(total, 4)), dim3(256), 0, st, __VA_ARGS__);            \
        else if ((ok4) && total % 4 == 0 && hw >= 4)                                                                           \
            hipLaunchKernelGGL((KERNEL<T1, T2, 4, true>), dim3(gn_grid(total, 4)), dim3(256), 0, st, __VA_ARGS__);             \
        else                                                                                                                   \
            hipLaunchKernelGGL((KERNEL<T1, T2, 1, false>), dim3(gn_grid(total, 1)), dim3(256), 0, st, __VA_ARGS__);            \
    } while (0)

template <typename TI>
void gn_launch_stats(float* part, const void* x, int64_t planes, int hw, hipStream_t st, GnFinish fin = GnFinish{}) {
    const int S = gn_slices(hw), sl = gn_slice_len(hw);
    if (gn_aligned(x))
        hipLaunchKernelGGL((gn_plane_stats_kernel<TI, 4>), dim3((unsigned)planes, S), dim3(256), 0, st, part, (const TI*)x, hw, sl, fin);
    else
        hipLaunchKernelGGL((gn_plane_stats_kernel<TI, 1>), dim3((unsigned)planes, S), dim3(256), 0, st, part, (const TI*)x, hw, sl, fin);
}

template <typename TI, typename TG>
void gn_launch_bwd_plane(float* part, const void* g, const void* g_lp, const void* x, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, const float* ymask, int64_t planes, int C, int cpg, int hw, int relu, hipStream_t st,
                         int* counters = nullptr, float* coef = nullptr, float* psum = nullptr, const unsigned char* bits = nullptr,
                         int bn_batch = 0, float* bn_dgamma = nullptr, float* bn_dbeta = nullptr) {
    const int S = gn_slices(hw), sl = gn_slice_len(hw);
    if (gn_aligned(x) && gn_aligned(g) && gn_aligned(g_lp) && gn_aligned(ymask))
        hipLaunchKernelGGL((gn_bwd_plane_kernel<TI, TG, 4>), dim3((unsigned)planes, S), dim3(256), 0, st, part, (const TG*)g,
                           (const TI*)g_lp, (const TI*)x, mean, rstd, gamma, beta, ymask, C, cpg, hw, sl, relu, counters, coef, psum, bits, bn_batch, bn_dgamma, bn_dbeta);
    else
        hipLaunchKernelGGL((gn_bwd_plane_kernel<TI, TG, 1>), dim3((unsigned)planes, S), dim3(256), 0, st, part, (const TG*)g,
                           (const TI*)g_lp, (const TI*)x, mean, rstd, gamma, beta, ymask, C, cpg, hw, sl, relu, counters, coef, psum, bits, bn_batch, bn_dgamma, bn_dbeta);
}

template <typename TI, typename TO>
void gn_fwd_run(void* y, void* y_lp, float* mean, float* rstd, float* ws, const void* x, const float* res, const float* gamma,
                const float* beta, int batch, int C, int hw, int groups, float eps, int relu, int* counters, unsigned char* bits,
                hipStream_t st) {
    const int cpg = C / groups, rows = batch * groups;
    const int64_t planes = (int64_t)batch * C, total = planes * hw;
    const int S = gn_slices(hw);
    float* ab = ws;                // [planes][2]
    float* part = ws + 5 * planes; // [planes][S][3]
    int nit, nw;
    if constexpr (sizeof(TI) == 2) {
        if (counters && gn_group_plan(cpg, hw, &nit, &nw) && gn_aligned(x) && gn_aligned(y) && gn_aligned(y_lp) && gn_aligned(res)) {
            const GnGroupArgs a{hw, cpg, groups, relu, eps};
#define GN_GROUP_FWD(N) hipLaunchKernelGGL((gn_group_fwd_kernel<TI, TO, N>), dim3(rows), dim3(64 * nw), 0, st, (TO*)y, (TI*)y_lp, mean, \
                                           rstd, (const TI*)x, res, gamma, beta, a, bits)
            if (nit == 1) GN_GROUP_FWD(1); else if (nit == 2) GN_GROUP_FWD(2); else GN_GROUP_FWD(4);
#undef GN_GROUP_FWD
            return;
        }
    }
    if (counters) {
        gn_launch_stats<TI>(part, x, planes, hw, st, GnFinish{counters, mean, rstd, ab, gamma, beta, groups, cpg, eps});
    } else {
        gn_launch_stats<TI>(part, x, planes, hw, st);
        hipLaunchKernelGGL(gn_row_finish_kernel, dim3(sis_cdiv(rows, 64)), dim3(64), 0, st, mean, rstd, ab, part, gamma, beta, rows,
                           groups, cpg, S, eps);
    }
    GN_DISPATCH_APPLY(gn_apply_kernel, TI, TO, gn_aligned(x) && gn_aligned(y) && gn_aligned(y_lp) && gn_aligned(res), (TO*)y,
                      (TI*)y_lp, (const TI*)x, ab, res, hw, total, relu, bits);
}

template <typename TI, typename TG>
void gn_bwd_run(void* dx, float* dres, float* dgamma, float* dbeta, float* ws, const void* g, const void* g_lp, const void* x,
                const float* ymask, const float* mean, const float* rstd, const float* gamma, const float* beta, int batch, int C,
                int hw, int groups, int relu, int* counters, const unsigned char* bits, hipStream_t st) {
    const bool gated = ymask || bits;   // the ReLU gate of the residual form: the saved output or the forward's bit per element
    const int cpg = C / groups, rows = batch * groups;
    const int64_t planes = (int64_t)batch * C, total = planes * hw;
    const int S = gn_slices(hw);
    float* psum = ws;              // [planes][2]
    float* coef = ws + 2 * planes; // [planes][3]
    float* part = ws + 5 * planes; // [planes][S][2]
    float* no_param = nullptr;
    int nit, nw;
    if constexpr (sizeof(TI) == 2) {
        if (counters && gn_group_plan(cpg, hw, &nit, &nw) && gn_aligned(dx) && gn_aligned(g) && gn_aligned(g_lp) && gn_aligned(x) &&
            gn_aligned(ymask) && gn_aligned(dres) && (dres == nullptr || gated)) {
            const GnGroupArgs a{hw, cpg, groups, relu, 0.f};
            // The in-launch hand-over is ONE device-scope counter that every workgroup increments (after draining its stores):
            // ~12 ns per arrival (MI355X_MICROARCH.md, "fanin"), i.e. 98 us for the 8 192 one-plane groups of block3's projection
            // norm (measured 117 us for 67 MB; 59 us for block2's 4 096) against ~3 us for a launch of its own.  Beyond 1 024
            // workgroups the plane sums are added by gn_param_reduce_kernel instead.
            int* const done = rows <= 1024 ? counters : nullptr;
#define GN_GROUP_BWD(N, LP, MK) hipLaunchKernelGGL((gn_group_bwd_kernel<TI, TG, N, LP, MK>), dim3(rows), dim3(64 * nw), 0, st, (TI*)dx, dres, \
                                                   dgamma, dbeta, psum, done, (const TG*)g, (const TI*)g_lp, (const TI*)x, ymask, mean,  \
                                                   rstd, gamma, beta, a, batch, bits)
#define GN_GROUP_BWD_N(LP, MK) do { if (nit == 1) GN_GROUP_BWD(1, LP, MK); else if (nit == 2) GN_GROUP_BWD(2, LP, MK); else GN_GROUP_BWD(4, LP, MK); } while (0)
            if (g_lp && gated) GN_GROUP_BWD_N(true, true);
            else if (g_lp) GN_GROUP_BWD_N(true, false);
            else if (gated) GN_GROUP_BWD_N(false, true);
            else GN_GROUP_BWD_N(false, false);
#undef GN_GROUP_BWD_N
#undef GN_GROUP_BWD
            if (!done) hipLaunchKernelGGL(gn_param_reduce_kernel, dim3(sis_cdiv(C, 256)), dim3(256), 0, st, dgamma, dbeta, psum, batch, C);
            return;
        }
    }
    if (counters) {
        gn_launch_bwd_plane<TI, TG>(part, g, g_lp, x, mean, rstd, gamma, beta, ymask, planes, C, cpg, hw, relu, st, counters, coef, psum, bits);
    } else {
        gn_launch_bwd_plane<TI, TG>(part, g, g_lp, x, mean, rstd, gamma, beta, ymask, planes, C, cpg, hw, relu, st, nullptr, nullptr, nullptr, bits);
        hipLaunchKernelGGL(gn_bwd_row_kernel, dim3(sis_cdiv(rows, 64)), dim3(64), 0, st, coef, psum, part, rstd, gamma, rows, groups,
                           cpg, hw, S);
    }
    GN_DISPATCH_APPLY(gn_bwd_apply_kernel, TI, TG,
                      gn_aligned(dx) && gn_aligned(g) && gn_aligned(g_lp) && gn_aligned(x) && gn_aligned(ymask) && gn_aligned(dres),
                      (TI*)dx, (const TG*)g, (const TI*)g_lp, (const TI*)x, coef, mean, rstd, gamma, beta, ymask, dres, C, cpg, hw, total, relu,
                      counters ? dgamma : no_param, dbeta, (const float*)psum, batch, bits);
    if (!counters) hipLaunchKernelGGL(gn_param_reduce_kernel, dim3(sis_cdiv(C, 256)), dim3(256), 0, st, dgamma, dbeta, psum, batch, C);
}

template <typename TI, typename TO>
void bn_fwd_run(void* y, float* mean, float* rstd, float* rm, float* rv, float* ws, const void* x, const float* gamma,
                const float* beta, int batch, int C, int hw, float eps, float momentum, int relu, int* counters, hipStream_t st) {
    const float* res = nullptr;
    const int64_t planes = (int64_t)batch * C, total = planes * hw;
    const int S = gn_slices(hw);
    float* ab = ws;
    float* part = ws + 5 * planes;
    if (counters) {   // the workgroup that completes a channel merges it (no launch of its own for C threads of work)
        GnFinish fin{counters, mean, rstd, ab, gamma, beta, C, 0, eps};
        fin.bn_batch = batch; fin.running_mean = rm; fin.running_var = rv; fin.momentum = momentum;
        gn_launch_stats<TI>(part, x, planes, hw, st, fin);
    } else {
        gn_launch_stats<TI>(part, x, planes, hw, st);
        hipLaunchKernelGGL(bn_chan_finish_kernel, dim3(sis_cdiv(C, 64)), dim3(64), 0, st, mean, rstd, ab, rm, rv, part, gamma, beta,
                           batch, C, S, eps, momentum);
    }
    GN_DISPATCH_APPLY(gn_apply_kernel, TI, TO, gn_aligned(x) && gn_aligned(y), (TO*)y, (TI*)nullptr, (const TI*)x, ab, res, hw, total,
                      relu, (unsigned char*)nullptr);
}

template <typename TI, typename TG>
void bn_bwd_run(void* dx, float* dgamma, float* dbeta, float* ws, const void* g, const void* x, const float* mean,
                const float* rstd, const float* gamma, const float* beta, int batch, int C, int hw, int relu, int* counters,
                hipStream_t st) {
    const int64_t planes = (int64_t)batch * C, total = planes * hw;
    const int S = gn_slices(hw);
    float* coef = ws + 2 * planes;
    float* part = ws + 5 * planes;
    const float* none = nullptr;
    float* no_dres = nullptr;
    if (counters) {
        gn_launch_bwd_plane<TI, TG>(part, g, nullptr, x, mean, rstd, gamma, beta, none, planes, C, 0, hw, relu, st, counters, coef, nullptr,
                                    nullptr, batch, dgamma, dbeta);
    } else {
        gn_launch_bwd_plane<TI, TG>(part, g, nullptr, x, mean, rstd, gamma, beta, none, planes, C, 0, hw, relu, st);
        hipLaunchKernelGGL(bn_bwd_chan_kernel, dim3(sis_cdiv(C, 64)), dim3(64), 0, st, coef, dgamma, dbeta, part, rstd, gamma, batch,
                           C, hw, S);
    }
    GN_DISPATCH_APPLY(gn_bwd_apply_kernel, TI, TG, gn_aligned(dx) && gn_aligned(g) && gn_aligned(x), (TI*)dx, (const TG*)g,
                      (const TI*)nullptr, (const TI*)x, coef, mean, rstd, gamma, beta, none, no_dres, C, 0, hw, total, relu, no_dres, no_dres,
                      none, batch, (const unsigned char*)nullptr);
}

}  // namespace

extern "C" int64_t sis_group_norm_workspace_floats(int batch, int channels, int hw) {
    return (int64_t)batch * channels * (5 + 3 * gn_slices(hw));
}

extern "C" int sis_group_norm_fwd(void* y, void* y_lp, float* mean, float* rstd, float* workspace, const void* x,
                                  const float* residual,
                                  const float* gamma, const float* beta, int x_dtype, int y_dtype, int batch, int channels,
                                  int hw, int groups, float eps, int relu, int* counters, void* relu_bits, void* stream) {
    if (batch == 0) return 0;
    SIS_REQUIRE(y && mean && rstd && workspace && x && gamma && beta, "sis_group_norm_fwd: null pointer");
    SIS_REQUIRE(!relu_bits || (relu && (reinterpret_cast<uintptr_t>(relu_bits) & 7) == 0), "sis_group_norm_fwd: the gate bits go with a ReLU (8-byte aligned buffer)");
    unsigned char* bits = (unsigned char*)relu_bits;
    SIS_REQUIRE(batch > 0 && channels > 0 && hw > 0 && groups > 0 && channels % groups == 0,
                "sis_group_norm_fwd: bad sizes (C %d, groups %d)", channels, groups);
    SIS_REQUIRE(y_dtype == x_dtype || y_dtype == SIS_F32, "sis_group_norm_fwd: output dtype must be the input's or f32");
    SIS_REQUIRE(!residual || y_dtype == SIS_F32, "sis_group_norm_fwd: a residual needs a float32 output");
    SIS_REQUIRE(!y_lp || (y_dtype == SIS_F32 && x_dtype != SIS_F32),
                "sis_group_norm_fwd: the 16-bit copy goes with a float32 output of a 16-bit input");
    hipStream_t st = (hipStream_t)stream;
#define GN_FWD(TI)                                                                                                       \
    if (y_dtype == SIS_F32) gn_fwd_run<TI, float>(y, y_lp, mean, rstd, workspace, x, residual, gamma, beta, batch, channels, hw, groups, eps, relu, counters, bits, st); \
    else gn_fwd_run<TI, TI>(y, nullptr, mean, rstd, workspace, x, nullptr, gamma, beta, batch, channels, hw, groups, eps, relu, counters, bits, st);
    switch (x_dtype) {
        case SIS_F32: GN_FWD(float) break;
        case SIS_F16: GN_FWD(__half) break;
        case SIS_BF16: GN_FWD(__hip_bfloat16) break;
        default: return sis_fail("sis_group_norm_fwd: dtype code %d not supported (f32, f16, bf16)", x_dtype);
    }
#undef GN_FWD
    SIS_CHECK_LAUNCH("gn_fwd");
    return 0;
}

extern "C" int sis_group_norm_bwd(void* dx, float* dresidual, float* dgamma, float* dbeta, float* workspace, const void* grad_y,
                                  const void* grad_y_lp, const void* x, const float* y_mask, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, int x_dtype, int g_dtype, int batch, int channels, int hw, int groups,
                                  int relu, int* counters, const void* relu_bits, void* stream) {
    if (batch == 0) return 0;
    SIS_REQUIRE(dx && dgamma && dbeta && workspace && grad_y && x && mean && rstd && gamma && beta,
                "sis_group_norm_bwd: null pointer");
    SIS_REQUIRE(!(relu_bits && y_mask), "sis_group_norm_bwd: the ReLU gate comes from y_mask OR from relu_bits");
    const unsigned char* bits = (const unsigned char*)relu_bits;
    SIS_REQUIRE(batch > 0 && channels > 0 && hw > 0 && groups > 0 && channels % groups == 0,
                "sis_group_norm_bwd: bad sizes (C %d, groups %d)", channels, groups);
    SIS_REQUIRE(g_dtype == x_dtype || g_dtype == SIS_F32, "sis_group_norm_bwd: gradient dtype must be the input's or f32");
    SIS_REQUIRE(!grad_y_lp || (g_dtype == SIS_F32 && x_dtype != SIS_F32),
                "sis_group_norm_bwd: the 16-bit gradient goes with a float32 gradient of a 16-bit input");
    hipStream_t st = (hipStream_t)stream;
#define GN_BWD(TI)                                                                                                        \
    if (g_dtype == SIS_F32) gn_bwd_run<TI, float>(dx, dresidual, dgamma, dbeta, workspace, grad_y, grad_y_lp, x, y_mask, mean, rstd, gamma, beta, batch, channels, hw, groups, relu, counters, bits, st); \
    else gn_bwd_run<TI, TI>(dx, dresidual, dgamma, dbeta, workspace, grad_y, nullptr, x, y_mask, mean, rstd, gamma, beta, batch, channels, hw, groups, relu, counters, bits, st);
    switch (x_dtype) {
        case SIS_F32: GN_BWD(float) break;
        case SIS_F16: GN_BWD(__half) break;
        case SIS_BF16: GN_BWD(__hip_bfloat16) break;
        default: return sis_fail("sis_group_norm_bwd: dtype code %d not supported (f32, f16, bf16)", x_dtype);
    }
#undef GN_BWD
    SIS_CHECK_LAUNCH("gn_bwd");
    return 0;
}

// ---- nn.BatchNorm2d in training mode (+ ReLU) with 16-bit or fp32 tensors: the TransUNet decoder's Conv2dReLU blocks
// (vit_seg_modeling.py:265-287).  Same three-launch structure, statistics per channel over (batch, H, W); running
// statistics updated with `momentum` (unbiased variance), may be NULL.  Workspace as for group norm.
extern "C" int sis_batch_norm_fwd(void* y, float* mean, float* rstd, float* running_mean, float* running_var, float* workspace,
                                  const void* x, const float* gamma, const float* beta, int x_dtype, int y_dtype, int batch,
                                  int channels, int hw, float eps, float momentum, int relu, int* counters, void* stream) {
    if (batch == 0) return 0;
    SIS_REQUIRE(y && mean && rstd && workspace && x && gamma && beta, "sis_batch_norm_fwd: null pointer");
    SIS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sis_batch_norm_fwd: running statistics come in pairs");
    SIS_REQUIRE(batch > 0 && channels > 0 && hw > 0, "sis_batch_norm_fwd: non-positive size");
    SIS_REQUIRE(y_dtype == x_dtype || y_dtype == SIS_F32, "sis_batch_norm_fwd: output dtype must be the input's or f32");
    hipStream_t st = (hipStream_t)stream;
#define BN_FWD(TI)                                                                                                        \
    if (y_dtype == SIS_F32) bn_fwd_run<TI, float>(y, mean, rstd, running_mean, running_var, workspace, x, gamma, beta, batch, channels, hw, eps, momentum, relu, counters, st); \
    else bn_fwd_run<TI, TI>(y, mean, rstd, running_mean, running_var, workspace, x, gamma, beta, batch, channels, hw, eps, momentum, relu, counters, st);
    switch (x_dtype) {
        case SIS_F32: BN_FWD(float) break;
        case SIS_F16: BN_FWD(__half) break;
        case SIS_BF16: BN_FWD(__hip_bfloat16) break;
        default: return sis_fail("sis_batch_norm_fwd: dtype code %d not supported (f32, f16, bf16)", x_dtype);
    }
#undef BN_FWD
    SIS_CHECK_LAUNCH("bn_fwd");
    return 0;
}

extern "C" int sis_batch_norm_bwd(void* dx, float* dgamma, float* dbeta, float* workspace, const void* grad_y, const void* x,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta, int x_dtype,
                                  int g_dtype, int batch, int channels, int hw, int relu, int* counters, void* stream) {
    if (batch == 0) return 0;
    SIS_REQUIRE(dx && dgamma && dbeta && workspace && grad_y && x && mean && rstd && gamma && beta,
                "sis_batch_norm_bwd: null pointer");
    SIS_REQUIRE(batch > 0 && channels > 0 && hw > 0, "sis_batch_norm_bwd: non-positive size");
    SIS_REQUIRE(g_dtype == x_dtype || g_dtype == SIS_F32, "sis_batch_norm_bwd: gradient dtype must be the input's or f32");
    hipStream_t st = (hipStream_t)stream;
#define BN_BWD(TI)                                                                                                        \
    if (g_dtype == SIS_F32) bn_bwd_run<TI, float>(dx, dgamma, dbeta, workspace, grad_y, x, mean, rstd, gamma, beta, batch, channels, hw, relu, counters, st); \
    else bn_bwd_run<TI, TI>(dx, dgamma, dbeta, workspace, grad_y, x, mean, rstd, gamma, beta, batch, channels, hw, relu, counters, st);
    switch (x_dtype) {
        case SIS_F32: BN_BWD(float) break;
        case SIS_F16: BN_BWD(__half) break;
        case SIS_BF16: BN_BWD(__hip_bfloat16) break;
        default: return sis_fail("sis_batch_norm_bwd: dtype code %d not supported (f32, f16, bf16)", x_dtype);
    }
#undef BN_BWD
    SIS_CHECK_LAUNCH("bn_bwd");
    return 0;
}

/* bytes of the one-bit-per-element ReLU gate of sis_group_norm_fwd / _bwd (relu_bits), padded to whole 8-byte words */
extern "C" int64_t sis_group_norm_gate_bytes(int batch, int channels, int hw) {
    return (((int64_t)batch * channels * hw + 63) / 64) * 8;
}
