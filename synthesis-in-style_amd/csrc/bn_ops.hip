// Batch normalisation fused with its neighbours, fp32 NCHW: the HBM-bound half of the EMANet training step.
//
// Reference code path: every convolution of networks/ema_net/network.py is followed by
// `norm_layer` (= F.batch_norm with batch statistics, momentum 3e-4, eps 1e-5: bn_lib/nn/modules/batchnorm.py:51-56),
// usually a ReLU, and in the bottlenecks a residual add + ReLU (network.py:37-56) -- on the reference each of these is
// its own full-tensor pass (and its own backward pass).  Here:
//
//   forward   bn_stats_kernel     one read of x   -> per-(channel, slice) count/mean/M2 partials (two sweeps over a
//                                                    16K-element slice that the second time comes from L2), merged in
//                                                    fixed order with Chan's formula: no E[x^2]-E[x]^2 cancellation,
//                                                    bitwise reproducible; also updates running_mean / running_var
//             bn_act_fwd_kernel   one read of x (+ residual), one write of y = relu(gamma*xhat + beta + residual)
//   backward  bn_bwd_reduce       one read of dy, y, x -> sum(dy') and sum(dy'*xhat) per channel (dy' = dy * [y > 0])
//             bn_bwd_apply        one read of dy, y, x, one write of dx (and of the residual gradient dy')
//
// Algorithmic HBM bytes per element: forward 4*(2 reads + 1 write) (+4 with residual), backward 4*(6 reads + 1 write)
// (+4 for the residual gradient); unfused ATen does forward BN (2r+1w) + ReLU (1r+1w) + add (2r+1w) + ReLU (1r+1w).
// ReLU gate as a bit mask: the backward needs y only for its sign.  The forward apply pass can leave one bit per element
// (four 64-bit ballots per wave and float4 column: word (i / 64) * 4 + e, bit i % 64 for component e of float4 i), and both
// backward passes then read 1/8 byte instead of 4 bytes per element for the gate: backward 4*(4 reads + 1 write) + 2/8.
#include <cstdlib>
#include "sis_common.h"

namespace {

constexpr int BN_SLICE = 16384;  // elements per (channel, slice) partial

struct BnGeom { int B, C, HW, slices_per_plane_group, S; int64_t n; };

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// A channel's data = B planes of HW floats.  Slice s of channel c covers elements [s*BN_SLICE, (s+1)*BN_SLICE) of
// the channel's B*HW logical elements (element e lives at plane e / HW, offset e % HW).  Lanes walk float4s; one
// division at the start, then plane / offset advance incrementally (HW % 4 == 0: a float4 never straddles planes).
struct ChanWalk {
    int64_t e, hi;
    int b, r, C, HW, c;
    __device__ __forceinline__ ChanWalk(int64_t lo, int64_t hi_, int c_, int C_, int HW_) : hi(hi_), C(C_), HW(HW_), c(c_) {
        e = lo + (int64_t)threadIdx.x * 4;
        b = (int)(e / HW);
        r = (int)(e - (int64_t)b * HW);
    }
    __device__ __forceinline__ bool valid() const { return e < hi; }
    __device__ __forceinline__ int64_t addr() const { return ((int64_t)b * C + c) * HW + r; }
    __device__ __forceinline__ void next() {
        e += 1024; r += 1024;
        while (r >= HW) { r -= HW; ++b; }
    }
};

// One workgroup per (channel, slice): a thread's 16 float4s of the slice are ALL requested before the first is used (one
// memory latency per workgroup instead of 16 dependent ones: these workgroups are short, the pass is latency-bound), then the
// mean and the centred second moment come out of the registers -- the slice is read once.
__global__ __launch_bounds__(256) void bn_stats_kernel(float* __restrict__ partial, const float* __restrict__ x, int C,
                                                       int HW, int64_t n, int S) {
    __shared__ float red[4];
    constexpr int PER = BN_SLICE / 1024;   // float4s per thread and slice
    const int c = blockIdx.x / S, s = blockIdx.x % S;
    const int64_t lo = (int64_t)s * BN_SLICE, hi = min(n, lo + BN_SLICE);
    float4 v[PER];
    {
        ChanWalk w(lo, hi, c, C, HW);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            v[k] = w.valid() ? *reinterpret_cast<const float4*>(x + w.addr()) : make_float4(0.f, 0.f, 0.f, 0.f);
            w.next();
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float cnt = (float)(hi - lo);
    const float mean = block_sum(sum, red) / cnt;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (lo + (int64_t)threadIdx.x * 4 + (int64_t)k * 1024 < hi) {
            const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
            m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    m2 = block_sum(m2, red);
    if (threadIdx.x == 0) {
        float* o = partial + ((int64_t)c * S + s) * 3;
        o[0] = cnt; o[1] = mean; o[2] = m2;
    }
}

__global__ __launch_bounds__(64) void bn_stats_finish_kernel(float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var,
                                                             const float* __restrict__ partial, int C, int S, float eps,
                                                             float momentum) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int s = 0; s < S; ++s) {  // Chan et al. pairwise merge, fixed order
        const float* p = partial + ((int64_t)c * S + s) * 3;
        const float nb = p[0], mb = p[1], m2b = p[2];
        const float nt = n + nb, delta = mb - mean;
        mean += delta * (nb / nt);
        m2 += m2b + delta * delta * (n * nb / nt);
        n = nt;
    }
    const float var = m2 / n;
    mean_out[c] = mean;
    invstd_out[c] = rsqrtf(var + eps);
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
    }
}

__device__ __forceinline__ bool gate_bit(const unsigned long long* __restrict__ mask, int64_t i4, int e) {
    return (mask[(i4 >> 6) * 4 + e] >> (i4 & 63)) & 1ull;
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(float4* __restrict__ y, const float4* __restrict__ x,
                                                         const float4* __restrict__ res, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int C, int HW4, int64_t total4,
                                                         unsigned long long* __restrict__ mask) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        const int c = (int)((i / HW4) % C);
        const float a = (gamma ? gamma[c] : 1.f) * invstd[c];
        const float b = (beta ? beta[c] : 0.f) - mean[c] * a;
        float4 v = x[i];
        v.x = v.x * a + b; v.y = v.y * a + b; v.z = v.z * a + b; v.w = v.w * a + b;
        if (RES) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        y[i] = v;
        if (RELU && mask) {   // (i - lane is a multiple of 64: the wave's 64 float4s fill one group of four words)
            const unsigned long long b0 = __ballot(v.x > 0.f), b1 = __ballot(v.y > 0.f), b2 = __ballot(v.z > 0.f), b3 = __ballot(v.w > 0.f);
            if ((threadIdx.x & 63) == 0) {
                unsigned long long* m = mask + (i >> 6) * 4;
                m[0] = b0; m[1] = b1; m[2] = b2; m[3] = b3;
            }
        }
    }
}

template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(float* __restrict__ partial, const float* __restrict__ dy,
                                                            const float* __restrict__ y, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            int C, int HW, int64_t n, int S,
                                                            const unsigned long long* __restrict__ mask) {
    __shared__ float red[4];
    const int c = blockIdx.x / S, s = blockIdx.x % S;
    const int64_t lo = (int64_t)s * BN_SLICE, hi = min(n, lo + BN_SLICE);
    const float mu = mean[c], is = invstd[c];
    float s1 = 0.f, s2 = 0.f;
    constexpr int UN = 4;   // float4 pairs requested before the first is used (memory-level parallelism, as in bn_stats_kernel)
    for (ChanWalk w(lo, hi, c, C, HW); w.valid();) {
        float4 gq[UN], xq[UN];
        int64_t aq[UN];
        bool ok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            ok[u] = w.valid();
            aq[u] = ok[u] ? w.addr() : 0;
            gq[u] = ok[u] ? *reinterpret_cast<const float4*>(dy + aq[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
            xq[u] = ok[u] ? *reinterpret_cast<const float4*>(x + aq[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
            w.next();
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (!ok[u]) continue;
            float4 g = gq[u];
            const float4 xv = xq[u];
            const int64_t a = aq[u];
            if (RELU) {
                if (mask) {
                    const int64_t i4 = a >> 2;
                    if (!gate_bit(mask, i4, 0)) g.x = 0.f;
                    if (!gate_bit(mask, i4, 1)) g.y = 0.f;
                    if (!gate_bit(mask, i4, 2)) g.z = 0.f;
                    if (!gate_bit(mask, i4, 3)) g.w = 0.f;
                } else {
                    const float4 o = *reinterpret_cast<const float4*>(y + a);
                    if (!(o.x > 0.f)) g.x = 0.f;
                    if (!(o.y > 0.f)) g.y = 0.f;
                    if (!(o.z > 0.f)) g.z = 0.f;
                    if (!(o.w > 0.f)) g.w = 0.f;
                }
            }
            s1 += (g.x + g.y) + (g.z + g.w);
            s2 += (g.x * ((xv.x - mu) * is) + g.y * ((xv.y - mu) * is)) + (g.z * ((xv.z - mu) * is) + g.w * ((xv.w - mu) * is));
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { partial[((int64_t)c * S + s) * 2] = s1; partial[((int64_t)c * S + s) * 2 + 1] = s2; }
}

__global__ __launch_bounds__(64) void bn_bwd_finish_kernel(float* __restrict__ sum_dy, float* __restrict__ sum_dy_xhat,
                                                           const float* __restrict__ partial, int C, int S) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int s = 0; s < S; ++s) { a += partial[((int64_t)c * S + s) * 2]; b += partial[((int64_t)c * S + s) * 2 + 1]; }
    sum_dy[c] = a; sum_dy_xhat[c] = b;
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float4* __restrict__ dx, float4* __restrict__ dres,
                                                           const float4* __restrict__ dy, const float4* __restrict__ y,
                                                           const float4* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sum_dy,
                                                           const float* __restrict__ sum_dy_xhat, int C, int HW4,
                                                           int64_t total4, float inv_n,
                                                           const unsigned long long* __restrict__ mask) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        const int c = (int)((i / HW4) % C);
        const float mu = mean[c], is = invstd[c];
        const float k = (gamma ? gamma[c] : 1.f) * is;
        const float m1 = sum_dy[c] * inv_n, m2 = sum_dy_xhat[c] * inv_n;
        float4 g = dy[i];
        if (RELU) {
            if (mask) {
                if (!gate_bit(mask, i, 0)) g.x = 0.f;
                if (!gate_bit(mask, i, 1)) g.y = 0.f;
                if (!gate_bit(mask, i, 2)) g.z = 0.f;
                if (!gate_bit(mask, i, 3)) g.w = 0.f;
            } else {
                const float4 o = y[i];
                if (!(o.x > 0.f)) g.x = 0.f;
                if (!(o.y > 0.f)) g.y = 0.f;
                if (!(o.z > 0.f)) g.z = 0.f;
                if (!(o.w > 0.f)) g.w = 0.f;
            }
        }
        if (RES) dres[i] = g;
        const float4 xv = x[i];
        float4 r;
        r.x = k * (g.x - m1 - (xv.x - mu) * is * m2);
        r.y = k * (g.y - m1 - (xv.y - mu) * is * m2);
        r.z = k * (g.z - m1 - (xv.z - mu) * is * m2);
        r.w = k * (g.w - m1 - (xv.w - mu) * is * m2);
        dx[i] = r;
    }
}

// ---- single-pass kernels for channels whose B * HW elements are ONE slice (<= 16 384: every 32 x 32 layer at batch 16 --
// 42 of EMANet-50's norms): one workgroup per channel keeps its 64 values per thread in registers, so the forward reads x
// once instead of twice (statistics + apply) and the backward reads dy and x once instead of twice; 1 launch instead of 3.
// Same expressions in the same order as the kernels above (forward: bitwise the same results; backward: to an ulp).  (HW % 256 == 0: a wave's 64 float4s
// stay inside one plane and fill one group of mask words.)
template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_fused_fwd_kernel(float* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, const float* __restrict__ x,
                                                           const float* __restrict__ res, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int C, int HW, int64_t n, float eps,
                                                           float momentum, unsigned long long* __restrict__ mask) {
    __shared__ float red[4];
    constexpr int PER = BN_SLICE / 1024;
    const int c = blockIdx.x;
    float4 v[PER];
    int64_t addr[PER];
    {
        ChanWalk w(0, n, c, C, HW);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            addr[k] = w.valid() ? w.addr() : -1;
            v[k] = w.valid() ? *reinterpret_cast<const float4*>(x + addr[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
            w.next();
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float cnt = (float)n;
    const float mean = block_sum(sum, red) / cnt;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (addr[k] >= 0) {
            const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
            m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    m2 = block_sum(m2, red);
    const float var = m2 / cnt;
    const float is = rsqrtf(var + eps);
    if (threadIdx.x == 0) {
        mean_out[c] = mean; invstd_out[c] = is;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (cnt > 1.f ? m2 / (cnt - 1.f) : var);
        }
    }
    const float a = (gamma ? gamma[c] : 1.f) * is;
    const float b = (beta ? beta[c] : 0.f) - mean * a;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (addr[k] < 0) continue;   // (wave-uniform: n % 256 == 0)
        float4 o = v[k];
        o.x = o.x * a + b; o.y = o.y * a + b; o.z = o.z * a + b; o.w = o.w * a + b;
        if (RES) { const float4 r = *reinterpret_cast<const float4*>(res + addr[k]); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(y + addr[k]) = o;
        if (RELU && mask) {
            const unsigned long long b0 = __ballot(o.x > 0.f), b1 = __ballot(o.y > 0.f), b2 = __ballot(o.z > 0.f), b3 = __ballot(o.w > 0.f);
            if ((threadIdx.x & 63) == 0) {
                unsigned long long* m = mask + ((addr[k] >> 2) >> 6) * 4;
                m[0] = b0; m[1] = b1; m[2] = b2; m[3] = b3;
            }
        }
    }
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void bn_fused_bwd_kernel(float* __restrict__ dx, float* __restrict__ dres, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, const float* __restrict__ dy,
                                                           const float* __restrict__ y, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, int C, int HW, int64_t n,
                                                           const unsigned long long* __restrict__ mask) {
    __shared__ float red[4];
    constexpr int PER = BN_SLICE / 1024;
    const int c = blockIdx.x;
    const float mu = mean[c], is = invstd[c];
    float4 g[PER], xv[PER];
    int64_t addr[PER];
    {
        ChanWalk w(0, n, c, C, HW);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            addr[k] = w.valid() ? w.addr() : -1;
            g[k] = w.valid() ? *reinterpret_cast<const float4*>(dy + addr[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
            xv[k] = w.valid() ? *reinterpret_cast<const float4*>(x + addr[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
            w.next();
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (addr[k] < 0) continue;
        if (RELU) {
            if (mask) {
                const int64_t i4 = addr[k] >> 2;
                if (!gate_bit(mask, i4, 0)) g[k].x = 0.f;
                if (!gate_bit(mask, i4, 1)) g[k].y = 0.f;
                if (!gate_bit(mask, i4, 2)) g[k].z = 0.f;
                if (!gate_bit(mask, i4, 3)) g[k].w = 0.f;
            } else {
                const float4 o = *reinterpret_cast<const float4*>(y + addr[k]);
                if (!(o.x > 0.f)) g[k].x = 0.f;
                if (!(o.y > 0.f)) g[k].y = 0.f;
                if (!(o.z > 0.f)) g[k].z = 0.f;
                if (!(o.w > 0.f)) g[k].w = 0.f;
            }
        }
        s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
        s2 += (g[k].x * ((xv[k].x - mu) * is) + g[k].y * ((xv[k].y - mu) * is)) + (g[k].z * ((xv[k].z - mu) * is) + g[k].w * ((xv[k].w - mu) * is));
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) { dbeta[c] = s1; dgamma[c] = s2; }
    const float inv_n = 1.f / (float)n;
    const float kk = (gamma ? gamma[c] : 1.f) * is;
    const float m1 = s1 * inv_n, m2 = s2 * inv_n;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (addr[k] < 0) continue;
        if (RES) *reinterpret_cast<float4*>(dres + addr[k]) = g[k];
        float4 r;
        r.x = kk * (g[k].x - m1 - (xv[k].x - mu) * is * m2);
        r.y = kk * (g[k].y - m1 - (xv[k].y - mu) * is * m2);
        r.z = kk * (g[k].z - m1 - (xv[k].z - mu) * is * m2);
        r.w = kk * (g[k].w - m1 - (xv[k].w - mu) * is * m2);
        *reinterpret_cast<float4*>(dx + addr[k]) = r;
    }
}

// ---- the same single-pass idea for channels of up to 65 536 values (EMANet's 64 x 64 layers at batch 16: layer1's ten norms,
// 0.44 GB of activations that the three-launch form reads three times forward and five times backward): one workgroup of
// 1 024 threads per channel, 64 values per thread.  Forward: the values stay in registers between the statistics and the apply
// step (as above).  Backward: 1 024 threads have 128 registers each, not enough for dy AND x -- dy stays in registers, x is read
// a second time for the apply step, out of the L2 this workgroup has just pulled its 256 KB through.  Statistics: the mean and the
// centred second moment of the whole channel directly (no per-slice merge): within 1e-6 of the three-launch form, not bitwise.
constexpr int BN_WIDE = 65536, BN_WIDE_THREADS = 1024, BN_WIDE_PER = BN_WIDE / (4 * BN_WIDE_THREADS);

template <int THREADS = BN_WIDE_THREADS>
struct ChanWalkWide {   // ChanWalk with a step of 4 * THREADS elements
    int64_t e, hi;
    int b, r, C, HW, c;
    __device__ __forceinline__ ChanWalkWide(int64_t hi_, int c_, int C_, int HW_) : hi(hi_), C(C_), HW(HW_), c(c_) {
        e = (int64_t)threadIdx.x * 4;
        b = (int)(e / HW);
        r = (int)(e - (int64_t)b * HW);
    }
    __device__ __forceinline__ bool valid() const { return e < hi; }
    __device__ __forceinline__ int64_t addr() const { return ((int64_t)b * C + c) * HW + r; }
    __device__ __forceinline__ void next() {
        e += 4 * THREADS; r += 4 * THREADS;
        while (r >= HW) { r -= HW; ++b; }
    }
};

template <int THREADS = BN_WIDE_THREADS>
__device__ __forceinline__ float block_sum_wide(float v, float* red) {   // THREADS / 64 waves, fixed order
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) s += red[w];
    return s;
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(BN_WIDE_THREADS) void bn_wide_fwd_kernel(float* __restrict__ y, float* __restrict__ mean_out,
                                                                     float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                                                     float* __restrict__ running_var, const float* __restrict__ x,
                                                                     const float* __restrict__ res, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, int C, int HW, int64_t n, float eps,
                                                                     float momentum, unsigned long long* __restrict__ mask) {
    __shared__ float red[BN_WIDE_THREADS / 64];
    const int c = blockIdx.x;
    float4 v[BN_WIDE_PER];
    {
        ChanWalkWide<> w(n, c, C, HW);
#pragma unroll
        for (int k = 0; k < BN_WIDE_PER; ++k) {
            v[k] = w.valid() ? *reinterpret_cast<const float4*>(x + w.addr()) : make_float4(0.f, 0.f, 0.f, 0.f);
            w.next();
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < BN_WIDE_PER; ++k) sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float cnt = (float)n;
    const float mean = block_sum_wide(sum, red) / cnt;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < BN_WIDE_PER; ++k) {
        if ((int64_t)threadIdx.x * 4 + (int64_t)k * 4 * BN_WIDE_THREADS < n) {
            const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
            m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    m2 = block_sum_wide(m2, red);
    const float var = m2 / cnt;
    const float is = rsqrtf(var + eps);
    if (threadIdx.x == 0) {
        mean_out[c] = mean; invstd_out[c] = is;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (cnt > 1.f ? m2 / (cnt - 1.f) : var);
        }
    }
    const float a = (gamma ? gamma[c] : 1.f) * is;
    const float b = (beta ? beta[c] : 0.f) - mean * a;
    ChanWalkWide<> w(n, c, C, HW);
#pragma unroll
    for (int k = 0; k < BN_WIDE_PER; ++k) {
        if (w.valid()) {   // (wave-uniform: n % 256 == 0)
            const int64_t ad = w.addr();
            float4 o = v[k];
            o.x = o.x * a + b; o.y = o.y * a + b; o.z = o.z * a + b; o.w = o.w * a + b;
            if (RES) { const float4 r = *reinterpret_cast<const float4*>(res + ad); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
            if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<float4*>(y + ad) = o;
            if (RELU && mask) {
                const unsigned long long b0 = __ballot(o.x > 0.f), b1 = __ballot(o.y > 0.f), b2 = __ballot(o.z > 0.f), b3 = __ballot(o.w > 0.f);
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long* m = mask + ((ad >> 2) >> 6) * 4;
                    m[0] = b0; m[1] = b1; m[2] = b2; m[3] = b3;
                }
            }
        }
        w.next();
    }
}

// THREADS x PER float4 = the channel's capacity (1 024 x 16: 65 536 values, x read twice; 512 x 8: 16 384 values, x kept in registers)
template <bool RELU, bool RES, int THREADS, int PER, bool KEEP_X>
__global__ __launch_bounds__(THREADS) void bn_wide_bwd_kernel(float* __restrict__ dx, float* __restrict__ dres,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                     const float* __restrict__ dy, const float* __restrict__ y,
                                                                     const float* __restrict__ x, const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                     int C, int HW, int64_t n, const unsigned long long* __restrict__ mask) {
    __shared__ float red[THREADS / 64];
    const int c = blockIdx.x;
    const float mu = mean[c], is = invstd[c];
    float4 g[PER], xk[KEEP_X ? PER : 1];
    float s1 = 0.f, s2 = 0.f;
    {
        ChanWalkWide<THREADS> w(n, c, C, HW);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w.valid()) {
                const int64_t ad = w.addr();
                g[k] = *reinterpret_cast<const float4*>(dy + ad);
                const float4 xv = *reinterpret_cast<const float4*>(x + ad);
                if (KEEP_X) xk[k] = xv;
                if (RELU) {
                    if (mask) {
                        const int64_t i4 = ad >> 2;
                        if (!gate_bit(mask, i4, 0)) g[k].x = 0.f;
                        if (!gate_bit(mask, i4, 1)) g[k].y = 0.f;
                        if (!gate_bit(mask, i4, 2)) g[k].z = 0.f;
                        if (!gate_bit(mask, i4, 3)) g[k].w = 0.f;
                    } else {
                        const float4 o = *reinterpret_cast<const float4*>(y + ad);
                        if (!(o.x > 0.f)) g[k].x = 0.f;
                        if (!(o.y > 0.f)) g[k].y = 0.f;
                        if (!(o.z > 0.f)) g[k].z = 0.f;
                        if (!(o.w > 0.f)) g[k].w = 0.f;
                    }
                }
                s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
                s2 += (g[k].x * ((xv.x - mu) * is) + g[k].y * ((xv.y - mu) * is)) + (g[k].z * ((xv.z - mu) * is) + g[k].w * ((xv.w - mu) * is));
            }
            w.next();
        }
    }
    s1 = block_sum_wide<THREADS>(s1, red);
    s2 = block_sum_wide<THREADS>(s2, red);
    if (threadIdx.x == 0) { dbeta[c] = s1; dgamma[c] = s2; }
    const float inv_n = 1.f / (float)n;
    const float kk = (gamma ? gamma[c] : 1.f) * is;
    const float m1 = s1 * inv_n, m2 = s2 * inv_n;
    ChanWalkWide<THREADS> w(n, c, C, HW);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (w.valid()) {
            const int64_t ad = w.addr();
            const float4 xv = KEEP_X ? xk[k] : *reinterpret_cast<const float4*>(x + ad);   // (second read: L2)
            if (RES) *reinterpret_cast<float4*>(dres + ad) = g[k];
            float4 r;
            r.x = kk * (g[k].x - m1 - (xv.x - mu) * is * m2);
            r.y = kk * (g[k].y - m1 - (xv.y - mu) * is * m2);
            r.z = kk * (g[k].z - m1 - (xv.z - mu) * is * m2);
            r.w = kk * (g[k].w - m1 - (xv.w - mu) * is * m2);
            *reinterpret_cast<float4*>(dx + ad) = r;
        }
        w.next();
    }
}

bool bn_wide_ok(int batch, int hw) {
    const char* e = getenv("SIS_BN_SINGLE_PASS");   // (read per call, as bn_fused_ok)
    const char* w = getenv("SIS_BN_WIDE");          // 0: channels above 16 384 values keep the three-launch form (A/B runs)
    const int64_t n = (int64_t)batch * hw;
    return !(e && e[0] == '0') && !(w && w[0] == '0') && n > BN_SLICE && n <= BN_WIDE && hw % 256 == 0;
}

bool bn_fused_ok(int batch, int hw) {
    const char* e = getenv("SIS_BN_SINGLE_PASS");   // 0: the three-launch form (A/B runs, the equality test); read per call
    return !(e && e[0] == '0') && (int64_t)batch * hw <= BN_SLICE && hw % 256 == 0;
}

int slices(int64_t n) { return (int)((n + BN_SLICE - 1) / BN_SLICE); }
int ew_blocks(int64_t total4) { const int64_t b = (total4 + 255) / 256; return (int)(b < 4096 ? b : 4096); }

int check_geom(const char* name, int B, int C, int HW) {
    SIS_REQUIRE(B > 0 && C > 0 && HW > 0, "%s: non-positive size", name);
    SIS_REQUIRE(HW % 4 == 0, "%s: H*W must be a multiple of 4 (16-byte rows)", name);
    return 0;
}

}  // namespace

extern "C" int64_t sis_bn_workspace_floats(int batch, int channels, int hw) {
    return (int64_t)channels * slices((int64_t)batch * hw) * 3;
}

extern "C" int sis_bn_stats(float* mean, float* invstd, float* running_mean, float* running_var, const float* x,
                            float* workspace, int batch, int channels, int hw, float eps, float momentum, void* stream) {
    if (check_geom("sis_bn_stats", batch, channels, hw)) return 1;
    SIS_REQUIRE(mean && invstd && x && workspace, "sis_bn_stats: null pointer");
    const int64_t n = (int64_t)batch * hw;
    const int S = slices(n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(channels * S), dim3(256), 0, st, workspace, x, channels, hw, n, S);
    SIS_CHECK_LAUNCH("bn_stats_kernel");
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(sis_cdiv(channels, 64)), dim3(64), 0, st, mean, invstd, running_mean,
                       running_var, workspace, channels, S, eps, momentum);
    SIS_CHECK_LAUNCH("bn_stats_finish_kernel");
    return 0;
}

extern "C" int64_t sis_bn_mask_words(int batch, int channels, int hw) {
    return (((int64_t)batch * channels * hw / 4 + 63) / 64) * 4;
}

extern "C" int sis_bn_act_fwd(float* y, const float* x, const float* residual, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, int batch, int channels, int hw, int relu,
                              void* relu_mask, void* stream) {
    if (check_geom("sis_bn_act_fwd", batch, channels, hw)) return 1;
    SIS_REQUIRE(y && x && mean && invstd, "sis_bn_act_fwd: null pointer");
    SIS_REQUIRE((((uintptr_t)y | (uintptr_t)x | (uintptr_t)residual) & 15) == 0, "sis_bn_act_fwd: 16-byte alignment");
    const int64_t total4 = (int64_t)batch * channels * hw / 4;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ew_blocks(total4)), blk(256);
#define SIS_BN_FWD(R, S_)                                                                                              \
    hipLaunchKernelGGL((bn_act_fwd_kernel<R, S_>), grid, blk, 0, st, (float4*)y, (const float4*)x, (const float4*)residual, \
                       mean, invstd, gamma, beta, channels, hw / 4, total4, (unsigned long long*)relu_mask)
    if (relu && residual) SIS_BN_FWD(true, true);
    else if (relu) SIS_BN_FWD(true, false);
    else if (residual) SIS_BN_FWD(false, true);
    else SIS_BN_FWD(false, false);
#undef SIS_BN_FWD
    SIS_CHECK_LAUNCH("bn_act_fwd_kernel");
    return 0;
}

extern "C" int sis_bn_act_bwd(float* dx, float* dresidual, float* dgamma, float* dbeta, const float* dy, const float* y,
                              const float* x, const float* mean, const float* invstd, const float* gamma, float* workspace,
                              int batch, int channels, int hw, int relu, const void* relu_mask, void* stream) {
    if (check_geom("sis_bn_act_bwd", batch, channels, hw)) return 1;
    SIS_REQUIRE(dx && dgamma && dbeta && dy && x && mean && invstd && workspace, "sis_bn_act_bwd: null pointer");
    SIS_REQUIRE(!relu || y || relu_mask, "sis_bn_act_bwd: the ReLU gate needs the forward output or its sign mask");
    const unsigned long long* mk = (const unsigned long long*)relu_mask;
    SIS_REQUIRE((((uintptr_t)dx | (uintptr_t)dresidual | (uintptr_t)dy | (uintptr_t)y | (uintptr_t)x) & 15) == 0,
                "sis_bn_act_bwd: 16-byte alignment");
    const int64_t n = (int64_t)batch * hw;
    const int S = slices(n);
    hipStream_t st = (hipStream_t)stream;
    if (bn_wide_ok(batch, hw)) {
#define SIS_BN_WBWD(R, S_)                                                                                                          \
    hipLaunchKernelGGL((bn_wide_bwd_kernel<R, S_, BN_WIDE_THREADS, BN_WIDE_PER, false>), dim3(channels), dim3(BN_WIDE_THREADS), 0, st, dx, dresidual, dgamma, dbeta, dy, y, x, \
                       mean, invstd, gamma, channels, hw, n, mk)
        if (relu && dresidual) SIS_BN_WBWD(true, true);
        else if (relu) SIS_BN_WBWD(true, false);
        else if (dresidual) SIS_BN_WBWD(false, true);
        else SIS_BN_WBWD(false, false);
#undef SIS_BN_WBWD
        SIS_CHECK_LAUNCH("bn_wide_bwd_kernel");
        sis_kernel_name = "bn_wide_bwd_kernel";
        return 0;
    }
    static const bool bwd512 = !(getenv("SIS_BN_BWD512") && getenv("SIS_BN_BWD512")[0] == '0');   // 0: the 256-thread form (A/B runs)
    if (bn_fused_ok(batch, hw) && bwd512) {
        // 512 threads x 8 float4 (dy and x in registers, ~100 VGPRs: 16 waves per CU instead of the 8 the 256-thread form's
        // ~200 VGPRs allow) -- the same sums in a different association: to an ulp of the 256-thread form
#define SIS_BN_MBWD(R, S_)                                                                                                   \
    hipLaunchKernelGGL((bn_wide_bwd_kernel<R, S_, 512, BN_SLICE / 2048, true>), dim3(channels), dim3(512), 0, st, dx, dresidual, dgamma, \
                       dbeta, dy, y, x, mean, invstd, gamma, channels, hw, n, mk)
        if (relu && dresidual) SIS_BN_MBWD(true, true);
        else if (relu) SIS_BN_MBWD(true, false);
        else if (dresidual) SIS_BN_MBWD(false, true);
        else SIS_BN_MBWD(false, false);
#undef SIS_BN_MBWD
        SIS_CHECK_LAUNCH("bn_wide_bwd_kernel<512>");
        sis_kernel_name = "bn_wide_bwd_kernel";   // the kernel that runs (rocprof / PMC tables are joined by this name)
        return 0;
    }
    if (bn_fused_ok(batch, hw)) {
#define SIS_BN_FBWD(R, S_)                                                                                                  \
    hipLaunchKernelGGL((bn_fused_bwd_kernel<R, S_>), dim3(channels), dim3(256), 0, st, dx, dresidual, dgamma, dbeta, dy, y, x, mean, \
                       invstd, gamma, channels, hw, n, mk)
        if (relu && dresidual) SIS_BN_FBWD(true, true);
        else if (relu) SIS_BN_FBWD(true, false);
        else if (dresidual) SIS_BN_FBWD(false, true);
        else SIS_BN_FBWD(false, false);
#undef SIS_BN_FBWD
        SIS_CHECK_LAUNCH("bn_fused_bwd_kernel");
        sis_kernel_name = "bn_fused_bwd_kernel";
        return 0;
    }
    if (relu)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, dim3(channels * S), dim3(256), 0, st, workspace, dy, y, x, mean, invstd,
                           channels, hw, n, S, mk);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, dim3(channels * S), dim3(256), 0, st, workspace, dy, y, x, mean,
                           invstd, channels, hw, n, S, mk);
    SIS_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(sis_cdiv(channels, 64)), dim3(64), 0, st, dbeta, dgamma, workspace, channels, S);
    SIS_CHECK_LAUNCH("bn_bwd_finish_kernel");
    const int64_t total4 = (int64_t)batch * channels * hw / 4;
    const dim3 grid(ew_blocks(total4)), blk(256);
    const float inv_n = 1.f / (float)n;
#define SIS_BN_BWD(R, S_)                                                                                              \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<R, S_>), grid, blk, 0, st, (float4*)dx, (float4*)dresidual, (const float4*)dy, \
                       (const float4*)y, (const float4*)x, mean, invstd, gamma, dbeta, dgamma, channels, hw / 4, total4, inv_n, mk)
    if (relu && dresidual) SIS_BN_BWD(true, true);
    else if (relu) SIS_BN_BWD(true, false);
    else if (dresidual) SIS_BN_BWD(false, true);
    else SIS_BN_BWD(false, false);
#undef SIS_BN_BWD
    SIS_CHECK_LAUNCH("bn_bwd_apply_kernel");
    sis_kernel_name = "bn_bwd_reduce_kernel+bn_bwd_apply_kernel";
    return 0;
}

/* Statistics + apply of a training-mode batch norm in ONE launch when a channel's batch * hw elements fit one workgroup's
 * registers (sis_bn_fused_supported: batch * hw <= 16 384 and hw % 256 == 0): mean / invstd / running statistics as
 * sis_bn_stats, y (and relu_mask) as sis_bn_act_fwd, bitwise the same values.  (sis_bn_act_bwd takes its single-pass form
 * under the same condition by itself.) */
extern "C" int sis_bn_fused_supported(int batch, int channels, int hw) {
    return (batch > 0 && channels > 0 && hw > 0 && (bn_fused_ok(batch, hw) || bn_wide_ok(batch, hw))) ? 1 : 0;
}

extern "C" int sis_bn_fused_fwd(float* y, float* mean, float* invstd, float* running_mean, float* running_var, const float* x,
                                const float* residual, const float* gamma, const float* beta, int batch, int channels, int hw,
                                float eps, float momentum, int relu, void* relu_mask, void* stream) {
    if (check_geom("sis_bn_fused_fwd", batch, channels, hw)) return 1;
    SIS_REQUIRE(y && mean && invstd && x, "sis_bn_fused_fwd: null pointer");
    SIS_REQUIRE(bn_fused_ok(batch, hw) || bn_wide_ok(batch, hw), "sis_bn_fused_fwd: batch * hw = %lld does not fit one workgroup / hw %% 256",
                (long long)batch * hw);
    SIS_REQUIRE((((uintptr_t)y | (uintptr_t)x | (uintptr_t)residual) & 15) == 0, "sis_bn_fused_fwd: 16-byte alignment");
    const int64_t n = (int64_t)batch * hw;
    hipStream_t st = (hipStream_t)stream;
    if (bn_wide_ok(batch, hw)) {
#define SIS_BN_WFWD(R, S_)                                                                                                        \
    hipLaunchKernelGGL((bn_wide_fwd_kernel<R, S_>), dim3(channels), dim3(BN_WIDE_THREADS), 0, st, y, mean, invstd, running_mean,     \
                       running_var, x, residual, gamma, beta, channels, hw, n, eps, momentum, (unsigned long long*)relu_mask)
        if (relu && residual) SIS_BN_WFWD(true, true);
        else if (relu) SIS_BN_WFWD(true, false);
        else if (residual) SIS_BN_WFWD(false, true);
        else SIS_BN_WFWD(false, false);
#undef SIS_BN_WFWD
        SIS_CHECK_LAUNCH("bn_wide_fwd_kernel");
        sis_kernel_name = "bn_wide_fwd_kernel";
        return 0;
    }
#define SIS_BN_FFWD(R, S_)                                                                                              \
    hipLaunchKernelGGL((bn_fused_fwd_kernel<R, S_>), dim3(channels), dim3(256), 0, st, y, mean, invstd, running_mean, running_var, \
                       x, residual, gamma, beta, channels, hw, n, eps, momentum, (unsigned long long*)relu_mask)
    if (relu && residual) SIS_BN_FFWD(true, true);
    else if (relu) SIS_BN_FFWD(true, false);
    else if (residual) SIS_BN_FFWD(false, true);
    else SIS_BN_FFWD(false, false);
#undef SIS_BN_FFWD
    SIS_CHECK_LAUNCH("bn_fused_fwd_kernel");
    sis_kernel_name = "bn_fused_fwd_kernel";
    return 0;
}
