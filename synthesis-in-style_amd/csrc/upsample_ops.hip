// Bilinear upsampling with align_corners=True (nn.UpsamplingBilinear2d), forward and backward -- the TransUNet decoder
// (networks/trans_u_net/vit_seg_modeling.py:290-329: every DecoderBlock and the SegmentationHead upsample by 2).
// ATen's kernel runs these four tensors at ~1/8 of the HBM roofline and its backward scatters with float atomics;
// here the forward is one pass (x2: source tile through LDS, 16-byte stores; other ratios: each lane 4 consecutive outputs
// of a row, the two source rows come from L2) and the backward is a gather (one lane per input pixel sums the <= 4 x 4 outputs whose footprint touches
// it): deterministic, no atomics.  Index rule = PyTorch's: src = dst * (in - 1) / (out - 1), i0 = (int) src,
// i1 = i0 + (i0 < in - 1), lambda1 = src - i0.  Arithmetic in fp32 for f32 / bf16 / f16 tensors.
#include "sis_common.h"

namespace {

struct Lerp1 { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp1 lerp1(int dst, float scale, int in_size) {
    const float src = scale * (float)dst;
    Lerp1 r;
    r.i0 = (int)src;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_up_fwd_kernel(T* __restrict__ out, const T* __restrict__ x, int h, int w,
                                                              int oh, int ow, float sy, float sx, int64_t total4) {
    const int ow4 = (ow + 3) >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        const int c4 = (int)(i % ow4);
        const int64_t row = i / ow4;
        const int oy = (int)(row % oh);
        const int64_t plane = row / oh;
        const Lerp1 ly = lerp1(oy, sy, h);
        const T* r0 = x + (plane * h + ly.i0) * w;
        const T* r1 = x + (plane * h + ly.i1) * w;
        T* o = out + (plane * oh + oy) * (int64_t)ow + 4 * c4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ox = 4 * c4 + e;
            const Lerp1 lx = lerp1(ox < ow ? ox : ow - 1, sx, w);
            v[e] = ly.l0 * (lx.l0 * sis_ld(r0, lx.i0) + lx.l1 * sis_ld(r0, lx.i1)) +
                   ly.l1 * (lx.l0 * sis_ld(r1, lx.i0) + lx.l1 * sis_ld(r1, lx.i1));
        }
        if (sizeof(T) == 4 && (ow & 3) == 0) {  // 16-byte store (rows are 16-byte aligned when ow % 4 == 0)
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else if (sizeof(T) == 2 && (ow & 3) == 0) {  // four 16-bit results as one 8-byte store
            T t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) sis_st(t, e, v[e]);
            uint2 q;
            __builtin_memcpy(&q, t, 8);
            *reinterpret_cast<uint2*>(o) = q;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * c4 + e < ow) sis_st(o, e, v[e]);
        }
    }
}

// Forward of the x2 case (every call of the TransUNet decoder), tiled through LDS: a workgroup owns 16 output rows x <= 512
// output columns of one plane.  The <= 10 x 258 source pixels they read arrive once (coalesced) and are kept as fp32; every
// lane then produces 8 consecutive outputs per store (16 bytes of bf16) from LDS reads -- the kernel above issues 16
// two-byte global loads per 4 outputs and ran at 1.2 TB/s of its traffic.  Same fp32 expression per output as above.
// `out_image_stride`: elements between consecutive images of `out` (>= planes_per_image * oh * ow): the result can land in the
// leading channels of a wider tensor (the decoder's concatenation with the skip feature needs no second copy of it).
constexpr int UF_OR = 16, UF_OC = 512, UF_SR = UF_OR / 2 + 2, UF_SC = UF_OC / 2 + 2;

template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_fwd_tiled_kernel(T* __restrict__ out, const T* __restrict__ x, int h, int w,
                                                                     int oh, int ow, float sy, float sx, int tiles_x, int tiles_y,
                                                                     int planes_per_image, int64_t out_image_stride) {
    __shared__ float tile[UF_SR][UF_SC + 1];
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int64_t plane = b / tiles_y;
    const int oy0 = ty_i * UF_OR, ox0 = tx_i * UF_OC;
    const int oy1 = min(oh, oy0 + UF_OR) - 1, ox1 = min(ow, ox0 + UF_OC) - 1;   // last output row / column of the tile
    const int r_lo = lerp1(oy0, sy, h).i0, r_hi = lerp1(oy1, sy, h).i1;         // <= UF_SR rows: src advances < 1/2 per output
    const int c_lo = lerp1(ox0, sx, w).i0, c_hi = lerp1(ox1, sx, w).i1;
    const int nr = r_hi - r_lo + 1, nc = c_hi - c_lo + 1;
    const T* src = x + plane * h * (int64_t)w;
    for (int e = threadIdx.x; e < nr * nc; e += 256) {
        const int r = e / nc, c = e - r * nc;
        tile[r][c] = sis_ld(src, (int64_t)(r_lo + r) * w + c_lo + c);
    }
    __syncthreads();
    const int64_t n = plane / planes_per_image, ch = plane - n * planes_per_image;
    T* dst = out + n * out_image_stride + ch * oh * (int64_t)ow;
    constexpr int GROUPS = UF_OC / 8;
    static_assert(256 % GROUPS == 0, "a thread keeps its column group over the rows it handles");
    // a thread's 8 output columns are the same for all its rows: their source columns / weights are formed once
    const int g = threadIdx.x % GROUPS, ox = ox0 + 8 * g;
    if (ox >= ow) return;
    int xa[8], xb[8];
    float xl0[8], xl1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const Lerp1 lx = lerp1(min(ox + k, ow - 1), sx, w);
        xa[k] = lx.i0 - c_lo; xb[k] = lx.i1 - c_lo; xl0[k] = lx.l0; xl1[k] = lx.l1;
    }
    for (int ry = threadIdx.x / GROUPS; ry < UF_OR; ry += 256 / GROUPS) {
        const int oy = oy0 + ry;
        if (oy >= oh) break;
        const Lerp1 ly = lerp1(oy, sy, h);
        const float* t0 = tile[ly.i0 - r_lo];
        const float* t1 = tile[ly.i1 - r_lo];
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = ly.l0 * (xl0[k] * t0[xa[k]] + xl1[k] * t0[xb[k]]) + ly.l1 * (xl0[k] * t1[xa[k]] + xl1[k] * t1[xb[k]]);
        T* o = dst + (int64_t)oy * ow + ox;
        if (ox + 8 <= ow && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            if constexpr (sizeof(T) == 2) {
                T t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) sis_st(t, k, v[k]);
                uint4 q;
                __builtin_memcpy(&q, t, 16);
                *reinterpret_cast<uint4*>(o) = q;
            } else {
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (ox + k < ow) sis_st(o, k, v[k]);
        }
    }
}

// grad_x[y][x] = sum over outputs (oy, ox) of wy(oy -> y) * wx(ox -> x) * grad_out[oy][ox]
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up_bwd_kernel(T* __restrict__ gx, const T* __restrict__ gout, int h, int w,
                                                              int oh, int ow, float sy, float sx, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % w);
    const int64_t row = i / w;
    const int y = (int)(row % h);
    const int64_t plane = row / h;
    // outputs o with floor(s * o) in {y - 1, y}
    const float inv_sy = sy > 0.f ? 1.f / sy : 0.f, inv_sx = sx > 0.f ? 1.f / sx : 0.f;
    // (floor / ceil of the real bounds already bracket every contributor; a rounding error of the quotient can only
    // move them outwards by one, never inwards past a contributor, and non-contributors get weight 0 below)
    int oy_lo = sy > 0.f ? (int)floorf((float)(y - 1) * inv_sy) : 0;
    int oy_hi = sy > 0.f ? (int)ceilf((float)(y + 1) * inv_sy) : oh - 1;
    int ox_lo = sx > 0.f ? (int)floorf((float)(xx - 1) * inv_sx) : 0;
    int ox_hi = sx > 0.f ? (int)ceilf((float)(xx + 1) * inv_sx) : ow - 1;
    oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0); oy_hi = min(oy_hi, oh - 1); ox_hi = min(ox_hi, ow - 1);
    const T* g = gout + plane * oh * (int64_t)ow;
    float acc = 0.f;
    if (ox_hi - ox_lo < 6 && oy_hi - oy_lo < 6) {
        // x2 upsampling: at most 6 candidates per axis; the column weights are formed once, not once per row
        float wxs[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int ox = ox_lo + k;
            const Lerp1 lx = lerp1(ox <= ox_hi ? ox : ox_hi, sx, w);
            float wx = 0.f;
            if (ox <= ox_hi && lx.i0 == xx) wx += lx.l0;
            if (ox <= ox_hi && lx.i1 == xx) wx += lx.l1;
            wxs[k] = wx;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int oy = oy_lo + j;
            if (oy > oy_hi) break;
            const Lerp1 ly = lerp1(oy, sy, h);
            float wy = 0.f;
            if (ly.i0 == y) wy += ly.l0;
            if (ly.i1 == y) wy += ly.l1;
            if (wy == 0.f) continue;
            const T* grow = g + (int64_t)oy * ow + ox_lo;
            float racc = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (wxs[k] != 0.f) racc += wxs[k] * sis_ld(grow, k);
            acc += wy * racc;
        }
        sis_st(gx, i, acc);
        return;
    }
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const Lerp1 ly = lerp1(oy, sy, h);
        float wy = 0.f;
        if (ly.i0 == y) wy += ly.l0;
        if (ly.i1 == y) wy += ly.l1;
        if (wy == 0.f) continue;
        float racc = 0.f;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const Lerp1 lx = lerp1(ox, sx, w);
            float wx = 0.f;
            if (lx.i0 == xx) wx += lx.l0;
            if (lx.i1 == xx) wx += lx.l1;
            if (wx != 0.f) racc += wx * sis_ld(g, (int64_t)oy * ow + ox);
        }
        acc += wy * racc;
    }
    sis_st(gx, i, acc);
}

// Backward of the x2 case (oh = 2h, ow = 2w: every call of the TransUNet decoder), tiled through LDS: a workgroup owns an
// 8 x 64 tile of grad_x of one plane.  The <= 22 x 144 grad_out window its footprint can touch arrives by 16-byte row loads
// (coalesced; the gather kernel above issues up to 36 two-byte loads per input pixel and ran at ~0.6 TB/s), is reduced
// along y with the row weights (8 x 144 partial sums in LDS), then along x.  Same fp32 weights as the gather kernel;
// rows are summed before columns.
constexpr int UB_TY = 8, UB_TX = 64, UB_OR = 2 * UB_TY + 6, UB_OC = 2 * UB_TX + 16;  // window rows / columns (8-aligned start)

template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_tiled_kernel(T* __restrict__ gx, const T* __restrict__ gout, int h, int w,
                                                                     int oh, int ow, float sy, float sx, int tiles_x, int tiles_y,
                                                                     int planes_per_image, int64_t gout_image_stride) {
    __shared__ float tile[UB_OR][UB_OC + 1];
    __shared__ float tmp[UB_TY][UB_OC + 1];
    __shared__ float wrow[UB_TY][UB_OR];
    int b = blockIdx.x;
    const int tx_i = b % tiles_x; b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int64_t plane = b / tiles_y;
    const int x0 = tx_i * UB_TX, y0 = ty_i * UB_TY;
    const int oy0 = 2 * y0 - 3, ox0 = 2 * x0 - 8;  // first window row / column (ox0 is a multiple of 8)
    // (`gout_image_stride`: the gradient may be the leading channels of a wider tensor -- the concatenation's gradient)
    const T* g = gout + (plane / planes_per_image) * gout_image_stride + (plane % planes_per_image) * oh * (int64_t)ow;
    // window -> LDS (fp32), zero outside the image
    constexpr int GROUPS = UB_OC / 8;
    for (int e = threadIdx.x; e < UB_OR * GROUPS; e += 256) {
        const int r = e / GROUPS, cgrp = e % GROUPS;
        const int oy = oy0 + r, ox = ox0 + 8 * cgrp;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        if (oy >= 0 && oy < oh && ox >= 0 && ox + 8 <= ow) {
            const T* src = g + (int64_t)oy * ow + ox;
            if constexpr (sizeof(T) == 2) {
                const uint4 q = *reinterpret_cast<const uint4*>(src);
                T t[8];
                __builtin_memcpy(t, &q, 16);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = sis_ld(t, k);
            } else {
                const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tile[r][8 * cgrp + k] = v[k];
    }
    // row weights: wrow[ty][r] = weight of window row r for input row y0 + ty
    for (int e = threadIdx.x; e < UB_TY * UB_OR; e += 256) {
        const int ty = e / UB_OR, r = e % UB_OR;
        const int oy = oy0 + r, y = y0 + ty;
        float wgt = 0.f;
        if (oy >= 0 && oy < oh && y < h) {
            const Lerp1 ly = lerp1(oy, sy, h);
            if (ly.i0 == y) wgt += ly.l0;
            if (ly.i1 == y) wgt += ly.l1;
        }
        wrow[ty][r] = wgt;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < UB_TY * UB_OC; e += 256) {
        const int ty = e / UB_OC, c = e % UB_OC;
        // input row y0 + ty is touched by window rows 2 ty .. 2 ty + 5 (outputs 2y - 3 .. 2y + 2; all others have weight 0)
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += wrow[ty][2 * ty + k] * tile[2 * ty + k][c];
        tmp[ty][c] = acc;
    }
    __syncthreads();
    const int ty = threadIdx.x >> 5, txl = (threadIdx.x & 31) * 2;
    const int y = y0 + ty;
    if (y >= h) return;
    float out[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int x = x0 + txl + j;
        float acc = 0.f;
        if (x < w) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {  // outputs 2x - 3 .. 2x + 2
                const int ox = 2 * x - 3 + k;
                if (ox >= 0 && ox < ow) {
                    const Lerp1 lx = lerp1(ox, sx, w);
                    float wgt = 0.f;
                    if (lx.i0 == x) wgt += lx.l0;
                    if (lx.i1 == x) wgt += lx.l1;
                    acc += wgt * tmp[ty][ox - ox0];
                }
            }
        }
        out[j] = acc;
    }
    T* dst = gx + (plane * h + y) * (int64_t)w + x0 + txl;
    if (x0 + txl < w) sis_st(dst, 0, out[0]);
    if (x0 + txl + 1 < w) sis_st(dst, 1, out[1]);
}

// ---- x2 without LDS (round 4).  For an exact doubling PyTorch's index rule is static: with s = (w - 1) / (2w - 1),
// floor(s * 2j) = j - 1 (j >= 1) and floor(s * (2j + 1)) = j, so output columns 2j, 2j + 1 read sources (j - 1, j) and (j, j + 1)
// -- no gather.  A lane owns 4 source columns of one source row i and produces the 2 x 8 outputs they centre: it loads its 4
// values of rows i - 1, i, i + 1 (8 or 16 contiguous bytes each, fully coalesced over the wave), takes columns 4g - 1 and 4g + 4
// from its neighbours by a wave shuffle, interpolates each row horizontally, then the two output rows vertically -- the same
// fp32 expression, in the same association, as the kernels above (a weight pair (l0, l1) formed around the neighbouring index
// where the fp32 floor lands on the other side gives the same value to 1e-7: linear interpolation is continuous there).
// The tiled kernel ran the decoder's four stages at 0.7 / 1.3 / 2.0 / 2.7 TB/s (a workgroup per 16 x 512 outputs of ONE plane:
// 7 of 8 lanes idle at 64 columns); backward likewise a gather of <= 4 x 4 outputs per source pixel with static candidates.
template <typename T> struct Vec4Io;
template <> struct Vec4Io<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    static __device__ __forceinline__ void store4(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec4Io<__hip_bfloat16> {
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[4]) {
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        v[0] = __builtin_bit_cast(float, q.x << 16); v[1] = __builtin_bit_cast(float, q.x & 0xFFFF0000u);
        v[2] = __builtin_bit_cast(float, q.y << 16); v[3] = __builtin_bit_cast(float, q.y & 0xFFFF0000u);
    }
    static __device__ __forceinline__ void store4(__hip_bfloat16* p, const float* v) {
        __hip_bfloat16 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = __float2bfloat16(v[k]);
        uint2 q;
        __builtin_memcpy(&q, t, 8);
        *reinterpret_cast<uint2*>(p) = q;
    }
};
template <> struct Vec4Io<__half> {
    static __device__ __forceinline__ void load(const __half* p, float (&v)[4]) {
        __half t[4];
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        __builtin_memcpy(t, &q, 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __half2float(t[k]);
    }
    static __device__ __forceinline__ void store4(__half* p, const float* v) {
        __half t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = __float2half(v[k]);
        uint2 q;
        __builtin_memcpy(&q, t, 8);
        *reinterpret_cast<uint2*>(p) = q;
    }
};

// lanes_per_row = w / 4 divides 64 or is a multiple of it (host): a lane's left / right neighbour in the row is lane -/+ 1
template <typename T>
__global__ __launch_bounds__(256) void bilinear_up2_fwd_direct_kernel(T* __restrict__ out, const T* __restrict__ x, int h, int w,
                                                                      float sy, float sx, int64_t total, int planes_per_image,
                                                                      int64_t out_image_stride) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lpr = w >> 2;
    const int64_t tc = t < total ? t : total - 1;   // (the grid is rounded up: surplus lanes recompute the last item, store nothing)
    const int g = (int)(tc % lpr);
    const int64_t rowi = tc / lpr;
    const int i = (int)(rowi % h);
    const int64_t plane = rowi / h;
    const T* src = x + plane * h * (int64_t)w + 4 * g;
    const int wl = threadIdx.x & 63;
    const int rm = i > 0 ? i - 1 : 0, rp = i < h - 1 ? i + 1 : h - 1;
    float a[3][4];
    Vec4Io<T>::load(src + (int64_t)rm * w, a[0]);
    Vec4Io<T>::load(src + (int64_t)i * w, a[1]);
    Vec4Io<T>::load(src + (int64_t)rp * w, a[2]);
    // horizontal weights of the 8 output columns 8g .. 8g + 7 (j = 4g + m: even output 2j from (j - 1, j), odd from (j, j + 1))
    float e0[4], e1[4], o0[4], o1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int j = 4 * g + m;
        const float se = sx * (float)(2 * j), so = sx * (float)(2 * j + 1);
        e1[m] = se - (float)(j - 1); e0[m] = 1.f - e1[m];
        o1[m] = so - (float)j; o0[m] = 1.f - o1[m];
    }
    float hrow[3][8];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float left = __shfl_up(a[r][3], 1), right = __shfl_down(a[r][0], 1);
        const T* rowp = src + (int64_t)(r == 0 ? rm : r == 1 ? i : rp) * w;
        if (g == 0) left = a[r][0];            // column -1: weight e0 = 0 at j = 0; any finite value
        else if (wl == 0) left = sis_ld(rowp, -1);     // rows wider than a wave: the neighbour sits in another wave
        if (g == lpr - 1) right = a[r][3];     // column w: PyTorch clamps i1 to w - 1
        else if (wl == 63) right = sis_ld(rowp, 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float sm1 = m == 0 ? left : a[r][m - 1], sp1 = m == 3 ? right : a[r][m + 1];
            hrow[r][2 * m] = e0[m] * sm1 + e1[m] * a[r][m];
            hrow[r][2 * m + 1] = o0[m] * a[r][m] + o1[m] * sp1;
        }
    }
    if (t >= total) return;
    // vertical: output row 2i from source rows (i - 1, i), row 2i + 1 from (i, i + 1)
    const float ve1 = sy * (float)(2 * i) - (float)(i - 1), ve0 = 1.f - ve1;
    const float vo1 = sy * (float)(2 * i + 1) - (float)i, vo0 = 1.f - vo1;
    float r0[8], r1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        r0[k] = ve0 * hrow[0][k] + ve1 * hrow[1][k];
        r1[k] = vo0 * hrow[1][k] + vo1 * hrow[2][k];
    }
    const int64_t n = plane / planes_per_image, ch = plane - n * planes_per_image;
    const int ow = 2 * w;
    T* dst = out + n * out_image_stride + (ch * (int64_t)(2 * h) + 2 * i) * ow + 8 * g;
    Vec4Io<T>::store4(dst, r0); Vec4Io<T>::store4(dst + 4, r0 + 4);
    Vec4Io<T>::store4(dst + ow, r1); Vec4Io<T>::store4(dst + ow + 4, r1 + 4);
}

// Backward of the same: a lane owns 4 source columns of RP consecutive source rows (RP = 2 when h is even: the 2 RP + 2 output rows
// 2 i0 - 1 .. 2 i0 + 2 RP they gather from are loaded once for both -- 6 row loads per 2 source rows instead of 8); columns
// 8g - 1 .. 8g + 8.  Weights by the index rule itself (lerp1 + comparison, as bilinear_up_bwd_kernel): a candidate that does not
// reference the pixel gets weight 0, the clamped last column / row gets both of its weights.
template <typename T, int RP>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_direct_kernel(T* __restrict__ gx, const T* __restrict__ gout, int h, int w,
                                                                      float sy, float sx, int64_t total, int planes_per_image,
                                                                      int64_t gout_image_stride) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lpr = w >> 2, oh = 2 * h, ow = 2 * w, hp = h / RP;
    const int64_t tc = t < total ? t : total - 1;
    const int g = (int)(tc % lpr);
    const int64_t rowi = tc / lpr;
    const int i0 = (int)(rowi % hp) * RP;
    const int64_t plane = rowi / hp;
    const int64_t n = plane / planes_per_image, ch = plane - n * planes_per_image;
    const T* src = gout + n * gout_image_stride + ch * (int64_t)oh * ow + 8 * g;
    // column weights: wx[m][k] = weight of output column 8g - 1 + 2m + k (k = 0..3: 2j - 1 .. 2j + 2) on source column j = 4g + m
    float wx[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int j = 4 * g + m;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ox = 2 * j - 1 + k;
            float wgt = 0.f;
            if (ox >= 0 && ox < ow) {
                const Lerp1 lx = lerp1(ox, sx, w);
                if (lx.i0 == j) wgt += lx.l0;
                if (lx.i1 == j) wgt += lx.l1;
            }
            wx[m][k] = wgt;
        }
    }
    float acc[RP][4];
#pragma unroll
    for (int q = 0; q < RP; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[q][m] = 0.f;
#pragma unroll
    for (int r = 0; r < 2 * RP + 2; ++r) {
        const int oy = 2 * i0 - 1 + r;
        const bool in = oy >= 0 && oy < oh;
        const int oyc = in ? oy : (oy < 0 ? 0 : oh - 1);
        float wy[RP];
#pragma unroll
        for (int q = 0; q < RP; ++q) {   // source row i0 + q gathers output rows 2 (i0 + q) - 1 .. + 2 = r in [2q, 2q + 3]
            wy[q] = 0.f;
            if (in && r >= 2 * q && r <= 2 * q + 3) {
                const Lerp1 ly = lerp1(oy, sy, h);
                if (ly.i0 == i0 + q) wy[q] += ly.l0;
                if (ly.i1 == i0 + q) wy[q] += ly.l1;
            }
        }
        float v[8];
        {
            float lo[4], hi[4];
            Vec4Io<T>::load(src + (int64_t)oyc * ow, lo);
            Vec4Io<T>::load(src + (int64_t)oyc * ow + 4, hi);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = lo[k]; v[4 + k] = hi[k]; }
        }
        float left = __shfl_up(v[7], 1), right = __shfl_down(v[0], 1);
        if (g == 0) left = 0.f;               // output column -1 does not exist (its weight is 0 as well)
        else if ((threadIdx.x & 63) == 0) left = sis_ld(src + (int64_t)oyc * ow, -1);
        if (g == lpr - 1) right = 0.f;        // nor does column 2w
        else if ((threadIdx.x & 63) == 63) right = sis_ld(src + (int64_t)oyc * ow, 8);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // outputs 8g + 2m - 1 .. 8g + 2m + 2
            const float c0 = m == 0 ? left : v[2 * m - 1], c1 = v[2 * m], c2 = v[2 * m + 1], c3 = m == 3 ? right : v[2 * m + 2];
            const float hsum = ((wx[m][0] * c0 + wx[m][1] * c1) + wx[m][2] * c2) + wx[m][3] * c3;
#pragma unroll
            for (int q = 0; q < RP; ++q)
                if (r >= 2 * q && r <= 2 * q + 3) acc[q][m] += wy[q] * hsum;
        }
    }
    if (t >= total) return;
#pragma unroll
    for (int q = 0; q < RP; ++q) Vec4Io<T>::store4(gx + (plane * h + i0 + q) * (int64_t)w + 4 * g, acc[q]);
}

template <typename T>
int launch_up(void* out, const void* x, int64_t planes, int h, int w, int oh, int ow, int backward, int planes_per_image,
              int64_t image_stride, hipStream_t st) {
    // image_stride: elements between the images of the UPSAMPLED-size tensor (forward: out, backward: grad_out = x)
    const float sy = oh > 1 ? (float)(h - 1) / (float)(oh - 1) : 0.f;
    const float sx = ow > 1 ? (float)(w - 1) / (float)(ow - 1) : 0.f;
    const bool dense = image_stride == (int64_t)planes_per_image * oh * ow;
    const bool x2 = oh == 2 * h && ow == 2 * w;
    // the LDS-free x2 kernels: 4 source columns per lane, whole rows per wave (w / 4 divides 64 or is a multiple of it), every
    // vector access aligned (SIS_UP2_DIRECT=0: the tiled kernels, for A/B runs)
    static const bool direct_on = !(getenv("SIS_UP2_DIRECT") && getenv("SIS_UP2_DIRECT")[0] == '0');
    const int lpr = w / 4;
    const int align = 16 / (int)sizeof(T);   // elements per 16 bytes
    const bool direct = direct_on && x2 && h >= 2 && w % 4 == 0 && lpr > 0 && (64 % lpr == 0 || lpr % 64 == 0) && image_stride % align == 0 &&
                        (reinterpret_cast<uintptr_t>(out) % 16) == 0 && (reinterpret_cast<uintptr_t>(x) % 16) == 0 &&
                        ((int64_t)oh * ow) % align == 0 && planes * (int64_t)h * lpr < ((int64_t)1 << 38);
    if (direct) {
        const int64_t total = planes * (int64_t)h * lpr;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        if (!backward) {
            hipLaunchKernelGGL(bilinear_up2_fwd_direct_kernel<T>, dim3(blocks), dim3(256), 0, st, (T*)out, (const T*)x, h, w, sy, sx, total,
                               planes_per_image, image_stride);
            SIS_CHECK_LAUNCH("bilinear_up2_fwd_direct_kernel");
        } else if (h % 2 == 0) {
            const int64_t total2 = total / 2;
            hipLaunchKernelGGL((bilinear_up2_bwd_direct_kernel<T, 2>), dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, st, (T*)out,
                               (const T*)x, h, w, sy, sx, total2, planes_per_image, image_stride);
            SIS_CHECK_LAUNCH("bilinear_up2_bwd_direct_kernel");
        } else {
            hipLaunchKernelGGL((bilinear_up2_bwd_direct_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, (T*)out, (const T*)x, h, w, sy, sx, total,
                               planes_per_image, image_stride);
            SIS_CHECK_LAUNCH("bilinear_up2_bwd_direct_kernel");
        }
        return 0;
    }
    if (!backward) {
        const int tiles_x = sis_cdiv(ow, UF_OC), tiles_y = sis_cdiv(oh, UF_OR);
        if (x2 && h >= 2 && w >= 2 && planes * tiles_x * tiles_y < ((int64_t)1 << 31)) {
            hipLaunchKernelGGL(bilinear_up2_fwd_tiled_kernel<T>, dim3((unsigned)(planes * tiles_x * tiles_y)), dim3(256), 0, st,
                               (T*)out, (const T*)x, h, w, oh, ow, sy, sx, tiles_x, tiles_y, planes_per_image, image_stride);
            SIS_CHECK_LAUNCH("bilinear_up2_fwd_tiled_kernel");
            return 0;
        }
        SIS_REQUIRE(dense, "sis_upsample_bilinear: a strided result needs the x2 case");
        const int64_t total4 = planes * oh * ((ow + 3) >> 2);
        const int64_t blocks = (total4 + 255) / 256;
        hipLaunchKernelGGL(bilinear_up_fwd_kernel<T>, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                           (T*)out, (const T*)x, h, w, oh, ow, sy, sx, total4);
        SIS_CHECK_LAUNCH("bilinear_up_fwd_kernel");
    } else if (x2 && ow % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && image_stride % 8 == 0 &&
               planes * sis_cdiv(w, UB_TX) * sis_cdiv(h, UB_TY) < ((int64_t)1 << 31)) {
        const int tiles_x = sis_cdiv(w, UB_TX), tiles_y = sis_cdiv(h, UB_TY);
        hipLaunchKernelGGL(bilinear_up2_bwd_tiled_kernel<T>, dim3((unsigned)(planes * tiles_x * tiles_y)), dim3(256), 0, st,
                           (T*)out, (const T*)x, h, w, oh, ow, sy, sx, tiles_x, tiles_y, planes_per_image, image_stride);
        SIS_CHECK_LAUNCH("bilinear_up2_bwd_tiled_kernel");
    } else {  // out = grad_x [planes][h][w], x = grad_out [planes][oh][ow]
        SIS_REQUIRE(dense, "sis_upsample_bilinear: a strided gradient needs the aligned x2 case");
        const int64_t total = planes * h * w;
        hipLaunchKernelGGL(bilinear_up_bwd_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (T*)out,
                           (const T*)x, h, w, oh, ow, sy, sx, total);
        SIS_CHECK_LAUNCH("bilinear_up_bwd_kernel");
    }
    return 0;
}

}  // namespace

extern "C" int sis_upsample_bilinear_strided(void* out, const void* x, int dtype, int batch, int channels, int h, int w, int out_h,
                                             int out_w, int64_t image_stride, int backward, void* stream) {
    const int64_t planes = (int64_t)batch * channels;
    if (planes == 0) return 0;
    SIS_REQUIRE(out && x, "sis_upsample_bilinear: null pointer");
    SIS_REQUIRE(batch > 0 && channels > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0, "sis_upsample_bilinear: non-positive size");
    SIS_REQUIRE(planes * (int64_t)out_h * out_w < ((int64_t)1 << 40), "sis_upsample_bilinear: tensor too large");
    SIS_REQUIRE(image_stride >= (int64_t)channels * out_h * out_w, "sis_upsample_bilinear: image stride below the size of an image");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case SIS_F32: return launch_up<float>(out, x, planes, h, w, out_h, out_w, backward, channels, image_stride, st);
        case SIS_F16: return launch_up<__half>(out, x, planes, h, w, out_h, out_w, backward, channels, image_stride, st);
        case SIS_BF16: return launch_up<__hip_bfloat16>(out, x, planes, h, w, out_h, out_w, backward, channels, image_stride, st);
        default: return sis_fail("sis_upsample_bilinear: dtype code %d not supported (f32, f16, bf16)", dtype);
    }
}

extern "C" int sis_upsample_bilinear(void* out, const void* x, int dtype, int64_t planes, int h, int w, int out_h,
                                     int out_w, int backward, void* stream) {
    if (planes == 0) return 0;
    SIS_REQUIRE(planes > 0 && planes < ((int64_t)1 << 31), "sis_upsample_bilinear: plane count");
    return sis_upsample_bilinear_strided(out, x, dtype, 1, (int)planes, h, w, out_h, out_w, planes * (int64_t)out_h * out_w, backward, stream);
}
