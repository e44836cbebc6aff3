// K2: upfirdn2d for gfx950 (+ the Blur/noise/bias/leaky-ReLU fusion the generator uses).
//
// Semantics: networks/stylegan2/op/upfirdn2d_kernel.cu:83-134 (polyphase walk, flipped taps),
// output size :167-168.  Two kernels:
//   * updn_direct_kernel<T>  -- every (up, down, pad, taps, minor) combination, one output per lane,
//     input read through L1/L2.  There is no "unsupported mode" (the reference returns garbage
//     outside its 6 templates, upfirdn2d_kernel.cu:172-175).
//   * blur_tile_kernel<FUSE> -- the generator's hot case (up = down = 1, taps <= 4x4, minor = 1,
//     fp32): a 32x64 output tile per 256-lane workgroup, input tile staged once in LDS with
//     coalesced row reads, each lane slides a 4x4 register window down 8 rows (4 LDS reads per
//     output instead of 16).  FUSE adds NoiseInjection + FusedLeakyReLU (model.py:338-340) so the
//     (2H+1)^2 intermediate is read exactly once and the activation written exactly once.
// Both are HBM-bound: 4*(in + out) bytes per plane.
#include "sis_common.h"

namespace {

// floor(a / b) for an up-sampling factor b >= 1 (C division truncates toward zero: shift negative numerators first)
__host__ __device__ __forceinline__ int floor_div(int a, int b) { return (a >= 0 ? a : a - b + 1) / b; }

struct UpdnParams {
    int major, in_h, in_w, minor, kh, kw;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
    int out_h, out_w;
};

template <typename T>
__global__ __launch_bounds__(256) void updn_direct_kernel(T* __restrict__ out, const T* __restrict__ in,
                                                          const T* __restrict__ taps, UpdnParams p, int64_t total) {
    typedef typename sis_acc<T>::type A;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int64_t r = i;
        const int mn = (int)(r % p.minor); r /= p.minor;
        const int ox = (int)(r % p.out_w); r /= p.out_w;
        const int oy = (int)(r % p.out_h); r /= p.out_h;
        const int64_t mj = r;
        const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
        const int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
        const int ix0 = floor_div(mid_x, p.up_x), iy0 = floor_div(mid_y, p.up_y);
        const int kx0 = (ix0 + 1) * p.up_x - mid_x - 1, ky0 = (iy0 + 1) * p.up_y - mid_y - 1;
        A v = (A)0;
        for (int fy = ky0, iy = iy0; fy < p.kh; fy += p.up_y, ++iy) {
            if (iy < 0 || iy >= p.in_h) continue;
            const T* row = in + ((mj * p.in_h + iy) * (int64_t)p.in_w) * p.minor + mn;
            const T* trow = taps + (p.kh - 1 - fy) * p.kw;
            for (int fx = kx0, ix = ix0; fx < p.kw; fx += p.up_x, ++ix) {
                if (ix < 0 || ix >= p.in_w) continue;
                v += sis_ld(row, (int64_t)ix * p.minor) * sis_ld(trow, p.kw - 1 - fx);
            }
        }
        sis_st(out, i, v);
    }
}

constexpr int BT_H = 32, BT_W = 64, BT_K = 4;
constexpr int BT_LH = BT_H + BT_K - 1, BT_LW = BT_W + BT_K - 1;

struct BlurParams {
    int planes, channels, in_h, in_w, out_h, out_w, kh, kw, pad_x0, pad_y0;
    int tiles_x, tiles_y;
    int64_t noise_bstride;
    float slope, ascale;
};

template <bool FUSE>
__global__ __launch_bounds__(256) void blur_tile_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                        const float* __restrict__ taps, const float* __restrict__ noise,
                                                        const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                        BlurParams p) {
    __shared__ float tile[BT_LH * BT_LW];
    __shared__ float kf[BT_K * BT_K];  // flipped taps, zero padded to 4x4
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx_i = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y; bid /= p.tiles_y;
    const int plane = bid;
    const int oy0 = ty_i * BT_H, ox0 = tx_i * BT_W;
    if (tid < BT_K * BT_K) {
        const int fy = tid / BT_K, fx = tid % BT_K;
        kf[tid] = (fy < p.kh && fx < p.kw) ? taps[(p.kh - 1 - fy) * p.kw + (p.kw - 1 - fx)] : 0.f;
    }
    const float* src = in + (int64_t)plane * p.in_h * p.in_w;
    const int iy_base = oy0 - p.pad_y0, ix_base = ox0 - p.pad_x0;
    for (int e = tid; e < BT_LH * BT_LW; e += 256) {
        const int r = e / BT_LW, c = e - r * BT_LW;
        const int iy = iy_base + r, ix = ix_base + c;
        float v = 0.f;
        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v = src[iy * p.in_w + ix];
        tile[e] = v;
    }
    __syncthreads();
    const int lx = tid & 63, lyg = tid >> 6;
    const int ox = ox0 + lx;
    float k[BT_K][BT_K];
#pragma unroll
    for (int a = 0; a < BT_K; ++a)
#pragma unroll
        for (int b = 0; b < BT_K; ++b) k[a][b] = kf[a * BT_K + b];
    float win[BT_K][BT_K];
    const int r0 = lyg * 8;
#pragma unroll
    for (int a = 0; a < BT_K - 1; ++a)
#pragma unroll
        for (int b = 0; b < BT_K; ++b) win[a + 1][b] = tile[(r0 + a) * BT_LW + lx + b];
    float nw = 0.f, bb = 0.f;
    const float* nz = nullptr;
    if (FUSE) {
        if (noise) {
            nw = noise_w[0];
            nz = noise + (int64_t)(plane / p.channels) * p.noise_bstride;
        }
        if (bias) bb = bias[plane % p.channels];
    }
    float* dst = out + (int64_t)plane * p.out_h * p.out_w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int a = 0; a < BT_K - 1; ++a)
#pragma unroll
            for (int b = 0; b < BT_K; ++b) win[a][b] = win[a + 1][b];
#pragma unroll
        for (int b = 0; b < BT_K; ++b) win[BT_K - 1][b] = tile[(r0 + j + BT_K - 1) * BT_LW + lx + b];
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < BT_K; ++a)
#pragma unroll
            for (int b = 0; b < BT_K; ++b) v += win[a][b] * k[a][b];
        const int oy = oy0 + r0 + j;
        if (oy < p.out_h && ox < p.out_w) {
            if (FUSE) {
                if (nz) v += nw * nz[oy * p.out_w + ox];
                v += bb;
                v = (v > 0.f ? v : v * p.slope) * p.ascale;
            }
            dst[oy * p.out_w + ox] = v;
        }
    }
}

// Row-streaming blur for the generator's hot shape: 4x4 taps, up = down = 1, output width a multiple of 4,
// input rows 16-byte aligned (row stride % 4 == 0; the transposed conv writes its (2H+1) x (2W+1) result
// into rows padded to 2W+4 floats for exactly this).  One lane = 4 consecutive output columns x 8 rows:
// per input row ONE aligned float4 load, the 3 halo values come from the neighbouring lanes by shuffle
// (explicit loads only at row / wave edges), per output row one float4 store.  No LDS, no index division
// in the loop.
struct BlurRowParams {
    int planes, channels, in_h, in_w, in_rs, out_h, out_w, pad_x0, pad_y0;
    int cols4, row_groups;
    int64_t noise_bstride, total;
    float slope, ascale;
};
constexpr int BR_ROWS = 8;

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Round 2: every load of the lane's 8 + 3 input rows (and of its 8 noise rows) is in flight before the first one is used.
// The first version waited after each row's float4 (the halo shuffles consume it at once, and the edge lanes' scalar loads sat
// behind branches): one 16-byte load in flight per lane, 32 KB per CU -- by Little's law about the 4.8 TB/s it measured.  Loads go
// through buffer descriptors so that a masked element (row outside the image, lane that takes its halo from a neighbour) is an
// out-of-range OFFSET, not a branch and not a select on loaded data.
template <bool FUSE>
__global__ __launch_bounds__(256) void blur_rows_kernel(float* __restrict__ out, const float* __restrict__ in,
                                                        const float* __restrict__ taps, const float* __restrict__ noise,
                                                        const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                        BlurRowParams p) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NR = BR_ROWS + 3;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = g < p.total;
    const int64_t gg = active ? g : p.total - 1;
    const int xi = (int)(gg % p.cols4);
    const int64_t r1 = gg / p.cols4;
    const int rg = (int)(r1 % p.row_groups);
    const int plane = (int)(r1 / p.row_groups);
    const int lane = threadIdx.x & 63;
    float k[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) k[a][b] = taps[(3 - a) * 4 + (3 - b)];  // flipped taps
    // The wave's descriptor starts at the plane of its first lane (a wave of a narrow map spans a few planes: small offsets).
    const int plane0 = __builtin_amdgcn_readfirstlane(plane);
    const int64_t plane_elems = (int64_t)p.in_h * p.in_rs;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(in + plane0 * plane_elems), 0, 0x7FFFFFFF, 0x00020000);
    const unsigned base = (unsigned)(((int64_t)(plane - plane0) * plane_elems + 4 * xi) * 4);  // bytes: row 0, column 4 xi
    const int oy0 = rg * BR_ROWS;
    const bool left_edge = xi == 0, right_edge = xi == p.cols4 - 1;
    // pad_x0 == 1: the lane's aligned float4 covers window columns 1..4; column 0 is the left neighbour's .w, columns 5..6 the right
    // neighbour's .x / .y -- by shuffle, except at wave edges (from memory) and image edges (zero / from memory up to in_w).
    const bool mem_l = lane == 0 && !left_edge, mem_r = lane == 63 || right_edge;
    f32x4_t ra[NR];
    float rl[NR];
    f32x2_t rr[NR];
#pragma unroll
    for (int t = 0; t < NR; ++t) {
        const int iy = oy0 - p.pad_y0 + t;
        const bool ok = iy >= 0 && iy < p.in_h;
        const unsigned ro = base + (unsigned)(iy * p.in_rs * 4);
        ra[t] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? ro : OOB, 0, 0));
        rl[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && mem_l) ? ro - 4u : OOB, 0, 0));
        rr[t] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (ok && mem_r) ? ro + 16u : OOB, 0, 0));
    }
    float nw = 0.f, bb = 0.f;
    f32x4_t nz[BR_ROWS];
    if (FUSE) {
        const bool has_noise = noise != nullptr;
        if (has_noise) nw = noise_w[0];
        if (bias) bb = bias[plane % p.channels];
        const int b0 = __builtin_amdgcn_readfirstlane(plane / p.channels);
        const __amdgpu_buffer_rsrc_t ns = __builtin_amdgcn_make_buffer_rsrc((void*)(noise + (has_noise ? b0 * p.noise_bstride : 0)), 0, 0x7FFFFFFF, 0x00020000);
        const unsigned nbase = (unsigned)(((int64_t)(plane / p.channels - b0) * p.noise_bstride + 4 * xi) * 4);
#pragma unroll
        for (int j = 0; j < BR_ROWS; ++j) {
            const int oy = oy0 + j;
            nz[j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ns, (has_noise && oy < p.out_h) ? nbase + (unsigned)(oy * p.out_w * 4) : OOB, 0, 0));
        }
    }
    const bool r0_ok = 4 * xi + 4 < p.in_w, r1_ok = 4 * xi + 5 < p.in_w;  // (only the image's right edge can fail these)
    float win[4][7];
    auto window_row = [&](int t, float* w) {
        const float sl = __shfl_up(ra[t].w, 1, 64), s0 = __shfl_down(ra[t].x, 1, 64), s1 = __shfl_down(ra[t].y, 1, 64);
        w[0] = left_edge ? 0.f : (lane == 0 ? rl[t] : sl);
        w[1] = ra[t].x; w[2] = ra[t].y; w[3] = ra[t].z; w[4] = ra[t].w;
        w[5] = mem_r ? (r0_ok ? rr[t].x : 0.f) : s0;
        w[6] = mem_r ? (r1_ok ? rr[t].y : 0.f) : s1;
    };
#pragma unroll
    for (int t = 0; t < 3; ++t) window_row(t, win[t + 1]);
    float* dst = out + (int64_t)plane * p.out_h * p.out_w;
#pragma unroll
    for (int j = 0; j < BR_ROWS; ++j) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < 7; ++c) win[t][c] = win[t + 1][c];
        window_row(j + 3, win[3]);
        float o[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            float v = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) v += win[a][x + b] * k[a][b];
            o[x] = v;
        }
        const int oy = oy0 + j;
        if (FUSE) {
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                float v = o[x] + nw * nz[j][x] + bb;
                o[x] = (v > 0.f ? v : v * p.slope) * p.ascale;
            }
        }
        if (active && oy < p.out_h) *reinterpret_cast<float4*>(dst + (int64_t)oy * p.out_w + 4 * xi) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

int launch_blur_rows(float* out, const float* in, const float* taps, const float* noise, int64_t nbs,
                     const float* noise_w, const float* bias, int planes, int channels, int in_h, int in_w, int in_rs,
                     int out_h, int out_w, int pad0, bool fuse, hipStream_t st) {
    BlurRowParams p;
    p.planes = planes; p.channels = channels; p.in_h = in_h; p.in_w = in_w; p.in_rs = in_rs; p.out_h = out_h;
    p.out_w = out_w; p.pad_x0 = pad0; p.pad_y0 = pad0;
    p.cols4 = out_w / 4; p.row_groups = sis_cdiv(out_h, BR_ROWS);
    p.noise_bstride = nbs; p.slope = 0.2f; p.ascale = 1.4142135623730951f;
    p.total = (int64_t)planes * p.row_groups * p.cols4;
    const int64_t blocks = (p.total + 255) / 256;
    SIS_REQUIRE(blocks > 0 && blocks < ((int64_t)1 << 31), "blur: grid too large");
    if (fuse)
        hipLaunchKernelGGL(blur_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, out, in, taps, noise, noise_w,
                           bias, p);
    else
        hipLaunchKernelGGL(blur_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, out, in, taps, noise,
                           noise_w, bias, p);
    SIS_CHECK_LAUNCH("blur_rows_kernel");
    return 0;
}

template <typename T>
int launch_direct(void* out, const void* in, const void* taps, const UpdnParams& p, hipStream_t st) {
    const int64_t total = (int64_t)p.major * p.out_h * p.out_w * p.minor;
    if (total == 0) return 0;
    const int64_t want = (total + 255) / 256;
    const int blocks = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(updn_direct_kernel<T>, dim3(blocks), dim3(256), 0, st, (T*)out, (const T*)in, (const T*)taps, p,
                       total);
    SIS_CHECK_LAUNCH("sis_upfirdn2d");
    return 0;
}

int launch_blur_tile(float* out, const float* in, const float* taps, const float* noise, int64_t nbs,
                     const float* noise_w, const float* bias, int planes, int channels, int in_h, int in_w, int out_h,
                     int out_w, int kh, int kw, int pad_x0, int pad_y0, bool fuse, hipStream_t st) {
    BlurParams p;
    p.planes = planes; p.channels = channels; p.in_h = in_h; p.in_w = in_w; p.out_h = out_h; p.out_w = out_w;
    p.kh = kh; p.kw = kw; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    p.tiles_x = sis_cdiv(out_w, BT_W); p.tiles_y = sis_cdiv(out_h, BT_H);
    p.noise_bstride = nbs; p.slope = 0.2f; p.ascale = 1.4142135623730951f;
    const int64_t blocks = (int64_t)planes * p.tiles_x * p.tiles_y;
    if (blocks == 0) return 0;
    SIS_REQUIRE(blocks < ((int64_t)1 << 31), "blur: grid too large");
    if (fuse)
        hipLaunchKernelGGL(blur_tile_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, out, in, taps, noise,
                           noise_w, bias, p);
    else
        hipLaunchKernelGGL(blur_tile_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, out, in, taps, noise,
                           noise_w, bias, p);
    SIS_CHECK_LAUNCH("blur_tile_kernel");
    return 0;
}

}  // namespace

extern "C" int sis_upfirdn2d_out_size(int in_size, int up, int down, int pad0, int pad1, int k) {
    return (in_size * up + pad0 + pad1 - k + down) / down;
}

extern "C" int sis_upfirdn2d(void* out, const void* in, const void* taps, int dtype, int major, int in_h, int in_w,
                             int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                             int pad_x1, int pad_y0, int pad_y1, void* stream) {
    SIS_REQUIRE(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "sis_upfirdn2d: up/down factors must be >= 1");
    SIS_REQUIRE(kh >= 1 && kw >= 1, "sis_upfirdn2d: empty tap matrix");
    SIS_REQUIRE(major >= 0 && in_h >= 0 && in_w >= 0 && minor >= 0, "sis_upfirdn2d: negative size");
    UpdnParams p;
    p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kh; p.kw = kw;
    p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
    p.out_h = sis_upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh);
    p.out_w = sis_upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw);
    SIS_REQUIRE(p.out_h >= 0 && p.out_w >= 0, "sis_upfirdn2d: negative output size %dx%d", p.out_h, p.out_w);
    if ((int64_t)major * p.out_h * p.out_w * minor == 0) return 0;
    SIS_REQUIRE(out && in && taps, "sis_upfirdn2d: null pointer");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case SIS_F32:
            // hot case of the generator (Blur after the up-convolution, model.py:262): LDS-tiled kernel
            if (up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && minor == 1 && kh <= BT_K && kw <= BT_K &&
                p.out_h * p.out_w >= 1024)
                return launch_blur_tile((float*)out, (const float*)in, (const float*)taps, nullptr, 0, nullptr, nullptr,
                                        major, 1, in_h, in_w, p.out_h, p.out_w, kh, kw, pad_x0, pad_y0, false, st);
            return launch_direct<float>(out, in, taps, p, st);
        case SIS_F64: return launch_direct<double>(out, in, taps, p, st);
        case SIS_F16: return launch_direct<__half>(out, in, taps, p, st);
        case SIS_BF16: return launch_direct<__hip_bfloat16>(out, in, taps, p, st);
        default: return sis_fail("sis_upfirdn2d: unsupported dtype code %d", dtype);
    }
}

extern "C" int sis_blur_noise_act(float* out, const float* in, const float* taps, const float* noise,
                                  int64_t noise_batch_stride, const float* noise_weight, const float* bias, int batch,
                                  int channels, int in_h, int in_w, int in_row_stride, int kh, int kw, int pad0, int pad1,
                                  int fuse_act, void* stream) {
    SIS_REQUIRE(kh >= 1 && kw >= 1 && kh <= BT_K && kw <= BT_K, "sis_blur_noise_act: taps must be at most %dx%d", BT_K, BT_K);
    const int out_h = in_h + pad0 + pad1 - kh + 1, out_w = in_w + pad0 + pad1 - kw + 1;
    SIS_REQUIRE(out_h >= 0 && out_w >= 0, "sis_blur_noise_act: negative output size");
    if ((int64_t)batch * channels * out_h * out_w == 0) return 0;
    SIS_REQUIRE(out && in && taps, "sis_blur_noise_act: null pointer");
    if (noise) SIS_REQUIRE(noise_weight, "sis_blur_noise_act: noise given without noise_weight");
    if (in_row_stride <= 0) in_row_stride = in_w;
    SIS_REQUIRE(in_row_stride >= in_w, "sis_blur_noise_act: row stride %d smaller than the row %d", in_row_stride, in_w);
    const bool aligned = (((uintptr_t)in | (uintptr_t)out | (uintptr_t)noise) & 15) == 0;
    if (kh == 4 && kw == 4 && pad0 == 1 && in_row_stride % 4 == 0 && out_w % 4 == 0 && out_w >= 8 && aligned &&
        (!noise || (noise_batch_stride % 4) == 0))
        return launch_blur_rows(out, in, taps, noise, noise_batch_stride, noise_weight, bias, batch * channels, channels,
                                in_h, in_w, in_row_stride, out_h, out_w, pad0, fuse_act != 0, (hipStream_t)stream);
    SIS_REQUIRE(in_row_stride == in_w, "sis_blur_noise_act: padded rows need 4x4 taps, pad0 = 1 and a 4-aligned width");
    return launch_blur_tile(out, in, taps, noise, noise_batch_stride, noise_weight, bias, batch * channels, channels,
                            in_h, in_w, out_h, out_w, kh, kw, pad0, pad0, fuse_act != 0, (hipStream_t)stream);
}
