// The root convolution of TransUNet's hybrid trunk on the matrix cores: 7 x 7, stride 2, padding 3, THREE input channels
// (the image) -> 64 channels (reference: networks/trans_u_net/vit_seg_modeling_resnet_skip.py:115-125, StdConv2d(3, width, 7, 2,
// padding=3) of ResNetV2.root).  conv_bf16.hip contracts over input channels in chunks of 16 and declines a 3-channel layer; the
// library ran it as NHWC implicit GEMMs between layout transposes (forward 67 us + weight gradient 77 us + 0.12 ms of transposes
// and casts per step at 512 x 512, B = 8 -- the last library kernels of the bf16 step).
//
// GEMM view.  K = (ci, ky, kx) with kx padded 7 -> 8: 21 groups g = 7 ci + ky of 8 taps, padded to 22 groups = 11 K-steps of
// v_mfma_f32_32x32x16_bf16 (lanes 0-31 hold group 2 s, lanes 32-63 group 2 s + 1).  At stride 2 the 8 taps of a group for
// output pixel ox are 8 CONSECUTIVE input pixels 2 ox - 3 .. 2 ox + 4 of row 2 oy - 3 + ky: the B fragment of the forward is
// five aligned dwords of the LDS-staged row funnel-shifted by 16 bits (the run starts on an odd element).
//   forward   Y[co][pixel] = sum_K Wp[co][K] * patch[K][pixel]           A = packed weights (global, 22 KB), B = patches
//   dW        dW[co][K]    = sum_pixels dY[co][pixel] * patch[pixel][K]  A = 8 consecutive pixels of a dY row (one 16-byte load),
//             B = for column K = (g, kx) the 8 input pixels 2 (ox0 + j) - 3 + kx, j = 0..7: every second element of a 16-element
//             run -- eight dwords, their low or high halves picked by v_perm_b32 according to the parity of kx.
// The image may be float32 (converted while it is staged: no cast launch, no bf16 copy kept for the backward) or bfloat16.
#include "sis_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 st_bf16x8;
typedef __attribute__((ext_vector_type(16))) float st_f32x16;
typedef unsigned short u16;

constexpr int ST_CO = 64, ST_G = 21, ST_STEPS = 11, ST_ROWS = 13, ST_N = 168;   // channels out, (ci, ky) groups, K-steps, staged rows per 4 output rows, dW columns
constexpr int ST_FW = 72;                                                       // staged columns of the forward's 32-pixel tile (2 * 32 + 8)

__device__ __forceinline__ u16 st_bf16(float v) { __hip_bfloat16 b = __float2bfloat16(v); return *reinterpret_cast<u16*>(&b); }
__device__ __forceinline__ u16 st_load(const float* p) { return st_bf16(*p); }
__device__ __forceinline__ u16 st_load(const u16* p) { return *p; }

// rows 2 oy0 - 3 .. 2 oy0 + 9, columns c0 .. c0 + tw - 1 of the three image planes of sample b -> tile[ci][row][tw] (bf16; zeros outside)
template <typename TX>
__device__ __forceinline__ void st_stage(u16* tile, const TX* x, int b, int H, int W, int oy0, int c0, int tw, int threads) {
    const int total = 3 * ST_ROWS * tw;
    for (int e = threadIdx.x; e < total; e += threads) {
        const int c = e % tw, rr = e / tw, r = rr % ST_ROWS, ci = rr / ST_ROWS;
        const int iy = 2 * oy0 - 3 + r, ix = c0 + c;
        u16 v = 0;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = st_load(x + (((int64_t)b * 3 + ci) * H + iy) * W + ix);
        tile[e] = v;
    }
}

// wpk[step][half][co][8]: taps kx = 0..6 (7: zero) of group g = 2 step + half (g = 21: zero) of output channel co
template <typename TW>
__global__ __launch_bounds__(256) void stem7_pack_kernel(u16* __restrict__ wpk, const TW* __restrict__ w) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ST_STEPS * 2 * ST_CO * 8) return;
    const int kx = i & 7, co = (i >> 3) & 63, g = i >> 9;
    u16 v = 0;
    if (g < ST_G && kx < 7) v = st_load(w + ((int64_t)co * ST_G + g) * 7 + kx);   // w [co][ci][ky][kx], g = 7 ci + ky
    wpk[i] = v;
}

// workgroup = 4 output rows x 32 output pixels x 64 channels; wave w = row oy0 + w
template <typename TX>
__global__ __launch_bounds__(256) void stem7_fwd_kernel(u16* __restrict__ y, const TX* __restrict__ x, const u16* __restrict__ wpk, int H, int W,
                                                        int Ho, int Wo) {
    __shared__ __attribute__((aligned(16))) u16 tile[3 * ST_ROWS * ST_FW];
    const int b = blockIdx.z, oy0 = blockIdx.y * 4, ox0 = blockIdx.x * 32;
    st_stage(tile, x, b, H, W, oy0, 2 * ox0 - 4, ST_FW, 256);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, kh = lane >> 5;
    const int oy = oy0 + wave;
    if (oy >= Ho) return;
    st_f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    const unsigned* t32 = reinterpret_cast<const unsigned*>(tile);
#pragma unroll
    for (int s = 0; s < ST_STEPS; ++s) {
        const int g = 2 * s + kh;
        uint4 frag = make_uint4(0u, 0u, 0u, 0u);
        if (g < ST_G) {
            const int ci = g / 7, ky = g - 7 * ci;
            const unsigned* p = t32 + ((ci * ST_ROWS + 2 * wave + ky) * ST_FW) / 2 + n;   // tile columns 2 n .. 2 n + 9; taps = columns 2 n + 1 .. 2 n + 8
            const unsigned d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
            frag = make_uint4(__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16), __builtin_amdgcn_alignbit(d3, d2, 16),
                              __builtin_amdgcn_alignbit(d4, d3, 16) & 0x0000FFFFu);   // (the 8th tap has a zero weight: its pixel must not reach the product as inf / NaN)
        }
        const st_bf16x8 bf = __builtin_bit_cast(st_bf16x8, frag);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const uint4 a = *reinterpret_cast<const uint4*>(wpk + ((int64_t)(2 * s + kh) * ST_CO + mt * 32 + n) * 8);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(st_bf16x8, a), bf, acc[mt], 0, 0, 0);
        }
    }
    const int ox = ox0 + n;
    if (ox >= Wo) return;
    const int64_t plane = (int64_t)Ho * Wo;
    u16* yb = y + (int64_t)b * ST_CO * plane + (int64_t)oy * Wo + ox;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            yb[co * plane] = st_bf16(acc[mt][i]);
        }
}

// workgroup = 4 output rows of one sample, all their pixels; wave w = M tile w & 1, N tiles (w >> 1) + {0, 2, 4}; partial dW
// [co][168] of the workgroup -> slab[workgroup]
template <typename TX>
__global__ __launch_bounds__(256) void stem7_wgrad_kernel(float* __restrict__ slab, const TX* __restrict__ x, const u16* __restrict__ gy, int H, int W,
                                                          int Ho, int Wo, int tw) {
    extern __shared__ __attribute__((aligned(16))) u16 wtile[];   // [3][13][tw]
    const int b = blockIdx.y, oy0 = blockIdx.x * 4;
    st_stage(wtile, x, b, H, W, oy0, -4, tw, 256);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nn = lane & 31, kh = lane >> 5;
    const int mt = wave & 1, ntb = wave >> 1;
    int rowbase[3], cbase[3];
    bool valid[3], odd[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (ntb + 2 * t) * 32 + nn, g = n >> 3, kx = n & 7;
        valid[t] = g < ST_G && kx < 7;
        const int gg = valid[t] ? g : 0, ci = gg / 7, ky = gg - 7 * ci;
        rowbase[t] = (ci * ST_ROWS + ky) * tw;
        odd[t] = ((1 + kx) & 1) != 0;          // tile column of tap kx for pixel ox: 2 ox + 1 + kx
        cbase[t] = (1 + kx) & ~1;
    }
    st_f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const unsigned* t32 = reinterpret_cast<const unsigned*>(wtile);
    const int64_t plane = (int64_t)Ho * Wo;
    const u16* gyc = gy + ((int64_t)b * ST_CO + mt * 32 + nn) * plane;
    const bool vec = (Wo & 7) == 0;
    for (int r = 0; r < 4; ++r) {
        const int oy = oy0 + r;
        if (oy >= Ho) break;
        const u16* grow = gyc + (int64_t)oy * Wo;
        for (int ox0 = 0; ox0 < Wo; ox0 += 16) {
            const int px = ox0 + 8 * kh;
            uint4 a = make_uint4(0u, 0u, 0u, 0u);
            if (vec) {
                if (px < Wo) a = *reinterpret_cast<const uint4*>(grow + px);
            } else {
                unsigned e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = px + j < Wo ? grow[px + j] : 0u;
                a = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            }
            const st_bf16x8 af = __builtin_bit_cast(st_bf16x8, a);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                uint4 frag = make_uint4(0u, 0u, 0u, 0u);
                if (valid[t]) {
                    const unsigned* p = t32 + (rowbase[t] + 2 * r * tw + 2 * px + cbase[t]) / 2;
                    unsigned d[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = p[j];
                    // every second element of the 16-element run: the high halves (odd start) or the low halves of the eight dwords
                    const unsigned sel = odd[t] ? 0x07060302u : 0x05040100u;
                    frag = make_uint4(__builtin_amdgcn_perm(d[1], d[0], sel), __builtin_amdgcn_perm(d[3], d[2], sel),
                                      __builtin_amdgcn_perm(d[5], d[4], sel), __builtin_amdgcn_perm(d[7], d[6], sel));
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(st_bf16x8, frag), acc[t], 0, 0, 0);
            }
        }
    }
    float* out = slab + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * ST_CO * ST_N;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (ntb + 2 * t) * 32 + nn;
        if (n >= ST_N) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            out[co * ST_N + n] = acc[t][i];
        }
    }
}

// dW[co][ci][ky][kx] = sum over the workgroups' partial tiles, in a fixed order: six lane groups add every sixth partial in order,
// lane group 0 adds the six sums in order.  One workgroup per output channel.
template <typename TW>
__global__ __launch_bounds__(1024) void stem7_wgrad_reduce_kernel(TW* __restrict__ dw, const float* __restrict__ slab, int parts) {
    __shared__ float sums[6][ST_N];
    const int co = blockIdx.x, n = threadIdx.x % ST_N, q = threadIdx.x / ST_N;
    if (q < 6) {
        float s = 0.f;
        for (int p = q; p < parts; p += 6) s += slab[((int64_t)p * ST_CO + co) * ST_N + n];
        sums[q][n] = s;
    }
    __syncthreads();
    if (q == 0) {
        const float s = ((((sums[0][n] + sums[1][n]) + sums[2][n]) + sums[3][n]) + sums[4][n]) + sums[5][n];
        const int g = n >> 3, kx = n & 7;
        if (kx < 7) sis_st(dw, ((int64_t)co * ST_G + g) * 7 + kx, s);
    }
}

bool stem_shape_ok(int cin, int cout, int ksize, int stride, int pad, int h, int w) {
    return cin == 3 && cout == ST_CO && ksize == 7 && stride == 2 && pad == 3 && h >= 7 && w >= 7 && h <= 4096 && w <= 1016;
}

}  // namespace

extern "C" int sis_stem_conv_supported(int cin, int cout, int ksize, int stride, int padding, int h, int w) {
    return stem_shape_ok(cin, cout, ksize, stride, padding, h, w) ? 1 : 0;
}

extern "C" int64_t sis_stem_conv_packed_elems(void) { return (int64_t)ST_STEPS * 2 * ST_CO * 8; }

extern "C" int sis_stem_conv_pack(void* packed, const void* weight, int weight_dtype, void* stream) {
    SIS_REQUIRE(packed && weight, "sis_stem_conv_pack: null pointer");
    SIS_REQUIRE(weight_dtype == SIS_F32 || weight_dtype == SIS_BF16, "sis_stem_conv_pack: weight must be float32 or bfloat16");
    const int total = ST_STEPS * 2 * ST_CO * 8;
    if (weight_dtype == SIS_F32)
        hipLaunchKernelGGL(stem7_pack_kernel<float>, dim3(sis_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (u16*)packed, (const float*)weight);
    else
        hipLaunchKernelGGL(stem7_pack_kernel<u16>, dim3(sis_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (u16*)packed, (const u16*)weight);
    SIS_CHECK_LAUNCH("stem7_pack_kernel");
    return 0;
}

extern "C" int sis_stem_conv_fwd(void* y, const void* x, int x_dtype, const void* packed, int batch, int h, int w, void* stream) {
    if (batch <= 0) return 0;
    SIS_REQUIRE(y && x && packed, "sis_stem_conv_fwd: null pointer");
    SIS_REQUIRE(x_dtype == SIS_F32 || x_dtype == SIS_BF16, "sis_stem_conv_fwd: the image must be float32 or bfloat16");
    SIS_REQUIRE(stem_shape_ok(3, ST_CO, 7, 2, 3, h, w), "sis_stem_conv_fwd: %d x %d image not supported", h, w);
    SIS_REQUIRE((((uintptr_t)packed) & 15) == 0, "sis_stem_conv_fwd: the packed weights must be 16-byte aligned");
    SIS_REQUIRE(batch <= 65535, "sis_stem_conv_fwd: more than 65 535 samples");
    const int ho = (h + 6 - 7) / 2 + 1, wo = (w + 6 - 7) / 2 + 1;
    const dim3 grid(sis_cdiv(wo, 32), sis_cdiv(ho, 4), batch);
    if (x_dtype == SIS_F32)
        hipLaunchKernelGGL(stem7_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (u16*)y, (const float*)x, (const u16*)packed, h, w, ho, wo);
    else
        hipLaunchKernelGGL(stem7_fwd_kernel<u16>, grid, dim3(256), 0, (hipStream_t)stream, (u16*)y, (const u16*)x, (const u16*)packed, h, w, ho, wo);
    SIS_CHECK_LAUNCH("stem7_fwd_kernel");
    sis_kernel_name = "stem7_fwd_kernel";
    return 0;
}

extern "C" int64_t sis_stem_conv_wgrad_workspace_bytes(int batch, int h, int w) {
    const int ho = (h + 6 - 7) / 2 + 1;
    return (int64_t)batch * sis_cdiv(ho, 4) * ST_CO * ST_N * 4;
}

extern "C" int sis_stem_conv_wgrad(void* dw, int dw_dtype, const void* x, int x_dtype, const void* grad_y, int batch, int h, int w,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    if (batch <= 0) return 0;
    SIS_REQUIRE(dw && x && grad_y && workspace, "sis_stem_conv_wgrad: null pointer");
    SIS_REQUIRE(dw_dtype == SIS_F32 || dw_dtype == SIS_BF16, "sis_stem_conv_wgrad: dW must be float32 or bfloat16");
    SIS_REQUIRE(x_dtype == SIS_F32 || x_dtype == SIS_BF16, "sis_stem_conv_wgrad: the image must be float32 or bfloat16");
    SIS_REQUIRE(stem_shape_ok(3, ST_CO, 7, 2, 3, h, w), "sis_stem_conv_wgrad: %d x %d image not supported", h, w);
    SIS_REQUIRE(workspace_bytes >= sis_stem_conv_wgrad_workspace_bytes(batch, h, w), "sis_stem_conv_wgrad: workspace too small");
    SIS_REQUIRE((((uintptr_t)grad_y) & 15) == 0, "sis_stem_conv_wgrad: dL/dy must be 16-byte aligned");
    SIS_REQUIRE(batch <= 65535, "sis_stem_conv_wgrad: more than 65 535 samples");
    const int ho = (h + 6 - 7) / 2 + 1, wo = (w + 6 - 7) / 2 + 1;
    const int tw = 2 * (sis_cdiv(wo, 16) * 16) + 8;                    // staged columns -4 .. 2 * ceil16(Wo) + 3 (even: rows stay dword aligned)
    const size_t lds = (size_t)3 * ST_ROWS * tw * sizeof(u16);
    SIS_REQUIRE(lds <= 160 * 1024, "sis_stem_conv_wgrad: image too wide");
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem7_wgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem7_wgrad_kernel<u16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return sis_fail("sis_stem_conv_wgrad: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const dim3 grid(sis_cdiv(ho, 4), batch);
    if (x_dtype == SIS_F32)
        hipLaunchKernelGGL(stem7_wgrad_kernel<float>, grid, dim3(256), lds, st, (float*)workspace, (const float*)x, (const u16*)grad_y, h, w, ho, wo, tw);
    else
        hipLaunchKernelGGL(stem7_wgrad_kernel<u16>, grid, dim3(256), lds, st, (float*)workspace, (const u16*)x, (const u16*)grad_y, h, w, ho, wo, tw);
    SIS_CHECK_LAUNCH("stem7_wgrad_kernel");
    const int parts = batch * sis_cdiv(ho, 4);
    if (dw_dtype == SIS_F32)
        hipLaunchKernelGGL(stem7_wgrad_reduce_kernel<float>, dim3(ST_CO), dim3(1024), 0, st, (float*)dw, (const float*)workspace, parts);
    else
        hipLaunchKernelGGL(stem7_wgrad_reduce_kernel<__hip_bfloat16>, dim3(ST_CO), dim3(1024), 0, st, (__hip_bfloat16*)dw, (const float*)workspace, parts);
    SIS_CHECK_LAUNCH("stem7_wgrad_reduce_kernel");
    sis_kernel_name = "stem7_wgrad_kernel";
    return 0;
}
