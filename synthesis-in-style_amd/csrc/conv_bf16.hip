// bf16 convolutions of the segmentation training step on the CDNA4 matrix cores, direct on NCHW tensors.
//
// Replaces, for TransUNet under bf16 autocast (BASELINE.json configs[4]), the library path
// NCHW -> NHWC transpose -> implicit-GEMM kernel -> NHWC -> NCHW transpose of every convolution
// (reference call sites: networks/trans_u_net/vit_seg_modeling_resnet_skip.py:20-37,40-75 StdConv2d 3x3 / 1x1,
// vit_seg_modeling.py:265-287 decoder Conv2dReLU, :324-329 segmentation head, :125-168 patch embedding).
//
// GEMM view per image:  D[co][pixel] = sum_{ci,tap} W[co][ci,tap] * X[ci][pixel + tap],  fp32 accumulation on
// v_mfma_f32_32x32x16_bf16 (A = weights, B = activations, 16 input channels per instruction and tap).
//
// The MFMA wants 8 consecutive K elements (= input channels) per lane, NCHW keeps a channel's pixels consecutive.
// The transposition happens once per staged element on its way into LDS, in registers:
//   * X tile: [c8][row][x][8 channels] -- one 16-byte unit per pixel and group of 8 channels.  A thread loads 4 pixels
//     of 8 channel rows (8 x 8 bytes), interleaves them with 16 v_perm_b32 and writes 4 units (ds_write_b128).  Any tap
//     shift is then a whole number of units: every B fragment is ONE ds_read_b128, consecutive lanes = consecutive pixels
//     (conflict free), all taps of a chunk addressed from one base register with immediate offsets.
//   * W tile: prepacked once per weight update by conv_pack_kernel into exactly the LDS image
//     [co tile][channel chunk][tap][row][unit ^ swizzle(row)][8 channels], copied linearly by LDS-DMA
//     (global_load_lds_dwordx4); the XOR swizzle makes the 16 rows of a ds_read_b128 lane group hit 16 different banks.
// Double-buffered stages, one barrier per chunk; X loads for chunk c+1 are issued before the MFMAs of chunk c and
// written to LDS after them (issue early / write late), the weight DMA flies underneath.
//
// The data gradient is the same kernel on adjoint-packed weights (channel roles swapped, taps rotated by 180 degrees);
// stride-2 layers read their B fragments at a pixel stride of two units.
#include <algorithm>
#include <type_traits>
#include "sis_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short u16;

template <int MT_, int NPIX_, int KC_, int KH_, int S_, int TW_>
struct ConvCfg {
    static constexpr int MT = MT_, NPIX = NPIX_, KC = KC_, KH = KH_, KW = KH_, S = S_, TW = TW_;
    static constexpr int PAD = KH / 2, TAPS = KH * KW;
    static constexpr int TR = NPIX / TW;                 // output rows per tile (NPIX output pixels per workgroup)
    // 8 waves as WM x WN over (MT / 32) x (NPIX / 32) MFMA blocks
    static constexpr int WM = MT == 128 ? (NPIX == 256 ? 2 : 4) : MT == 64 ? (NPIX == 256 ? 1 : 2) : 1;
    static constexpr int WN = 8 / WM;
    static constexpr int MB = MT / 32 / WM;              // 32-row MFMA blocks per wave along M
    static constexpr int NB = NPIX / 32 / WN;            // 32-pixel MFMA blocks per wave along N
    static_assert(MB >= 1 && NB >= 1 && MB * WM * 32 == MT && NB * WN * 32 == NPIX, "wave layout");
    static constexpr int HALO = PAD > 0 ? 8 : 0;         // left / right margin of the staged rows, in pixels (keeps 8-pixel alignment)
    static constexpr int RI = (TR - 1) * S + KH;         // staged input rows
    static constexpr int LW = TW * S + 2 * HALO;         // staged input pixels per row
    static constexpr int P = KC / 8;                     // channel planes per chunk = 16-byte units per weight row
    static constexpr int XBYTES = P * RI * LW * 16;
    static constexpr int WBYTES = TAPS * MT * KC * 2;
    static constexpr int STAGE = XBYTES + WBYTES;
    static constexpr int TASKS = P * RI * (LW / 4);      // staging tasks of 8 channels x 4 pixels
    static constexpr int NT = (TASKS + 511) / 512;
    static constexpr int WDMA = WBYTES / 1024;           // 1 KiB LDS-DMA pieces per chunk
    static_assert(WBYTES % 1024 == 0, "weight stage must be whole DMA pieces");
    static_assert(2 * STAGE <= 160 * 1024, "stages exceed the LDS");
};

__host__ __device__ constexpr int swz_shift(int units) { return units == 2 ? 3 : units == 4 ? 2 : 1; }

// unit index inside a weight row -> position in the LDS / packed image (see the header)
__host__ __device__ __forceinline__ int swz(int unit, int row, int units) {
    return unit ^ ((row >> swz_shift(units)) & (units - 1));
}

struct ConvParams {
    const u16* x;        // [N, Cin, H, W] bf16
    const u16* wp;       // packed weights
    const float* bias;   // [Cout] or null
    u16* y;              // [N, Cout, Ho, Wo] bf16
    int N, Cin, Cout, H, W, Ho, Wo;
    int tiles_x, tiles_y, co_tiles;
    int aligned;         // rows / planes / tile origins 8-byte aligned and no partially valid 4-pixel group
};

// ---------------------------------------------------------------------------------------------------- weight packing
// w: [Mrole... see sis_conv_bf16_pack.  One thread per packed element.
// (`out2` / `total2`: the adjoint packing of the same weight written by the same launch, MT2 = its M tile)
template <typename T>
__global__ __launch_bounds__(256) void conv_pack_kernel(u16* __restrict__ out, const T* __restrict__ w, int Cout, int Cin,
                                                        int KH, int MT, int KC, int adjoint, int64_t total,
                                                        u16* __restrict__ out2, int MT2, int64_t total2) {
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total + total2) return;
    if (e >= total) { e -= total; out = out2; MT = MT2; adjoint = 1; }
    const int taps = KH * KH, units = KC / 8;
    const int M = adjoint ? Cin : Cout, K = adjoint ? Cout : Cin;
    const int nchunks = K / KC;
    int64_t r = e;
    const int j = r % 8; r /= 8;
    const int upos = r % units; r /= units;
    const int row = r % MT; r /= MT;
    const int tap = r % taps; r /= taps;
    const int chunk = r % nchunks; r /= nchunks;
    const int mt = (int)r;
    const int unit = swz(upos, row, units);  // the swizzle is an involution on the unit index
    const int m = mt * MT + row, k = chunk * KC + unit * 8 + j;
    float v = 0.f;
    if (m < M) {
        const int ky = tap / KH, kx = tap % KH;
        if (!adjoint) v = (float)w[(((int64_t)m * Cin + k) * KH + ky) * KH + kx];
        else v = (float)w[(((int64_t)k * Cin + m) * KH + (KH - 1 - ky)) * KH + (KH - 1 - kx)];
    }
    __hip_bfloat16 b = __float2bfloat16(v);
    out[e] = *reinterpret_cast<u16*>(&b);
}

// Weight standardisation of ALL the StdConv2d layers of a network and the packing of the results in ONE launch
// (vit_seg_modeling_resnet_skip.py:20-27: every forward standardises every weight; 52 layers in R50-ViT-B/16 = 52
// weight_std launches + 55 conv_pack launches per step before this).  Table of 14 int64 per layer:
//   w (float32 [cout][cin][k][k]), w_hat (bf16, same shape), invstd (float32 [cout]), packed, adjoint (0 = none),
//   cout, cin, k, mt, kc, mt2, rows (= M tiles * mt: workgroups of the layer), row_begin (prefix sum of rows),
//   filter_begin (prefix sum of cout: the backward's workgroups).
// One workgroup per packed row: two-pass mean / variance of filter `co` in fp32 (as weight_std_fwd_kernel), then the bf16
// values go to w_hat (what the weight-gradient kernels and the backward read), to row `co` of the forward image and to
// column `co` of the adjoint image (taps rotated by 180 degrees).  Rows beyond cout (tile padding) are written as zeros.
constexpr int WSP_FIELDS = 14;

__global__ __launch_bounds__(256) void weight_std_pack_multi_kernel(const long long* __restrict__ table, int n_layers, float eps) {
    __shared__ float red[4];
    int layer = 0;
    while (layer + 1 < n_layers && (int)blockIdx.x >= (int)table[(layer + 1) * WSP_FIELDS + 12]) ++layer;
    const long long* d = table + (int64_t)layer * WSP_FIELDS;
    const float* w = reinterpret_cast<const float*>(d[0]);
    u16* what = reinterpret_cast<u16*>(d[1]);
    float* invstd = reinterpret_cast<float*>(d[2]);
    u16* packed = reinterpret_cast<u16*>(d[3]);
    u16* adj = reinterpret_cast<u16*>(d[4]);
    const int Cout = (int)d[5], Cin = (int)d[6], KH = (int)d[7], MT = (int)d[8], KC = (int)d[9], MT2 = (int)d[10];
    const int co = (int)blockIdx.x - (int)d[12];
    const int taps = KH * KH, n = Cin * taps, units = KC / 8;
    const int nchunks = Cin / KC;
    const int mt = co / MT, row = co % MT;
    auto fwd_pos = [&](int ci, int tap) {
        const int chunk = ci / KC, unit = (ci % KC) >> 3, j = ci & 7;
        return ((((int64_t)(mt * nchunks + chunk) * taps + tap) * MT + row) * units + swz(unit, row, units)) * 8 + j;
    };
    if (co >= Cout) {   // padding rows of the last M tile
        for (int i = threadIdx.x; i < n; i += 256) packed[fwd_pos(i / taps, i % taps)] = 0;
        return;
    }
    const float* src = w + (int64_t)co * n;
    const bool plain = what == nullptr;   // a layer without standardisation (the decoder's Conv2d): bf16 rounding + packing only
    float mean = 0.f, sd = 1.f;
    if (!plain) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += src[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
        __syncthreads();
        float m2 = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) { const float dd = src[i] - mean; m2 += dd * dd; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m2;
        __syncthreads();
        const float var = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
        sd = sqrtf(var + eps);
        if (threadIdx.x == 0) invstd[co] = 1.f / sd;
    }
    // adjoint image: M = Cin (row = ci), K = Cout (this filter is element k = co of every row)
    const int nchunks2 = MT2 ? Cout / KC : 0, chunk2 = co / KC, unit2 = (co % KC) >> 3, j2 = co & 7;
    for (int i = threadIdx.x; i < n; i += 256) {
        __hip_bfloat16 b = __float2bfloat16(plain ? src[i] : (src[i] - mean) / sd);
        const u16 v = *reinterpret_cast<u16*>(&b);
        if (!plain) what[(int64_t)co * n + i] = v;
        const int ci = i / taps, tap = i - ci * taps;
        packed[fwd_pos(ci, tap)] = v;
        if (MT2) {
            const int ky = tap / KH, kx = tap - ky * KH, tap2 = (KH - 1 - ky) * KH + (KH - 1 - kx);
            const int mt2 = ci / MT2, row2 = ci % MT2;
            adj[((((int64_t)(mt2 * nchunks2 + chunk2) * taps + tap2) * MT2 + row2) * units + swz(unit2, row2, units)) * 8 + j2] = v;
        }
    }
    if (MT2 && Cin % MT2) {   // padding rows of the adjoint image's last M tile: zeros in this filter's column
        const int pad0 = Cin, pad1 = (Cin + MT2 - 1) / MT2 * MT2;
        for (int i = threadIdx.x; i < (pad1 - pad0) * taps; i += 256) {
            const int ci = pad0 + i / taps, tap2 = i % taps;
            const int mt2 = ci / MT2, row2 = ci % MT2;
            adj[((((int64_t)(mt2 * nchunks2 + chunk2) * taps + tap2) * MT2 + row2) * units + swz(unit2, row2, units)) * 8 + j2] = 0;
        }
    }
}

// Backward of the standardisation for all layers in one launch: dw = invstd * (g - mean(g) - w_hat * mean(g * w_hat)) per filter
// (as weight_std_bwd_kernel, w_hat recomputed from the fp32 master weight).  The gradient / result pointers change from step
// to step, so they travel as kernel arguments (<= 64 layers per launch); everything static comes from the forward's table.
constexpr int WSB_MAX = 64;
struct WsBwdPtrs { const u16* g[WSB_MAX]; float* dw[WSB_MAX]; };

__global__ __launch_bounds__(256) void weight_std_bwd_multi_kernel(const long long* __restrict__ table, int layer0, int n_layers, WsBwdPtrs ptrs) {
    __shared__ float red[4];
    auto block_sum = [&](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    const int base = (int)table[(int64_t)layer0 * WSP_FIELDS + 13];
    int layer = layer0;
    while (layer + 1 < layer0 + n_layers && (int)blockIdx.x + base >= (int)table[(int64_t)(layer + 1) * WSP_FIELDS + 13]) ++layer;
    const long long* d = table + (int64_t)layer * WSP_FIELDS;
    const u16* g = ptrs.g[layer - layer0];
    float* dw = ptrs.dw[layer - layer0];
    if (!g) return;
    const int co = (int)blockIdx.x + base - (int)d[13];
    const int n = (int)d[6] * (int)d[7] * (int)d[7];
    const float* row = reinterpret_cast<const float*>(d[0]) + (int64_t)co * n;
    const u16* grow = g + (int64_t)co * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += row[i];
    const float mean = block_sum(s) / (float)n;
    const float is = reinterpret_cast<const float*>(d[2])[co];
    float sg = 0.f, sgw = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = __builtin_bit_cast(float, (unsigned)grow[i] << 16), wh = (row[i] - mean) * is;
        sg += gi; sgw += gi * wh;
    }
    const float mg = block_sum(sg) / (float)n;
    const float mgw = block_sum(sgw) / (float)n;
    float* o = dw + (int64_t)co * n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = __builtin_bit_cast(float, (unsigned)grow[i] << 16), wh = (row[i] - mean) * is;
        o[i] = is * (gi - mg - wh * mgw);
    }
}

// ---------------------------------------------------------------------------------------------------- the convolution
template <typename C, bool ALIGNED>
__global__ __launch_bounds__(512, 2) void conv_bf16_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave % C::WM, wn = wave / C::WM;

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (id % 8 = XCD group).  The output-channel
    // tiles that share an input tile take CONSECUTIVE slots of ONE group: the input tile is pulled into that XCD's L2 once
    // (with the channel tile as the slowest grid index every channel tile re-fetched it from beyond L2).
    const int slot = blockIdx.x >> 3;
    int t = (slot / p.co_tiles) * 8 + (blockIdx.x & 7);
    if (t >= p.N * p.tiles_y * p.tiles_x) return;
    const int co_t = slot % p.co_tiles;
    const int tx_i = t % p.tiles_x; t /= p.tiles_x;
    const int ty_i = t % p.tiles_y; t /= p.tiles_y;
    const int n = t;
    const int ox0 = tx_i * C::TW, oy0 = ty_i * C::TR;
    const int ix0 = ox0 * C::S - C::HALO, iy0 = oy0 * C::S - C::PAD;
    const int nchunks = p.Cin / C::KC;

    const u16* xin = p.x + (int64_t)n * p.Cin * p.H * p.W;
    const u16* wsrc = p.wp + (int64_t)co_t * nchunks * (C::WBYTES / 2);

    // ---- staging tasks of this thread (fixed over the chunk loop): 8 channels x 4 pixels each
    int task_goff[C::NT];   // element offset of (channel 0 of the chunk's plane, row, first pixel) inside the image, or -1
    int task_lds[C::NT];    // byte offset of the first unit inside a stage's X region
    int task_mask[C::NT];   // validity of the 4 pixels (bit e), for the unaligned / ragged path
#pragma unroll
    for (int i = 0; i < C::NT; ++i) {
        const int tk = tid + i * 512;
        task_goff[i] = -1; task_lds[i] = 0; task_mask[i] = 0;
        if (tk < C::TASKS) {
            const int q = tk % (C::LW / 4);
            const int ry = (tk / (C::LW / 4)) % C::RI;
            const int pl = tk / ((C::LW / 4) * C::RI);
            const int iy = iy0 + ry, ix = ix0 + 4 * q;
            task_lds[i] = ((pl * C::RI + ry) * C::LW + 4 * q) * 16;
            int mask = 0;
            if (iy >= 0 && iy < p.H)
#pragma unroll
                for (int e = 0; e < 4; ++e) mask |= (ix + e >= 0 && ix + e < p.W) ? (1 << e) : 0;
            task_mask[i] = mask;
            task_goff[i] = mask ? ((pl * 8) * p.H + iy) * p.W + ix : -1;  // fits 31 bits: one image's planes of a chunk
            if (!mask) task_goff[i] = -2;  // in range of the tile but outside the image: zeros
        }
    }

    uint2 xr[C::NT][8];  // staged pixels: [task][channel] = 4 bf16

    auto load_x = [&](int chunk) {
        const u16* base = xin + (int64_t)chunk * C::KC * p.H * p.W;
        const int plane = p.H * p.W;
#pragma unroll
        for (int i = 0; i < C::NT; ++i) {
            if (task_goff[i] >= 0) {
                const u16* g = base + task_goff[i];
                if constexpr (ALIGNED) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) xr[i][c] = *reinterpret_cast<const uint2*>(g + (int64_t)c * plane);
                } else {
                    const int mask = task_mask[i];
                    if (mask == 15) {
                        // odd planes / rows (127 x 127 maps): the 4 pixels start on any 2-byte boundary.  Three dword loads
                        // from the address rounded down to 4 bytes, funnel-shifted by 16 bits when it was odd.
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const u16* gc = g + (int64_t)c * plane;
                            const bool odd = (reinterpret_cast<uintptr_t>(gc) & 2) != 0;
                            const unsigned* d = reinterpret_cast<const unsigned*>(gc - (odd ? 1 : 0));
                            const unsigned d0 = d[0], d1 = d[1], d2 = d[2];
                            xr[i][c] = odd ? make_uint2(__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16))
                                           : make_uint2(d0, d1);
                        }
                    } else {  // ragged group at a row end: element by element
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const u16* gc = g + (int64_t)c * plane;
                            const unsigned e0 = (mask & 1) ? gc[0] : 0, e1 = (mask & 2) ? gc[1] : 0;
                            const unsigned e2 = (mask & 4) ? gc[2] : 0, e3 = (mask & 8) ? gc[3] : 0;
                            xr[i][c] = make_uint2(e0 | (e1 << 16), e2 | (e3 << 16));
                        }
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) xr[i][c] = make_uint2(0u, 0u);
            }
        }
    };

    auto store_x = [&](int stage) {
        unsigned char* xs = lds + stage * C::STAGE + C::WBYTES;
#pragma unroll
        for (int i = 0; i < C::NT; ++i) {
            if (task_goff[i] != -1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint4 u;
                    const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                    if (e < 2) {
                        u.x = __builtin_amdgcn_perm(xr[i][1].x, xr[i][0].x, sel);
                        u.y = __builtin_amdgcn_perm(xr[i][3].x, xr[i][2].x, sel);
                        u.z = __builtin_amdgcn_perm(xr[i][5].x, xr[i][4].x, sel);
                        u.w = __builtin_amdgcn_perm(xr[i][7].x, xr[i][6].x, sel);
                    } else {
                        u.x = __builtin_amdgcn_perm(xr[i][1].y, xr[i][0].y, sel);
                        u.y = __builtin_amdgcn_perm(xr[i][3].y, xr[i][2].y, sel);
                        u.z = __builtin_amdgcn_perm(xr[i][5].y, xr[i][4].y, sel);
                        u.w = __builtin_amdgcn_perm(xr[i][7].y, xr[i][6].y, sel);
                    }
                    *reinterpret_cast<uint4*>(xs + task_lds[i] + e * 16) = u;
                }
            }
        }
    };

    auto dma_w = [&](int chunk, int stage) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(wsrc) + (int64_t)chunk * C::WBYTES;
        unsigned char* dst = lds + stage * C::STAGE;
        for (int piece = wave; piece < C::WDMA; piece += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    };

    // ---- fragment addresses
    // A: row = wm * 32 * MB + mb * 32 + r (inside the MT tile), unit = ks * 2 + h (swizzled), tap stride = MT * KC * 2 bytes
    int a_off[C::MB][C::KC / 16];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
        const int row = (wm * C::MB + mb) * 32 + r;
#pragma unroll
        for (int ks = 0; ks < C::KC / 16; ++ks) a_off[mb][ks] = row * (C::KC * 2) + swz(ks * 2 + h, row, C::P) * 16;
    }
    // B: pixel block nb -> (tile row, first pixel); unit = plane (ks * 2 + h), row ty * S + ky, pixel (tx + r) * S + kx + HALO - PAD
    int b_off[C::NB];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) {
        const int blk = wn * C::NB + nb;
        const int ty = blk / (C::TW / 32), tx = (blk % (C::TW / 32)) * 32;
        b_off[nb] = C::WBYTES + (((h * C::RI) + ty * C::S) * C::LW + (tx + r) * C::S + C::HALO - C::PAD) * 16;
    }

    f32x16 acc[C::MB][C::NB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    // ---- prologue: chunk 0 into stage 0
    dma_w(0, 0);
    load_x(0);
    store_x(0);
    __syncthreads();  // (its fence waits for the LDS-DMA as well)

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        if (more) {
            dma_w(chunk + 1, cur ^ 1);
            load_x(chunk + 1);
        }
        const unsigned char* st = lds + cur * C::STAGE;
#pragma unroll
        for (int tap = 0; tap < C::TAPS; ++tap) {
            const int ky = tap / C::KW, kx = tap % C::KW;
#pragma unroll
            for (int ks = 0; ks < C::KC / 16; ++ks) {
                bf16x8 a[C::MB], b[C::NB];
#pragma unroll
                for (int mb = 0; mb < C::MB; ++mb)
                    a[mb] = *reinterpret_cast<const bf16x8*>(st + tap * (C::MT * C::KC * 2) + a_off[mb][ks]);
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb)
                    b[nb] = *reinterpret_cast<const bf16x8*>(st + b_off[nb] + (ks * 2 * C::RI * C::LW + ky * C::LW + kx) * 16);
#pragma unroll
                for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < C::NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb], b[nb], acc[mb][nb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the write-late half of the staging behind the MFMAs
        if (more) store_x(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, bf16, NCHW stores (lanes 0-31 of a register write 32 consecutive pixels of one channel row).  The
    // bias of a wave's rows is fetched once per 32-row block, a row's address is the block's base plus a multiple of the plane
    // stride, and full channel tiles (the usual case) carry no per-row bounds checks.
    const int plane = p.Ho * p.Wo;
    auto store_tile = [&](auto checked) {
#pragma unroll
        for (int mb = 0; mb < C::MB; ++mb) {
            const int co0 = co_t * C::MT + (wm * C::MB + mb) * 32 + 4 * h;  // row i = co0 + (i & 3) + 8 * (i >> 2)
            float bv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) bv[i] = 0.f;
            if (p.bias) {
#pragma unroll
                for (int i = 0; i < 16; ++i) bv[i] = p.bias[min(co0 + (i & 3) + 8 * (i >> 2), p.Cout - 1)];  // (rows >= Cout are not stored)
            }
#pragma unroll
            for (int nb = 0; nb < C::NB; ++nb) {
                const int blk = wn * C::NB + nb;
                const int oy = oy0 + blk / (C::TW / 32), ox = ox0 + (blk % (C::TW / 32)) * 32 + r;
                if (oy >= p.Ho || ox >= p.Wo) continue;
                u16* yb = p.y + ((int64_t)n * p.Cout + co0) * plane + oy * p.Wo + ox;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ro = (i & 3) + 8 * (i >> 2);
                    if (!decltype(checked)::value || co0 + ro < p.Cout) {
                        __hip_bfloat16 bvv = __float2bfloat16(acc[mb][nb][i] + bv[i]);
                        yb[ro * plane] = *reinterpret_cast<u16*>(&bvv);
                    }
                }
            }
        }
    };
    if ((co_t + 1) * C::MT <= p.Cout) store_tile(std::false_type());
    else store_tile(std::true_type());
}

template <typename C>
int launch_conv(const ConvParams& p, hipStream_t st, const char* name) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<C, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::STAGE);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<C, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * C::STAGE);
        if (e != hipSuccess) return sis_fail("%s: cannot raise the LDS limit: %s", name, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(8 * p.co_tiles * sis_cdiv((int64_t)p.N * p.tiles_y * p.tiles_x, 8));
    SIS_OCC_REPORT((conv_bf16_kernel<C, true>), 512, 2 * C::STAGE);
    if (p.aligned) hipLaunchKernelGGL((conv_bf16_kernel<C, true>), grid, dim3(512), 2 * C::STAGE, st, p);
    else hipLaunchKernelGGL((conv_bf16_kernel<C, false>), grid, dim3(512), 2 * C::STAGE, st, p);
    SIS_CHECK_LAUNCH(name);
    sis_kernel_name = name;
    return 0;
}

struct Plan { int mt, kc, tw, npix; };

// Tile plan of a layer: M tile by output channels, channel chunk by kernel size, tile width by image width, pixels per
// workgroup (256, 128 or 64) so that the grid fills the chip's 256 compute units when the layer is small.
bool conv_plan(int batch, int cin, int cout, int h, int w, int ksize, int stride, Plan* plan) {
    if (!((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2))) return false;
    const bool flat = ksize == 1 && stride == 1;      // pointwise: the image is one row of H*W pixels
    const int kc = flat ? 64 : 16;
    if (cin % kc) return false;
    int mt = cout > 64 ? 128 : cout > 32 ? 64 : 32;
    if (stride == 2) mt = cout > 64 ? 128 : 64;
    const int pad = ksize / 2;
    const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    int npix = 256;
    auto groups = [&](int np) {
        const int tw = flat ? np : (stride == 1 && wo <= 32 ? 32 : 64);
        const int64_t tiles = flat ? sis_cdiv((int64_t)ho * wo, np) : (int64_t)sis_cdiv(wo, tw) * sis_cdiv(ho, np / tw);
        return (int64_t)(batch > 0 ? batch : 1) * tiles * sis_cdiv(cout, mt);
    };
    if (stride == 1 && mt >= 64) {
        while (npix > (mt == 128 ? 64 : 128) && groups(npix) < 256) npix /= 2;
        if (!flat && mt == 128 && npix == 64 && wo > 32) npix = 128;  // 64-pixel tiles exist as 32 x 2 only
    }
    plan->mt = mt; plan->kc = kc; plan->npix = npix;
    plan->tw = flat ? npix : (stride == 1 && wo <= 32 ? 32 : 64);
    return true;
}

}  // namespace

extern "C" int sis_conv_bf16_supported(int cin, int cout, int h, int w, int ksize, int stride) {
    Plan pl;
    return conv_plan(1, cin, cout, h, w, ksize, stride, &pl) ? 1 : 0;
}

extern "C" int64_t sis_conv_bf16_packed_elems(int cin, int cout, int h, int w, int ksize, int stride, int adjoint) {
    const int M = adjoint ? cin : cout, K = adjoint ? cout : cin;
    Plan pl;
    if (!conv_plan(1, K, M, h, w, ksize, stride, &pl)) return -1;  // (the packing depends on mt and kc only)
    const int64_t mtiles = (M + pl.mt - 1) / pl.mt;
    return mtiles * pl.mt * (int64_t)K * ksize * ksize;
}

extern "C" int sis_conv_bf16_pack(void* packed, const void* weight, int weight_dtype, int cin, int cout, int h, int w,
                                  int ksize, int stride, int adjoint, void* stream) {
    SIS_REQUIRE(packed && weight, "sis_conv_bf16_pack: null pointer");
    const int M = adjoint ? cin : cout, K = adjoint ? cout : cin;
    Plan pl;
    SIS_REQUIRE(conv_plan(1, K, M, h, w, ksize, stride, &pl), "sis_conv_bf16_pack: unsupported layer %d->%d k%d s%d", cin, cout, ksize, stride);
    SIS_REQUIRE(!(adjoint && stride != 1), "sis_conv_bf16_pack: the adjoint packing is for stride-1 layers");
    const int64_t total = sis_conv_bf16_packed_elems(cin, cout, h, w, ksize, stride, adjoint);
    hipStream_t st = (hipStream_t)stream;
    const int blocks = sis_cdiv(total, 256);
    if (weight_dtype == SIS_F32)
        hipLaunchKernelGGL(conv_pack_kernel<float>, dim3(blocks), dim3(256), 0, st, (u16*)packed, (const float*)weight, cout, cin,
                           ksize, pl.mt, pl.kc, adjoint, total, (u16*)nullptr, 0, (int64_t)0);
    else if (weight_dtype == SIS_BF16)
        hipLaunchKernelGGL(conv_pack_kernel<__hip_bfloat16>, dim3(blocks), dim3(256), 0, st, (u16*)packed,
                           (const __hip_bfloat16*)weight, cout, cin, ksize, pl.mt, pl.kc, adjoint, total, (u16*)nullptr, 0, (int64_t)0);
    else
        return sis_fail("sis_conv_bf16_pack: weights must be float32 or bfloat16");
    SIS_CHECK_LAUNCH("conv_pack_kernel");
    return 0;
}

extern "C" int sis_conv_bf16_pack_both(void* packed, void* packed_adjoint, const void* weight, int weight_dtype, int cin, int cout,
                                       int h, int w, int ksize, void* stream) {
    SIS_REQUIRE(packed && packed_adjoint && weight, "sis_conv_bf16_pack_both: null pointer");
    Plan pf, pa;
    SIS_REQUIRE(conv_plan(1, cin, cout, h, w, ksize, 1, &pf) && conv_plan(1, cout, cin, h, w, ksize, 1, &pa) && pf.kc == pa.kc,
                "sis_conv_bf16_pack_both: unsupported stride-1 layer %d->%d k%d", cin, cout, ksize);
    const int64_t total = sis_conv_bf16_packed_elems(cin, cout, h, w, ksize, 1, 0);
    const int64_t total2 = sis_conv_bf16_packed_elems(cin, cout, h, w, ksize, 1, 1);
    hipStream_t st = (hipStream_t)stream;
    const int blocks = sis_cdiv(total + total2, 256);
    if (weight_dtype == SIS_F32)
        hipLaunchKernelGGL(conv_pack_kernel<float>, dim3(blocks), dim3(256), 0, st, (u16*)packed, (const float*)weight, cout, cin,
                           ksize, pf.mt, pf.kc, 0, total, (u16*)packed_adjoint, pa.mt, total2);
    else if (weight_dtype == SIS_BF16)
        hipLaunchKernelGGL(conv_pack_kernel<__hip_bfloat16>, dim3(blocks), dim3(256), 0, st, (u16*)packed,
                           (const __hip_bfloat16*)weight, cout, cin, ksize, pf.mt, pf.kc, 0, total, (u16*)packed_adjoint, pa.mt, total2);
    else
        return sis_fail("sis_conv_bf16_pack_both: weights must be float32 or bfloat16");
    SIS_CHECK_LAUNCH("conv_pack_kernel");
    return 0;
}

extern "C" int sis_conv_bf16(void* y, const void* x, const void* packed, const float* bias, int batch, int cin, int cout,
                             int h, int w, int ksize, int stride, void* stream) {
    if (batch <= 0) return 0;
    SIS_REQUIRE(y && x && packed, "sis_conv_bf16: null pointer");
    Plan pl;
    SIS_REQUIRE(conv_plan(batch, cin, cout, h, w, ksize, stride, &pl), "sis_conv_bf16: unsupported layer %d->%d k%d s%d", cin, cout, ksize, stride);
    const int pad = ksize / 2;
    ConvParams p;
    p.x = (const u16*)x; p.wp = (const u16*)packed; p.bias = bias; p.y = (u16*)y;
    p.N = batch; p.Cin = cin; p.Cout = cout;
    p.Ho = (h + 2 * pad - ksize) / stride + 1; p.Wo = (w + 2 * pad - ksize) / stride + 1;
    p.H = h; p.W = w;
    if (ksize == 1 && stride == 1) {  // pointwise: one row of H*W pixels
        p.H = 1; p.W = h * w; p.Ho = 1; p.Wo = h * w;
    }
    SIS_REQUIRE((int64_t)cin * h * w < (1LL << 31) && (int64_t)cout * p.Ho * p.Wo < (1LL << 31), "sis_conv_bf16: image planes exceed 2^31 elements");
    const int tr = pl.npix / pl.tw;
    p.tiles_x = sis_cdiv(p.Wo, pl.tw); p.tiles_y = sis_cdiv(p.Ho, tr);
    p.co_tiles = sis_cdiv(cout, pl.mt);
    p.aligned = (p.W % 4 == 0) && (((int64_t)p.H * p.W) % 4 == 0) && ((((uintptr_t)x) & 7) == 0);
    hipStream_t st = (hipStream_t)stream;
#define CONV_CASE(MT, NPIX, KC, KH, S, TW)                                                                      \
    if (pl.mt == MT && pl.npix == NPIX && pl.kc == KC && ksize == KH && stride == S && pl.tw == TW)            \
        return launch_conv<ConvCfg<MT, NPIX, KC, KH, S, TW>>(p, st, "conv_bf16_kernel<" #MT "," #NPIX "," #KC "," #KH "," #S "," #TW ">");
    CONV_CASE(128, 256, 16, 3, 1, 64) CONV_CASE(128, 256, 16, 3, 1, 32) CONV_CASE(128, 128, 16, 3, 1, 64)
    CONV_CASE(128, 128, 16, 3, 1, 32) CONV_CASE(128, 64, 16, 3, 1, 32)
    CONV_CASE(64, 256, 16, 3, 1, 64) CONV_CASE(64, 256, 16, 3, 1, 32) CONV_CASE(64, 128, 16, 3, 1, 64) CONV_CASE(64, 128, 16, 3, 1, 32)
    CONV_CASE(32, 256, 16, 3, 1, 64) CONV_CASE(32, 256, 16, 3, 1, 32)
    CONV_CASE(128, 256, 64, 1, 1, 256) CONV_CASE(128, 128, 64, 1, 1, 128) CONV_CASE(128, 64, 64, 1, 1, 64)
    CONV_CASE(64, 256, 64, 1, 1, 256) CONV_CASE(64, 128, 64, 1, 1, 128) CONV_CASE(32, 256, 64, 1, 1, 256)
    CONV_CASE(128, 256, 16, 3, 2, 64) CONV_CASE(64, 256, 16, 3, 2, 64) CONV_CASE(128, 256, 16, 1, 2, 64) CONV_CASE(64, 256, 16, 1, 2, 64)
#undef CONV_CASE
    return sis_fail("sis_conv_bf16: no kernel instance for tile plan mt=%d npix=%d kc=%d k=%d s=%d tw=%d", pl.mt, pl.npix, pl.kc, ksize, stride, pl.tw);
}

/* One launch for the standardised + packed weights of all StdConv2d layers (see weight_std_pack_multi_kernel).
 * sis_weight_std_pack_plan: the packing parameters of a layer (they depend on cout, cin, ksize, stride only): mt, kc, mt2 (0: no
 * adjoint image: stride 2, or cout not a multiple of kc) and the element counts of the two images; returns 0 when the layer
 * has no plan (its weight then goes through sis_weight_std_fwd / sis_conv_bf16_pack as before). */
extern "C" int sis_weight_std_pack_plan(int cin, int cout, int ksize, int stride, int* mt, int* kc, int* mt2, int64_t* packed_elems,
                                        int64_t* adjoint_elems) {
    Plan pf, pa;
    if (!conv_plan(1, cin, cout, 64, 64, ksize, stride, &pf)) return 0;
    *mt = pf.mt; *kc = pf.kc; *mt2 = 0; *adjoint_elems = 0;
    *packed_elems = (int64_t)sis_cdiv(cout, pf.mt) * pf.mt * cin * ksize * ksize;
    if (stride == 1 && conv_plan(1, cout, cin, 64, 64, ksize, 1, &pa) && pa.kc == pf.kc) {
        *mt2 = pa.mt;
        *adjoint_elems = (int64_t)sis_cdiv(cin, pa.mt) * pa.mt * cout * ksize * ksize;
    }
    return 1;
}

extern "C" int sis_weight_std_pack_multi(const void* table, int n_layers, int total_rows, float eps, void* stream) {
    if (n_layers <= 0 || total_rows <= 0) return 0;
    SIS_REQUIRE(table, "sis_weight_std_pack_multi: null table");
    hipLaunchKernelGGL(weight_std_pack_multi_kernel, dim3(total_rows), dim3(256), 0, (hipStream_t)stream, (const long long*)table, n_layers, eps);
    SIS_CHECK_LAUNCH("weight_std_pack_multi_kernel");
    return 0;
}

/* Backward of sis_weight_std_pack_multi's standardisation: grads[i] (bf16, layer i's dL/dw_hat, NULL = layer skipped) ->
 * dw[i] (float32, the layer's weight shape); `grads` / `dw` are HOST arrays of n_layers device pointers, `filters_total` =
 * sum of cout over the layers (= the table's last filter_begin + cout). */
extern "C" int sis_weight_std_bwd_multi(const void* table, const void* const* grads, void* const* dw, const int* couts, int n_layers,
                                        void* stream) {
    if (n_layers <= 0) return 0;
    SIS_REQUIRE(table && grads && dw && couts, "sis_weight_std_bwd_multi: null pointer");
    for (int l0 = 0; l0 < n_layers; l0 += WSB_MAX) {
        const int nl = std::min(WSB_MAX, n_layers - l0);
        WsBwdPtrs ptrs;
        int rows = 0;
        for (int i = 0; i < WSB_MAX; ++i) {
            ptrs.g[i] = i < nl ? (const u16*)grads[l0 + i] : nullptr;
            ptrs.dw[i] = i < nl ? (float*)dw[l0 + i] : nullptr;
            SIS_REQUIRE(i >= nl || !ptrs.g[i] || ptrs.dw[i], "sis_weight_std_bwd_multi: layer %d has a gradient but no result", l0 + i);
            if (i < nl) rows += couts[l0 + i];
        }
        hipLaunchKernelGGL(weight_std_bwd_multi_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const long long*)table, l0, nl, ptrs);
        SIS_CHECK_LAUNCH("weight_std_bwd_multi_kernel");
    }
    return 0;
}
