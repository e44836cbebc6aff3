// Weight gradient of the stride-1 3x3 bf16 convolutions on the CDNA4 matrix cores, direct on NCHW tensors.
//
//   dW[co][ci][ky][kx] = sum_{n,y,x} dY[n][co][y][x] * X[n][ci][y + ky - 1][x + kx - 1]
//
// (reference call sites: the backward of every StdConv2d 3x3, vit_seg_modeling_resnet_skip.py:20-37, and of the decoder's
// Conv2dReLU, vit_seg_modeling.py:265-287 -- autograd's convolution_backward, which the library runs as NHWC
// implicit-GEMM kernels between layout transposes).
//
// GEMM view: M = co (A = dY), N = ci (B = X), K = pixels, one 32 x 32 fp32 accumulator tile PER TAP (9 of them, 144
// registers per wave) on v_mfma_f32_32x32x16_bf16.  K = pixels is the contiguous axis of both operands in NCHW, so both
// LDS images are plain copies of tensor rows -- no transposition anywhere:
//   * A fragment = 8 consecutive pixels of a dY row: one ds_read_b128 (row pitch padded to 9 units: conflict free);
//   * B fragment for tap (ky, kx) = 8 consecutive pixels of X row y + ky - 1 starting at x + kx - 1.  kx = 1 is the
//     aligned 16-byte group itself; kx = 0 / 2 straddle two groups and are assembled in registers with v_alignbit_b32
//     from the centre group and its left / right neighbour (5 funnel shifts per 16-pixel step serve both).  A lane reads
//     the 9 groups its four K-steps of a row touch once (9 ds_read_b128 per input row and 12 MFMAs).
// A workgroup walks down the rows of one 64-pixel (or 32-pixel) column strip of one image: per step it stages ONE new dY
// row and ONE new X row (ring of 4 X rows in LDS: rows y-1, y, y+1 in use, y+2 arriving), 36 (18) MFMAs per wave between
// barriers, loads issued before the MFMAs and written to LDS after them.
// Split-K over (image, strip, row block): every unit writes its partial [tap][co][ci] tile to a slab; a second kernel
// adds the slabs in unit order (deterministic, no atomics) and writes dW[co][ci][3][3].
#include <algorithm>
#include "sis_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short u16;

template <int WM_, int WN_, int KS_, int NWV_ = 8>
struct WgCfg {
    // waves along co / ci / K (the K waves split a row step's 16-pixel K-steps and write partial tiles of their own:
    // the narrow tail of the decoder, 16 or 3 output channels, has no other parallelism); K-steps per row step
    // NWV = waves per workgroup: 8, or 4 -- two workgroups per CU (the accumulators allow two waves per SIMD), one wave of each on
    // every SIMD, so that one's barrier and staging fall under the other's MFMAs
    static constexpr int WM = WM_, WN = WN_, NWV = NWV_, WK = NWV_ / (WM_ * WN_), KS = KS_, THREADS = 64 * NWV_;
    static_assert(WM * WN * WK == NWV && KS % WK == 0, "wave layout");
    static constexpr int MT = 32 * WM, NT = 32 * WN;         // co x ci tile of the workgroup
    static constexpr int SW = 16 * KS;                       // strip width in pixels
    static constexpr int DY_UNITS = SW / 8 + 1;              // 16-byte units per dY row (one pad unit)
    static constexpr int DY_ROW = DY_UNITS * 16;             // bytes
    static constexpr int DY_BUF = MT * DY_ROW;
    static constexpr int XR_UNITS = SW / 8 + 2 + 1;          // strip + one halo group each side + pad
    static constexpr int X_CI = (4 * XR_UNITS + 1) * 16;     // bytes per input channel: ring of 4 rows (+1 unit: odd pitch)
    static constexpr int X_BYTES = NT * X_CI;
    static constexpr int LDS = 2 * DY_BUF + X_BYTES;
    static constexpr int DY_CHUNKS = MT * (SW / 8);          // 16-byte loads per dY row
    static constexpr int X_CHUNKS = NT * (SW / 8 + 2);
    static constexpr int NDY = (DY_CHUNKS + THREADS - 1) / THREADS, NX = (X_CHUNKS + THREADS - 1) / THREADS;
    static_assert(LDS <= 160 * 1024, "LDS");
};

constexpr int WG_JOBS_MAX = 16;   // layers of one shape per launch (grid.z), operands through pointer tables

struct WgParams {
    const u16* x;    // [N, Cin, H, W]
    const u16* gy;   // [N, Cout, H, W]
    float* slab;     // [units][9][co_pad][ci_pad]
    int jobs;        // > 0: grid.z layers of this shape: x / gy from the tables, slab of layer z at slab + z * slab_job_stride
    long long slab_job_stride;
    const u16* xj[WG_JOBS_MAX]; const u16* gyj[WG_JOBS_MAX];
    int N, Cin, Cout, H, W;
    int strips, row_blocks, rows_per_block;
    int co_tiles, ci_tiles;
    int aligned;
};

__device__ __forceinline__ uint4 load_chunk(const u16* row, int x0, int W, bool row_ok, bool aligned) {
    // 8 pixels [x0, x0 + 8) of a tensor row (zeros outside [0, W) or when the row itself is outside the image)
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!row_ok) return v;
    if (aligned) {
        if (x0 >= 0 && x0 + 8 <= W) v = *reinterpret_cast<const uint4*>(row + x0);
        return v;  // W % 8 == 0: a group is inside or outside as a whole
    }
    if (x0 >= 0 && x0 + 8 <= W) {
        // whole group inside the row but on an arbitrary 2-byte boundary (127-wide maps): five dword loads from the address
        // rounded down to 4 bytes, funnel-shifted by 16 bits when it was odd
        const u16* g = row + x0;
        const bool odd = (reinterpret_cast<uintptr_t>(g) & 2) != 0;
        const unsigned* d = reinterpret_cast<const unsigned*>(g - (odd ? 1 : 0));
        const unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
        return odd ? make_uint4(__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                                __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(d4, d3, 16))
                   : make_uint4(d0, d1, d2, d3);
    }
    unsigned e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (x0 + j >= 0 && x0 + j < W) ? row[x0 + j] : 0u;
    return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
}

// the three kx taps of one input row and one 16-pixel K-step: centre group as is, its two neighbours funnel-shifted in
__device__ __forceinline__ void tap_row(f32x16 (&acc)[9], int ky, bf16x8 a, uint4 lft, uint4 c, uint4 rgt) {
    const unsigned t0 = __builtin_amdgcn_alignbit(c.x, lft.w, 16), t1 = __builtin_amdgcn_alignbit(c.y, c.x, 16);
    const unsigned t2 = __builtin_amdgcn_alignbit(c.z, c.y, 16), t3 = __builtin_amdgcn_alignbit(c.w, c.z, 16);
    const unsigned t4 = __builtin_amdgcn_alignbit(rgt.x, c.w, 16);
    const uint4 b0 = make_uint4(t0, t1, t2, t3), b2 = make_uint4(t1, t2, t3, t4);
    acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b0), acc[ky * 3 + 0], 0, 0, 0);
    acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, c), acc[ky * 3 + 1], 0, 0, 0);
    acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b2), acc[ky * 3 + 2], 0, 0, 0);
}

template <typename C, bool ALIGNED>
__global__ __launch_bounds__(C::THREADS, 2) void conv_wgrad_bf16_kernel(WgParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const dy_lds = lds;
    unsigned char* const x_lds = lds + 2 * C::DY_BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave % C::WM, wn = (wave / C::WM) % C::WN, wk = wave / (C::WM * C::WN);

    int u = blockIdx.x;
    const int rb = u % p.row_blocks; u /= p.row_blocks;
    const int strip = u % p.strips; u /= p.strips;
    const int n = u;
    const int co_t = blockIdx.y % p.co_tiles, ci_t = blockIdx.y / p.co_tiles;
    const int x0 = strip * C::SW;
    const int y_begin = rb * p.rows_per_block, y_end = min(p.H, y_begin + p.rows_per_block);
    const int64_t plane = (int64_t)p.H * p.W;
    const u16* gy_base = (p.jobs ? p.gyj[blockIdx.z] : p.gy) + ((int64_t)n * p.Cout + co_t * C::MT) * plane;
    const u16* x_base = (p.jobs ? p.xj[blockIdx.z] : p.x) + ((int64_t)n * p.Cin + ci_t * C::NT) * plane;

    uint4 dyr[C::NDY], xr[C::NX];

    auto load_dy = [&](int y) {  // dY row y of the tile's channels, strip columns
#pragma unroll
        for (int i = 0; i < C::NDY; ++i) {
            const int c = tid + i * C::THREADS;
            if (c < C::DY_CHUNKS) {
                const int ch = c / (C::SW / 8), g = c % (C::SW / 8);
                const bool ok = co_t * C::MT + ch < p.Cout;
                dyr[i] = load_chunk(gy_base + (int64_t)ch * plane + (int64_t)y * p.W, x0 + 8 * g, p.W, ok, ALIGNED);
            }
        }
    };
    auto store_dy = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::NDY; ++i) {
            const int c = tid + i * C::THREADS;
            if (c < C::DY_CHUNKS) {
                const int ch = c / (C::SW / 8), g = c % (C::SW / 8);
                *reinterpret_cast<uint4*>(dy_lds + buf * C::DY_BUF + ch * C::DY_ROW + g * 16) = dyr[i];
            }
        }
    };
    auto load_x = [&](int y) {  // X row y (may be -1 or H: zeros), strip columns plus one group each side
#pragma unroll
        for (int i = 0; i < C::NX; ++i) {
            const int c = tid + i * C::THREADS;
            if (c < C::X_CHUNKS) {
                const int ch = c / (C::SW / 8 + 2), g = c % (C::SW / 8 + 2);
                const bool ok = y >= 0 && y < p.H && ci_t * C::NT + ch < p.Cin;
                xr[i] = load_chunk(x_base + (int64_t)ch * plane + (int64_t)y * p.W, x0 + 8 * (g - 1), p.W, ok, ALIGNED);
            }
        }
    };
    auto store_x = [&](int y) {
        const int slot = (y + 1) & 3;
#pragma unroll
        for (int i = 0; i < C::NX; ++i) {
            const int c = tid + i * C::THREADS;
            if (c < C::X_CHUNKS) {
                const int ch = c / (C::SW / 8 + 2), g = c % (C::SW / 8 + 2);
                *reinterpret_cast<uint4*>(x_lds + ch * C::X_CI + (slot * C::XR_UNITS + g) * 16) = xr[i];
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // ---- prologue: X rows y_begin - 1, y_begin, y_begin + 1 and dY row y_begin
    load_x(y_begin - 1); store_x(y_begin - 1);
    load_x(y_begin); store_x(y_begin);
    load_x(y_begin + 1); store_x(y_begin + 1);
    load_dy(y_begin); store_dy(y_begin & 1);
    __syncthreads();

    const unsigned char* a_base = dy_lds + (wm * 32 + r) * C::DY_ROW + h * 16;
    const unsigned char* b_base = x_lds + (wn * 32 + r) * C::X_CI + h * 16;

    for (int y = y_begin; y < y_end; ++y) {
        const bool more = y + 1 < y_end;
        if (more) {
            load_dy(y + 1);
            load_x(y + 2);
        }
        constexpr int KW = C::KS / C::WK;  // K-steps of this wave: wk, wk + WK, ...
        bf16x8 a[KW];
#pragma unroll
        for (int i = 0; i < KW; ++i)
            a[i] = *reinterpret_cast<const bf16x8*>(a_base + (y & 1) * C::DY_BUF + (wk + i * C::WK) * 32);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int slot = (y + ky) & 3;  // image row y + ky - 1
            const unsigned char* row = b_base + slot * C::XR_UNITS * 16;
            if constexpr (C::WK == 1) {
                uint4 g[2 * C::KS + 1];      // groups h - 1 .. h + 2 KS - 1 of the strip (LDS group index + 1), each read once
#pragma unroll
                for (int i = 0; i < 2 * C::KS + 1; ++i) g[i] = *reinterpret_cast<const uint4*>(row + i * 16);
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks) tap_row(acc, ky, a[ks], g[2 * ks], g[2 * ks + 1], g[2 * ks + 2]);
            } else {
#pragma unroll
                for (int i = 0; i < KW; ++i) {
                    const int ks = wk + i * C::WK;
                    const uint4 lft = *reinterpret_cast<const uint4*>(row + (2 * ks) * 16);
                    const uint4 c = *reinterpret_cast<const uint4*>(row + (2 * ks + 1) * 16);
                    const uint4 rgt = *reinterpret_cast<const uint4*>(row + (2 * ks + 2) * 16);
                    tap_row(acc, ky, a[i], lft, c, rgt);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            store_dy((y + 1) & 1);
            store_x(y + 2);
        }
        __syncthreads();
    }

    // ---- partial tile -> slab[unit][tap][co][ci] (ci on the lanes: 128-byte runs)
    const int co_pad = p.co_tiles * C::MT, ci_pad = p.ci_tiles * C::NT;
    float* out = p.slab + (p.jobs ? (int64_t)blockIdx.z * p.slab_job_stride : 0) + (int64_t)blockIdx.x * 9 * co_pad * ci_pad;
    if constexpr (C::WK == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co_t * C::MT + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                const int ci = ci_t * C::NT + wn * 32 + r;
                out[((int64_t)t * co_pad + co) * ci_pad + ci] = acc[t][i];
            }
    } else {
        // the K waves of a (co, ci) block first add their tiles through the (now idle) staging LDS, tap by tap, in wave
        // order; one partial tile per workgroup reaches the slab
        constexpr int NP = C::WM * C::WN;
        static_assert(C::WK * NP * 1024 * 4 <= C::LDS, "reduction scratch");
        float* red = reinterpret_cast<float*>(lds);  // [WK][NP][16][64]
        const int pair = wm + C::WM * wn;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();  // everyone is out of the main loop / has read the previous tap
#pragma unroll
            for (int i = 0; i < 16; ++i) red[((wk * NP + pair) * 16 + i) * 64 + lane] = acc[t][i];
            __syncthreads();
            for (int e = tid; e < NP * 1024; e += C::THREADS) {
                const int pr = e >> 10, i = (e >> 6) & 15, ln = e & 63;
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < C::WK; ++k) sum += red[((k * NP + pr) * 16 + i) * 64 + ln];
                const int co = co_t * C::MT + (pr % C::WM) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);
                const int ci = ci_t * C::NT + (pr / C::WM) * 32 + (ln & 31);
                out[((int64_t)t * co_pad + co) * ci_pad + ci] = sum;
            }
        }
    }
}

// dW[co][ci][tap] = sum over units of slab[unit][tap][co][ci].  One workgroup per (tap, co, 64 input channels): the
// four waves add units w, w + 4, ... in order (coalesced 256-byte reads), then (s0 + s1) + (s2 + s3): a fixed order.
struct WgDwTab { void* dw[WG_JOBS_MAX]; };

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(T* __restrict__ dw, const float* __restrict__ slab, int units,
                                                                int Cout, int Cin, int co_pad, int ci_pad, int taps,
                                                                long long slab_job_stride = 0, WgDwTab tab = WgDwTab{}) {
    __shared__ float red[4][64];
    if (slab_job_stride) {   // grid.y layers of one shape
        dw = reinterpret_cast<T*>(tab.dw[blockIdx.y]);
        slab += (int64_t)blockIdx.y * slab_job_stride;
    }
    const int cchunks = (Cin + 63) / 64;
    int b = blockIdx.x;
    const int cc = b % cchunks; b /= cchunks;
    const int co = b % Cout, t = b / Cout;
    const int ln = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ci = cc * 64 + ln;
    float s = 0.f;
    if (ci < Cin) {
        const float* src = slab + ((int64_t)t * co_pad + co) * ci_pad + ci;
        const int64_t stride = (int64_t)taps * co_pad * ci_pad;
#pragma unroll 4
        for (int u = w; u < units; u += 4) s += src[u * stride];
    }
    red[w][ln] = s;
    __syncthreads();
    if (w == 0 && ci < Cin) sis_st(dw, ((int64_t)co * Cin + ci) * taps + t, (red[0][ln] + red[1][ln]) + (red[2][ln] + red[3][ln]));
}

// ---------------------------------------------------------------------------------------------------- 1 x 1 layers
// dW[co][ci] = sum_{n,p} dY[n][co][p] * X[n][ci][p]: a GEMM whose reduction axis (the pixels of a plane) is the contiguous
// axis of BOTH NCHW operands, so both LDS images are copies of plane segments and every fragment is one ds_read_b128 (8
// consecutive pixels of a channel).  (reference call sites: the backward of conv1x1 / StdConv2d 1x1 in every bottleneck,
// vit_seg_modeling_resnet_skip.py:30-37,40-75.)  These layers are memory-bound (co ci / (co + ci) FLOP per byte, 25-100),
// so the workgroup tile is as large as the layer allows -- 64 x 64 per wave = 2 x 2 MFMA blocks, WM x WN waves, the
// remaining 8 / (WM WN) waves split the 16-pixel K-steps of a stage -- and units = (image, pixel range) fill the chip;
// unit slabs are added in unit order by conv_wgrad_reduce_kernel (deterministic).  Planes of odd size (127 x 127: 2-byte
// aligned rows) go through load_chunk's funnel-shift path.
template <int WM_, int WN_, int KSW_>
struct PwCfg {
    static constexpr int WM = WM_, WN = WN_, WK = 8 / (WM_ * WN_), KSW = KSW_;   // KSW: K-steps per wave and stage
    static_assert(WM * WN * WK == 8, "wave layout");
    static constexpr int MT = 64 * WM, NT = 64 * WN;
    static constexpr int KP = 16 * WK * KSW;                 // pixels per stage
    static constexpr int PITCH = KP * 2 + 16;                // bytes per staged channel row (+1 unit: conflict-free b128 reads)
    static constexpr int A_BYTES = MT * PITCH, STAGE = (MT + NT) * PITCH, LDS = 2 * STAGE;
    static constexpr int A_CHUNKS = MT * (KP / 8), B_CHUNKS = NT * (KP / 8);
    static constexpr int NA = (A_CHUNKS + 511) / 512, NB = (B_CHUNKS + 511) / 512;
    static_assert(LDS <= 160 * 1024 && WK * WM * WN * 4096 <= LDS, "LDS");
};

struct PwParams {
    const u16* x; const u16* gy; float* slab;
    int jobs; long long slab_job_stride;   // as WgParams
    const u16* xj[WG_JOBS_MAX]; const u16* gyj[WG_JOBS_MAX];
    int N, Cin, Cout, P;              // P = pixels per plane
    int units_per_image, unit_len;    // unit u: image u / units_per_image, pixels [k * unit_len, min(P, (k + 1) * unit_len))
    int co_tiles, ci_tiles;
};

template <typename C, bool ALIGNED>
__global__ __launch_bounds__(512, 2) void conv1x1_wgrad_bf16_kernel(PwParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave % C::WM, wn = (wave / C::WM) % C::WN, wk = wave / (C::WM * C::WN);
    const int n = blockIdx.x / p.units_per_image, uk = blockIdx.x % p.units_per_image;
    const int co_t = blockIdx.y % p.co_tiles, ci_t = blockIdx.y / p.co_tiles;
    const int p_begin = uk * p.unit_len, p_end = min(p.P, p_begin + p.unit_len);
    const u16* gy_base = (p.jobs ? p.gyj[blockIdx.z] : p.gy) + ((int64_t)n * p.Cout + co_t * C::MT) * p.P;
    const u16* x_base = (p.jobs ? p.xj[blockIdx.z] : p.x) + ((int64_t)n * p.Cin + ci_t * C::NT) * p.P;
    const int stages = (p_end - p_begin + C::KP - 1) / C::KP;

    uint4 ar[C::NA], br[C::NB];
    auto load = [&](int s) {   // pixels beyond the unit's end read as zeros (p_end plays the row width)
        const int p0 = p_begin + s * C::KP;
#pragma unroll
        for (int i = 0; i < C::NA; ++i) {
            const int c = tid + i * 512;
            if (C::A_CHUNKS % 512 == 0 || c < C::A_CHUNKS) {
                const int ch = c / (C::KP / 8), g = c % (C::KP / 8);
                ar[i] = load_chunk(gy_base + (int64_t)ch * p.P, p0 + 8 * g, p_end, co_t * C::MT + ch < p.Cout, ALIGNED);
            }
        }
#pragma unroll
        for (int i = 0; i < C::NB; ++i) {
            const int c = tid + i * 512;
            if (C::B_CHUNKS % 512 == 0 || c < C::B_CHUNKS) {
                const int ch = c / (C::KP / 8), g = c % (C::KP / 8);
                br[i] = load_chunk(x_base + (int64_t)ch * p.P, p0 + 8 * g, p_end, ci_t * C::NT + ch < p.Cin, ALIGNED);
            }
        }
    };
    auto store = [&](int buf) {
        unsigned char* a = lds + buf * C::STAGE;
        unsigned char* b = a + C::A_BYTES;
#pragma unroll
        for (int i = 0; i < C::NA; ++i) {
            const int c = tid + i * 512;
            if (C::A_CHUNKS % 512 == 0 || c < C::A_CHUNKS) *reinterpret_cast<uint4*>(a + (c / (C::KP / 8)) * C::PITCH + (c % (C::KP / 8)) * 16) = ar[i];
        }
#pragma unroll
        for (int i = 0; i < C::NB; ++i) {
            const int c = tid + i * 512;
            if (C::B_CHUNKS % 512 == 0 || c < C::B_CHUNKS) *reinterpret_cast<uint4*>(b + (c / (C::KP / 8)) * C::PITCH + (c % (C::KP / 8)) * 16) = br[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    load(0); store(0);
    __syncthreads();
    const int a_off = (wm * 64 + r) * C::PITCH + h * 16, b_off = C::A_BYTES + (wn * 64 + r) * C::PITCH + h * 16;
    for (int s = 0; s < stages; ++s) {
        const bool more = s + 1 < stages;
        if (more) load(s + 1);
        const unsigned char* st = lds + (s & 1) * C::STAGE;
#pragma unroll
        for (int i = 0; i < C::KSW; ++i) {
            const int ks = wk + i * C::WK;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(st + a_off + ks * 32);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(st + a_off + 32 * C::PITCH + ks * 32);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(st + b_off + ks * 32);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(st + b_off + 32 * C::PITCH + ks * 32);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store((s + 1) & 1);
        __syncthreads();
    }

    // ---- the K waves add their tiles through the staging LDS in wave order, block by block; one tile per unit reaches the slab
    const int co_pad = p.co_tiles * C::MT, ci_pad = p.ci_tiles * C::NT;
    float* out = p.slab + (p.jobs ? (int64_t)blockIdx.z * p.slab_job_stride : 0) + (int64_t)blockIdx.x * co_pad * ci_pad;
    constexpr int NP = C::WM * C::WN;
    float* red = reinterpret_cast<float*>(lds);  // [WK][NP][16][64]
    const int pair = wm + C::WM * wn;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ba = t >> 1, bb = t & 1;
        if (t) __syncthreads();   // the previous block has been read
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((wk * NP + pair) * 16 + i) * 64 + lane] = acc[ba][bb][i];
        __syncthreads();
        for (int e = tid; e < NP * 1024; e += 512) {
            const int pr = e >> 10, i = (e >> 6) & 15, ln = e & 63;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < C::WK; ++k) sum += red[((k * NP + pr) * 16 + i) * 64 + ln];
            const int co = co_t * C::MT + (pr % C::WM) * 64 + ba * 32 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);
            const int ci = ci_t * C::NT + (pr / C::WM) * 64 + bb * 32 + (ln & 31);
            out[(int64_t)co * ci_pad + ci] = sum;
        }
    }
}

struct PwPlan { int wm, wn, units_per_image, unit_len, co_tiles, ci_tiles, units; int64_t slab_bytes; };

bool pw_plan(int batch, int cin, int cout, int pixels, int64_t workspace_bytes, PwPlan* pl, int jobs = 1) {
    workspace_bytes /= jobs;   // every layer of the launch has its own slabs
    if (cin < 8 || cout < 8 || pixels < 8 || batch < 1) return false;
    pl->wm = cout > 64 ? 2 : 1;
    pl->wn = cin > 64 ? 2 : 1;
    const int kp = pl->wm * pl->wn == 1 ? 128 : 64;
    const int mt = 64 * pl->wm, nt = 64 * pl->wn;
    pl->co_tiles = sis_cdiv(cout, mt); pl->ci_tiles = sis_cdiv(cin, nt);
    const int64_t tile_bytes = (int64_t)pl->co_tiles * mt * pl->ci_tiles * nt * 4;
    const int tiles = pl->co_tiles * pl->ci_tiles;
    // pixel ranges per image: enough workgroups for ~2 per CU, at least 4 stages each, slabs within the workspace
    int upi = 1;
    while ((int64_t)batch * upi * tiles * jobs < 512 && pixels / (upi * 2) >= 4 * kp && (int64_t)batch * upi * 2 * tile_bytes <= workspace_bytes) upi *= 2;
    pl->unit_len = sis_cdiv(sis_cdiv(pixels, upi), kp) * kp;
    pl->units_per_image = sis_cdiv(pixels, pl->unit_len);
    pl->units = batch * pl->units_per_image;
    pl->slab_bytes = (int64_t)pl->units * tile_bytes;
    return pl->slab_bytes <= workspace_bytes && (int64_t)batch * std::max(cin, cout) * pixels < (1LL << 31);
}

template <typename C>
int launch_pw(const PwParams& p, int units, bool aligned, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_wgrad_bf16_kernel<C, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_wgrad_bf16_kernel<C, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return sis_fail("conv1x1_wgrad_bf16_kernel: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(units, p.co_tiles * p.ci_tiles, p.jobs ? p.jobs : 1);
    if (aligned) hipLaunchKernelGGL((conv1x1_wgrad_bf16_kernel<C, true>), grid, dim3(512), C::LDS, st, p);
    else hipLaunchKernelGGL((conv1x1_wgrad_bf16_kernel<C, false>), grid, dim3(512), C::LDS, st, p);
    SIS_CHECK_LAUNCH("conv1x1_wgrad_bf16_kernel");
    sis_kernel_name = "conv1x1_wgrad_bf16_kernel";
    return 0;
}

struct WgPlan { int wm, wn, ks, nwv, strips, row_blocks, rows_per_block, co_tiles, ci_tiles, units, partials; int64_t slab_bytes; };

bool wgrad_plan(int batch, int cin, int cout, int h, int w, int64_t workspace_bytes, WgPlan* pl, int jobs = 1) {
    if (cin % 8 || cin < 16) return false;
    workspace_bytes /= jobs;   // every layer of the launch has its own slabs
    // SIS_WGRAD_WAVES: 8 / 4 force the 8-wave tiles / the 4-wave 64 x 64 tile (two workgroups per CU) wherever it applies; default:
    // the 4-wave tile only where the 8-wave one wastes half its 128 input channels (64 -> 64 layers on aligned maps: 114 -> 84 us
    // at 256 x 256; every wider layer measured 5-20 % SLOWER on 64 x 64 tiles -- the X rows are staged twice as often)
    static const int waves = getenv("SIS_WGRAD_WAVES") ? atoi(getenv("SIS_WGRAD_WAVES")) : 0;
    pl->nwv = 8;
    if (cout <= 32) {                 // narrow tail of the decoder: one co block, the spare waves split K
        pl->wm = 1;
        pl->wn = cin > 32 ? 2 : 1;
        pl->ks = cin > 32 ? 4 : 8;
        // wide maps: strips twice as wide -- the K waves of these one-block layers then hold two K-steps (18 MFMAs) per barrier
        // instead of one (SIS_WGRAD_NARROW_KS=0: the narrower strips)
        static const bool wide = !(getenv("SIS_WGRAD_NARROW_KS") && getenv("SIS_WGRAD_NARROW_KS")[0] == '0');
        if (wide && w >= 32 * pl->ks) pl->ks *= 2;
    } else {
        pl->wm = cout >= 128 ? 4 : 2;
        pl->wn = 8 / pl->wm;
        pl->ks = w <= 32 ? 2 : 4;
        if (waves == 4 || (waves == 0 && pl->wm == 2 && cin <= 64 && w % 8 == 0)) { pl->wm = 2; pl->wn = 2; pl->nwv = 4; }
    }
    const int sw = 16 * pl->ks;
    const int mt = 32 * pl->wm, nt = 32 * pl->wn;
    pl->co_tiles = sis_cdiv(cout, mt); pl->ci_tiles = sis_cdiv(cin, nt);
    pl->strips = sis_cdiv(w, sw);
    const int64_t tile_bytes = (int64_t)9 * pl->co_tiles * mt * pl->ci_tiles * nt * 4;
    const int tiles = pl->co_tiles * pl->ci_tiles;
    // row blocks: enough workgroups to fill the chip (~2 per CU), at least 8 rows each, slabs within the workspace
    int rbk = 1;
    // (the one-block layers of the decoder's tail are slab-bound: one round of workgroups, 127 / 71 / 67 us against 143 / 81 / 78 with two)
    static const int min_wg_env = getenv("SIS_WGRAD_MINWG") ? atoi(getenv("SIS_WGRAD_MINWG")) : 0;
    const int min_wg = min_wg_env ? min_wg_env : (cout <= 32 ? 256 : 512);
    while (rbk < h / 8 && (int64_t)batch * pl->strips * rbk * tiles * jobs < min_wg &&
           (int64_t)batch * pl->strips * (rbk * 2) * tile_bytes <= workspace_bytes) rbk *= 2;
    pl->rows_per_block = sis_cdiv(h, rbk);
    pl->row_blocks = sis_cdiv(h, pl->rows_per_block);
    pl->units = batch * pl->strips * pl->row_blocks;
    pl->partials = pl->units;
    pl->slab_bytes = (int64_t)pl->units * tile_bytes;
    return pl->slab_bytes <= workspace_bytes;
}

template <typename C>
int launch_wgrad(const WgParams& p, int units, hipStream_t st, const char* name) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_bf16_kernel<C, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_bf16_kernel<C, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return sis_fail("%s: cannot raise the LDS limit: %s", name, hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(units, p.co_tiles * p.ci_tiles, p.jobs ? p.jobs : 1);
    SIS_OCC_REPORT((conv_wgrad_bf16_kernel<C, true>), C::THREADS, C::LDS);
    if (p.aligned) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<C, true>), grid, dim3(C::THREADS), C::LDS, st, p);
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<C, false>), grid, dim3(C::THREADS), C::LDS, st, p);
    SIS_CHECK_LAUNCH(name);
    sis_kernel_name = name;
    return 0;
}

}  // namespace

extern "C" int sis_conv_bf16_wgrad_supported(int batch, int cin, int cout, int h, int w, int64_t workspace_bytes) {
    WgPlan pl;
    return wgrad_plan(batch, cin, cout, h, w, workspace_bytes, &pl) ? 1 : 0;
}

// n_jobs layers of ONE shape: one launch of the tile kernel (grid.z = layer) and one of the reduction (grid.y = layer).
static int conv_wgrad_jobs(void* const* dw, int dw_dtype, const void* const* x, const void* const* grad_y, int n_jobs, int batch, int cin,
                           int cout, int h, int w, void* workspace, int64_t workspace_bytes, void* stream, const char* who) {
    WgPlan pl;
    SIS_REQUIRE(wgrad_plan(batch, cin, cout, h, w, workspace_bytes, &pl, n_jobs),
                "%s: no tile plan for %d x (%d->%d @%dx%d) within %lld workspace bytes", who, n_jobs, cin, cout, h, w, (long long)workspace_bytes);
    WgParams p;
    p.x = (const u16*)x[0]; p.gy = (const u16*)grad_y[0]; p.slab = (float*)workspace;
    p.N = batch; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w;
    p.strips = pl.strips; p.row_blocks = pl.row_blocks; p.rows_per_block = pl.rows_per_block;
    p.co_tiles = pl.co_tiles; p.ci_tiles = pl.ci_tiles;
    p.jobs = n_jobs > 1 ? n_jobs : 0;
    p.slab_job_stride = pl.slab_bytes / 4;
    uintptr_t bits = 0;
    WgDwTab tab = {};
    for (int j = 0; j < n_jobs; ++j) {
        SIS_REQUIRE(dw[j] && x[j] && grad_y[j], "%s: null pointer in layer %d", who, j);
        p.xj[j] = (const u16*)x[j]; p.gyj[j] = (const u16*)grad_y[j]; tab.dw[j] = dw[j];
        bits |= (uintptr_t)x[j] | (uintptr_t)grad_y[j];
    }
    p.aligned = (w % 8 == 0) && (bits & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (pl.nwv == 4 && pl.ks == 4) rc = launch_wgrad<WgCfg<2, 2, 4, 4>>(p, pl.units, st, "conv_wgrad_bf16_kernel<2,2,4,4>");
    else if (pl.nwv == 4) rc = launch_wgrad<WgCfg<2, 2, 2, 4>>(p, pl.units, st, "conv_wgrad_bf16_kernel<2,2,2,4>");
    else if (pl.wm == 4 && pl.ks == 4) rc = launch_wgrad<WgCfg<4, 2, 4>>(p, pl.units, st, "conv_wgrad_bf16_kernel<4,2,4>");
    else if (pl.wm == 4 && pl.ks == 2) rc = launch_wgrad<WgCfg<4, 2, 2>>(p, pl.units, st, "conv_wgrad_bf16_kernel<4,2,2>");
    else if (pl.wm == 2 && pl.ks == 4) rc = launch_wgrad<WgCfg<2, 4, 4>>(p, pl.units, st, "conv_wgrad_bf16_kernel<2,4,4>");
    else if (pl.wm == 2) rc = launch_wgrad<WgCfg<2, 4, 2>>(p, pl.units, st, "conv_wgrad_bf16_kernel<2,4,2>");
    else if (pl.wn == 2 && pl.ks == 8) rc = launch_wgrad<WgCfg<1, 2, 8>>(p, pl.units, st, "conv_wgrad_bf16_kernel<1,2,8>");
    else if (pl.wn == 2) rc = launch_wgrad<WgCfg<1, 2, 4>>(p, pl.units, st, "conv_wgrad_bf16_kernel<1,2,4>");
    else if (pl.ks == 16) rc = launch_wgrad<WgCfg<1, 1, 16>>(p, pl.units, st, "conv_wgrad_bf16_kernel<1,1,16>");
    else rc = launch_wgrad<WgCfg<1, 1, 8>>(p, pl.units, st, "conv_wgrad_bf16_kernel<1,1,8>");
    if (rc) return rc;
    const int mt = 32 * pl.wm, nt = 32 * pl.wn;
    const int blocks = 9 * cout * sis_cdiv(cin, 64);
    const long long stride = p.jobs ? p.slab_job_stride : 0;
    if (dw_dtype == SIS_F32)
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel<float>, dim3(blocks, n_jobs), dim3(256), 0, st, (float*)dw[0], (const float*)workspace,
                           pl.partials, cout, cin, pl.co_tiles * mt, pl.ci_tiles * nt, 9, stride, tab);
    else
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel<__hip_bfloat16>, dim3(blocks, n_jobs), dim3(256), 0, st, (__hip_bfloat16*)dw[0],
                           (const float*)workspace, pl.partials, cout, cin, pl.co_tiles * mt, pl.ci_tiles * nt, 9, stride, tab);
    SIS_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
    return 0;
}

// layers per launch: at most WG_JOBS_MAX, and as many as leave every layer its slabs in the workspace
template <typename PlanFits>
static int jobs_per_launch(int n_jobs, PlanFits fits) {
    int n = n_jobs < WG_JOBS_MAX ? n_jobs : WG_JOBS_MAX;
    while (n > 1 && !fits(n)) n = (n + 1) / 2;
    return n;
}

extern "C" int sis_conv_bf16_wgrad(void* dw, int dw_dtype, const void* x, const void* grad_y, int batch, int cin, int cout,
                                   int h, int w, void* workspace, int64_t workspace_bytes, void* stream) {
    if (batch <= 0) return 0;
    SIS_REQUIRE(dw && x && grad_y && workspace, "sis_conv_bf16_wgrad: null pointer");
    SIS_REQUIRE(dw_dtype == SIS_F32 || dw_dtype == SIS_BF16, "sis_conv_bf16_wgrad: dW must be float32 or bfloat16");
    return conv_wgrad_jobs(&dw, dw_dtype, &x, &grad_y, 1, batch, cin, cout, h, w, workspace, workspace_bytes, stream, "sis_conv_bf16_wgrad");
}

/* The same for n_jobs layers of ONE shape (the trunk's repeated bottleneck units, queued during the backward): `dw`, `x`,
 * `grad_y` are HOST arrays of n_jobs device pointers.  One tile launch + one reduction launch per <= 16 layers; the tile plan
 * counts the layers' workgroups together, so every layer is cut into fewer, longer units than it would be alone. */
extern "C" int sis_conv_bf16_wgrad_multi(void* const* dw, int dw_dtype, const void* const* x, const void* const* grad_y, int n_jobs,
                                         int batch, int cin, int cout, int h, int w, void* workspace, int64_t workspace_bytes, void* stream) {
    if (batch <= 0 || n_jobs <= 0) return 0;
    SIS_REQUIRE(dw && x && grad_y && workspace, "sis_conv_bf16_wgrad_multi: null pointer");
    SIS_REQUIRE(dw_dtype == SIS_F32 || dw_dtype == SIS_BF16, "sis_conv_bf16_wgrad_multi: dW must be float32 or bfloat16");
    for (int j0 = 0; j0 < n_jobs;) {
        const int left = n_jobs - j0;
        const int n = jobs_per_launch(left, [&](int k) { WgPlan pl; return wgrad_plan(batch, cin, cout, h, w, workspace_bytes, &pl, k); });
        const int rc = conv_wgrad_jobs(dw + j0, dw_dtype, x + j0, grad_y + j0, n, batch, cin, cout, h, w, workspace, workspace_bytes, stream,
                                       "sis_conv_bf16_wgrad_multi");
        if (rc) return rc;
        j0 += n;
    }
    return 0;
}

/* 1x1 stride-1 layers: dW [Cout][Cin] (float32 or bfloat16) from bf16 NCHW x [B][Cin][pixels] and dL/dy [B][Cout][pixels]. */
extern "C" int sis_conv1x1_bf16_wgrad_supported(int batch, int cin, int cout, int pixels, int64_t workspace_bytes) {
    PwPlan pl;
    return pw_plan(batch, cin, cout, pixels, workspace_bytes, &pl) ? 1 : 0;
}

static int conv1x1_wgrad_jobs(void* const* dw, int dw_dtype, const void* const* x, const void* const* grad_y, int n_jobs, int batch, int cin,
                              int cout, int pixels, void* workspace, int64_t workspace_bytes, void* stream, const char* who) {
    PwPlan pl;
    SIS_REQUIRE(pw_plan(batch, cin, cout, pixels, workspace_bytes, &pl, n_jobs),
                "%s: no plan for %d x (%d->%d, %d pixels) within %lld workspace bytes", who, n_jobs, cin, cout, pixels, (long long)workspace_bytes);
    PwParams p;
    p.x = (const u16*)x[0]; p.gy = (const u16*)grad_y[0]; p.slab = (float*)workspace;
    p.N = batch; p.Cin = cin; p.Cout = cout; p.P = pixels;
    p.units_per_image = pl.units_per_image; p.unit_len = pl.unit_len; p.co_tiles = pl.co_tiles; p.ci_tiles = pl.ci_tiles;
    p.jobs = n_jobs > 1 ? n_jobs : 0;
    p.slab_job_stride = pl.slab_bytes / 4;
    uintptr_t bits = 0;
    WgDwTab tab = {};
    for (int j = 0; j < n_jobs; ++j) {
        SIS_REQUIRE(dw[j] && x[j] && grad_y[j], "%s: null pointer in layer %d", who, j);
        p.xj[j] = (const u16*)x[j]; p.gyj[j] = (const u16*)grad_y[j]; tab.dw[j] = dw[j];
        bits |= (uintptr_t)x[j] | (uintptr_t)grad_y[j];
    }
    const bool aligned = (pixels % 8 == 0) && (bits & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (pl.wm == 2 && pl.wn == 2) rc = launch_pw<PwCfg<2, 2, 2>>(p, pl.units, aligned, st);
    else if (pl.wm == 2) rc = launch_pw<PwCfg<2, 1, 1>>(p, pl.units, aligned, st);
    else if (pl.wn == 2) rc = launch_pw<PwCfg<1, 2, 1>>(p, pl.units, aligned, st);
    else rc = launch_pw<PwCfg<1, 1, 1>>(p, pl.units, aligned, st);
    if (rc) return rc;
    const int mt = 64 * pl.wm, nt = 64 * pl.wn;
    const int blocks = cout * sis_cdiv(cin, 64);
    const long long stride = p.jobs ? p.slab_job_stride : 0;
    if (dw_dtype == SIS_F32)
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel<float>, dim3(blocks, n_jobs), dim3(256), 0, st, (float*)dw[0], (const float*)workspace,
                           pl.units, cout, cin, pl.co_tiles * mt, pl.ci_tiles * nt, 1, stride, tab);
    else
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel<__hip_bfloat16>, dim3(blocks, n_jobs), dim3(256), 0, st, (__hip_bfloat16*)dw[0],
                           (const float*)workspace, pl.units, cout, cin, pl.co_tiles * mt, pl.ci_tiles * nt, 1, stride, tab);
    SIS_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
    return 0;
}

extern "C" int sis_conv1x1_bf16_wgrad(void* dw, int dw_dtype, const void* x, const void* grad_y, int batch, int cin, int cout,
                                      int pixels, void* workspace, int64_t workspace_bytes, void* stream) {
    if (batch <= 0) return 0;
    SIS_REQUIRE(dw && x && grad_y && workspace, "sis_conv1x1_bf16_wgrad: null pointer");
    SIS_REQUIRE(dw_dtype == SIS_F32 || dw_dtype == SIS_BF16, "sis_conv1x1_bf16_wgrad: dW must be float32 or bfloat16");
    return conv1x1_wgrad_jobs(&dw, dw_dtype, &x, &grad_y, 1, batch, cin, cout, pixels, workspace, workspace_bytes, stream, "sis_conv1x1_bf16_wgrad");
}

extern "C" int sis_conv1x1_bf16_wgrad_multi(void* const* dw, int dw_dtype, const void* const* x, const void* const* grad_y, int n_jobs,
                                            int batch, int cin, int cout, int pixels, void* workspace, int64_t workspace_bytes,
                                            void* stream) {
    if (batch <= 0 || n_jobs <= 0) return 0;
    SIS_REQUIRE(dw && x && grad_y && workspace, "sis_conv1x1_bf16_wgrad_multi: null pointer");
    SIS_REQUIRE(dw_dtype == SIS_F32 || dw_dtype == SIS_BF16, "sis_conv1x1_bf16_wgrad_multi: dW must be float32 or bfloat16");
    for (int j0 = 0; j0 < n_jobs;) {
        const int left = n_jobs - j0;
        const int n = jobs_per_launch(left, [&](int k) { PwPlan pl; return pw_plan(batch, cin, cout, pixels, workspace_bytes, &pl, k); });
        const int rc = conv1x1_wgrad_jobs(dw + j0, dw_dtype, x + j0, grad_y + j0, n, batch, cin, cout, pixels, workspace, workspace_bytes, stream,
                                          "sis_conv1x1_bf16_wgrad_multi");
        if (rc) return rc;
        j0 += n;
    }
    return 0;
}
