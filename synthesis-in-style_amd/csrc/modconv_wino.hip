// Stride-1 3x3 modulated convolution in Winograd F(2x2, 3x3) form on the fp32 matrix cores.
//
// The direct kernel (modconv_mfma2.hip) already keeps the MFMA pipe ~85 % busy at the clock the chip holds, so
// the remaining lever is the amount of MFMA work: F(2x2,3x3) produces a 2x2 output tile from 16 multiplies per
// (co, ci) instead of 36, i.e. 2.25x fewer matrix FLOPs for the 7 stride-1 layers (64.4 of the 90.2 GFLOP/image).
//
//   U[ci][xi][co] = (G g G^T)[xi]        prepacked once per checkpoint (sis_modconv_prepack_wino); stored as
//                                        [ci][q][ih][co][il][jj] with xi = 4 (2 ih + il) + 2 q + jj, so that the four
//                                        A operands a lane needs for two Winograd rows sit in ONE 16-byte LDS read
//   V[xi]         = (B^T d B)[xi]        per lane, in registers: the lane owning tile t (= MFMA column) and
//                                        channel 2cp + (lane>>5) reads its 4x4 input patch d from the raw LDS
//                                        tile (8 x ds_read_b64), scales it by the style s[b,ci] and does the
//                                        32 add/sub of the transform -- no second LDS image, no extra pass
//   M[xi]        += U[xi] (co x ci) * V[xi] (ci x tile)      16 independent MFMA chains (v_mfma_f32_32x32x2_f32)
//   Y             = A^T M A              epilogue, again lane-local (all 16 xi of a (co, tile) sit in one lane),
//                                        then demodulation, noise, bias, leaky-ReLU * sqrt(2), 8-byte stores
//
// Workgroup = 8 waves = 64 co x 64 tiles (256 pixels), see the kernel's header comment; same LDS-DMA double
// buffering and one barrier per 8-channel chunk as the direct kernel.  (A first version with 4 waves x 16 xi
// = 256 accumulator VGPRs, one wave per SIMD, reached only 46 % MFMA efficiency: nothing covered the patch
// reads and transform adds.)  Numerics: G has
// 1/2 entries (exact in binary), the transforms only add; measured error vs the fp64 oracle is within the same
// 2e-5 per-layer bound as the direct kernel (tests/test_generator_gpu.py).
#include "modconv_common.h"

// Timing ablations (WRONG results) exist for development builds only: `tools/build_variant.sh WORK <tag> -DSIS_ABLATIONS
// -DSIS_WINO_NOSTORE ...`.  Without -DSIS_ABLATIONS the switches are refused at compile time, so the shipped library cannot
// contain a wrong-result path.
#if !defined(SIS_ABLATIONS) && (defined(SIS_WINO_NODMA) || defined(SIS_WINO_NOTRANSFORM) || defined(SIS_WINO_NOBARRIER) || \
                                defined(SIS_WINO_NOSTORE) || defined(SIS_WINO_STAGGER))
#error "SIS_WINO_* ablation switches need -DSIS_ABLATIONS (development builds only)"
#endif

namespace {

constexpr int WCC = 8;      // input channels per chunk
#ifndef SIS_WINO_BAR_SLOT
#define SIS_WINO_BAR_SLOT 27
#endif
constexpr int S_BAR = SIS_WINO_BAR_SLOT;  // MFMA slot of a chunk behind which its barrier sits (32: at the chunk end, the form before r05)
constexpr int WMBLK = 64;   // output channels per workgroup
constexpr int WTILES = 64;  // 2x2 output tiles per workgroup (256 pixels)

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// 8-byte LDS read of two adjacent floats, typed as double on purpose: the compiler's wait-count pass puts
// s_waitcnt vmcnt(0) in front of every LDS read it thinks may alias an in-flight LDS-DMA write, and type-based alias
// info is what tells it otherwise -- float2 (a struct) aliases everything, which stalled every chunk on the DMA it
// had just issued; float and double reads do not.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 lds_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }  // ds_read_b128, typed like lds_ld2
__device__ __forceinline__ void lds_st4(float* p, f32x2 a, f32x2 b) { *reinterpret_cast<f32x4*>(p) = f32x4{a.x, a.y, b.x, b.y}; }
__device__ __forceinline__ float2 lds_ld2(const float* p) {
    const f32x2 d = *reinterpret_cast<const f32x2*>(p);
    return make_float2(d.x, d.y);
}
// LDS-DMA through a buffer descriptor: address = descriptor base + per-lane byte offset (VGPR) + wave-uniform byte
// offset (SGPR), so the per-chunk address arithmetic is scalar (no 64-bit VALU adds, which a SIMD partner's MFMA stream
// throttles to one instruction per ~10 cycles), and a lane whose offset is >= num_records gets ZEROS written to its
// LDS slot (tools/micro/buffer_lds_oob.hip): image halos and partial blocks need neither exec masks nor a zero fill.
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ void bufld16(__amdgpu_buffer_rsrc_t r, float* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
// Packed fp32 VALU (two lanes of work per instruction), written as assembly because the compiler splits v2f32 arithmetic
// back into scalar instructions.  np = (-1, 1).
__device__ __forceinline__ f32x2 pk_lo_np_plus(f32x2 y, f32x2 np, f32x2 x) {  // (x.lo - y.lo, x.hi + y.lo)
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(y), "v"(np), "v"(x));
    return r;
}
__device__ __forceinline__ f32x2 pk_hi_np_cross(f32x2 x, f32x2 np, f32x2 y) {  // (y.lo - x.hi, x.hi - y.hi)
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(np), "v"(y));
    return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_neg_sub(f32x2 a, f32x2 b) {  // -a - b
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[1,1] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_scale(f32x2 a, f32x2 s_lo) {  // a * s_lo.lo
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(s_lo));
    return r;
}
__device__ __forceinline__ void glds4(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

// u[ci][q][ih][co][il][jj] (xi = 4 (2 ih + il) + 2 q + jj) from w[co][ci][3][3]
// ADJOINT = true builds the weights of the data gradient instead: the convolution that maps dL/dy back to dL/dx
// uses w'[ci][co][r][c] = w[co][ci][2-r][2-c] (roles of the channel axes swapped, taps rotated by 180 degrees);
// `cout` / `cin` are then the output / input channels of THAT convolution (= cin / cout of w's own layout).
template <bool ADJOINT>
__global__ __launch_bounds__(256) void wino_prepack_kernel(float* __restrict__ u, const float* __restrict__ w, int cout,
                                                           int cin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // i = ci * cout + co
    if (i >= (int64_t)cout * cin) return;
    const int co = (int)(i % cout), ci = (int)(i / cout);
    const float* gsrc = ADJOINT ? w + ((int64_t)ci * cout + co) * 9 : w + ((int64_t)co * cin + ci) * 9;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = gsrc[ADJOINT ? 8 - k : k];
    float t[4][3];  // G g
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
        t[0][c] = g0;
        t[1][c] = 0.5f * (g0 + g1 + g2);
        t[2][c] = 0.5f * (g0 - g1 + g2);
        t[3][c] = g2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = t[r][0], b = t[r][1], c = t[r][2];
        const float o[4] = {a, 0.5f * (a + b + c), 0.5f * (a - b + c), c};
#pragma unroll
        for (int j = 0; j < 4; ++j)  // xi = 4 r + j;  r = 2 ih + il,  j = 2 q + jj  ->  [ci][q][ih][co][il][jj]
            u[(((int64_t)ci * 2 + (j >> 1)) * 2 + (r >> 1)) * cout * 4 + (int64_t)co * 4 + (r & 1) * 2 + (j & 1)] = o[j];
    }
}

// Forward AND adjoint images of a training step's weight from one launch (blockIdx.y: 0 forward, 1 adjoint): the backward of the
// same step needs the adjoint anyway (19 launches per EMANet step saved).
__global__ __launch_bounds__(256) void wino_prepack_both_kernel(float* __restrict__ u, float* __restrict__ u_adj,
                                                                const float* __restrict__ w, int cout_w, int cin_w) {
    const bool adjoint = blockIdx.y == 1;
    const int cout = adjoint ? cin_w : cout_w, cin = adjoint ? cout_w : cin_w;   // roles in the convolution this image serves
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // i = ci * cout + co
    if (i >= (int64_t)cout * cin) return;
    const int co = (int)(i % cout), ci = (int)(i / cout);
    const float* gsrc = adjoint ? w + ((int64_t)ci * cout + co) * 9 : w + ((int64_t)co * cin + ci) * 9;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = gsrc[adjoint ? 8 - k : k];
    float t[4][3];  // G g
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
        t[0][c] = g0;
        t[1][c] = 0.5f * (g0 + g1 + g2);
        t[2][c] = 0.5f * (g0 - g1 + g2);
        t[3][c] = g2;
    }
    float* dst = adjoint ? u_adj : u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = t[r][0], b = t[r][1], c = t[r][2];
        const float o[4] = {a, 0.5f * (a + b + c), 0.5f * (a - b + c), c};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dst[(((int64_t)ci * 2 + (j >> 1)) * 2 + (r >> 1)) * cout * 4 + (int64_t)co * 4 + (r & 1) * 2 + (j & 1)] = o[j];
    }
}

// The same for EVERY 3x3 layer of a network in one launch (EMANet-50: 19 layers, 19 launches of 5-17 us per step before).
// table: n_layers rows of WPM_FIELDS int64 = (w, u, u_adjoint, cout, cin, first block); blockIdx.y: 0 forward, 1 adjoint.
constexpr int WPM_FIELDS = 6;
__global__ __launch_bounds__(256) void wino_prepack_multi_kernel(const long long* __restrict__ table, int n_layers) {
    int layer = 0;
    while (layer + 1 < n_layers && (long long)blockIdx.x >= table[(layer + 1) * WPM_FIELDS + 5]) ++layer;
    const long long* d = table + (int64_t)layer * WPM_FIELDS;
    const float* w = reinterpret_cast<const float*>(d[0]);
    const int cout_w = (int)d[3], cin_w = (int)d[4];
    const bool adjoint = blockIdx.y == 1;
    const int cout = adjoint ? cin_w : cout_w, cin = adjoint ? cout_w : cin_w;
    const int64_t i = ((int64_t)blockIdx.x - d[5]) * 256 + threadIdx.x;  // i = ci * cout + co
    if (i >= (int64_t)cout * cin) return;
    const int co = (int)(i % cout), ci = (int)(i / cout);
    const float* gsrc = adjoint ? w + ((int64_t)ci * cout + co) * 9 : w + ((int64_t)co * cin + ci) * 9;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = gsrc[adjoint ? 8 - k : k];
    float t[4][3];  // G g
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
        t[0][c] = g0;
        t[1][c] = 0.5f * (g0 + g1 + g2);
        t[2][c] = 0.5f * (g0 - g1 + g2);
        t[3][c] = g2;
    }
    float* dst = reinterpret_cast<float*>(adjoint ? d[2] : d[1]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float a = t[r][0], b = t[r][1], c = t[r][2];
        const float o[4] = {a, 0.5f * (a + b + c), 0.5f * (a - b + c), c};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dst[(((int64_t)ci * 2 + (j >> 1)) * 2 + (r >> 1)) * cout * 4 + (int64_t)co * 4 + (r & 1) * 2 + (j & 1)] = o[j];
    }
}

// 8 waves: (co half wm) x (tile half wn) x (xi column pair q).  Wave q owns the Winograd columns j in {2q, 2q+1}
// of M (xi = 4 i + j), i.e. 8 of the 16 MFMA chains = 128 accumulator VGPRs, so two waves share a SIMD and one
// wave's patch reads / transform adds run under the other's MFMAs.  A^T M is column-local; only the final
// "* A" mixes columns, so the q = 1 waves hand 4 partial values per (co, tile) to their q = 0 partner through
// the (by then idle) weight staging LDS and the q = 0 waves run the layer tail.
constexpr int WNTHR = 512;

__global__ __launch_bounds__(WNTHR, 2) void modconv_wino_kernel(const ConvParams p, const int xt_max) {
    constexpr int WF = WCC * 16 * WMBLK;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ul = lds;                    // [2][WF]   (64 KB; reused for the column exchange in the epilogue)
    float* Xl = lds + 2 * WF;           // [2][WCC * xt]
    float* Sl = Xl + 2 * WCC * xt_max;  // [nb][Cin]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int q = __builtin_amdgcn_readfirstlane(wave & 1), wn = (wave >> 1) & 1, wm = wave >> 2;
    const int wbase = tid & ~63;

    // XCD-aware order: output-channel block fastest.  Workgroups are dealt round-robin over the 8 XCDs, so with
    // n_co = Cout / MBLK in {2,4,8} every XCD keeps working on the same weight slice (<= 2.4 MB: stays in its
    // 4 MiB L2) while the n_co workgroups that share an input tile run at the same time on different XCDs
    // (one HBM read, the rest MALL hits).  Pixel-tile-fastest order measured a 48 % L2 miss rate on this kernel.
    const int n_co = (p.Cout + WMBLK - 1) / WMBLK;
    int pt = blockIdx.x / n_co;
    const int o0 = (blockIdx.x % n_co) * WMBLK;
    const TileClass tc = p.cls[0];
    const int twi = pt % tc.ntw; pt /= tc.ntw;
    const int thi = pt % tc.nth;
    const int bt = pt / tc.nth;
    const int thl = tc.th_log2, twl = tc.tw_log2;
    const int th = 1 << thl, tw = 1 << twl;
    const int b0 = bt * tc.nb, h0 = thi << thl, w0 = twi << twl;
    // staged input tile: rows h0-1 .. h0+th, columns w0-4 .. w0+tw+3 (16-byte aligned superset of the 1-pixel
    // halo: W and w0 are multiples of 4, so every aligned float4 is entirely inside or entirely outside the image)
    const int eh = th + 2, ew = tw + 8;
    const int xt = tc.xt;
    const int HW = p.H * p.W;
    const int k_lo = blockIdx.y * p.kchunk;
    const int k_hi = min(p.Cin, k_lo + p.kchunk);

    for (int e = tid; e < 2 * WCC * xt; e += WNTHR) Xl[e] = 0.f;
    {  // four elements per lane and trip, requested together (clamped; a predicated load per trip was one round trip per element)
        const int sl_total = p.s ? tc.nb * p.Cin : 0;  // plain convolution (p.s == nullptr): no style rows
        for (int e0 = tid; e0 < sl_total; e0 += 4 * WNTHR) {
            float sv4[4];
            bool in_batch[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * WNTHR, sl_total - 1);
                const int n = e / p.Cin, ci = e - n * p.Cin;
                in_batch[u] = b0 + n < p.B;
                sv4[u] = p.s[(int64_t)min(b0 + n, p.B - 1) * p.Cin + ci];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * WNTHR < sl_total) Sl[e0 + u * WNTHR] = in_batch[u] ? sv4[u] : 0.f;
        }
    }
    // one float4 chunk of the tile per lane per channel (<= 512 chunks: host-checked)
    int st_goff = -1;
    {
        const int ew4 = ew >> 2;
        if (tid < (xt >> 2)) {
            const int n = tid / (eh * ew4), rem = tid - n * (eh * ew4);
            const int r = rem / ew4, c4 = rem - r * ew4;
            const int b = b0 + n, h = h0 - 1 + r, w = w0 - 4 + 4 * c4;
            if (b < p.B && h >= 0 && h < p.H && w >= 0 && w < p.W) st_goff = b * p.Cin * HW + h * p.W + w;
        }
    }
    // weights of a chunk: 32 rows (ci, q, ih) x 64 co x 4 floats; float4 e = it * 512 + tid sits in row e / 64 at co e % 64
    constexpr int WV4 = WF / 4, WIT = WV4 / WNTHR;  // 4 float4 per lane per chunk
    int w_goff[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int e = it * WNTHR + tid;
        const int row = e / WMBLK, co = e - row * WMBLK;
        w_goff[it] = (o0 + co < p.Cout) ? (row * p.Cout + o0 + co) * 4 : -1;
    }
    auto stage = [&](int ci0, int buf) {
        const float* usrc = p.wpk + (int64_t)ci0 * 16 * p.Cout;
        float* udst = Ul + buf * WF + wbase * 4;
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            if (w_goff[it] >= 0) glds16(usrc + w_goff[it], udst + it * WNTHR * 4);
        const float* xsrc = p.x + (int64_t)ci0 * HW;
        float* xdst = Xl + buf * WCC * xt + wbase * 4;
        if (st_goff >= 0) {
#pragma unroll
            for (int j = 0; j < WCC; ++j) glds16(xsrc + st_goff + j * HW, xdst + j * xt);
        }
    };

    const int tpl = thl + twl - 2;  // log2(tiles per sample)
    const int t = wn * 32 + l31;
    const int tn = t >> tpl, trem = t & ((1 << tpl) - 1);
    const int ty = trem >> (twl - 1), tx = trem & ((tw >> 1) - 1);
    const int xo = min(tn, tc.nb - 1) * eh * ew + 2 * ty * ew + 2 * tx + 2 + half * xt;  // even: 8-byte aligned reads
    const int so = min(tn, tc.nb - 1) * p.Cin + half;
    const int aoff = (half * 4 + q * 2) * (WMBLK * 4) + (wm * 32 + l31) * 4;  // + (8 cp + ih) * WMBLK * 4: float4 (il, jj)

    f32x16 acc[4][2];  // [row i of M][column jj of this wave's pair]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][jj][j] = 0.f;

    __syncthreads();
    stage(k_lo, 0);
    __syncthreads();

    int buf = 0;
    for (int ci0 = k_lo; ci0 < k_hi; ci0 += WCC, buf ^= 1) {
        if (ci0 + WCC < k_hi) stage(ci0 + WCC, buf ^ 1);
        const float* Ub = Ul + buf * WF + aoff;
        const float* Xb = Xl + buf * WCC * xt + xo;
#pragma unroll
        for (int cp = 0; cp < WCC / 2; ++cp) {
            const float sv = p.s ? Sl[so + ci0 + 2 * cp] : 1.f;
            const float* xb = Xb + 2 * cp * xt;
            float d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // patch columns sit at odd offsets: three aligned 8-byte reads, middle 4 used
                const float2 a0 = lds_ld2(xb + r * ew);
                const float2 a1 = lds_ld2(xb + r * ew + 2);
                const float2 a2 = lds_ld2(xb + r * ew + 4);
                d[r][0] = a0.y; d[r][1] = a1.x; d[r][2] = a1.y; d[r][3] = a2.x;
            }
            float tt[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {  // B^T d
                tt[0][c] = d[0][c] - d[2][c];
                tt[1][c] = d[1][c] + d[2][c];
                tt[2][c] = d[2][c] - d[1][c];
                tt[3][c] = d[1][c] - d[3][c];
            }
            const f32x4 ua0 = lds_ld4(Ub + (8 * cp) * WMBLK * 4);      // rows i = 0, 1
            const f32x4 ua1 = lds_ld4(Ub + (8 * cp + 1) * WMBLK * 4);  // rows i = 2, 3
            const float ua[4][2] = {{ua0.x, ua0.y}, {ua0.z, ua0.w}, {ua1.x, ua1.y}, {ua1.z, ua1.w}};
            float v0[4], v1[4];
            if (q == 0) {  // wave-uniform: this wave's two columns of (B^T d) B, scaled by the style
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = (tt[i][0] - tt[i][2]) * sv; v1[i] = (tt[i][1] + tt[i][2]) * sv; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = (tt[i][2] - tt[i][1]) * sv; v1[i] = (tt[i][1] - tt[i][3]) * sv; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[i][0], v0[i], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[i][1], v1[i], acc[i][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  m[r][jj] = (A^T M)[r][column 2q+jj];  Y[r][0] = m0 + m1 + m2,  Y[r][1] = m1 - m2 - m3.
    // q = 0 contributes (m0 + m1, m1), q = 1 contributes (m2, -m2 - m3).
    float part[16][4];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float m[2][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            m[0][jj] = acc[0][jj][j] + acc[1][jj][j] + acc[2][jj][j];
            m[1][jj] = acc[1][jj][j] - acc[2][jj][j] - acc[3][jj][j];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            part[j][2 * r] = q == 0 ? m[r][0] + m[r][1] : m[r][0];
            part[j][2 * r + 1] = q == 0 ? m[r][1] : -m[r][0] - m[r][1];
        }
    }
    float* xch = Ul + (wave >> 1) * (64 * 64);  // [16 j][4][64 lanes] per wave pair
    if (q == 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) xch[(j * 4 + e) * 64 + lane] = part[j][e];
    }
    __syncthreads();
    if (q == 1) return;

    const bool partial = p.ksplit > 1;
    const int b = b0 + tn, oh = h0 + 2 * ty, ow = w0 + 2 * tx;
    if (tn >= tc.nb || b >= p.B || oh >= p.H || ow >= p.W) return;
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (!partial && p.fuse && p.noise) {
        const float nw = p.noise_w[0];
        const float* np = p.noise + (int64_t)b * p.noise_bstride + oh * p.W + ow;
        nz[0] = nw * np[0]; nz[1] = nw * np[1]; nz[2] = nw * np[p.W]; nz[3] = nw * np[p.W + 1];
    }
    float* obase = (partial ? p.slab + (int64_t)blockIdx.y * p.B * p.Cout * HW : p.out) + (int64_t)b * p.Cout * HW +
                   oh * p.W + ow;
    const float* db = p.dscale + (int64_t)b * p.Cout;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int co = o0 + wm * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
        if (co >= p.Cout) continue;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = part[j][e] + xch[(j * 4 + e) * 64 + lane];
        if (!partial) {
            const float dd = p.dscale ? db[co] : 1.f;
            const float bb = (p.fuse && p.bias) ? p.bias[co] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float val = y[e] * dd;
                if (p.fuse) {
                    val += nz[e];
                    val += bb;
                    val = (val > 0.f ? val : val * 0.2f) * 1.4142135623730951f;
                }
                y[e] = val;
            }
        }
        float* oc = obase + (int64_t)co * HW;
        *reinterpret_cast<float2*>(oc) = make_float2(y[0], y[1]);
        *reinterpret_cast<float2*>(oc + p.W) = make_float2(y[2], y[3]);
    }
}

#ifdef SIS_WINO_TRACE
// Development build only (tools/wino_trace.sh): per-wave cycle stamps of the steady-state chunk loop of 4 workgroups.
constexpr int TR_WG0 = 500, TR_NWG = 4, TR_CHUNKS = 64, TR_SLOTS = 4;
__device__ unsigned int sis_wino_trace[TR_NWG][8][TR_CHUNKS][TR_SLOTS];
__device__ unsigned int sis_wino_trace_tile[TR_NWG][8][16][8];
#define WINO_TRACE_TILE(slot)                                                                                     \
    do {                                                                                                          \
        if (blockIdx.y == 0 && blockIdx.x >= TR_WG0 && blockIdx.x < TR_WG0 + TR_NWG && k < 16) {                    \
            const unsigned int now_ = (unsigned int)__builtin_readcyclecounter();                                 \
            if (lane == 0) sis_wino_trace_tile[blockIdx.x - TR_WG0][wave][k][slot] = now_;                        \
        }                                                                                                         \
    } while (0)
#define WINO_TRACE(slot)                                                                                          \
    do {                                                                                                          \
        if (tr_on && c < TR_CHUNKS) {                                                                             \
            const unsigned int now_ = (unsigned int)__builtin_readcyclecounter();                                 \
            if (lane == 0) sis_wino_trace[blockIdx.x - TR_WG0][wave][c][slot] = now_;                             \
        }                                                                                                         \
    } while (0)
#else
#define WINO_TRACE(slot) do {} while (0)
#define WINO_TRACE_TILE(slot) do {} while (0)
#endif

// Pipelined variant (default when its 158 KB of LDS fit): the input transform is done ONCE per (channel, tile) --
// one patch per lane per chunk, 512 lanes = 8 channels x 64 tiles -- into a second LDS image V[xi][channel][tile],
// one chunk ahead of the MFMAs, so the matrix loop is nothing but ds_read_b32 pairs and MFMAs (in the kernel above
// every patch is re-read and re-transformed by the 4 waves that share it: 48.5 % MFMA-busy measured).  Per
// iteration c: DMA weights(c+1), DMA input(c+2), transform(c+1) -> V, MFMA(c); one barrier.
// Template: XI = 64-lane parts of the input tile (power of two >= xt / 256), STYLED = modulated (style rows in LDS).
template <int XI, bool STYLED>
__global__ __launch_bounds__(WNTHR, 2) void modconv_wino2_kernel(const ConvParams p, const int xt_max, const int tiles_per_wg, const int xcd_group) {
    constexpr int WF = WCC * 16 * WMBLK;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int VF = 16 * WCC * WTILES;  // transformed input of one chunk: [channel][q][ih][tile][il][jj] (as the weights)
    float* Ul = lds;                    // [2][WF]   (64 KB)
    float* Vl = lds + 2 * WF;           // [2][VF]   (64 KB; reused for the column exchange in the epilogue)
    float* Xl = Vl + 2 * VF;            // [2][WCC * xt]
    float* Sl = Xl + 2 * WCC * xt_max;  // [nb][Cin]   (absent for a plain convolution)
    float* Tl = Sl + (p.s ? p.nb_max * p.Cin : 0);  // layer-tail operands of the current tile:
    float* Dl = Tl;                      //   [nb][WMBLK] scale * demodulation
    float* Bl = Dl + p.nb_max * WMBLK;   //   [WMBLK]     bias
    float* Nl = Bl + WMBLK;              //   [WTILES][4] noise_weight * noise of the tile's pixels

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int q = wave & 1, wn = (wave >> 1) & 1, wm = wave >> 2;

    // XCD-aware order: output-channel block fastest.  Workgroups are dealt round-robin over the 8 XCDs, so with
    // n_co = Cout / MBLK in {2,4,8} every XCD keeps working on the same weight slice (<= 2.4 MB: stays in its
    // 4 MiB L2) while the n_co workgroups that share an input tile run at the same time on different XCDs
    // (one HBM read, the rest MALL hits).  Pixel-tile-fastest order measured a 48 % L2 miss rate on this kernel.
    // When ALL the transformed weights fit one L2 (128 / 256 channels: 1 / 4 MB) the order is the other way
    // round (xcd_group): logical workgroup id = (hardware id % 8) * (grid / 8) + hardware id / 8, so the n_co
    // workgroups of a pixel tile sit on ONE XCD and the input tile is fetched once instead of n_co times.
#ifdef SIS_WINO_STAGGER  // experiment: workgroups start out of phase so their epilogue store bursts do not coincide
    for (int i = ((int)(blockIdx.x >> 3) & 7) * SIS_WINO_STAGGER; i > 0; --i) __builtin_amdgcn_s_sleep(127);
#endif
    const int n_co = (p.Cout + WMBLK - 1) / WMBLK;
    const int wg = xcd_group ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int o0 = (wg % n_co) * WMBLK;
    const TileClass tc = p.cls[0];
    const int thl = tc.th_log2, twl = tc.tw_log2;
    const int th = 1 << thl, tw = 1 << twl;
    // staged input tile: rows h0-1 .. h0+th, columns w0-4 .. w0+tw+3 (16-byte aligned superset of the 1-pixel
    // halo: W and w0 are multiples of 4, so every aligned float4 is entirely inside or entirely outside the image)
    const int eh = th + 2, ew = tw + 8;
    const int xt = tc.xt;
    const int HW = p.H * p.W;
    const int k_lo = blockIdx.y * p.kchunk;
    const int k_hi = min(p.Cin, k_lo + p.kchunk);
    const int tpl0 = thl + twl - 2;  // log2(tiles per sample)

    // A workgroup walks `tiles_per_wg` pixel tiles of its output-channel block (persistent over
    // tiles): the next tile's first DMA flies under the current tile's epilogue, and the per-workgroup launch /
    // first-touch latency is paid once per `tiles_per_wg` tiles instead of once per tile (7 us against 16 chunks
    // x 2.5 us on the 128-channel 256^2 layer).
    // Input tile DMA: the tile is xt / 4 float4 per channel, cut into XI parts of 64 lanes (XI rounded up to a power of
    // two); wave w moves part w % XI of the channels w / XI + k * (8 / XI) -- every wave issues the same number of DMA
    // instructions (issue back-pressure on two waves was the critical path of the chunk) from one offset register.
    constexpr int x_chstep = WCC / XI;
    const int x_part = wave & (XI - 1), x_ch0 = wave / XI;
    // The last part is shifted back so that it ends with the tile (it overlaps its neighbour, which writes the same
    // values): every lane of every DMA instruction owns a float4 of the tile -- no exec mask, no branch in the loop.
    const int x_base = min(x_part * 256, xt - 256);  // floats; xt >= 256 (a tile has 64 patches) and xt % 4 == 0
    const int x_f4 = (x_base >> 2) + lane;  // this lane's float4 of the tile
    // Tile state: the tile being multiplied and (from the top of its chunk loop on) the next one of this workgroup, whose
    // first chunks are staged by the current tile's LAST chunks -- one flat (tile, chunk) pipeline, no start-up per tile.
    struct TileState {
        int b0, h0, w0;
        unsigned x_voff;                // byte offset of this lane's float4 inside the channel plane, from the tile's first sample
        __amdgpu_buffer_rsrc_t x_rsrc;  // descriptor based at the tile's first sample
    } T, Tn;
    // (sample, row, float4 column) of this lane's float4 inside a tile: the same for every tile
    const int ew4_ = ew >> 2;
    const int xl_n = x_f4 / (eh * ew4_), xl_r = (x_f4 - xl_n * (eh * ew4_)) / ew4_, xl_c4 = x_f4 - xl_n * (eh * ew4_) - xl_r * ew4_;
    auto tile_setup = [&](int pt, TileState& S) {
        const int twi = pt % tc.ntw; pt /= tc.ntw;
        const int thi = pt % tc.nth;
        S.b0 = (pt / tc.nth) * tc.nb; S.h0 = thi << thl; S.w0 = twi << twl;
        const int b = S.b0 + xl_n, h = S.h0 - 1 + xl_r, w = S.w0 - 4 + 4 * xl_c4;
        S.x_voff = (b < p.B && h >= 0 && h < p.H && w >= 0 && w < p.W) ? (unsigned)(xl_n * p.Cin * HW + h * p.W + w) * 4u : BUF_OOB;
        S.x_rsrc = dma_rsrc(p.x + (int64_t)S.b0 * p.Cin * HW);
    };
    // Tile k of workgroup g is pixel tile g + k * (#workgroups per channel block): the workgroups running at the same
    // time cover NEIGHBOURING tiles, whose halos they share through L2 (consecutive tiles per workgroup measured 57 %
    // more fetched bytes).
    const int pt_first = wg / n_co, pt_step = gridDim.x / n_co;
    tile_setup(pt_first, T);
    // weights: chunk = 32 rows (ci, q, ih) of 64 co x 4 floats = one 1 KB DMA instruction per row, 4 rows per wave
    // (row = it * 8 + wave); every lane moves the float4 of output channel o0 + lane
    constexpr int WIT = WF / 4 / WNTHR;
    const __amdgpu_buffer_rsrc_t u_rsrc = dma_rsrc(p.wpk + (int64_t)o0 * 4);
    const unsigned u_voff = (o0 + lane < p.Cout) ? (unsigned)lane * 16u : BUF_OOB;
    const unsigned u_row_bytes = (unsigned)p.Cout * 16u;
    auto stage_u = [&](int ci0, int buf) {
        float* udst = Ul + buf * WF + wave * 256;
        const unsigned s0 = (unsigned)(ci0 * 4 + wave) * u_row_bytes;
#pragma unroll
        for (int it = 0; it < WIT; ++it) bufld16(u_rsrc, udst + it * 8 * 256, u_voff, s0 + (unsigned)(it * 8) * u_row_bytes);
    };
    auto stage_x = [&](int ci0, int buf, const TileState& S) {
        float* xdst = Xl + buf * WCC * xt + x_base;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int j = x_ch0 + k * x_chstep;
            bufld16(S.x_rsrc, xdst + j * xt, S.x_voff, (unsigned)((ci0 + j) * HW) * 4u);
        }
    };
    auto stage_u_piece = [&](int ci0, int buf, int it) {  // one 1 KB DMA instruction of stage_u
        bufld16(u_rsrc, Ul + buf * WF + wave * 256 + it * 8 * 256, u_voff, (unsigned)(ci0 * 4 + wave + it * 8) * u_row_bytes);
    };
    auto stage_x_piece = [&](int ci0, int buf, int k, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {  // one DMA instruction of stage_x
        const int j = x_ch0 + k * x_chstep;
        bufld16(rsrc, Xl + buf * WCC * xt + x_base + j * xt, voff, (unsigned)((ci0 + j) * HW) * 4u);
    };
    // Input transform, ONE patch per lane per chunk: lane = tile (0..63), wave = channel of the chunk.  V = B^T d B
    // scaled by the style of the tile's sample, written as [channel][q][ih][tile][il][jj] (xi = 4 (2 ih + il) + 2 q + jj):
    // four 16-byte LDS writes per patch, lane-contiguous.  Packed fp32 arithmetic, see the chunk loop.
    const int ttile = lane, tch = wave;
    const int ttn = ttile >> tpl0, ttrem = ttile & ((1 << tpl0) - 1);
    const int tty = ttrem >> (twl - 1), ttx = ttrem & ((tw >> 1) - 1);
    const int txo = min(ttn, tc.nb - 1) * eh * ew + 2 * tty * ew + 2 * ttx + 2 + tch * xt;
    const int tso = min(ttn, tc.nb - 1) * p.Cin + tch;
    const int tvo = tch * (16 * WTILES) + ttile * 4;  // + (2 q + ih) * WTILES * 4
    const f32x2 negpos = {-1.f, 1.f};
    auto transform = [&](int ci0, int xbuf, int vbuf) {
        f32x2 sv2 = {1.f, 1.f};
        if (STYLED) sv2.x = Sl[tso + ci0];
        const float* xb = Xl + xbuf * WCC * xt + txo + 1;
        f32x2 t01[4], t23[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 px = {xb[r * ew], xb[r * ew + 1]}, py = {xb[r * ew + 2], xb[r * ew + 3]};
            t01[r] = pk_lo_np_plus(py, negpos, px);
            t23[r] = pk_hi_np_cross(px, negpos, py);
        }
        float* vb = Vl + vbuf * VF + tvo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // column pair (2h, 2h+1) = q
            const f32x2* tq = h == 0 ? t01 : t23;
            const f32x2 a0_ = pk_sub(tq[0], tq[2]), a1_ = pk_add(tq[1], tq[2]), a2_ = pk_sub(tq[2], tq[1]), a3_ = pk_sub(tq[1], tq[3]);
            lds_st4(vb + (2 * h) * WTILES * 4, pk_scale(a0_, sv2), pk_scale(a1_, sv2));      // rows i = 0, 1
            lds_st4(vb + (2 * h + 1) * WTILES * 4, pk_scale(a2_, sv2), pk_scale(a3_, sv2));  // rows i = 2, 3
        }
    };

    const int tpl = tpl0;
    const int t = wn * 32 + l31;
    const int tn = t >> tpl, trem = t & ((1 << tpl) - 1);
    const int ty = trem >> (twl - 1), tx = trem & ((tw >> 1) - 1);
    // A / B operands of this lane: float4 (il, jj) of row (channel 2 cp + half, q, ih) at column co / tile
    const int aoff = (half * 4 + q * 2) * (WMBLK * 4) + (wm * 32 + l31) * 4;    // + (8 cp + ih) * WMBLK * 4
    const int voff = (half * 4 + q * 2) * (WTILES * 4) + (wn * 32 + l31) * 4;   // + (8 cp + ih) * WTILES * 4

    f32x16 acc[4][2];  // [row i of M][column jj of this wave's pair]
    const bool partial = p.ksplit > 1;

    // First DMA of a tile: every global access of the start-up is in flight before the first wait (one memory
    // round trip).  Out-of-image float4 slots get their zeros from the DMA itself (out-of-range buffer offsets).
    // Style rows of the tile's samples: nb * Cin consecutive floats of p.s.  Up to 1024 of them travel through two registers
    // per lane: loaded BEFORE the tile's first DMA (a wait for them must not cover the DMA issued after them) and parked in
    // LDS by style_store() once nobody reads the previous tile's rows any more.
    const int sl_n = STYLED ? tc.nb * p.Cin : 0;
    const bool sl_in_regs = sl_n <= 2 * WNTHR;
    float sl_reg[2] = {0.f, 0.f};
    auto style_load = [&](const TileState& S) {
        const int avail = (p.B - S.b0) * p.Cin;  // rows of samples beyond the batch read as zero
        const float* src = p.s + (int64_t)S.b0 * p.Cin;
        if (sl_in_regs) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * WNTHR;
                sl_reg[i] = (e < sl_n && e < avail) ? src[e] : 0.f;
            }
        } else {
            for (int e = tid; e < sl_n; e += WNTHR) Sl[e] = e < avail ? src[e] : 0.f;
        }
    };
    auto style_store = [&]() {
        if (!sl_in_regs) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + i * WNTHR < sl_n) Sl[tid + i * WNTHR] = sl_reg[i];
    };
    auto tile_first_dma = [&]() {  // the workgroup's first tile only
        if (STYLED) style_load(T);
        stage_u(k_lo, 0);
        stage_x(k_lo, 0, T);
        if (k_lo + WCC < k_hi) stage_x(k_lo + WCC, 1, T);
    };
    // Layer-tail operands (demodulation, bias, noise) of a tile: fetched into registers a whole tile ahead, parked in
    // LDS once nobody reads the previous tile's any more, read back in the epilogue -- no dependent global loads
    // in the tail and no registers held across the matrix loop.
    // (Nothing here may USE a loaded value -- a use is a wait for every load in flight, the next tile's input DMA included:
    // the noise is scaled when it is parked.)
    float t_noise = 0.f, t_scale = 1.f, t_bias = 0.f;
    const float noise_w0 = (!partial && p.fuse && p.noise) ? p.noise_w[0] : 0.f;
    auto tail_load = [&](const TileState& S) {
        t_noise = 0.f; t_scale = 1.f; t_bias = 0.f;
        if (partial) return;
        if (p.fuse && p.noise && tid < WTILES * 4) {
            const int tt_ = tid >> 2, e = tid & 3;
            const int n = tt_ >> tpl0, rem = tt_ & ((1 << tpl0) - 1);
            const int yy = S.h0 + 2 * (rem >> (twl - 1)) + (e >> 1), xx = S.w0 + 2 * (rem & ((tw >> 1) - 1)) + (e & 1);
            if (n < tc.nb && S.b0 + n < p.B && yy < p.H && xx < p.W)
                t_noise = p.noise[(int64_t)(S.b0 + n) * p.noise_bstride + yy * p.W + xx];
        }
        if (tid < tc.nb * WMBLK) {
            const int n = tid / WMBLK, co = o0 + tid % WMBLK;
            if (p.dscale && S.b0 + n < p.B && co < p.Cout) t_scale = p.dscale[(int64_t)(S.b0 + n) * p.Cout + co];
            if (n == 0 && p.fuse && p.bias && co < p.Cout) t_bias = p.bias[co];
        }
    };
    auto tail_store = [&]() {
        if (tid < WTILES * 4) Nl[tid] = noise_w0 * t_noise;
        if (tid < tc.nb * WMBLK) Dl[tid] = t_scale;
        if (tid < WMBLK) Bl[tid] = t_bias;
    };

    tile_first_dma();
    tail_load(T);
    style_store();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunk 0 (and input chunk 1) landed, styles visible
    tail_store();
    transform(k_lo, 0, 0);
    __syncthreads();  // V(0) visible

    for (int k = 0;; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[i][jj][j] = 0.f;

        WINO_TRACE_TILE(0);
#ifdef SIS_WINO_TRACE
        const bool tr_on = k == 1 && blockIdx.y == 0 && blockIdx.x >= TR_WG0 && blockIdx.x < TR_WG0 + TR_NWG;
#endif
        // The next tile of this workgroup: its descriptor now, its first two input chunks by DMA from this tile's last two
        // chunks, its first transform in this tile's last chunk (style factor from a register), its weights for free (chunk
        // 0 of the weights is the same for every tile: the last chunk's prefetch index wraps).  Measured before this: 4-5 k
        // cycles per tile for the next tile's setup + first DMA burst, 1.7 k for its first transform and the barriers around it.
        const bool has_next = k + 1 < tiles_per_wg;  // (the host keeps tiles_per_wg at 1 when a tile has a single chunk)
        float sv_next = 1.f;
        if (has_next) {
            tile_setup(pt_first + (k + 1) * pt_step, Tn);
            if (STYLED) sv_next = (Tn.b0 + ttn < p.B) ? p.s[(int64_t)(Tn.b0 + min(ttn, tc.nb - 1)) * p.Cin + k_lo + tch] : 0.f;
        }
        // Single-phase software-pipelined chunk.  Measured (tools/wino_trace.py, tools/micro/mfma_valu_overlap.hip): on this
        // SIMD nothing a wave issues is hidden under the matrix pipe -- a chunk costs 64 MFMAs x 64 cycles plus ~5 cycles
        // for EVERY other instruction of its two waves -- and a wave that transforms while its partner multiplies crawls
        // (the staggered two-phase loop of round 1 had the partners multiplying one after the other).  So every wave runs
        // ONE straight-line block per chunk with as few instructions as the data flow allows: 16-byte operand reads (one
        // per operand per two MFMA pairs), the DMA with scalar addressing, the transform with 16-byte writes, buffer
        // parity a compile-time constant (LDS addresses are immediates).
        const int k_last = k_hi - WCC, k_len = k_hi - k_lo;
        // One slot after each of the 32 MFMAs.  Issue costs add up inside an MFMA gap and only ~48 cycles of them hide under a
        // 64-cycle MFMA (MI355X_MICROARCH.md, constants: an LDS-DMA piece costs 60-185 cycles to issue, a packed-f32 VALU
        // instruction ~3x a scalar one), so the side work is spread one expensive item per slot.
        f32x4 ou[3], ov[3];
        auto chunk_barrier = [&]() {
#ifdef SIS_WINO_NOBARRIER  // timing experiment only (results are garbage): what does the per-chunk barrier cost?
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            // (the wait for this wave's DMA pieces is spelled out: the compiler places its own where it sees an LDS read that may
            // alias a DMA destination, which need not be in front of the barrier)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // V(c+1) written, DMA retired, everyone done with U(c) / V(c)
#endif
        };
        if (S_BAR < 32) {  // a tile starts on parity 0
            ou[0] = lds_ld4(Ul + aoff);
            ov[0] = lds_ld4(Vl + voff);
        }
        auto chunk = [&](auto parity, const int ci0, const int c) {
            constexpr int cur = decltype(parity)::value, nxt = cur ^ 1;
            WINO_TRACE(0);
            const bool last = ci0 + WCC >= k_hi;          // the transform below then works for the next tile's chunk 0
            const int uci = last ? k_lo : ci0 + WCC;      // weights: the same for every tile
            const bool x_next = ci0 + 2 * WCC >= k_hi && has_next;  // input chunk c + 2 belongs to the next tile
            const int xci = x_next ? ci0 + 2 * WCC - k_len : min(ci0 + 2 * WCC, k_last);
            const __amdgpu_buffer_rsrc_t xr = x_next ? Tn.x_rsrc : T.x_rsrc;
            const unsigned xv = x_next ? Tn.x_voff : T.x_voff;
            const float* xb = Xl + nxt * WCC * xt + txo + 1;
            const float* Ub = Ul + cur * WF + aoff;
            const float* Vb = Vl + cur * VF + voff;
            float* vw = Vl + nxt * VF + tvo;
            auto operands = [&](int g, int slot) {  // group g = (cp, ih): MFMA rows i = 2 ih, 2 ih + 1, both jj
                ou[slot] = lds_ld4(Ub + (8 * (g >> 1) + (g & 1)) * WMBLK * 4);
                ov[slot] = lds_ld4(Vb + (8 * (g >> 1) + (g & 1)) * WTILES * 4);
            };
            float sv = 1.f, d[4][4], e[4][4], o[4][4];  // patch, d B, B^T (d B) scaled
            // slot tables: first slot of each kind of side work
#ifndef SIS_WINO_XDMA_EARLY
#define SIS_WINO_XDMA_EARLY 0
#endif
            // input pieces every (second) slot, weight pieces every second slot; EARLY: up to four input pieces in the odd slots
            // between the weight pieces (the barrier at slot S_BAR waits for all of them: 26 slots of cover instead of 19)
            constexpr bool X_EARLY = SIS_WINO_XDMA_EARLY && XI <= 4;
            constexpr int X_STEP = X_EARLY ? 2 : (XI > 2 ? 1 : 2);
            constexpr int S_UDMA = 0, S_XDMA = X_EARLY ? 1 : 8;
            static_assert((X_EARLY || S_UDMA + 2 * (WIT - 1) < S_XDMA) && S_XDMA + X_STEP * (XI - 1) < 27, "DMA pieces must fit in front of the barrier");
            constexpr int S_READ = 1, S_RSTEP = 2;  // patch rows: every second slot from slot 1
            constexpr int S_A = 11;                 // d B: 4 slots
            static_assert(S_BAR >= 27 || S_BAR < 0, "slot 0 of the operand registers is in use until slot 27");
            constexpr int S_B = 15;                 // B^T (d B) and scale, column j: slots S_B + 2 j, S_B + 2 j + 1; writes follow
            // operands run two groups (four MFMA pairs) ahead; group 0 of this chunk was requested behind the PREVIOUS chunk's
            // barrier (or at the tile start), under that chunk's last MFMAs
            if (S_BAR >= 32) operands(0, 0);
            operands(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            WINO_TRACE(1);
#pragma unroll
            for (int sl = 0; sl < 32; ++sl) {
                const int g = sl >> 2, part = sl & 3, slot = g % 3;
                const int i = 2 * (g & 1) + (part >> 1), jj = part & 1;
                const float ua = part == 0 ? ou[slot].x : part == 1 ? ou[slot].y : part == 2 ? ou[slot].z : ou[slot].w;
                const float va = part == 0 ? ov[slot].x : part == 1 ? ov[slot].y : part == 2 ? ov[slot].z : ov[slot].w;
                acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua, va, acc[i][jj], 0, 0, 0);
                if (part == 1 && g < 6) operands(g + 2, (g + 2) % 3);
#ifndef SIS_WINO_NODMA
#pragma unroll
                for (int it = 0; it < WIT; ++it)
                    if (sl == S_UDMA + 2 * it) stage_u_piece(uci, nxt, it);
#pragma unroll
                for (int kx = 0; kx < XI; ++kx)
                    if (sl == S_XDMA + X_STEP * kx) stage_x_piece(xci, cur, kx, xr, xv);
#endif
#ifndef SIS_WINO_NOTRANSFORM
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (sl == S_READ + S_RSTEP * r) {  // patch row r: columns 1..4 of the 8-byte aligned row start
                        if (r == 0 && STYLED) { const float sv_here = Sl[tso + min(ci0 + WCC, k_last)]; sv = last ? sv_next : sv_here; }
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) d[r][cc] = xb[r * ew + cc];
                    }
                    if (sl == S_A + r) {  // (d B)[r][.]
                        e[r][0] = d[r][0] - d[r][2]; e[r][1] = d[r][1] + d[r][2]; e[r][2] = d[r][2] - d[r][1]; e[r][3] = d[r][1] - d[r][3];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (sl == S_B + 2 * j) {
                        o[0][j] = e[0][j] - e[2][j]; o[1][j] = e[1][j] + e[2][j]; o[2][j] = e[2][j] - e[1][j]; o[3][j] = e[1][j] - e[3][j];
                    }
                    if (sl == S_B + 2 * j + 1) {
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) o[ii][j] *= sv;
                        if (j & 1) {  // column pair q = j / 2 complete: rows (0,1) and (2,3) as two 16-byte writes
                            const int qq = j >> 1;
                            *reinterpret_cast<f32x4*>(vw + (2 * qq) * WTILES * 4) = f32x4{o[0][j - 1], o[0][j], o[1][j - 1], o[1][j]};
                            *reinterpret_cast<f32x4*>(vw + (2 * qq + 1) * WTILES * 4) = f32x4{o[2][j - 1], o[2][j], o[3][j - 1], o[3][j]};
                        }
                    }
                }
#endif
                if (sl == S_BAR) {
                    // The chunk's barrier, BEFORE its last MFMAs: by this slot the wave has issued every DMA piece, written its
                    // V(c+1) and read its last operands of U(c) / V(c) (slot 21), so the barrier means what it meant at the chunk
                    // end -- and the first operand pair of chunk c + 1 is requested right behind it, its LDS latency (all eight
                    // waves ask at once) under the MFMAs of slots S_BAR + 1 .. 31 instead of in front of the next chunk's first.
                    chunk_barrier();
                    ou[0] = lds_ld4(Ul + nxt * WF + aoff);
                    ov[0] = lds_ld4(Vl + nxt * VF + voff);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            WINO_TRACE(2);
            WINO_TRACE(3);
            if (S_BAR >= 32) chunk_barrier();
        };
        // (A `late` placement for waves 4-7 -- their DMA in the second half of the chunk, while their SIMD partners are between
        // pieces -- measured 6-8 % SLOWER on the same device: the late pieces are not landed at the barrier.)
        // Buffer parity is a compile-time constant: a tile starts on parity 0 -- workgroups that walk several tiles have an
        // even chunk count per tile (host-checked) -- and the loop body is a pair of chunks.
        for (int ci0 = k_lo, c = 0; ci0 < k_hi; ci0 += 2 * WCC, c += 2) {
            chunk(std::integral_constant<int, 0>(), ci0, c);
            if (ci0 + WCC < k_hi) chunk(std::integral_constant<int, 1>(), ci0 + WCC, c + 1);
        }

        WINO_TRACE_TILE(1);
        if (has_next) {  // the next tile's style rows and tail operands: global loads that land under this tile's epilogue
            if (STYLED) style_load(Tn);
            tail_load(Tn);
        }
        const int ob = T.b0 + tn, oh = T.h0 + 2 * ty, ow = T.w0 + 2 * tx;  // this tile's output pixel
        const bool live = tn < tc.nb && ob < p.B && oh < p.H && ow < p.W;
        WINO_TRACE_TILE(3);

        // ---- epilogue.  m[r][jj] = (A^T M)[r][column 2q+jj];  Y[r][0] = m0 + m1 + m2,  Y[r][1] = m1 - m2 - m3.
        // q = 0 contributes (m0 + m1, m1), q = 1 contributes (m2, -m2 - m3).  The two waves of a pair swap halves:
        // wave q finalises accumulator rows j in [8q, 8q+8) and hands its partial sums of the other 8 rows to its
        // partner through the (by now idle) V image LDS, so both run the layer tail and the stores.
        float mine[8][4];
        // Exchange area: the weight and V buffers of the chunk just finished (the last chunk's parity; the other pair already holds the
        // next tile's chunk 0): wave pairs 0, 1 in the weight buffer, 2, 3 in the V buffer, 16 KB each.
        const int pl = ((k_len / WCC) - 1) & 1, pair = wave >> 1;
        float* xch = (pair < 2 ? Ul + pl * WF : Vl + pl * VF) + (pair & 1) * (64 * 64);  // [sender q][8 rows][4][64 lanes]
        // (Packed fp32 over accumulator-row pairs (j, j+1): no MFMA is in flight on the CU here, so v_pk_add_f32 is two results per
        // issue slot; same operations in the same order as the scalar form.)
        auto reduce_and_send = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                f32x2 m[2][2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const f32x2 a0 = {acc[0][jj][j], acc[0][jj][j + 1]};
                    const f32x2 a1 = {acc[1][jj][j], acc[1][jj][j + 1]};
                    const f32x2 a2 = {acc[2][jj][j], acc[2][jj][j + 1]};
                    const f32x2 a3 = {acc[3][jj][j], acc[3][jj][j + 1]};
                    m[0][jj] = pk_add(pk_add(a0, a1), a2);
                    m[1][jj] = pk_sub(pk_sub(a1, a2), a3);
                }
                f32x2 pr[4];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    pr[2 * r] = Q == 0 ? pk_add(m[r][0], m[r][1]) : m[r][0];
                    pr[2 * r + 1] = Q == 0 ? m[r][1] : pk_neg_sub(m[r][0], m[r][1]);
                }
                if ((j >> 3) == Q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        mine[j & 7][e] = pr[e].x;
                        mine[(j & 7) + 1][e] = pr[e].y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xch[(Q * 32 + (j & 7) * 4 + e) * 64 + lane] = pr[e].x;
                        xch[(Q * 32 + ((j & 7) + 1) * 4 + e) * 64 + lane] = pr[e].y;
                    }
                }
            }
        };
        if (q == 0) reduce_and_send(std::integral_constant<int, 0>());
        else reduce_and_send(std::integral_constant<int, 1>());
        WINO_TRACE_TILE(4);
        __syncthreads();
        WINO_TRACE_TILE(5);

        if (live) {
            // One straight-line path for the three output kinds (split-K slab: raw sums; plain; fused layer tail).  The parked tail
            // operands are already neutral where a kind does not use them (scale 1, bias 0, noise 0: tail_load), so the kind
            // only selects the two activation constants: no branches, every LDS read issued before the first use.
            const bool tail_on = !partial && p.fuse;
            const float slope = tail_on ? 0.2f : 1.f, gain = tail_on ? 1.4142135623730951f : 1.f;
            const int cl0 = wm * 32 + 16 * q + 4 * half;  // row j = 8 q + jj sits at channel cl0 + (jj & 3) + 8 * (jj >> 2)
            float* obase = (partial ? p.slab + (int64_t)blockIdx.y * p.B * p.Cout * HW : p.out) +
                           ((int64_t)ob * p.Cout + o0 + cl0) * HW + oh * p.W + ow;
            const float* xin = xch + (1 - q) * 32 * 64 + lane;
            float y[8][4], dd[8], bbv[8], nz[4];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[jj][e] = xin[(jj * 4 + e) * 64];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int cl = cl0 + (jj & 3) + 8 * (jj >> 2);
                dd[jj] = Dl[tn * WMBLK + cl];
                bbv[jj] = Bl[cl];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) nz[e] = Nl[t * 4 + e];
            auto finalise = [&](auto checked) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float val = (mine[jj][e] + y[jj][e]) * dd[jj];
                        val += nz[e];
                        val += bbv[jj];
                        y[jj][e] = fmaxf(val, val * slope) * gain;  // = (val > 0 ? val : 0.2 val) * sqrt(2) when the tail is on
                    }
                    const int ro = (jj & 3) + 8 * (jj >> 2);
                    if (!decltype(checked)::value || o0 + cl0 + ro < p.Cout) {
                        float* oc = obase + ro * HW;
#ifdef SIS_WINO_NOSTORE  // ablation: the epilogue without its global stores (results wrong)
                        if (y[jj][0] + y[jj][1] + y[jj][2] + y[jj][3] != 12345.678f) continue;
#endif
                        *reinterpret_cast<float2*>(oc) = make_float2(y[jj][0], y[jj][1]);
                        *reinterpret_cast<float2*>(oc + p.W) = make_float2(y[jj][2], y[jj][3]);
                    }
                }
            };
            if (o0 + WMBLK <= p.Cout) finalise(std::false_type());
            else finalise(std::true_type());
        }
        if (!has_next) break;
        WINO_TRACE_TILE(6);
        __syncthreads();  // everyone is done with the exchange area and with this tile's tail operands
        WINO_TRACE_TILE(7);
        tail_store();
        style_store();  // (the old rows' last readers were this tile's transforms)
        T = Tn;
        __syncthreads();  // tail operands and style rows of the next tile visible
        WINO_TRACE_TILE(2);
    }
}

}  // namespace

#ifdef SIS_WINO_TRACE
extern "C" int sis_wino_trace_read(unsigned int* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sis_wino_trace), sizeof(unsigned int) * TR_NWG * 8 * TR_CHUNKS * TR_SLOTS);
}
extern "C" int sis_wino_trace_tile_read(unsigned int* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sis_wino_trace_tile), sizeof(unsigned int) * TR_NWG * 8 * 16 * 8);
}
#endif

extern "C" int sis_modconv_prepack_wino(float* u, const float* w, int cout, int cin, void* stream) {
    SIS_REQUIRE(u && w, "sis_modconv_prepack_wino: null pointer");
    SIS_REQUIRE(cout > 0 && cin > 0, "sis_modconv_prepack_wino: bad sizes");
    hipLaunchKernelGGL(wino_prepack_kernel<false>, dim3(sis_cdiv((int64_t)cout * cin, 256)), dim3(256), 0,
                       (hipStream_t)stream, u, w, cout, cin);
    SIS_CHECK_LAUNCH("sis_modconv_prepack_wino");
    return 0;
}

extern "C" int sis_conv3x3_prepack(float* u, const float* w, int cout, int cin, int adjoint, void* stream) {
    SIS_REQUIRE(u && w, "sis_conv3x3_prepack: null pointer");
    SIS_REQUIRE(cout > 0 && cin > 0, "sis_conv3x3_prepack: bad sizes");
    const dim3 grid(sis_cdiv((int64_t)cout * cin, 256));
    if (adjoint)
        hipLaunchKernelGGL(wino_prepack_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, u, w, cout, cin);
    else
        hipLaunchKernelGGL(wino_prepack_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, u, w, cout, cin);
    SIS_CHECK_LAUNCH("sis_conv3x3_prepack");
    return 0;
}

extern "C" int sis_conv3x3_prepack_both(float* u, float* u_adjoint, const float* w, int cout, int cin, void* stream) {
    SIS_REQUIRE(u && u_adjoint && w, "sis_conv3x3_prepack_both: null pointer");
    SIS_REQUIRE(cout > 0 && cin > 0, "sis_conv3x3_prepack_both: bad sizes");
    hipLaunchKernelGGL(wino_prepack_both_kernel, dim3(sis_cdiv((int64_t)cout * cin, 256), 2), dim3(256), 0, (hipStream_t)stream, u,
                       u_adjoint, w, cout, cin);
    SIS_CHECK_LAUNCH("sis_conv3x3_prepack_both");
    return 0;
}

extern "C" int sis_conv3x3_prepack_multi(const void* table, int n_layers, int total_blocks, void* stream) {
    if (n_layers <= 0 || total_blocks <= 0) return 0;
    SIS_REQUIRE(table, "sis_conv3x3_prepack_multi: null table");
    hipLaunchKernelGGL(wino_prepack_multi_kernel, dim3(total_blocks, 2), dim3(256), 0, (hipStream_t)stream, (const long long*)table, n_layers);
    SIS_CHECK_LAUNCH("sis_conv3x3_prepack_multi");
    return 0;
}

void modconv_splitk_finish_launch(const ConvParams& p, hipStream_t st);

// Returns 0 / 1 like the other launchers, -1 when the shape is not eligible (odd sizes, unaligned, tiny Cin).
int modconv_wino_launch(ConvParams& p, hipStream_t st, void* workspace, int64_t workspace_bytes, bool plan_only) {
    if (p.H % 2 || p.W % 4 || p.Cin % WCC != 0 || !p.cout_vec4 || (((uintptr_t)p.x | (uintptr_t)p.wpk | (uintptr_t)p.out) & 15))
        return -1;
    if (p.fuse && p.noise && (((uintptr_t)p.noise & 3) != 0)) return -1;
    const TileClass& tc = p.cls[0];
    if (p.ncls != 1 || tc.th_log2 < 1 || tc.tw_log2 < 2 || tc.xt > WNTHR * 4) return -1;  // xt: padded-row tile, see mc_add_class ext
    // split-K plan (same rule as the direct kernel, 64-channel blocks)
    p.ksplit = 1; p.kchunk = p.Cin; p.slab = nullptr;
    const int64_t blocks = (int64_t)p.npos_tiles * sis_cdiv(p.Cout, WMBLK);
    if (blocks < 256 && p.Cin >= 4 * WCC && workspace) {
        int want = (int)((384 + blocks - 1) / blocks);
        const int max_split = p.Cin / (2 * WCC);
        if (want > max_split) want = max_split;
        const int64_t out_bytes = (int64_t)p.B * p.Cout * p.OH * p.ORS * 4;
        if ((int64_t)want * out_bytes > workspace_bytes) want = (int)(workspace_bytes / out_bytes);
        if (want >= 2) {
            p.kchunk = sis_cdiv(sis_cdiv(p.Cin, want), WCC) * WCC;
            p.ksplit = sis_cdiv(p.Cin, p.kchunk);
            p.slab = (float*)workspace;
        }
    }
    const size_t lds = (size_t)(2 * WCC * 16 * WMBLK + 2 * WCC * tc.xt + (p.s ? p.nb_max * p.Cin : 0)) * sizeof(float);
    const size_t lds2 = lds + (size_t)(2 * 16 * WCC * WTILES + p.nb_max * WMBLK + WMBLK + WTILES * 4) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    if (plan_only) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_wino_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return sis_fail("modconv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    static const bool pipelined = !(getenv("SIS_WINO_PIPE") && getenv("SIS_WINO_PIPE")[0] == '0');
    if (pipelined && lds2 <= 160 * 1024 && tc.xt >= 256) {
        sis_kernel_name = "modconv_wino2_kernel";
        // pixel tiles per workgroup: as many as keep >= 1024 workgroups (4 per CU) in flight
        static const int tpw_cap = getenv("SIS_WINO_TPW") ? atoi(getenv("SIS_WINO_TPW")) : 16;
        int tpw = 1;
        static const int min_wg = getenv("SIS_WINO_MINWG") ? atoi(getenv("SIS_WINO_MINWG")) : 1024;
        // (the tile-to-tile pipeline stages the next tile from the last two chunks and keeps the buffer parity: an even number of
        // chunks per tile, otherwise one tile per workgroup)
        while (p.kchunk % (2 * WCC) == 0 && tpw * 2 <= tpw_cap && p.npos_tiles % (tpw * 2) == 0 && blocks / (tpw * 2) >= min_wg) tpw *= 2;
        // all co-blocks of a pixel tile on one XCD when the whole transformed weight tensor stays L2-resident there
        static const double swz_mb = getenv("SIS_WINO_XCD_MB") ? atof(getenv("SIS_WINO_XCD_MB")) : 5.0;
        const int64_t grid = blocks / tpw;
        const int n_co_h = (p.Cout + WMBLK - 1) / WMBLK;
        const int xcd_group = (double)p.Cin * p.Cout * 16 * sizeof(float) <= swz_mb * 1048576.0 && n_co_h > 1 && grid % (8 * n_co_h) == 0;
        typedef void (*kern_t)(const ConvParams, const int, const int, const int);
        static const kern_t table[4][2] = {{modconv_wino2_kernel<1, false>, modconv_wino2_kernel<1, true>},
                                           {modconv_wino2_kernel<2, false>, modconv_wino2_kernel<2, true>},
                                           {modconv_wino2_kernel<4, false>, modconv_wino2_kernel<4, true>},
                                           {modconv_wino2_kernel<8, false>, modconv_wino2_kernel<8, true>}};
        static bool table_attr[4][2] = {};
        const int xi_log2 = tc.xt <= 256 ? 0 : tc.xt <= 512 ? 1 : tc.xt <= 1024 ? 2 : 3;
        const kern_t kern = table[xi_log2][p.s ? 1 : 0];
        bool& have_attr = table_attr[xi_log2][p.s ? 1 : 0];
        if (!have_attr) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return sis_fail("modconv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            have_attr = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid, p.ksplit), dim3(WNTHR), lds2, st, p, tc.xt, tpw, xcd_group);
    } else {
        sis_kernel_name = "modconv_wino_kernel";
        hipLaunchKernelGGL(modconv_wino_kernel, dim3((unsigned)blocks, p.ksplit), dim3(WNTHR), lds, st, p, tc.xt);
    }
    SIS_CHECK_LAUNCH("modconv_wino_kernel");
    if (p.ksplit > 1) modconv_splitk_finish_launch(p, st);
    return 0;
}
