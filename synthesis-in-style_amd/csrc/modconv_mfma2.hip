// Fast path of the modulated convolution: LDS-DMA (global_load_lds) staging, double-buffered
// LDS, one barrier per K chunk, optional split-K for the launch-starved low-resolution layers.
//
// Same GEMM mapping as modconv_mfma.hip (see its header); what changes is how operands arrive:
//   * weights   global_load_lds_dwordx4: one wave instruction moves two 512-byte rows
//               [ci][tap][co0..co0+127] straight into LDS (no VGPRs, no ds_write);
//   * input     global_load_lds_dwordx4, EXEC-masked, over the 16-byte aligned superset of the halo'd tile
//               (W % 4 == 0: every aligned float4 is wholly inside or outside the image); out-of-image
//               chunks are never written, they keep the zeros stored once before the K loop (the set of
//               padded positions of a tile is the same for every channel chunk).  Dword-granular DMA of
//               the exact halo cost the transposed kernel 22 % (3x the DMA instructions);
//   * style     s[b, :] of the tile's samples sits in LDS for the whole kernel and multiplies
//               the B operand after its ds_read (one v_mul per operand, hidden under the MFMAs),
//               so staging is a pure copy and needs no arithmetic.
// While chunk c is multiplied out of buffer c&1, the DMA for chunk c+1 fills the other buffer;
// `__syncthreads()` (s_waitcnt vmcnt(0) + s_barrier) at the end of the chunk both retires this
// wave's DMA and frees the buffer just read.  Two workgroups per CU (<= 256 VGPRs, ~50 KB LDS)
// still overlap each other's barriers.
//
// Split-K (ksplit > 1): blockIdx.y owns a slice of input channels and writes raw partial sums to
// slab[slice]; modconv_splitk_finish adds the slices in fixed order (bitwise reproducible, no
// atomics) and applies the epilogue.  Used when position-tiles x oc-blocks < ~2 per CU (4^2..16^2).
#include <type_traits>
#include "modconv_common.h"

namespace {

template <int MODE>
struct V2 {
    static constexpr int CC = MODE == 0 ? 4 : 8;  // input channels per chunk
};

// Wave layout of a workgroup: WM x WN waves, each owning MT x NT 32x32 accumulator tiles
// (MODE 1: NT = 1 and the 4 output phases take the place of the N tiles).
template <int MODE, int KS, int WM_, int WN_, int MT_, int NT_, int OCC_, int CC_ = V2<MODE>::CC>
struct V2Cfg {
    static constexpr int CC = CC_;
    static constexpr int NTAPS = KS * KS;
    static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_, OCC = OCC_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int MBLK = 32 * WM * MT;
    static constexpr int NPOS = 32 * WN * NT;
    static constexpr int NACC = MODE == 0 ? NT : 4;
    static constexpr int PAD_LO = MODE == 0 ? KS / 2 : 1;
    static constexpr int EXT = MODE == 0 ? KS - 1 : 1;
    static constexpr int XI = (512 + THREADS - 1) / THREADS;  // float4 chunks per lane per channel (xt <= 2048)
};

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// LDS-DMA through a buffer descriptor (see modconv_wino.hip): per-lane byte offset in a VGPR, wave-uniform offset in an
// SGPR -- the per-chunk address arithmetic is scalar -- and out-of-range lanes get zeros written to their LDS slot.
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ void bufld16(__amdgpu_buffer_rsrc_t r, float* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

#ifdef SIS_V2_TRACE
// Development build only (tools/v2_trace.py): cycle stamps of the chunk loop of 4 workgroups of the transposed kernel.
constexpr int V2TR_WG0 = 2000, V2TR_NWG = 4, V2TR_CHUNKS = 64, V2TR_SLOTS = 4;
__device__ unsigned int sis_v2_trace[V2TR_NWG][8][V2TR_CHUNKS][V2TR_SLOTS];
#define V2_TRACE(slot)                                                                                            \
    do {                                                                                                          \
        if (MODE == 1 && blockIdx.y == 0 && blockIdx.x >= V2TR_WG0 && blockIdx.x < V2TR_WG0 + V2TR_NWG && tc_ < V2TR_CHUNKS) { \
            const unsigned int now_ = (unsigned int)__builtin_readcyclecounter();                                 \
            if (lane == 0) sis_v2_trace[blockIdx.x - V2TR_WG0][wave][tc_][slot] = now_;                           \
        }                                                                                                         \
    } while (0)
extern "C" int sis_v2_trace_read(unsigned int* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sis_v2_trace), sizeof(unsigned int) * V2TR_NWG * 8 * V2TR_CHUNKS * V2TR_SLOTS);
}
#else
#define V2_TRACE(slot) do {} while (0)
#endif

template <int MODE, int KS, typename C>
__global__ __launch_bounds__(C::THREADS, C::OCC) void modconv_v2_kernel(const ConvParams p, const int xt_max) {
    constexpr int CC = C::CC;
    constexpr int NTHR = C::THREADS;
    constexpr int XI = C::XI;
    constexpr int WF = CC * C::NTAPS * C::MBLK;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;                   // [2][WF]
    float* Xl = lds + 2 * WF;          // [2][CC * xt]
    float* Sl = Xl + 2 * CC * xt_max;  // [nb][Cin]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / C::WN;
    const int wn = wave % C::WN;
    const int wbase = wave * 64;  // wave-uniform: LDS-DMA destinations are base + lane * size
    // 8-wave layouts stage through buffer descriptors with the DMA instructions dealt evenly over the waves (BUF); the
    // 4-wave layouts keep the flat-address DMA below.
    constexpr bool BUF = NTHR == 512;
    constexpr int NWAVES = NTHR / 64;

    // XCD-aware order: output-channel block fastest.  Workgroups are dealt round-robin over the 8 XCDs, so with
    // n_co = Cout / MBLK in {2,4,8} every XCD keeps working on the same weight slice (<= 2.4 MB: stays in its
    // 4 MiB L2) while the n_co workgroups that share an input tile run at the same time on different XCDs
    // (one HBM read, the rest MALL hits).  Pixel-tile-fastest order measured a 48 % L2 miss rate on this kernel.
    const int n_co = (p.Cout + C::MBLK - 1) / C::MBLK;
    int pt = blockIdx.x / n_co;
    const int o0 = (blockIdx.x % n_co) * C::MBLK;
    int ci_cls = 0;
#pragma unroll
    for (int c = 1; c < MC_MAX_CLS; ++c)
        if (c < p.ncls && pt >= p.cls[c].first_block) ci_cls = c;
    const TileClass tc = p.cls[ci_cls];
    pt -= tc.first_block;
    const int twi = pt % tc.ntw; pt /= tc.ntw;
    const int thi = pt % tc.nth;
    const int bt = pt / tc.nth;
    const int thl = tc.th_log2, twl = tc.tw_log2;
    const int th = 1 << thl, tw = 1 << twl;
    const int b0 = bt * tc.nb, h0 = tc.h0 + (thi << thl), w0 = tc.w0 + (twi << twl);
    constexpr int LP = C::PAD_LO ? 4 : 0;  // staged rows start at w0 - LP: 16-byte aligned superset of the halo
    const int eh = th + C::EXT, ew = LP + ((tw + C::EXT - C::PAD_LO + 3) & ~3);
    const int xt = tc.xt;
    const int HW = p.H * p.W;
    const int k_lo = blockIdx.y * p.kchunk;
    const int k_hi = min(p.Cin, k_lo + p.kchunk);

    // ---- one-time LDS init: zeros under the input tiles, style rows of this tile's samples
    for (int e = tid; e < 2 * CC * xt; e += NTHR) Xl[e] = 0.f;
    {  // four elements per lane and trip, requested together (clamped; a predicated load per trip was one round trip per element)
        const int sl_total = tc.nb * p.Cin;
        for (int e0 = tid; e0 < sl_total; e0 += 4 * NTHR) {
            float sv4[4];
            bool in_batch[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * NTHR, sl_total - 1);
                const int n = e / p.Cin, ci = e - n * p.Cin;
                in_batch[u] = b0 + n < p.B;
                sv4[u] = p.s[(int64_t)min(b0 + n, p.B - 1) * p.Cin + ci];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e0 + u * NTHR < sl_total) Sl[e0 + u * NTHR] = in_batch[u] ? sv4[u] : 0.f;
        }
    }

    // ---- per-lane DMA source offsets for the input tile, one float4 chunk each (-1: outside the image, never
    // written: those positions keep the zeros stored above)
    int st_goff[XI];
    {
        const int ew4 = ew >> 2;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int idx = tid + NTHR * i;
            st_goff[i] = -1;
            if (idx < (xt >> 2)) {
                const int n = idx / (eh * ew4), rem = idx - n * (eh * ew4);
                const int r = rem / ew4, c4 = rem - r * ew4;
                const int b = b0 + n, h = h0 - C::PAD_LO + r, w = w0 - LP + 4 * c4;
                if (b < p.B && h >= 0 && h < p.H && w >= 0 && w < p.W) st_goff[i] = b * p.Cin * HW + h * p.W + w;
            }
        }
    }
    // weight DMA: float4 e = it*256 + tid of the chunk's [CC*NTAPS][MBLK] slab
    constexpr int WV4 = WF / 4, WIT = (WV4 + NTHR - 1) / NTHR;
    int w_goff[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int e = it * NTHR + tid;
        const int row = e / (C::MBLK / 4), q = e - row * (C::MBLK / 4);
        w_goff[it] = (e < WV4 && o0 + q * 4 < p.Cout) ? row * p.Cout + o0 + q * 4 : -1;
    }

    // BUF: input tile = xt / 4 float4 per channel in `xparts` parts of 64 lanes (power of two <= 8); wave w moves part
    // w % xparts of the channels w / xparts + k * (8 / xparts).  One offset register, the same DMA count for every wave
    // (with all input DMA on the first waves their SIMD finished each chunk last and the other three idled at the barrier).
    const int xparts = xt <= 256 ? 1 : xt <= 512 ? 2 : xt <= 1024 ? 4 : 8;
    const int x_part = wave & (xparts - 1), x_ch0 = wave / xparts, x_chstep = NWAVES / xparts;
    const int x_f4 = x_part * 64 + lane;
    const bool x_lane = x_f4 < (xt >> 2);
    unsigned x_voff = BUF_OOB;
    if (BUF && x_lane) {
        const int ew4 = ew >> 2;
        const int n = x_f4 / (eh * ew4), rem = x_f4 - n * (eh * ew4);
        const int r = rem / ew4, c4 = rem - r * ew4;
        const int b = b0 + n, h = h0 - C::PAD_LO + r, w = w0 - LP + 4 * c4;
        if (b < p.B && h >= 0 && h < p.H && w >= 0 && w < p.W) x_voff = (unsigned)(n * p.Cin * HW + h * p.W + w) * 4u;
    }
    const __amdgpu_buffer_rsrc_t x_rsrc = dma_rsrc(p.x + (int64_t)b0 * p.Cin * HW);
    const __amdgpu_buffer_rsrc_t w_rsrc = dma_rsrc(p.wpk + o0);
    constexpr int WROW4 = C::MBLK / 4;  // float4 per weight row
    static_assert(!BUF || ((C::CC * C::NTAPS * C::MBLK / 4) % 64 == 0 && NTHR % WROW4 == 0), "weight chunk must split into whole waves / rows");
    const unsigned w_voff = (o0 + (tid % WROW4) * 4 < p.Cout) ? (unsigned)((tid / WROW4) * p.Cout + (tid % WROW4) * 4) * 4u : BUF_OOB;
    auto stage_buf = [&](int ci0, int buf) {
        float* wdst = Wl + buf * WF + wbase * 4;
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            if (it * NTHR + wbase < WV4)  // wave-uniform (WV4 % 64 == 0)
                bufld16(w_rsrc, wdst + it * NTHR * 4, w_voff, (unsigned)((ci0 * C::NTAPS + it * (NTHR / WROW4)) * p.Cout) * 4u);
        float* xdst = Xl + buf * CC * xt + x_part * 256;
        if (x_lane) {
            for (int j = x_ch0; j < CC; j += x_chstep) bufld16(x_rsrc, xdst + j * xt, x_voff, (unsigned)((ci0 + j) * HW) * 4u);
        }
    };
    auto stage = [&](int ci0, int buf) {
        if (BUF) { stage_buf(ci0, buf); return; }
        const float* wsrc = p.wpk + (int64_t)ci0 * C::NTAPS * p.Cout;
        float* wdst = Wl + buf * WF + wbase * 4;
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            if (w_goff[it] >= 0) glds16(wsrc + w_goff[it], wdst + it * NTHR * 4);
        const float* xsrc = p.x + (int64_t)ci0 * HW;
        float* xdst = Xl + buf * CC * xt + wbase * 4;
#pragma unroll
        for (int j = 0; j < CC; ++j)
#pragma unroll
            for (int i = 0; i < XI; ++i)
                if (st_goff[i] >= 0) glds16(xsrc + st_goff[i] + j * HW, xdst + j * xt + i * NTHR * 4);
    };

    // ---- per-lane operand offsets
    int xo[C::NT], so[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) {
        const int pp = (wn * C::NT + t) * 32 + l31;
        const int n = pp >> (thl + twl), rem = pp & ((1 << (thl + twl)) - 1);
        const int r = rem >> twl, c = rem & (tw - 1);
        xo[t] = n * eh * ew + r * ew + c + (LP - C::PAD_LO) + half * xt;
        so[t] = min(n, tc.nb - 1) * p.Cin + half;
    }
    const int aoff = half * C::NTAPS * C::MBLK + wm * C::MT * 32 + l31;
    // Transposed kernel: a wave whose 32 positions all lie outside its tile's samples (edge classes: the last column holds
    // 64 of the 128 positions of a tile, the corner 4) still stages and keeps the barriers but issues no multiplies -- the
    // matrix pipe of its SIMD goes to the other workgroup on the CU (edge tiles are 5 % of the 64x64 layer, 27 % of 16x16).
    const bool wave_live = MODE == 0 || ((wn * 32) >> (thl + twl)) < tc.nb;

    f32x16 acc[C::MT][C::NACC];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int a = 0; a < C::NACC; ++a)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][a][j] = 0.f;

    // Transposed kernel: the demodulation factors of this lane's 16 * MT output channels are fetched here, under the matrix loop
    // (fetched in the epilogue they were 16 dependent round trips to L2 per tile).
    const bool partial = p.ksplit > 1;
    float dsc[MODE == 1 ? C::MT : 1][16];
    if (MODE == 1) {
        const int n = (wn * 32 + l31) >> (thl + twl);
        const float* db = p.dscale + (int64_t)min(b0 + n, p.B - 1) * p.Cout;
#pragma unroll
        for (int m = 0; m < C::MT; ++m)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = o0 + (wm * C::MT + m) * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
                dsc[m][j] = partial ? 1.f : db[min(co, p.Cout - 1)];
            }
    }

    __syncthreads();  // zeros and style rows are in LDS before any DMA may land on them
    stage(k_lo, 0);
    __syncthreads();  // vmcnt(0) + barrier: chunk 0 landed

    int buf = 0;
#ifdef SIS_V2_TRACE
    int tc_ = 0;
#endif
    // Transposed kernel: the four output phases of a position are (2h, 2w), (2h, 2w+1), (2h+1, 2w), (2h+1, 2w+1).  The edge tile
    // classes (positions h = H or w = W) have no second row / column: their phases' multiplies (3 of 9 per channel pair for the last
    // row or column, 5 of 9 for the corner) are skipped -- a compile-time choice per copy of the chunk loop, so interior tiles run
    // the loop they always ran.  (An edge tile is a whole pass of a wave's MFMA chain behind the last full round of workgroups:
    // 18 % of the 16 x 16 layer's time.)
    auto chunk_loop = [&](auto col1c, auto row1c) {
    constexpr bool COL1 = decltype(col1c)::value, ROW1 = decltype(row1c)::value;
    for (int ci0 = k_lo; ci0 < k_hi; ci0 += CC, buf ^= 1) {
        V2_TRACE(0);
        if (ci0 + CC < k_hi) stage(ci0 + CC, buf ^ 1);
        V2_TRACE(1);
        const float* Wb = Wl + buf * WF;
        const float* Xb = Xl + buf * CC * xt;
        float sv[CC / 2][C::NT];
#pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp)
#pragma unroll
            for (int t = 0; t < C::NT; ++t) sv[cp][t] = Sl[so[t] + ci0 + 2 * cp];

        if (MODE == 0) {
#pragma unroll
            for (int tap = 0; tap < C::NTAPS; ++tap) {
                const int toff = (tap / KS) * ew + (tap % KS);
#pragma unroll
                for (int cp = 0; cp < CC / 2; ++cp) {
                    float a[C::MT], bv[C::NT];
#pragma unroll
                    for (int m = 0; m < C::MT; ++m) a[m] = Wb[(2 * cp * C::NTAPS + tap) * C::MBLK + aoff + m * 32];
#pragma unroll
                    for (int t = 0; t < C::NT; ++t) bv[t] = Xb[2 * cp * xt + xo[t] + toff] * sv[cp][t];
#pragma unroll
                    for (int m = 0; m < C::MT; ++m)
#pragma unroll
                        for (int t = 0; t < C::NT; ++t)
                            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[t], acc[m][t], 0, 0, 0);
                }
            }
        } else if (wave_live) {
#pragma unroll
            for (int cp = 0; cp < CC / 2; ++cp) {
                const float* xb = Xb + 2 * cp * xt + xo[0];
                const float s0 = sv[cp][0];
                const float x_ul = xb[0] * s0, x_u = xb[1] * s0, x_l = xb[ew] * s0, x_c = xb[ew + 1] * s0;
#pragma unroll
                for (int m = 0; m < C::MT; ++m) {
                    const float* wb = Wb + 2 * cp * C::NTAPS * C::MBLK + aoff + m * 32;
                    float a[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) a[t] = wb[t * C::MBLK];
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x_c, acc[m][0], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[6], x_u, acc[m][0], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], x_l, acc[m][0], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8], x_ul, acc[m][0], 0, 0, 0);
                    if constexpr (COL1) {
                        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], x_c, acc[m][1], 0, 0, 0);
                        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[7], x_u, acc[m][1], 0, 0, 0);
                    }
                    if constexpr (ROW1) {
                        acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], x_c, acc[m][2], 0, 0, 0);
                        acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[5], x_l, acc[m][2], 0, 0, 0);
                    }
                    if constexpr (COL1 && ROW1) acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4], x_c, acc[m][3], 0, 0, 0);
                }
            }
        }
        V2_TRACE(2);
        __syncthreads();  // next chunk's DMA retired (vmcnt 0) and everyone is done with this buffer
#ifdef SIS_V2_TRACE
        ++tc_;
#endif
    }
    };
    {
        const bool col1 = MODE == 0 || tc.w1 <= p.W, row1 = MODE == 0 || tc.h1 <= p.H;  // (MODE 1: does the class have 2w+1 / 2h+1 outputs)
        if (MODE == 0 || (col1 && row1)) chunk_loop(std::true_type(), std::true_type());
        else if (row1) chunk_loop(std::false_type(), std::true_type());
        else if (col1) chunk_loop(std::true_type(), std::false_type());
        else chunk_loop(std::false_type(), std::false_type());
    }

    // ---- epilogue (ksplit > 1: raw partial sums to this slice's slab)
    if (MODE == 0) {
        float nw = 0.f;
        if (!partial && p.fuse && p.noise) nw = p.noise_w[0];
        float* obase = partial ? p.slab + (int64_t)blockIdx.y * p.B * p.Cout * HW : p.out;
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const int pp = (wn * C::NT + t) * 32 + l31;
            const int n = pp >> (thl + twl), rem = pp & ((1 << (thl + twl)) - 1);
            const int b = b0 + n, h = h0 + (rem >> twl), w = w0 + (rem & (tw - 1));
            if (n >= tc.nb || b >= p.B || h >= p.H || w >= p.W) continue;
            float nz = 0.f;
            if (!partial && p.fuse && p.noise) nz = nw * p.noise[(int64_t)b * p.noise_bstride + h * p.W + w];
            float* ob = obase + (int64_t)b * p.Cout * HW + h * p.W + w;
            const float* db = p.dscale + (int64_t)b * p.Cout;
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int co = o0 + (wm * C::MT + m) * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
                    if (co < p.Cout) {
                        float v = acc[m][t][j];
                        if (!partial) {
                            v *= db[co];
                            if (p.fuse) {
                                v += nz;
                                if (p.bias) v += p.bias[co];
                                v = (v > 0.f ? v : v * 0.2f) * 1.4142135623730951f;
                            }
                        }
                        ob[(int64_t)co * HW] = v;
                    }
                }
        }
    } else {
        const int pp = wn * 32 + l31;
        const int n = pp >> (thl + twl), rem = pp & ((1 << (thl + twl)) - 1);
        const int b = b0 + n, h = h0 + (rem >> twl), w = w0 + (rem & (tw - 1));
        if (n < tc.nb && b < p.B && h < tc.h1 && w < tc.w1) {
            const int OHW = p.OH * p.ORS;
            float* ob = (partial ? p.slab + (int64_t)blockIdx.y * p.B * p.Cout * OHW : p.out) + (int64_t)b * p.Cout * OHW;
            const bool pair = (p.ORS & 1) == 0 && 2 * w + 1 < p.OW;  // both column phases valid, 8-byte aligned
            // Interior tiles (all four phases of every position exist, rows 8-byte aligned) of a full channel block: two 8-byte
            // stores per channel at the tile's base plus a multiple of the plane stride, no per-element checks.
            if (tc.h1 <= p.H && tc.w1 <= p.W && (p.ORS & 1) == 0 && o0 + C::MBLK <= p.Cout) {
                float* o00 = ob + (int64_t)(o0 + wm * C::MT * 32 + 4 * half) * OHW + (int64_t)(2 * h) * p.ORS + 2 * w;
#pragma unroll
                for (int m = 0; m < C::MT; ++m)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        float* oc = o00 + (m * 32 + (j & 3) + 8 * (j >> 2)) * OHW;
                        const float d = dsc[m][j];
                        *reinterpret_cast<float2*>(oc) = make_float2(acc[m][0][j] * d, acc[m][1][j] * d);
                        *reinterpret_cast<float2*>(oc + p.ORS) = make_float2(acc[m][2][j] * d, acc[m][3][j] * d);
                    }
                return;
            }
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int co = o0 + (wm * C::MT + m) * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
                    if (co < p.Cout) {
                        const float d = dsc[m][j];
                        float* oc = ob + (int64_t)co * OHW;
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const int oy = 2 * h + a;
                            if (oy >= p.OH) continue;
                            float* orow = oc + (int64_t)oy * p.ORS + 2 * w;
                            if (pair) {
                                *reinterpret_cast<float2*>(orow) = make_float2(acc[m][2 * a][j] * d, acc[m][2 * a + 1][j] * d);
                            } else {
                                orow[0] = acc[m][2 * a][j] * d;
                                if (2 * w + 1 < p.OW) orow[1] = acc[m][2 * a + 1][j] * d;
                            }
                        }
                    }
                }
        }
    }
}

// Adds the K slices in slice order and applies demodulation (+ noise, bias, leaky-ReLU when fused).
__global__ __launch_bounds__(256) void modconv_splitk_finish(const ConvParams p, int64_t plane_elems, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t ohw = (int64_t)p.OH * p.ORS;
    float nw = 0.f;
    if (p.fuse && p.noise) nw = p.noise_w[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        // Eight slices requested at a time, added in slice order (one load per trip was one L2 round trip per slice: 20 us for the
        // 4 x 4 layer's 32 slices).  Past the last slice a lane re-reads the last one and does not add it.
        float v = 0.f;
        for (int k = 0; k < p.ksplit; k += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p.slab[(int64_t)min(k + u, p.ksplit - 1) * plane_elems + i];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k + u < p.ksplit) v += t[u];
        }
        const int64_t bc = i / ohw, hw = i - bc * ohw;
        const int b = (int)(bc / p.Cout), co = (int)(bc - (int64_t)b * p.Cout);
        if (p.dscale) v *= p.dscale[bc];
        if (p.fuse) {
            if (p.noise) v += nw * p.noise[(int64_t)b * p.noise_bstride + hw];
            if (p.bias) v += p.bias[co];
            v = (v > 0.f ? v : v * 0.2f) * 1.4142135623730951f;
        }
        p.out[i] = v;
    }
}

template <int MODE, int KS, typename C>
int launch_v2(ConvParams& p, hipStream_t st) {
    constexpr int CC = C::CC;
    int xt_max = 0;
    for (int c = 0; c < p.ncls; ++c) xt_max = p.cls[c].xt > xt_max ? p.cls[c].xt : xt_max;
    if (xt_max > C::THREADS * C::XI * 4) return -1;
    const size_t lds = (size_t)(2 * CC * C::NTAPS * C::MBLK + 2 * CC * xt_max + p.nb_max * p.Cin) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_v2_kernel<MODE, KS, C>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return sis_fail("modconv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int64_t bx = (int64_t)p.npos_tiles * sis_cdiv(p.Cout, C::MBLK);
    SIS_REQUIRE(bx > 0 && bx < ((int64_t)1 << 31), "modconv: bad grid");
    sis_kernel_name = MODE == 1 ? "modconv_v2_kernel<1, 3>" : KS == 3 ? "modconv_v2_kernel<0, 3>" : "modconv_v2_kernel<0, 1>";
    SIS_OCC_REPORT((modconv_v2_kernel<MODE, KS, C>), C::THREADS, lds);
    hipLaunchKernelGGL((modconv_v2_kernel<MODE, KS, C>), dim3((unsigned)bx, p.ksplit), dim3(C::THREADS), lds, st, p, xt_max);
    SIS_CHECK_LAUNCH("modconv_v2_kernel");
    if (p.ksplit > 1) {
        const int64_t total = (int64_t)p.B * p.Cout * p.OH * p.ORS;
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(modconv_splitk_finish, dim3(blocks), dim3(256), 0, st, p, total, total);
        SIS_CHECK_LAUNCH("modconv_splitk_finish");
    }
    return 0;
}

}  // namespace

void modconv_splitk_finish_launch(const ConvParams& p, hipStream_t st) {
    const int64_t total = (int64_t)p.B * p.Cout * p.OH * p.ORS;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(modconv_splitk_finish, dim3(blocks), dim3(256), 0, st, p, total, total);
}

// Chooses the K split: enough (tile x oc-block x slice) workgroups to give every CU ~2, slices a
// multiple of the chunk size, and the slabs must fit the caller's workspace.
static void plan_splitk(ConvParams& p, int mblk, int cc, int64_t workspace_bytes) {
    p.ksplit = 1; p.kchunk = p.Cin; p.slab = nullptr;
    const int64_t blocks = (int64_t)p.npos_tiles * sis_cdiv(p.Cout, mblk);
    static const int min_blocks = getenv("SIS_SPLITK_MIN_BLOCKS") ? atoi(getenv("SIS_SPLITK_MIN_BLOCKS")) : 384;
    static const int target = getenv("SIS_SPLITK_TARGET") ? atoi(getenv("SIS_SPLITK_TARGET")) : 512;
    if (blocks >= min_blocks || p.Cin < 4 * cc) return;
    int want = (int)((target + blocks - 1) / blocks);
    const int max_split = p.Cin / (2 * cc);
    if (want > max_split) want = max_split;
    const int64_t out_bytes = (int64_t)p.B * p.Cout * p.OH * p.ORS * 4;
    if ((int64_t)want * out_bytes > workspace_bytes) want = (int)(workspace_bytes / out_bytes);
    if (want < 2) return;
    int kchunk = sis_cdiv(sis_cdiv(p.Cin, want), cc) * cc;
    p.kchunk = kchunk;
    p.ksplit = sis_cdiv(p.Cin, kchunk);
}

// Wave layouts (SIS_CONV_CFG=<mode0><mode1> digits select experiments; default "11": the 8-wave layouts,
// 4 waves per SIMD, measured 3-10 % faster than the 4-wave ones: more waves to cover barrier skew).
typedef V2Cfg<0, 3, 2, 2, 2, 4, 2> Conv3A;   // 256 thr, 128 co x 256 px, wave 64x128, 2 WG/CU
typedef V2Cfg<0, 3, 2, 4, 2, 2, 4> Conv3B;   // 512 thr, 128 co x 256 px, wave 64x64,  2 WG/CU (4 waves/SIMD)
typedef V2Cfg<0, 1, 2, 2, 2, 4, 2> Conv1A;
typedef V2Cfg<1, 3, 1, 4, 2, 1, 2> UpA;      // 256 thr, 64 co x 128 pos, wave 64 co x 32 pos x 4 phases
typedef V2Cfg<1, 3, 2, 4, 1, 1, 4> UpB;      // 512 thr, 64 co x 128 pos, wave 32 co x 32 pos x 4 phases
typedef V2Cfg<1, 3, 1, 8, 2, 1, 2> UpC;      // 512 thr, 64 co x 256 pos, wave 64 co x 32 pos x 4 phases

int modconv_v2_tile(int mode, int* mblk, int* npos) {
    static int cfg = -1;
    if (cfg < 0) {
        const char* e = getenv("SIS_CONV_CFG");
        cfg = (e && e[0] >= '0' && e[0] <= '9' && e[1] >= '0' && e[1] <= '9') ? (e[0] - '0') * 10 + (e[1] - '0') : 11;
    }
    const int c = mode == 0 ? cfg / 10 : cfg % 10;
    if (mode == 0) { *mblk = 128; *npos = 256; }
    else { *mblk = 64; *npos = c == 2 ? 256 : 128; }
    return c;
}

int modconv_v2_launch(ConvParams& p, int mode, int ks, hipStream_t st, void* workspace, int64_t workspace_bytes) {
    const int cc = mode == 0 ? V2<0>::CC : V2<1>::CC;  // (16-channel chunks for the transposed kernel measured 2x slower)
    if (p.Cin % cc != 0 || !p.cout_vec4 || (((uintptr_t)p.x | (uintptr_t)p.wpk) & 15) != 0 || p.W % 4 != 0) return -1;
    for (int c = 0; c < p.ncls; ++c)
        if (p.cls[c].w0 % 4 != 0) return -1;
    mc_set_padded_xt(p, mode == 0 ? ks / 2 : 1, mode == 0 ? ks - 1 : 1);
    int mblk, npos;
    const int c = modconv_v2_tile(mode, &mblk, &npos);
    plan_splitk(p, mblk, cc, workspace ? workspace_bytes : 0);
    if (p.ksplit > 1) p.slab = (float*)workspace;
    if (mode == 0 && ks == 3) return c == 1 ? launch_v2<0, 3, Conv3B>(p, st) : launch_v2<0, 3, Conv3A>(p, st);
    if (mode == 0 && ks == 1) return launch_v2<0, 1, Conv1A>(p, st);
    if (mode == 1 && ks == 3)
        return c == 1 ? launch_v2<1, 3, UpB>(p, st) : c == 2 ? launch_v2<1, 3, UpC>(p, st) : launch_v2<1, 3, UpA>(p, st);
    return -1;
}
