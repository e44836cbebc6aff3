// Transposed stride-2 3x3 modulated convolution (StyleGAN2's up-convolutions, reference networks/stylegan2/model.py:251-262:
// conv_transpose2d(stride 2) on per-sample weights) with 25 instead of 36 multiplies per 2 x 2 input positions: the fast-FIR
// form of the four output phases.
//
// Per axis, position h contributes out[2h] = g0 x[h] + g2 x[h-1] (a 2-tap FIR over the positions) and out[2h+1] = g1 x[h]
// (csrc/modconv_mfma2.hip computes exactly these, 3 multiplies per position and axis, 9 per position in 2-D).  For a PAIR of
// positions 2b, 2b+1 with d0 = x[2b-1], d1 = x[2b], d2 = x[2b+1] the two even outputs are F(2,2) of a 2-tap filter:
//     m0 = (d0 - d1) g2     m1 = d1 (g0 + g2)     m2 = (d2 - d1) g0        out[4b] = m0 + m1,  out[4b+2] = m1 + m2
//     m3 = d1 g1            m4 = d2 g1                                     out[4b+1] = m3,     out[4b+3] = m4
// 5 products instead of 6; in two dimensions 25 instead of 36 per 2 x 2 block of positions (-30.6 % matrix work), with
// 4 x 4 = 16 distinct transformed weight planes U[u][v] = sum_{ky in K(u), kx in K(v)} W[ky][kx], K = ({2}, {0, 2}, {0}, {1}),
// prepacked once per checkpoint, and a 4 x 4 data transform T = (row transform) (column transform) of the block's 3 x 3
// input patch, transform t = (d0 - d1, d1, d2 - d1, d2): 14 subtractions, done per lane in registers on the way from the
// LDS to the MFMA (no transformed image).  Product (p, q), p, q = 0..4, multiplies plane (pu[p], pu[q]) with
// T[dp[p]][dp[q]], pu = (0, 1, 2, 3, 3), dp = (0, 1, 2, 1, 3); the 4 x 4 outputs of a block are
// out[r][s] = sum_{p in R(r), q in R(s)} acc[p][q], R = ({0, 1}, {3}, {1, 2}, {4}).
//
// GEMM view: M = output channels, N = 2 x 2 position BLOCKS, K = input channels, 25 accumulator planes.  25 planes of a
// 32 x 32 tile (400 registers) leave one wave per SIMD; on v_mfma_f32_16x16x4_f32 (same rate: 64 FLOP/clk/SIMD, exact fp32) a
// wave holds 32 channels x 16 blocks x 25 planes in 200 registers and two waves share a SIMD.  Workgroup = 4 waves =
// 64 channels x 32 blocks (128 positions), 4 input channels per chunk, 58 KB of LDS: TWO workgroups per CU, one wave of each
// on every SIMD, so that one's barrier, patch transform, fragment reads, prologue and epilogue fall under the other's MFMAs
// (the 8-wave / 64-block / 8-channel tile, SIS_UPFIR_WAVES=8, has both waves of a SIMD behind the same barrier: 5.22 -> 4.94 ms
// on the three large layers).  Lane l of a wave: block l & 15, input channel (l >> 4) of the 4-deep MFMA step.
//
// Blocks are numbered row-major over (sample, block row, block column) -- (H/2 + 1) x (W/2 + 1) per sample: the transposed
// convolution has H + 1 position rows, the last block row / column holds one valid position -- and a workgroup takes 64
// CONSECUTIVE blocks (no tile classes: 0.2 % of the blocks of a 64 x 64 layer are padding).  Its input tile is the run of
// image rows those blocks touch, each staged whole as [4 zeros | W pixels | 4 zeros] by LDS-DMA through a buffer descriptor
// whose out-of-range lanes write the zeros (rows -1 and H, H + 1 and the pad columns alike); "virtual" row v of sample b,
// v = h + 1 in [0, H + 3), has index b (H + 3) + v, so a run that crosses from one sample into the next is still one range.
// Staging: one input channel per wave and chunk, wave w moves channel w (4 pieces of weights: 16 planes x 64 channels, and
// <= 4 pieces of input rows), double-buffered, one barrier per chunk (as modconv_mfma2.hip).  The style factor s[b, ci] multiplies the
// raw patch values after their LDS read; demodulation in the epilogue; noise / bias / activation belong to the blur kernel
// that reads this kernel's (2H+1) x (2W+4)-strided result.
#include <type_traits>
#include "modconv_common.h"

namespace {

typedef float uf_f32x4 __attribute__((ext_vector_type(4)));

constexpr int UF_MBLK = 64;                           // output channels per workgroup
// NW = waves per workgroup (8 or 4): NW input channels per chunk (one channel's DMA per wave) and 8 NW position blocks per
// workgroup.  NW = 4 halves the tile and the chunk (58 KB of LDS): TWO workgroups share a CU, one wave of each per SIMD, and
// one's barrier / transform / epilogue falls under the other's MFMAs.
constexpr int UF_PLANES = 16;
constexpr int UF_WROW = UF_PLANES * UF_MBLK + 32;     // floats per channel row of the weight stage [8 plane pairs][64 co][2]:
                                                      // +32 puts the two channels a 32-lane group of a ds_read_b64 touches 32
                                                      // banks (of 64) apart: conflict-free
constexpr int UF_XP_MAX = 4;                          // 1 KiB pieces per channel of the input tile (xs <= 1024 floats)
constexpr unsigned UF_OOB = 0x80000000u;

struct UpFirParams {
    const float* x; const float* u; const float* s; const float* dscale; float* out;
    int B, Cin, Cout, H, W, OH, ORS;
    int nbw, bps, total_blocks;    // block columns per row, blocks per sample, blocks in the batch
    int vr;                        // virtual rows per sample = H + 3
    int rw4;                       // float4 per staged row = W / 4 + 2
    int xs;                        // floats per channel of the staged tile (multiple of 256)
};

__device__ __forceinline__ void uf_dma16(__amdgpu_buffer_rsrc_t r, float* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

// ABL: timing ablations for development builds (tools/bench_upfir.py, SIS_UPFIR_ABL): bit 0 no barrier in the chunk loop, bit 1 no
// DMA in the loop, bit 2 no patch reads / transform in the loop, bit 3 no weight-fragment reads in the loop.  Any non-zero value
// computes WRONG results; the shipped kernel is ABL = 0.
template <int PIPE, int ABL = 0, int NW = 8>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void modconv_upfir_kernel(const UpFirParams p) {
    constexpr int UF_CC = NW, UF_NBLK = 8 * NW, UF_THREADS = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;                              // [2][CC][WROW]
    float* Xl = Wl + 2 * UF_CC * UF_WROW;         // [2][CC][xs]
    float* Sl = Xl + 2 * UF_CC * p.xs;            // [2 samples][Cin]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;

    // output-channel block fastest: the n_co workgroups that share an input tile run at the same time on different XCDs
    const int n_co = p.Cout / UF_MBLK;
    const int tile = blockIdx.x / n_co, o0 = (blockIdx.x % n_co) * UF_MBLK;
    const int g0 = tile * UF_NBLK;
    const int b_first = g0 / p.bps, bh_first = (g0 % p.bps) / p.nbw;
    const int R0 = b_first * p.vr + 2 * bh_first;     // first virtual row of the tile (position row 2 bh - 1)
    const int RW = 4 * p.rw4;
    const int HW = p.H * p.W;

    // ---- this lane's block
    const int g = g0 + wn * 16 + l15;
    const bool live = g < p.total_blocks;
    const int gc = live ? g : p.total_blocks - 1;
    const int b = gc / p.bps, rem = gc % p.bps, bh = rem / p.nbw, bw = rem % p.nbw;
    const int xoff = (b * p.vr + 2 * bh - R0) * RW + 3 + 2 * bw;    // patch element (0, 0): row 2 bh - 1, column 2 bw - 1 (+ 4 pad)
    const int soff = (b - b_first) * p.Cin;

    // ---- style rows of the (at most two) samples of this tile
    for (int e = tid; e < 2 * p.Cin; e += UF_THREADS) {
        const int n = e / p.Cin, ci = e - n * p.Cin;
        Sl[e] = p.s[(int64_t)min(b_first + n, p.B - 1) * p.Cin + ci];
    }

    // ---- DMA plan.  Weights: wave w moves channel ci0 + w, piece q = planes 4 q .. 4 q + 3 (lane >> 4) x 64 channels.
    // Input: wave w moves channel ci0 + w, piece q = float4 q * 64 + lane of the [rows][rw4] tile.
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.u + 2 * o0), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7FFFFFFF, 0x00020000);
    // weight image of a channel: [plane pair 0..7][co][2 planes]; a 1 KiB piece = two pairs x 64 channels x 2
    const unsigned w_voff = (unsigned)(((lane >> 5) * p.Cout * 2 + (lane & 31) * 4) * 4);
    const int xpieces = p.xs >> 8;
    unsigned x_voff[UF_XP_MAX];
#pragma unroll
    for (int q = 0; q < UF_XP_MAX; ++q) {
        const int i = q * 64 + lane;
        const int row = i / p.rw4, c4 = i - row * p.rw4;
        const int v = R0 + row;
        const int bb = v / p.vr, h = v - bb * p.vr - 1, col = 4 * (c4 - 1);
        const bool ok = q < xpieces && bb < p.B && h >= 0 && h < p.H && col >= 0 && col < p.W;
        x_voff[q] = ok ? (unsigned)((bb * p.Cin * HW + h * p.W + col) * 4) : UF_OOB;
    }
    auto stage = [&](int ci0, int buf) {
        float* wdst = Wl + (buf * UF_CC + wave) * UF_WROW;
        const unsigned wbase = (unsigned)((ci0 + wave) * UF_PLANES * p.Cout * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) uf_dma16(w_rsrc, wdst + q * 256, w_voff, wbase + (unsigned)(q * 2 * p.Cout * 2 * 4));
        float* xdst = Xl + (buf * UF_CC + wave) * p.xs;
        const unsigned xbase = (unsigned)((ci0 + wave) * HW * 4);
#pragma unroll
        for (int q = 0; q < UF_XP_MAX; ++q)
            if (q < xpieces) uf_dma16(x_rsrc, xdst + q * 256, x_voff[q], xbase);   // (wave-uniform)
    };

    uf_f32x4 acc[5][5][2];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[a][c][t] = uf_f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();   // style rows
    stage(0, 0);
    __syncthreads();   // vmcnt(0) + barrier: chunk 0 landed

    if constexpr (PIPE == 4) {
        // Chunks software-pipelined across the barrier (NW == 4, one MFMA step per chunk).  In the plain loop every chunk starts behind
        // its barrier with 10 LDS reads and ~25 VALU of transform before its first MFMA.  Here the barrier of chunk c sits behind the
        // request for its last fragment pair (pair 7, asked for in iteration 5): from there on nobody reads buffer c & 1 any more, chunk
        // c + 1 has landed (its DMA was issued a whole chunk earlier) -- so the DMA of chunk c + 2 goes into the freed buffer and the
        // transform of chunk c + 1 runs in slices between the 20 MFMAs of plane row 3, and the next chunk starts with MFMAs.
        static_assert(NW == 4 && ABL == 0, "pipelined form: 4-wave workgroups, no ablations");
        const int n_chunks = p.Cin / UF_CC;
        if (n_chunks > 1) stage(UF_CC, 1);
        __syncthreads();   // (chunk 1 landed as well: once per workgroup)
        float tA[4][4], tB[4][4], cm0[4], raw[3][3], svn;
        auto tr_read = [&](int c, int bufc) {   // patch and style of chunk c (clamped past the end: values unused)
            const int cl = min(c, n_chunks - 1) * UF_CC + kq;
            svn = Sl[soff + cl];
            const float* xb = Xl + (bufc * UF_CC + kq) * p.xs + xoff;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) raw[r][j] = xb[r * RW + j];
        };
        // column transform of patch row r: rows 1 and 2 ARE rows 1 and 3 of t, row 0 waits in cm0 for the row transform
        auto tr_col = [&](int r, float (&t)[4][4]) {
            const float d0 = raw[r][0] * svn, d1 = raw[r][1] * svn, d2 = raw[r][2] * svn;
            float* o = r == 0 ? cm0 : (r == 1 ? t[1] : t[3]);
            o[0] = d0 - d1; o[1] = d1; o[2] = d2 - d1; o[3] = d2;
            // (pinned where it stands: the results are used in the NEXT chunk only, and the compiler otherwise sinks the arithmetic
            // into that chunk's block, in front of its first MFMA)
            asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
        };
        auto tr_row = [&](float (&t)[4][4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = cm0[j] - t[1][j]; t[2][j] = t[3][j] - t[1][j];
                asm volatile("" : "+v"(t[0][j]), "+v"(t[2][j]));
            }
        };
        tr_read(0, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) tr_col(r, tA);
        tr_row(tA);
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int woff = kq * UF_WROW + (wm * 32 + l15) * 2;
        // ring of three fragment pairs, carried from chunk to chunk: pair 0 of chunk c + 1 takes pair 6's slot once that is used up
        f2 ra[3][2];
        ra[0][0] = *reinterpret_cast<const f2*>(Wl + woff); ra[0][1] = *reinterpret_cast<const f2*>(Wl + woff + 32);
        auto chunk = [&](auto parity, int c, const float (&t)[4][4], float (&tn)[4][4]) {
            constexpr int B = decltype(parity)::value;
            const float* wb = Wl + B * UF_CC * UF_WROW + woff;
            auto fetch = [&](int k) {
                ra[k % 3][0] = *reinterpret_cast<const f2*>(wb + k * 2 * UF_MBLK);
                ra[k % 3][1] = *reinterpret_cast<const f2*>(wb + k * 2 * UF_MBLK + 32);
            };
            fetch(1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k + 2 < 8) fetch(k + 2);
                if (k == 7) {   // (behind the barrier: chunk c + 1 has landed)
                    ra[0][0] = *reinterpret_cast<const f2*>(Wl + (B ^ 1) * UF_CC * UF_WROW + woff);
                    ra[0][1] = *reinterpret_cast<const f2*>(Wl + (B ^ 1) * UF_CC * UF_WROW + woff + 32);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * k + e, u = i >> 2, v = i & 3;
                    const float a0 = ra[k % 3][0][e], a1 = ra[k % 3][1][e];
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            if ((pp && u != 3) || (qq && v != 3)) continue;
                            const int pi = u + pp, qi = v + qq;
                            const int di = pi == 3 ? 1 : (pi == 4 ? 3 : pi), dj = qi == 3 ? 1 : (qi == 4 ? 3 : qi);
                            acc[pi][qi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, t[di][dj], acc[pi][qi][0], 0, 0, 0);
                            acc[pi][qi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, t[di][dj], acc[pi][qi][1], 0, 0, 0);
                        }
                    if (k >= 5) {   // side work of the chunk's tail, one piece behind each MFMA group
                        if (k == 5 && e == 1) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk c + 1 (the compiler does not see them across the back edge)
                            __syncthreads();
                            if (c + 2 < n_chunks) stage((c + 2) * UF_CC, B);
                            tr_read(c + 1, B ^ 1);
                        }
                        if (k == 6 && e == 0) { tr_col(0, tn); tr_col(1, tn); }
                        if (k == 6 && e == 1) tr_col(2, tn);
                        if (k == 7 && e == 0) tr_row(tn);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (k < 5) {
                    if (k + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    const int u = (2 * k) >> 2, vhi = ((2 * k) & 3) + 1;
                    if (u == 3 && vhi == 3) __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                    else if (u == 3) __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    else if (vhi == 3) __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
        };
        for (int c = 0; c < n_chunks; c += 2) {
            chunk(std::integral_constant<int, 0>(), c, tA, tB);
            if (c + 1 < n_chunks) chunk(std::integral_constant<int, 1>(), c + 1, tB, tA);
        }
    } else {
    int buf = 0;
    float t0[4][4] = {}, t1[4][4] = {};
    // PIPE == 3: the SIMD partners (waves w and w + 4) run out of step inside a chunk -- waves 0-3 issue their DMA for the next
    // chunk at its start, waves 4-7 between their two MFMA steps -- so that one partner's staging / transform phase falls under
    // the other's MFMAs instead of both idling the matrix pipe behind the barrier (MI355X_MICROARCH.md, two waves per SIMD,
    // item 9: split roles by wave number >= 4).
    const bool late_dma = PIPE == 3 && wave >= 4;
    for (int ci0 = 0; ci0 < p.Cin; ci0 += UF_CC, buf ^= 1) {
        if (!(ABL & 2) && !late_dma && ci0 + UF_CC < p.Cin) stage(ci0 + UF_CC, buf ^ 1);
        const float* Wb = Wl + buf * UF_CC * UF_WROW;
        const float* Xb = Xl + buf * UF_CC * p.xs;
        // Transform of one MFMA step (4 channels: this lane's is 4 ks + kq): raw 3 x 3 patch x style -> t[4][4]
        auto transform = [&](int ks, float (&t)[4][4]) {
            if constexpr (ABL & 4) {
                if (ci0 != 0) return;   // (the first chunk's values stay in the registers)
            }
            const int cl = 4 * ks + kq;
            const float sv = Sl[soff + ci0 + cl];
            const float* xb = Xb + cl * p.xs + xoff;
            float c[3][4];                                    // column transform of the three patch rows
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float d0 = xb[r * RW] * sv, d1 = xb[r * RW + 1] * sv, d2 = xb[r * RW + 2] * sv;
                c[r][0] = d0 - d1; c[r][1] = d1; c[r][2] = d2 - d1; c[r][3] = d2;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                     // row transform
                t[0][j] = c[0][j] - c[1][j]; t[1][j] = c[1][j]; t[2][j] = c[2][j] - c[1][j]; t[3][j] = c[2][j];
            }
        };
        // The 50 MFMAs of one step.  Planes (u, v) in order, one pair of A fragments each, used by every product (p, q) with
        // pu[p] = u, pu[q] = v (plane row / column 3 serves the products 3 and 4).  PIPE >= 1: the fragments of plane i + 2 are
        // requested before the MFMAs of plane i (a ring of three register pairs, pinned with sched_group_barrier: left to
        // itself the compiler reads each pair right in front of its MFMAs and waits for it -- 32 exposed LDS latencies per chunk).
        auto multiply = [&](int ks, const float (&t)[4][4]) {
            // A fragments: one ds_read2_b64 per PAIR of planes (u, 2 h) / (u, 2 h + 1) delivers both planes for both channel tiles
            typedef float f2 __attribute__((ext_vector_type(2)));
            const float* wb = Wb + (4 * ks + kq) * UF_WROW + (wm * 32 + l15) * 2;
            f2 ra[3][2];   // ring of three pairs x two channel tiles
            auto fetch = [&](int k) {
                if constexpr (ABL & 8) {
                    ra[k % 3][0] = f2{(float)(k + lane), (float)(k - lane)}; ra[k % 3][1] = f2{(float)lane, (float)k};
                } else {
                    ra[k % 3][0] = *reinterpret_cast<const f2*>(wb + k * 2 * UF_MBLK);
                    ra[k % 3][1] = *reinterpret_cast<const f2*>(wb + k * 2 * UF_MBLK + 32);
                }
            };
            if constexpr (PIPE >= 1) { fetch(0); fetch(1); }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (PIPE >= 1) { if (k + 2 < 8) fetch(k + 2); }
                else fetch(k);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * k + e, u = i >> 2, v = i & 3;
                    const float a0 = ra[k % 3][0][e], a1 = ra[k % 3][1][e];
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            if ((pp && u != 3) || (qq && v != 3)) continue;
                            const int pi = u + pp, qi = v + qq;               // product indices: u (or 4 for the second use of row 3)
                            const int di = pi == 3 ? 1 : (pi == 4 ? 3 : pi), dj = qi == 3 ? 1 : (qi == 4 ? 3 : qi);
                            acc[pi][qi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, t[di][dj], acc[pi][qi][0], 0, 0, 0);
                            acc[pi][qi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, t[di][dj], acc[pi][qi][1], 0, 0, 0);
                        }
                }
                if constexpr (PIPE >= 1) {
                    if (k + 2 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // the read of pair k + 2 ...
                    // ... then the MFMAs of pair k: planes (u, v) and (u, v + 1) with v even: 2 + 2, on plane row 3: 4 + 4,
                    // and with plane column 3 in the pair twice as many for that plane
                    const int u = (2 * k) >> 2, vhi = ((2 * k) & 3) + 1;
                    if (u == 3 && vhi == 3) __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                    else if (u == 3) __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    else if (vhi == 3) __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
        };
        if constexpr (NW == 4) {     // one MFMA step per chunk
            transform(0, t0);
            multiply(0, t0);
        } else if constexpr (PIPE == 2) {   // both steps' transforms first: the second one's loads and VALU go under the first step's MFMAs
            transform(0, t0);
            transform(1, t1);
            multiply(0, t0);
            multiply(1, t1);
        } else {
            transform(0, t0);
            multiply(0, t0);
            if (late_dma && ci0 + UF_CC < p.Cin) stage(ci0 + UF_CC, buf ^ 1);
            transform(1, t1);
            multiply(1, t1);
        }
        if constexpr (!(ABL & 1)) __syncthreads();   // next chunk's DMA retired (vmcnt 0) and everyone is done with this buffer
    }
    }

    // ---- epilogue: output transform, demodulation, 16-byte stores of the block's 4 x 4 outputs per channel
    // (PIPE == 4: the block's coordinates are worked out again from the lane number -- the loop has no register to carry them)
    int eg = g;
    if constexpr (PIPE == 4) { eg = g0 + wn * 16 + (int)(threadIdx.x & 15); asm volatile("" : "+v"(eg)); }
    if (eg >= p.total_blocks) return;
    int eb = b, ebh = bh, ebw = bw;
    if constexpr (PIPE == 4) { eb = eg / p.bps; const int erem = eg % p.bps; ebh = erem / p.nbw; ebw = erem % p.nbw; }
    const int OHW = p.OH * p.ORS;
    const float* db = p.dscale + (int64_t)eb * p.Cout;
    float* ob = p.out + (int64_t)eb * p.Cout * OHW + (int64_t)(4 * ebh) * p.ORS + 4 * ebw;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = o0 + wm * 32 + tt * 16 + 4 * kq + r;
            const float d = db[co];
            float cs[5][4];                                   // column sums per product row p
#pragma unroll
            for (int pi = 0; pi < 5; ++pi) {
                cs[pi][0] = acc[pi][0][tt][r] + acc[pi][1][tt][r];
                cs[pi][1] = acc[pi][3][tt][r];
                cs[pi][2] = acc[pi][1][tt][r] + acc[pi][2][tt][r];
                cs[pi][3] = acc[pi][4][tt][r];
            }
            float* oc = ob + (int64_t)co * OHW;
#pragma unroll
            for (int ro = 0; ro < 4; ++ro) {
                if (4 * ebh + ro >= p.OH) continue;
                float4 v;
                if (ro == 0) v = make_float4(cs[0][0] + cs[1][0], cs[0][1] + cs[1][1], cs[0][2] + cs[1][2], cs[0][3] + cs[1][3]);
                else if (ro == 1) v = make_float4(cs[3][0], cs[3][1], cs[3][2], cs[3][3]);
                else if (ro == 2) v = make_float4(cs[1][0] + cs[2][0], cs[1][1] + cs[2][1], cs[1][2] + cs[2][2], cs[1][3] + cs[2][3]);
                else v = make_float4(cs[4][0], cs[4][1], cs[4][2], cs[4][3]);
                *reinterpret_cast<float4*>(oc + ro * p.ORS) = make_float4(v.x * d, v.y * d, v.z * d, v.w * d);
            }
        }
}

// u[ci][plane pair 2 pu + pv / 2][co][pv % 2] = sum_{ky in K(pu), kx in K(pv)} w[co][ci][ky][kx]
__global__ __launch_bounds__(256) void upfir_prepack_kernel(float* __restrict__ u, const float* __restrict__ w, int cout, int cin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // i = ci * cout + co (co fastest: coalesced writes)
    if (i >= (int64_t)cout * cin) return;
    const int co = (int)(i % cout), ci = (int)(i / cout);
    const float* src = w + ((int64_t)co * cin + ci) * 9;
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = src[t];
    // rows: K(0) = {2}, K(1) = {0, 2}, K(2) = {0}, K(3) = {1}
    float rows[4][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        rows[0][kx] = g[2][kx]; rows[1][kx] = g[0][kx] + g[2][kx]; rows[2][kx] = g[0][kx]; rows[3][kx] = g[1][kx];
    }
#pragma unroll
    for (int pu = 0; pu < 4; ++pu) {
        const float v4[4] = {rows[pu][2], rows[pu][0] + rows[pu][2], rows[pu][0], rows[pu][1]};
#pragma unroll
        for (int h = 0; h < 2; ++h)   // plane pair (pu, 2 h), (pu, 2 h + 1): the two planes of a channel side by side
            *reinterpret_cast<float2*>(u + (((int64_t)ci * 8 + pu * 2 + h) * cout + co) * 2) = make_float2(v4[2 * h], v4[2 * h + 1]);
    }
}

bool upfir_plan(UpFirParams& p, int batch, int cin, int cout, int h, int w, int row_stride, size_t* lds_bytes, int nw = 8) {
    const int UF_CC = nw, UF_NBLK = 8 * nw;
    static const int min_side = getenv("SIS_UPFIR_MIN_SIDE") ? atoi(getenv("SIS_UPFIR_MIN_SIDE")) : 16;   // (32: the 16 x 16 layer on the 4-phase kernel)
    if (batch <= 0 || h < min_side || w < min_side || (h & 1) || (w & 3) || cin % UF_CC || cout % UF_MBLK) return false;
    if ((row_stride & 3) || row_stride < 2 * w + 4) return false;
    if ((int64_t)batch * cin * h * w * 4 >= (1LL << 31) || (int64_t)cin * UF_PLANES * cout * 4 >= (1LL << 31)) return false;
    p.B = batch; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w; p.OH = 2 * h + 1; p.ORS = row_stride;
    p.nbw = w / 2 + 1;
    const int nbh = h / 2 + 1;
    p.bps = nbh * p.nbw;
    if (p.bps < UF_NBLK) return false;      // a tile spans at most two samples
    p.total_blocks = batch * p.bps;
    p.vr = h + 3;
    p.rw4 = w / 4 + 2;
    const int n_tiles = sis_cdiv(p.total_blocks, UF_NBLK);
    int rows_max = 0;
    for (int t = 0; t < n_tiles; ++t) {
        const int g0 = t * UF_NBLK, g1 = (g0 + UF_NBLK - 1 < p.total_blocks ? g0 + UF_NBLK - 1 : p.total_blocks - 1);
        const int r0 = (g0 / p.bps) * p.vr + 2 * ((g0 % p.bps) / p.nbw);
        const int r1 = (g1 / p.bps) * p.vr + 2 * ((g1 % p.bps) / p.nbw) + 2;
        rows_max = r1 - r0 + 1 > rows_max ? r1 - r0 + 1 : rows_max;
    }
    p.xs = sis_cdiv(rows_max * 4 * p.rw4, 256) * 256;
    if (p.xs > 256 * UF_XP_MAX) return false;
    *lds_bytes = (size_t)(2 * UF_CC * UF_WROW + 2 * UF_CC * p.xs + 2 * cin) * sizeof(float);
    return *lds_bytes <= 160 * 1024;
}

}  // namespace

// SIS_UPFIR_WAVES: 4 (default) = half tiles, two workgroups per CU; 8 = the one-workgroup-per-CU tile (A/B runs, same results)
static int upfir_waves() {
    static const int nw = getenv("SIS_UPFIR_WAVES") ? atoi(getenv("SIS_UPFIR_WAVES")) : 4;
    return nw;
}

extern "C" int sis_modconv_up_fir_supported(int batch, int cin, int cout, int h, int w, int t_row_stride) {
    UpFirParams p;
    size_t lds;
    const int nw = upfir_waves();
    return (nw == 4 || nw == 8) && upfir_plan(p, batch, cin, cout, h, w, t_row_stride, &lds, nw) ? 1 : 0;
}

extern "C" int sis_modconv_up_fir_prepack(float* u, const float* w, int cout, int cin, void* stream) {
    SIS_REQUIRE(u && w, "sis_modconv_up_fir_prepack: null pointer");
    SIS_REQUIRE(cout > 0 && cin > 0, "sis_modconv_up_fir_prepack: bad sizes");
    const int64_t n = (int64_t)cout * cin;
    hipLaunchKernelGGL(upfir_prepack_kernel, dim3(sis_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, u, w, cout, cin);
    SIS_CHECK_LAUNCH("upfir_prepack_kernel");
    return 0;
}

extern "C" int sis_modconv2d_up_fir(float* t, const float* x, const float* u, const float* s, const float* dscale, int batch,
                                    int cin, int cout, int h, int w, int t_row_stride, void* stream) {
    if (batch == 0) return 0;
    SIS_REQUIRE(t && x && u && s && dscale, "sis_modconv2d_up_fir: null pointer");
    SIS_REQUIRE((((uintptr_t)t | (uintptr_t)x | (uintptr_t)u) & 15) == 0, "sis_modconv2d_up_fir: pointers must be 16-byte aligned");
    UpFirParams p;
    size_t lds;
    const int nw = upfir_waves();
    SIS_REQUIRE(nw == 4 || nw == 8, "sis_modconv2d_up_fir: SIS_UPFIR_WAVES must be 4 or 8");
    const int UF_NBLK = 8 * nw, UF_THREADS = 64 * nw;
    SIS_REQUIRE(upfir_plan(p, batch, cin, cout, h, w, t_row_stride, &lds, nw),
                "sis_modconv2d_up_fir: %d x (%d -> %d) on %d x %d with row stride %d is not supported (sis_modconv_up_fir_supported)", batch,
                cin, cout, h, w, t_row_stride);
    p.x = x; p.u = u; p.s = s; p.dscale = dscale; p.out = t;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<1, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<0, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<4, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return sis_fail("sis_modconv2d_up_fir: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int64_t grid = (int64_t)sis_cdiv(p.total_blocks, UF_NBLK) * (cout / UF_MBLK);
    SIS_REQUIRE(grid > 0 && grid < ((int64_t)1 << 31), "sis_modconv2d_up_fir: bad grid");
    // chunk-loop form: 4 (4-wave workgroups only) = chunks pipelined across the barrier, 1 = fragments a pair ahead, 0 = plain; 2 / 3: 8-wave experiments
    static const int pipe = getenv("SIS_UPFIR_PIPE") ? atoi(getenv("SIS_UPFIR_PIPE")) : 4;
#ifdef SIS_ABLATIONS   // development builds only (tools/build_variant.sh WORK <tag> -DSIS_ABLATIONS): the shipped library has no wrong-result path
    static const int abl = getenv("SIS_UPFIR_ABL") ? atoi(getenv("SIS_UPFIR_ABL")) : 0;      // timing ablations: WRONG results
    if (abl) {
        SIS_REQUIRE(nw == 8, "SIS_UPFIR_ABL: the ablations exist for SIS_UPFIR_WAVES=8 only");
#define UF_ABL(A) hipLaunchKernelGGL((modconv_upfir_kernel<1, A>), dim3((unsigned)grid), dim3(UF_THREADS), lds, (hipStream_t)stream, p)
        static bool abl_attr = false;
        if (!abl_attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_upfir_kernel<1, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            abl_attr = true;
        }
        if (abl == 1) UF_ABL(1); else if (abl == 2) UF_ABL(2); else if (abl == 4) UF_ABL(4); else if (abl == 8) UF_ABL(8); else UF_ABL(15);
#undef UF_ABL
        SIS_CHECK_LAUNCH("modconv_upfir_kernel<ablation>");
        sis_kernel_name = "modconv_upfir_kernel";
        return 0;
    }
#endif
    if (nw == 4 && pipe == 4) hipLaunchKernelGGL((modconv_upfir_kernel<4, 0, 4>), dim3((unsigned)grid), dim3(UF_THREADS), lds, (hipStream_t)stream, p);
    else if (nw == 4 && pipe == 0) hipLaunchKernelGGL((modconv_upfir_kernel<0, 0, 4>), dim3((unsigned)grid), dim3(UF_THREADS), lds, (hipStream_t)stream, p);
    else if (nw == 4) hipLaunchKernelGGL((modconv_upfir_kernel<1, 0, 4>), dim3((unsigned)grid), dim3(UF_THREADS), lds, (hipStream_t)stream, p);
    else if (pipe == 0) hipLaunchKernelGGL(modconv_upfir_kernel<0>, dim3((unsigned)grid), dim3(UF_THREADS), lds, (hipStream_t)stream, p);
    else if (pipe == 2) hipLaunchKernelGGL(modconv_upfir_kernel<2>, dim3((unsigned)grid), dim3(UF_THREADS), lds, (hipStream_t)stream, p);
    else if (pipe == 3) hipLaunchKernelGGL(modconv_upfir_kernel<3>, dim3((unsigned)grid), dim3(UF_THREADS), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(modconv_upfir_kernel<1>, dim3((unsigned)grid), dim3(UF_THREADS), lds, (hipStream_t)stream, p);
    SIS_CHECK_LAUNCH("modconv_upfir_kernel");
    sis_kernel_name = "modconv_upfir_kernel";
    return 0;
}
