// Weight gradient of a plain 3x3 convolution (stride 1, padding 1) in the Winograd F(2x2,3x3) domain on the fp32
// matrix cores -- the third leg of networks/ema_net/network.py's 3x3 layers (forward and data gradient:
// modconv_wino.hip through sis_conv3x3).
//
//   dW[co][ci] = G^T ( sum over tiles t of  (A dY_t A^T) (.) (B^T d_t B) ) G
//
// dY_t = 2x2 tile of dL/dy, d_t = the 4x4 input patch around it: 16 multiplies per (co, ci, tile) instead of 36.
// Per Winograd point xi this is a GEMM  S[xi] (co x ci) += E[xi] (co x tiles) * V[xi] (tiles x ci)  whose reduction
// axis is the tile index.  Workgroup = 8 waves = 64 co x 64 ci x 16 xi (same accumulator split over wave pairs as
// the forward kernel), K chunk = 8 consecutive tiles of the flattened (sample, row, column) tile order; per chunk every lane loads ONE dY tile and ONE input
// patch from global memory into registers (two chunks ahead), transforms both and writes them into two LDS images
// E / V [xi][k slot half][channel][step kp]: tile 2 kp + half of the chunk, so that ONE ds_read_b128 of a lane (its channel,
// its k slot) delivers the operands of the four MFMA steps of a Winograd point (lane-contiguous: conflict free; the
// transform's b32 writes land 2-way, which costs a ds_write_b32 nothing).  Round 2: one straight-line block per chunk --
// round 1's loop staggered the two waves of a SIMD (transform first / MFMA first), which on this SIMD makes them multiply
// one after the other (tools/wino_trace.py on the forward kernel) -- with the loads, the two transforms and their 32 LDS
// writes spread over the slots between the 32 MFMAs.
// Split-K over tile ranges (blockIdx.y) writes raw S slabs; conv_wgrad_finish adds them in slice order
// (deterministic) and applies G^T . G.
#include <cstdlib>
#include <type_traits>
#include "sis_common.h"

// Timing ablations (WRONG results): development builds only (-DSIS_ABLATIONS -DSIS_WG_NOLOAD ...), refused otherwise.
#if !defined(SIS_ABLATIONS) && (defined(SIS_WG_NOTRANSFORM) || defined(SIS_WG_NOLOAD) || defined(SIS_WG_NOBARRIER))
#error "SIS_WG_* ablation switches need -DSIS_ABLATIONS (development builds only)"
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GK = 8;     // tiles per chunk (MFMA K = 2 per instruction: 4 instructions per xi per chunk)
constexpr int GBLK = 64;  // channels per workgroup on both GEMM axes
constexpr int GTHR = 512;

constexpr int GW_JOBS_MAX = 16;   // layers of one shape per launch (grid.z), operands through pointer tables

struct WgradParams {
    const float* x; const float* gy; float* slab;
    int jobs;                   // > 0: grid.z layers: x / gy from the tables, slabs of layer z at slab + z * slab_job_stride floats
    long long slab_job_stride;
    const float* xj[GW_JOBS_MAX]; const float* gyj[GW_JOBS_MAX];
    int B, Cin, Cout, H, W;
    int chunks_total, chunks_per_slice, tiles_per_row, tiles_per_sample;
};

__global__ __launch_bounds__(GTHR, 2) void conv_wgrad_wino_kernel(const WgradParams p) {
    constexpr int IMG = 16 * 2 * GBLK * 4;  // floats per image buffer: [xi][half][channel][kp]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* El = lds;            // [2][IMG]
    float* Vl = lds + 2 * IMG;  // [2][IMG]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int q = __builtin_amdgcn_readfirstlane(wave & 1), wn = (wave >> 1) & 1, wm = wave >> 2;
    const int n_ci = p.Cin / GBLK;
    const int o0 = (blockIdx.x / n_ci) * GBLK, i0 = (blockIdx.x % n_ci) * GBLK;
    const int c_lo = blockIdx.y * p.chunks_per_slice;
    const int c_hi = min(p.chunks_total, c_lo + p.chunks_per_slice);
    const int HW = p.H * p.W;

    // transform role: tile k of the chunk, channel ch of the workgroup's 64 (on both axes)
    const int tk = lane & 7, ch = tid >> 3;

    // Two register sets of raw tiles: while chunk c multiplies, set (c + 1) & 1 (loaded during chunk c - 1: a whole chunk of
    // latency cover) is transformed and written to LDS and set c & 1 is refilled with chunk c + 2.
    float xl[2][4], xh[2][4];  // input patch rows 2ty-1 .. 2ty+2: columns 2tx-1 and 2tx+2 ...
    f32x2 xm[2][4];            // ... and the aligned pair (2tx, 2tx+1) between them
    f32x2 gr[2][2];            // dY tile rows
    // Tile t = chunk * GK + tk of the flattened (sample, tile row, tile column) order.  The two divisions go through the float
    // reciprocal with one correction step (exact: t < 2^24, host-checked).
    const float inv_tps = 1.f / (float)p.tiles_per_sample, inv_tpr = 1.f / (float)p.tiles_per_row;
    // Loads go through buffer descriptors: a masked element gets an out-of-range offset and the hardware returns 0 -- the mask
    // selects an ADDRESS, so nothing waits for loaded data before the transform a chunk later (a select on the loaded value is
    // a wait for every load in flight, at the slot where the load was issued).
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.jobs ? p.xj[blockIdx.z] : p.x), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.jobs ? p.gyj[blockIdx.z] : p.gy), 0, 0x7FFFFFFF, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned g_xo = 0, g_go = 0;  // byte offsets of (patch row 0, column 2tx) and of the dY tile
    bool g_left = false, g_right = false;
    int g_y0 = 0;
    auto geo = [&](int chunk, bool valid = true) {  // (!valid: every load of the tile masked -- the refill of the last two chunks)
        const int t = chunk * GK + tk;
        int b = (int)((float)t * inv_tps), rem = t - b * p.tiles_per_sample;
        if (rem < 0) { --b; rem += p.tiles_per_sample; } else if (rem >= p.tiles_per_sample) { ++b; rem -= p.tiles_per_sample; }
        int ty = (int)((float)rem * inv_tpr), tx = rem - ty * p.tiles_per_row;
        if (tx < 0) { --ty; tx += p.tiles_per_row; } else if (tx >= p.tiles_per_row) { ++ty; tx -= p.tiles_per_row; }
        g_xo = (unsigned)(((b * p.Cin + i0 + ch) * HW + (2 * ty - 1) * p.W + 2 * tx) * 4);
        g_go = (unsigned)(((b * p.Cout + o0 + ch) * HW + 2 * ty * p.W + 2 * tx) * 4);
        g_left = tx > 0; g_right = 2 * tx + 2 < p.W; g_y0 = valid ? 2 * ty - 1 : -8;
        if (!valid) g_go = OOB;
    };
    auto ld1 = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0)); };
    auto ld2 = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) { return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0)); };
    auto load_x = [&](auto setc, int r) {  // patch row r of the tile geo() described
        constexpr int S = decltype(setc)::value;
        const int y = g_y0 + r;
        const bool row = y >= 0 && y < p.H;
        const unsigned o = g_xo + (unsigned)(r * p.W * 4);
        xl[S][r] = ld1(x_rsrc, (row && g_left) ? o - 4u : OOB);
        xm[S][r] = ld2(x_rsrc, row ? o : OOB);
        xh[S][r] = ld1(x_rsrc, (row && g_right) ? o + 8u : OOB);
    };
    auto load_g = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
        gr[S][0] = ld2(g_rsrc, g_go);
        gr[S][1] = ld2(g_rsrc, g_go + (unsigned)(p.W * 4));  // (a masked offset stays beyond the 2^31 - 1 bytes of the descriptor)
    };
    auto load = [&](auto setc, int chunk, bool valid) {  // whole tile at once (prologue)
        geo(chunk, valid);
#pragma unroll
        for (int r = 0; r < 4; ++r) load_x(setc, r);
        load_g(setc);
    };
    // image offset of this lane's (tile, channel): [xi][half = tk & 1][ch][kp = tk >> 1]
    const int img_off = ((tk & 1) * GBLK + ch) * 4 + (tk >> 1);  // + xi * 2 * GBLK * 4
    constexpr int XI_STRIDE = 2 * GBLK * 4;
    // Transforms write their Winograd rows as soon as they exist (few values live between slots).
    float rp[4], rq[4];
    auto transform_e_cols = [&](auto setc) {  // A dY: E = A dY A^T,  A = [[1,0],[1,1],[1,-1],[0,-1]]
        constexpr int S = decltype(setc)::value;
        const float a = gr[S][0].x, b = gr[S][0].y, c = gr[S][1].x, d = gr[S][1].y;
        rp[0] = a; rp[1] = a + c; rp[2] = a - c; rp[3] = -c;
        rq[0] = b; rq[1] = b + d; rq[2] = b - d; rq[3] = -d;
    };
    auto write_e = [&](int buf, int r) {  // Winograd row r of the E image
        float* eb = El + buf * IMG + img_off + r * 4 * XI_STRIDE;
        eb[0] = rp[r]; eb[XI_STRIDE] = rp[r] + rq[r]; eb[2 * XI_STRIDE] = rp[r] - rq[r]; eb[3 * XI_STRIDE] = -rq[r];
    };
    float tt[4][4];
    auto transform_v_rows = [&](auto setc) {  // B^T d
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d0 = c == 0 ? xl[S][0] : c == 1 ? xm[S][0].x : c == 2 ? xm[S][0].y : xh[S][0];
            const float d1 = c == 0 ? xl[S][1] : c == 1 ? xm[S][1].x : c == 2 ? xm[S][1].y : xh[S][1];
            const float d2 = c == 0 ? xl[S][2] : c == 1 ? xm[S][2].x : c == 2 ? xm[S][2].y : xh[S][2];
            const float d3 = c == 0 ? xl[S][3] : c == 1 ? xm[S][3].x : c == 2 ? xm[S][3].y : xh[S][3];
            tt[0][c] = d0 - d2; tt[1][c] = d1 + d2; tt[2][c] = d2 - d1; tt[3][c] = d1 - d3;
        }
    };
    auto write_v = [&](int buf, int r) {  // (B^T d) B, row r, into the V image
        float* vb = Vl + buf * IMG + img_off + r * 4 * XI_STRIDE;
        vb[0] = tt[r][0] - tt[r][2]; vb[XI_STRIDE] = tt[r][1] + tt[r][2];
        vb[2 * XI_STRIDE] = tt[r][2] - tt[r][1]; vb[3 * XI_STRIDE] = tt[r][1] - tt[r][3];
    };

    f32x16 acc[4][2];  // [row i of the 4x4 Winograd point grid][column jj of this wave's pair]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][jj][j] = 0.f;

    // MFMA operands: A[m = co][k = tile 2kp + half] from E, B[k][n = ci] from V; xi = 4 i + 2 q + jj; one float4 = kp 0..3
    const int aoff = (half * GBLK + wm * 32 + l31) * 4;  // + xi * XI_STRIDE
    const int boff = (half * GBLK + wn * 32 + l31) * 4;
    const std::integral_constant<int, 0> set0;
    const std::integral_constant<int, 1> set1;

    if (c_lo < c_hi) {
        load(set0, c_lo, true);
        transform_e_cols(set0);
        transform_v_rows(set0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { write_e(0, r); write_v(0, r); }
        load(set1, c_lo + 1, c_lo + 1 < c_hi);
        load(set0, c_lo + 2, c_lo + 2 < c_hi);
    }
    __syncthreads();
    // operand pair of the chunk's first Winograd point: carried from chunk to chunk (requested behind the previous chunk's barrier)
    f32x4 a = *reinterpret_cast<const f32x4*>(El + aoff + 2 * q * XI_STRIDE);
    f32x4 b = *reinterpret_cast<const f32x4*>(Vl + boff + 2 * q * XI_STRIDE);

    // One chunk: 32 MFMAs in eight groups (one Winograd point each: four steps from one float4 pair, the next group's pair
    // requested first), the side work of the chunk in the sixteen half-group slots between them: set P ^ 1 (chunk c + 1,
    // loaded during chunk c - 2) is transformed and written to LDS, then refilled with chunk c + 3 -- a load has a chunk and
    // a half (3-4 us) before its first use, and nearly every 64-lane load here includes a line that comes from HBM (x and dL/dy
    // are each read by 8-32 workgroups: 9 % of the L2 requests miss, ~20 lines per instruction).  P is the chunk's parity inside the
    // slice, a compile-time constant (the loop body is a pair of chunks).  Loads are unconditional (masked past the slice end):
    // a branch around loads makes every later wait a wait for ALL loads, the counter state of the two paths differs.
    auto chunk_body = [&](auto parity, int chunk) {
        constexpr int P = decltype(parity)::value;
        const std::integral_constant<int, P ^ 1> setT;
        const bool more = chunk + 1 < c_hi;
        const float* Eb = El + P * IMG + aoff;
        const float* Vb = Vl + P * IMG + boff;
        const int xi0 = 2 * q;
#pragma unroll
        for (int g = 0; g < 8; ++g) {  // Winograd point xi = 4 (g >> 1) + 2 q + (g & 1)
            const int i = g >> 1, jj = g & 1;
            f32x4 an = a, bn = b;
            if (g < 7) {
                const int xin = 4 * ((g + 1) >> 1) + 2 * q + ((g + 1) & 1);
                an = *reinterpret_cast<const f32x4*>(Eb + xin * XI_STRIDE);
                bn = *reinterpret_cast<const f32x4*>(Vb + xin * XI_STRIDE);
            } else {  // behind the chunk's barrier (below, after group 6): the first pair of chunk c + 1, under this chunk's last MFMAs
                an = *reinterpret_cast<const f32x4*>(El + (P ^ 1) * IMG + aoff + xi0 * XI_STRIDE);
                bn = *reinterpret_cast<const f32x4*>(Vl + (P ^ 1) * IMG + boff + xi0 * XI_STRIDE);
            }
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i][jj], 0, 0, 0);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i][jj], 0, 0, 0);
#ifndef SIS_WG_NOTRANSFORM  // (ablation switches for timing experiments: results wrong)
            if (more) {
                if (g == 1) { transform_e_cols(setT); write_e(P ^ 1, 0); }
                if (g == 2) write_e(P ^ 1, 2);
                if (g == 3) transform_v_rows(setT);
                if (g == 4) write_v(P ^ 1, 1);
                if (g == 5) write_v(P ^ 1, 3);
            }
#endif
#ifndef SIS_WG_NOLOAD
            if (g == 0) geo(chunk + 3, chunk + 3 < c_hi);
            if (g == 4) load_x(setT, 1);
            if (g == 5) load_x(setT, 3);
#endif
            __builtin_amdgcn_sched_barrier(0);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i][jj], 0, 0, 0);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i][jj], 0, 0, 0);
#ifndef SIS_WG_NOTRANSFORM
            if (more) {
                if (g == 1) write_e(P ^ 1, 1);
                if (g == 2) write_e(P ^ 1, 3);
                if (g == 3) write_v(P ^ 1, 0);
                if (g == 4) write_v(P ^ 1, 2);
            }
#endif
#ifndef SIS_WG_NOLOAD
            if (g == 2) load_g(setT);
            if (g == 3) load_x(setT, 0);
            if (g == 4) load_x(setT, 2);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifndef SIS_WG_NOBARRIER
            // The chunk's barrier sits before its last group: the images of chunk c + 1 are written by group 5, the last operand pair
            // of chunk c was requested in group 6 -- and what follows the barrier is the request for chunk c + 1's first pair, whose
            // LDS latency (eight waves asking at once) then lies under group 7's MFMAs instead of in front of the next chunk's.
            if (g == 6) __syncthreads();
#endif
            a = an; b = bn;
        }
    };
    for (int chunk = c_lo; chunk < c_hi; chunk += 2) {
        chunk_body(set0, chunk);
        if (chunk + 1 < c_hi) chunk_body(set1, chunk + 1);
    }

    // raw partial sums: slab[slice][xi][co][ci]
    float* sb = p.slab + (p.jobs ? (int64_t)blockIdx.z * p.slab_job_stride : 0) + (int64_t)blockIdx.y * 16 * p.Cout * p.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int xi = 4 * i + 2 * q + jj;
            float* sx = sb + ((int64_t)xi * p.Cout + o0 + wm * 32) * p.Cin + i0 + wn * 32 + l31;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * half;
                sx[(int64_t)row * p.Cin] = acc[i][jj][j];
            }
        }
}

// dW[co][ci][3][3] = G^T (sum over slices of S[.][co][ci]) G,  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
struct GwDwTab { float* dw[GW_JOBS_MAX]; };

__global__ __launch_bounds__(256) void conv_wgrad_finish_kernel(float* __restrict__ dw, const float* __restrict__ slab,
                                                                int n_slices, int64_t n_pairs, long long slab_job_stride = 0,
                                                                GwDwTab tab = GwDwTab{}) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // i = co * Cin + ci
    if (i >= n_pairs) return;
    if (slab_job_stride) {   // grid.y layers of one shape
        dw = tab.dw[blockIdx.y];
        slab += (int64_t)blockIdx.y * slab_job_stride;
    }
    float s[4][4];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        float v = 0.f;
        for (int k = 0; k < n_slices; ++k) v += slab[((int64_t)k * 16 + xi) * n_pairs + i];
        s[xi >> 2][xi & 3] = v;
    }
    float t[3][4];  // G^T s
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        t[0][c] = s[0][c] + 0.5f * (s[1][c] + s[2][c]);
        t[1][c] = 0.5f * (s[1][c] - s[2][c]);
        t[2][c] = 0.5f * (s[1][c] + s[2][c]) + s[3][c];
    }
    float* o = dw + i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[r * 3 + 0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
        o[r * 3 + 1] = 0.5f * (t[r][1] - t[r][2]);
        o[r * 3 + 2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
    }
}

// slab[0][xi][pair] <- sum over slices (slice order: deterministic); one lane per (xi, pair), coalesced over pairs.
// In place: element (0, xi, pair) is read and written by the same lane only.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(float* __restrict__ slab, int n_slices, int64_t n_pairs,
                                                                long long slab_job_stride = 0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    slab += (int64_t)blockIdx.z * slab_job_stride;   // (grid.z layers of one shape)
    const int64_t plane = 16 * n_pairs;
    float* p = slab + (int64_t)blockIdx.y * n_pairs + i;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;  // four independent load chains, added back in slice order below
    int k = 0;
    float sum = 0.f;
    for (; k + 4 <= n_slices; k += 4) {
        v0 = p[(int64_t)k * plane]; v1 = p[(int64_t)(k + 1) * plane]; v2 = p[(int64_t)(k + 2) * plane]; v3 = p[(int64_t)(k + 3) * plane];
        sum += v0; sum += v1; sum += v2; sum += v3;
    }
    for (; k < n_slices; ++k) sum += p[(int64_t)k * plane];
    p[0] = sum;
}

int wgrad_plan(WgradParams& p, int* ksplit, int batch, int cin, int cout, int h, int w, int64_t workspace_bytes, int jobs = 1) {
    workspace_bytes /= jobs;   // every layer of the launch has its own slabs
    if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return -1;
    if (h % 2 || w % 2 || cin % GBLK || cout % GBLK) return -1;
    if (((int64_t)batch * (h / 2) * (w / 2)) % GK) return -1;  // whole chunks of GK tiles
    if ((int64_t)batch * (cin > cout ? cin : cout) * h * w >= ((int64_t)1 << 29)) return -1;  // 32-bit byte offsets below 2^31
    if ((int64_t)batch * (h / 2) * (w / 2) >= (1 << 24)) return -1;  // tile indices are divided through float reciprocals
    p.B = batch; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w;
    p.tiles_per_row = w / 2;
    p.tiles_per_sample = (h / 2) * (w / 2);
    p.chunks_total = (int)((int64_t)batch * p.tiles_per_sample / GK);
    const int64_t blocks = (int64_t)(cin / GBLK) * (cout / GBLK) * jobs;   // (the layers of a launch fill the chip together)
    const int64_t slab_bytes = (int64_t)16 * cin * cout * 4;
    // One workgroup per CU fits (128 KB of LDS): ONE round of workgroups over the 256 CUs.  (Two rounds, the round-1 plan, halve
    // every slice -- twice the slab traffic and twice the prologues for the same multiplies: 256 -> 256 channels on 16 x 16 sub-images
    // went from 127 to 151 TF with one round, 320 workgroups -- a second round for a quarter of the CUs -- fell to 106.)
    static const int target_wgs = getenv("SIS_WGRAD_WGS") ? atoi(getenv("SIS_WGRAD_WGS")) : 256;
    int want = (int)((target_wgs + blocks - 1) / blocks);
    while (want > 1 && want * blocks > target_wgs) --want;    // never past one round
    if (want > p.chunks_total / 4) want = p.chunks_total / 4;  // at least 4 chunks per slice
    if (want < 1) want = 1;
    if ((int64_t)want * slab_bytes > workspace_bytes) want = (int)(workspace_bytes / slab_bytes);
    if (want < 1) return -1;
    p.chunks_per_slice = (p.chunks_total + want - 1) / want;
    *ksplit = (p.chunks_total + p.chunks_per_slice - 1) / p.chunks_per_slice;
    return 0;
}

}  // namespace

extern "C" int sis_conv3x3_wgrad_eligible(int batch, int cin, int cout, int h, int w, int64_t workspace_bytes) {
    WgradParams p;
    int ks;
    return wgrad_plan(p, &ks, batch, cin, cout, h, w, workspace_bytes) == 0 ? 1 : 0;
}

static int gw_jobs(float* const* dw, const float* const* x, const float* const* gy, int n_jobs, int batch, int cin, int cout, int h, int w,
                   void* workspace, int64_t workspace_bytes, void* stream, const char* who) {
    WgradParams p;
    int ksplit = 1;
    SIS_REQUIRE(wgrad_plan(p, &ksplit, batch, cin, cout, h, w, workspace_bytes, n_jobs) == 0,
                "%s: needs even H and W, B*H*W/4 %% 8 == 0, channels %% 64 == 0 and a workspace of at least "
                "64 * Cin * Cout bytes per layer (got %d x %d x %dx%d, %d -> %d)", who, n_jobs, batch, h, w, cin, cout);
    const int64_t pairs = (int64_t)cin * cout;
    p.slab = (float*)workspace;
    p.jobs = n_jobs > 1 ? n_jobs : 0;
    p.slab_job_stride = (long long)ksplit * 16 * pairs;
    GwDwTab tab = {};
    for (int j = 0; j < n_jobs; ++j) {
        SIS_REQUIRE(dw[j] && x[j] && gy[j], "%s: null pointer in layer %d", who, j);
        SIS_REQUIRE(((((uintptr_t)x[j] | (uintptr_t)gy[j]) & 7) == 0), "%s: tensors must be 8-byte aligned", who);
        p.xj[j] = x[j]; p.gyj[j] = gy[j]; tab.dw[j] = dw[j];
    }
    p.x = x[0]; p.gy = gy[0];
    const size_t lds = (size_t)4 * 16 * 2 * GBLK * 4 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_wino_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return sis_fail("%s: cannot raise the dynamic LDS limit: %s", who, hipGetErrorString(e));
        attr_set = true;
    }
    sis_kernel_name = "conv_wgrad_wino_kernel";
    SIS_OCC_REPORT(conv_wgrad_wino_kernel, GTHR, lds);
    hipLaunchKernelGGL(conv_wgrad_wino_kernel, dim3((cin / GBLK) * (cout / GBLK), ksplit, n_jobs), dim3(GTHR), lds, (hipStream_t)stream, p);
    SIS_CHECK_LAUNCH("conv_wgrad_wino_kernel");
    const long long stride = p.jobs ? p.slab_job_stride : 0;
    if (ksplit > 4) {  // many thin slices (few channel pairs): reduce them with one lane per (xi, pair) first
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(sis_cdiv(pairs, 256), 16, n_jobs), dim3(256), 0, (hipStream_t)stream,
                           (float*)workspace, ksplit, pairs, stride);
        SIS_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
        ksplit = 1;
    }
    hipLaunchKernelGGL(conv_wgrad_finish_kernel, dim3(sis_cdiv(pairs, 256), n_jobs), dim3(256), 0, (hipStream_t)stream, dw[0],
                       (const float*)workspace, ksplit, pairs, stride, tab);
    SIS_CHECK_LAUNCH("conv_wgrad_finish_kernel");
    return 0;
}

extern "C" int sis_conv3x3_wgrad(float* dw, const float* x, const float* gy, int batch, int cin, int cout, int h, int w,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    SIS_REQUIRE(dw && x && gy && workspace, "sis_conv3x3_wgrad: null pointer");
    return gw_jobs(&dw, &x, &gy, 1, batch, cin, cout, h, w, workspace, workspace_bytes, stream, "sis_conv3x3_wgrad");
}

/* The same for n_jobs layers of ONE shape (EMANet's repeated bottleneck units, queued during the backward): `dw`, `x`, `gy` are
 * HOST arrays of n_jobs device pointers.  One tile launch (grid.z = layer), one finish launch (and one slice reduction where the
 * plan has more than four slices) per <= 16 layers -- as few layers per launch as leave each its slabs in the workspace. */
extern "C" int sis_conv3x3_wgrad_multi(float* const* dw, const float* const* x, const float* const* gy, int n_jobs, int batch, int cin,
                                       int cout, int h, int w, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_jobs <= 0) return 0;
    SIS_REQUIRE(dw && x && gy && workspace, "sis_conv3x3_wgrad_multi: null pointer");
    for (int j0 = 0; j0 < n_jobs;) {
        int n = n_jobs - j0 < GW_JOBS_MAX ? n_jobs - j0 : GW_JOBS_MAX;
        for (; n > 1; n = (n + 1) / 2) {
            WgradParams q;
            int ks;
            if (wgrad_plan(q, &ks, batch, cin, cout, h, w, workspace_bytes, n) == 0) break;
        }
        const int rc = gw_jobs(dw + j0, x + j0, gy + j0, n, batch, cin, cout, h, w, workspace, workspace_bytes, stream, "sis_conv3x3_wgrad_multi");
        if (rc) return rc;
        j0 += n;
    }
    return 0;
}
