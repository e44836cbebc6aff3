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
#include "sis_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GK = 8;     // tiles per chunk (MFMA K = 2 per instruction: 4 instructions per xi per chunk)
constexpr int GBLK = 64;  // channels per workgroup on both GEMM axes
constexpr int GTHR = 512;

struct WgradParams {
    const float* x; const float* gy; float* slab;
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
    const float* xplane = p.x + (int64_t)(i0 + ch) * HW;
    const float* gplane = p.gy + (int64_t)(o0 + ch) * HW;

    f32x2 xr[4][3];  // input patch rows 2ty-1 .. 2ty+2, columns 2tx-2 .. 2tx+3 as three aligned pairs
    f32x2 gr[2];     // dY tile rows
    auto load = [&](int chunk) {  // tile t = chunk * GK + tk of the flattened (sample, tile row, tile column) order
        const int t = chunk * GK + tk;
        const int b = t / p.tiles_per_sample, rem = t - b * p.tiles_per_sample;
        const int ty = rem / p.tiles_per_row, tx = rem - ty * p.tiles_per_row;
        const float* xb = xplane + (int64_t)b * p.Cin * HW + (2 * ty - 1) * p.W + 2 * tx - 2;
        const float* gb = gplane + (int64_t)b * p.Cout * HW + 2 * ty * p.W + 2 * tx;
        const bool left = tx > 0, right = 2 * tx + 2 < p.W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = 2 * ty - 1 + r;
            const bool row = y >= 0 && y < p.H;
            const f32x2 z = {0.f, 0.f};
            xr[r][0] = (row && left) ? *reinterpret_cast<const f32x2*>(xb + r * p.W) : z;
            xr[r][1] = row ? *reinterpret_cast<const f32x2*>(xb + r * p.W + 2) : z;
            xr[r][2] = (row && right) ? *reinterpret_cast<const f32x2*>(xb + r * p.W + 4) : z;
        }
        gr[0] = *reinterpret_cast<const f32x2*>(gb);
        gr[1] = *reinterpret_cast<const f32x2*>(gb + p.W);
    };
    // image offset of this lane's (tile, channel): [xi][half = tk & 1][ch][kp = tk >> 1]
    const int img_off = ((tk & 1) * GBLK + ch) * 4 + (tk >> 1);  // + xi * 2 * GBLK * 4
    constexpr int XI_STRIDE = 2 * GBLK * 4;
    float ev[16], vv[16];  // transformed values of the chunk in flight (registers between their slots)
    auto transform_e = [&]() {  // E = A dY A^T,  A = [[1,0],[1,1],[1,-1],[0,-1]]
        const float a = gr[0].x, b = gr[0].y, c = gr[1].x, d = gr[1].y;
        const float rp[4] = {a, a + c, a - c, -c}, rq[4] = {b, b + d, b - d, -d};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ev[r * 4 + 0] = rp[r]; ev[r * 4 + 1] = rp[r] + rq[r]; ev[r * 4 + 2] = rp[r] - rq[r]; ev[r * 4 + 3] = -rq[r];
        }
    };
    float tt[4][4];
    auto transform_v_rows = [&]() {  // B^T d
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d0 = c == 0 ? xr[0][0].y : c == 1 ? xr[0][1].x : c == 2 ? xr[0][1].y : xr[0][2].x;
            const float d1 = c == 0 ? xr[1][0].y : c == 1 ? xr[1][1].x : c == 2 ? xr[1][1].y : xr[1][2].x;
            const float d2 = c == 0 ? xr[2][0].y : c == 1 ? xr[2][1].x : c == 2 ? xr[2][1].y : xr[2][2].x;
            const float d3 = c == 0 ? xr[3][0].y : c == 1 ? xr[3][1].x : c == 2 ? xr[3][1].y : xr[3][2].x;
            tt[0][c] = d0 - d2; tt[1][c] = d1 + d2; tt[2][c] = d2 - d1; tt[3][c] = d1 - d3;
        }
    };
    auto transform_v_cols = [&](int r) {  // (B^T d) B, row r
        vv[r * 4 + 0] = tt[r][0] - tt[r][2]; vv[r * 4 + 1] = tt[r][1] + tt[r][2];
        vv[r * 4 + 2] = tt[r][2] - tt[r][1]; vv[r * 4 + 3] = tt[r][1] - tt[r][3];
    };
    auto write_images = [&](int buf, int r) {  // Winograd row r of both images
        float* eb = El + buf * IMG + img_off;
        float* vb = Vl + buf * IMG + img_off;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            eb[(r * 4 + q4) * XI_STRIDE] = ev[r * 4 + q4];
            vb[(r * 4 + q4) * XI_STRIDE] = vv[r * 4 + q4];
        }
    };
    auto transform = [&](int buf) {  // whole chunk at once (prologue)
        transform_e();
        transform_v_rows();
#pragma unroll
        for (int r = 0; r < 4; ++r) { transform_v_cols(r); write_images(buf, r); }
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

    if (c_lo < c_hi) {
        load(c_lo);
        transform(0);
        if (c_lo + 1 < c_hi) load(c_lo + 1);
    }
    __syncthreads();

    int it = 0;
    for (int chunk = c_lo; chunk < c_hi; ++chunk, ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        const bool more = chunk + 1 < c_hi;
        const float* Eb = El + cur * IMG + aoff;
        const float* Vb = Vl + cur * IMG + boff;
        // registers hold chunk + 1 (loaded one iteration ago); its transform and LDS writes are spread over this chunk's
        // MFMA slots, the loads of chunk + 2 follow once the registers are free.  (`more` is wave-uniform: the side work of
        // the last chunk is skipped by scalar branches around straight-line code.)
#pragma unroll
        for (int g = 0; g < 8; ++g) {  // Winograd point xi = 4 (g >> 1) + 2 q + (g & 1): four MFMA steps from one float4 pair
            const int i = g >> 1, jj = g & 1, xi = 4 * i + 2 * q + jj;
            const f32x4 a = *reinterpret_cast<const f32x4*>(Eb + xi * XI_STRIDE);
            const f32x4 b = *reinterpret_cast<const f32x4*>(Vb + xi * XI_STRIDE);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[i][jj], 0, 0, 0);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[i][jj], 0, 0, 0);
            if (more) {
                if (g == 0) transform_e();
                if (g == 1) transform_v_rows();
                if (g >= 2 && g < 6) transform_v_cols(g - 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[i][jj], 0, 0, 0);
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[i][jj], 0, 0, 0);
            if (more) {
                if (g >= 3 && g < 7) write_images(nxt, g - 3);
                if (g == 7 && chunk + 2 < c_hi) load(chunk + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // raw partial sums: slab[slice][xi][co][ci]
    float* sb = p.slab + (int64_t)blockIdx.y * 16 * p.Cout * p.Cin;
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
__global__ __launch_bounds__(256) void conv_wgrad_finish_kernel(float* __restrict__ dw, const float* __restrict__ slab,
                                                                int n_slices, int64_t n_pairs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // i = co * Cin + ci
    if (i >= n_pairs) return;
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
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(float* __restrict__ slab, int n_slices, int64_t n_pairs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
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

int wgrad_plan(WgradParams& p, int* ksplit, int batch, int cin, int cout, int h, int w, int64_t workspace_bytes) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return -1;
    if (h % 2 || w % 2 || cin % GBLK || cout % GBLK) return -1;
    if (((int64_t)batch * (h / 2) * (w / 2)) % GK) return -1;  // whole chunks of GK tiles
    if ((int64_t)batch * (cin > cout ? cin : cout) * h * w >= ((int64_t)1 << 31)) return -1;
    p.B = batch; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w;
    p.tiles_per_row = w / 2;
    p.tiles_per_sample = (h / 2) * (w / 2);
    p.chunks_total = (int)((int64_t)batch * p.tiles_per_sample / GK);
    const int64_t blocks = (int64_t)(cin / GBLK) * (cout / GBLK);
    const int64_t slab_bytes = (int64_t)16 * cin * cout * 4;
    int want = (int)((512 + blocks - 1) / blocks);            // ~2 workgroup rounds over 256 CUs
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

extern "C" int sis_conv3x3_wgrad(float* dw, const float* x, const float* gy, int batch, int cin, int cout, int h, int w,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    SIS_REQUIRE(dw && x && gy && workspace, "sis_conv3x3_wgrad: null pointer");
    SIS_REQUIRE(((((uintptr_t)x | (uintptr_t)gy) & 7) == 0), "sis_conv3x3_wgrad: tensors must be 8-byte aligned");
    WgradParams p;
    int ksplit = 1;
    SIS_REQUIRE(wgrad_plan(p, &ksplit, batch, cin, cout, h, w, workspace_bytes) == 0,
                "sis_conv3x3_wgrad: needs even H and W, B*H*W/4 %% 8 == 0, channels %% 64 == 0 and a workspace of at least "
                "64 * Cin * Cout bytes (got %d x %dx%d, %d -> %d)", batch, h, w, cin, cout);
    p.x = x; p.gy = gy; p.slab = (float*)workspace;
    const size_t lds = (size_t)4 * 16 * 2 * GBLK * 4 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_wino_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return sis_fail("sis_conv3x3_wgrad: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    sis_kernel_name = "conv_wgrad_wino_kernel";
    SIS_OCC_REPORT(conv_wgrad_wino_kernel, GTHR, lds);
    hipLaunchKernelGGL(conv_wgrad_wino_kernel, dim3((cin / GBLK) * (cout / GBLK), ksplit), dim3(GTHR), lds,
                       (hipStream_t)stream, p);
    SIS_CHECK_LAUNCH("conv_wgrad_wino_kernel");
    const int64_t pairs = (int64_t)cin * cout;
    if (ksplit > 4) {  // many thin slices (few channel pairs): reduce them with one lane per (xi, pair) first
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(sis_cdiv(pairs, 256), 16), dim3(256), 0, (hipStream_t)stream,
                           (float*)workspace, ksplit, pairs);
        SIS_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
        ksplit = 1;
    }
    hipLaunchKernelGGL(conv_wgrad_finish_kernel, dim3(sis_cdiv(pairs, 256)), dim3(256), 0, (hipStream_t)stream, dw,
                       (const float*)workspace, ksplit, pairs);
    SIS_CHECK_LAUNCH("conv_wgrad_finish_kernel");
    return 0;
}
