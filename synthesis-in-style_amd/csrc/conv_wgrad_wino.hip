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
// patch from global memory into registers (one chunk ahead), transforms both and writes them into two LDS images
// E / V [xi][tile][channel] (row stride 72 floats: the 8 tiles x 8 channels of a wave land 2-way on the banks,
// which costs a ds_write_b32 nothing); the MFMA operands are then plain conflict-free ds_read_b32.  The two waves
// sharing a SIMD are staggered (transform first / MFMA first) like in the forward kernel.
// Split-K over tile ranges (blockIdx.y) writes raw S slabs; conv_wgrad_finish adds them in slice order
// (deterministic) and applies G^T . G.
#include "sis_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int GK = 8;     // tiles per chunk (MFMA K = 2 per instruction: 4 instructions per xi per chunk)
constexpr int GBLK = 64;  // channels per workgroup on both GEMM axes
constexpr int GLD = 72;   // LDS row stride of the E / V images in floats
constexpr int GTHR = 512;

struct WgradParams {
    const float* x; const float* gy; float* slab;
    int B, Cin, Cout, H, W;
    int chunks_total, chunks_per_slice, tiles_per_row, tiles_per_sample;
};

__global__ __launch_bounds__(GTHR, 2) void conv_wgrad_wino_kernel(const WgradParams p) {
    constexpr int IMG = 16 * GK * GLD;  // floats per image buffer
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
    auto transform = [&](int buf) {
        float* eb = El + buf * IMG + tk * GLD + ch;
        float* vb = Vl + buf * IMG + tk * GLD + ch;
        {   // E = A dY A^T,  A = [[1,0],[1,1],[1,-1],[0,-1]]
            const float a = gr[0].x, b = gr[0].y, c = gr[1].x, d = gr[1].y;
            const float rp[4] = {a, a + c, a - c, -c}, rq[4] = {b, b + d, b - d, -d};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                eb[(r * 4 + 0) * GK * GLD] = rp[r];
                eb[(r * 4 + 1) * GK * GLD] = rp[r] + rq[r];
                eb[(r * 4 + 2) * GK * GLD] = rp[r] - rq[r];
                eb[(r * 4 + 3) * GK * GLD] = -rq[r];
            }
        }
        {   // V = B^T d B
            float dd[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { dd[r][0] = xr[r][0].y; dd[r][1] = xr[r][1].x; dd[r][2] = xr[r][1].y; dd[r][3] = xr[r][2].x; }
            float tt[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                tt[0][c] = dd[0][c] - dd[2][c];
                tt[1][c] = dd[1][c] + dd[2][c];
                tt[2][c] = dd[2][c] - dd[1][c];
                tt[3][c] = dd[1][c] - dd[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vb[(r * 4 + 0) * GK * GLD] = tt[r][0] - tt[r][2];
                vb[(r * 4 + 1) * GK * GLD] = tt[r][1] + tt[r][2];
                vb[(r * 4 + 2) * GK * GLD] = tt[r][2] - tt[r][1];
                vb[(r * 4 + 3) * GK * GLD] = tt[r][1] - tt[r][3];
            }
        }
    };

    f32x16 acc[4][2];  // [row i of the 4x4 Winograd point grid][column jj of this wave's pair]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][jj][j] = 0.f;

    // MFMA operands: A[m = co][k = tile 2kp + half] from E, B[k][n = ci] from V; xi = 4 i + 2 q + jj
    const int aoff = (2 * q * GK + half) * GLD + wm * 32 + l31;  // + (4 i + jj) * GK * GLD + 2 kp * GLD
    const int boff = (2 * q * GK + half) * GLD + wn * 32 + l31;

    if (c_lo < c_hi) {
        load(c_lo);
        transform(0);
        if (c_lo + 1 < c_hi) load(c_lo + 1);
    }
    __syncthreads();

    const bool late_transform = __builtin_amdgcn_readfirstlane(wave) >= 4;
    int it = 0;
    for (int chunk = c_lo; chunk < c_hi; ++chunk, ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        const bool more = chunk + 1 < c_hi;
        if (!late_transform && more) {
            transform(nxt);                       // registers hold chunk + 1 (loaded one iteration ago)
            if (chunk + 2 < c_hi) load(chunk + 2);
        }
        const float* Eb = El + cur * IMG + aoff;
        const float* Vb = Vl + cur * IMG + boff;
#pragma unroll
        for (int kp = 0; kp < GK / 2; ++kp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Eb[((4 * i) * GK + 2 * kp) * GLD], Vb[((4 * i) * GK + 2 * kp) * GLD], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Eb[((4 * i + 1) * GK + 2 * kp) * GLD], Vb[((4 * i + 1) * GK + 2 * kp) * GLD], acc[i][1], 0, 0, 0);
            }
        }
        if (late_transform && more) {
            transform(nxt);
            if (chunk + 2 < c_hi) load(chunk + 2);
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
    const size_t lds = (size_t)4 * 16 * GK * GLD * sizeof(float);
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
