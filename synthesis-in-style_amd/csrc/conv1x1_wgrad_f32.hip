// Weight gradient of EMANet's fp32 1x1 convolutions (networks/ema_net/network.py:24,29,106-107,219-249,271-289 through
// networks/hip_conv.py::_Pointwise) on v_mfma_f32_32x32x2_f32:
//
//   dW[co][ci] = sum over samples b and pixels p of  dY[b][co][p] * X[b][ci][p]
//
// Both operands are NCHW rows whose pixels are contiguous, i.e. K-contiguous for this product: a workgroup tile is
// TM output channels x TN input channels, its K axis the flattened (sample, 64-pixel block) stream.  Staging is pure
// LDS-DMA (buffer descriptors: per-lane offset in a VGPR, everything per chunk in SGPRs; one 1 KB piece = 4 rows x 256 B),
// with the 16-byte units of a row XOR-swizzled by the row number ON THE SOURCE SIDE so that the operand reads -- one
// ds_read_b128 = 4 consecutive pixels of a row = the operands of 4 MFMA steps -- are conflict free (the MFMA's two k
// slots take pixels 4h .. 4h+3 of each 8-pixel group for BOTH operands: any fixed pixel <-> (step, k slot) map is a valid
// contraction order).  Each of the 8 waves owns MB 32 x 32 accumulator tiles stacked along the output channels (its B
// operand read is shared by them); split-K over the pixel stream writes slabs that an ordered pass adds (deterministic,
// no atomics).
#include <cstdlib>
#include "sis_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PK = 64;  // pixels per chunk (one 256-byte row piece per channel)

constexpr int PWG_JOBS_MAX = 16;   // layers of one shape per launch (grid.z), operands through pointer tables

struct PwgParams {
    const float* gy; const float* x; float* out;  // out: dW or the slab of slice 0
    int B, Cout, Cin, HW;
    int chunks_total, chunks_per_slice;
    int64_t slab_stride;  // floats between slices (0: single slice, out = dW)
    int jobs;             // > 0: grid.z layers of this shape: gy / x / out from the tables (out of layer z: its dW or its first slab)
    const float* gyj[PWG_JOBS_MAX]; const float* xj[PWG_JOBS_MAX]; float* outj[PWG_JOBS_MAX];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pwg_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ void pwg_dma(__amdgpu_buffer_rsrc_t r, float* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

// WM x WN waves of MB stacked 32 x 32 tiles each: TM = 32 WM MB output channels, TN = 32 WN input channels
template <int WM, int WN, int MB>
__global__ __launch_bounds__(512, 2) void conv1x1_wgrad_f32_kernel(const PwgParams p) {
    static_assert(WM * WN == 8, "8 waves");
    constexpr int TM = 32 * WM * MB, TN = 32 * WN;
    constexpr int A_FLOATS = TM * PK, B_FLOATS = TN * PK, STAGE = A_FLOATS + B_FLOATS;
    constexpr int A_PIECES = TM / 4, B_PIECES = TN / 4;  // 1 KB pieces (4 rows) per chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int n_ci = p.Cin / TN;
    const int co0 = (blockIdx.x / n_ci) * TM, ci0 = (blockIdx.x % n_ci) * TN;
    const int c_lo = blockIdx.y * p.chunks_per_slice, c_hi = min(p.chunks_total, c_lo + p.chunks_per_slice);
    const int blocks_per_sample = p.HW / PK;

    // DMA: piece q covers rows 4q .. 4q+3; lane L -> row 4q + L/16, LDS unit L % 16, global unit (L % 16) ^ (row & 15).
    // row & 15 = (4 (q & 3) + L / 16): four offset registers, indexed by q & 3.
    unsigned voff[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int row = 4 * v + (lane >> 4);
        voff[v] = (unsigned)(row * p.HW) * 4u + (unsigned)(((lane & 15) ^ (row & 15)) * 16);
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = pwg_rsrc(p.jobs ? p.gyj[blockIdx.z] : p.gy), b_rsrc = pwg_rsrc(p.jobs ? p.xj[blockIdx.z] : p.x);
    auto stage = [&](int chunk, int buf) {
        const int b = chunk / blocks_per_sample, p0 = (chunk - b * blocks_per_sample) * PK;
        float* al = lds + buf * STAGE;
        float* bl = al + A_FLOATS;
        // rows of a piece group of 16 (= 4 pieces) are 16 * HW floats apart
        for (int q = wave; q < A_PIECES; q += 8)
            pwg_dma(a_rsrc, al + q * 256, voff[q & 3], (unsigned)(((int64_t)(b * p.Cout + co0 + 16 * (q >> 2)) * p.HW + p0) * 4));
        for (int q = wave; q < B_PIECES; q += 8)
            pwg_dma(b_rsrc, bl + q * 256, voff[q & 3], (unsigned)(((int64_t)(b * p.Cin + ci0 + 16 * (q >> 2)) * p.HW + p0) * 4));
    };

    // operand reads: row = tile row of this lane, 16-byte unit (2 j + half) ^ (row & 15), j = 0..7
    int a_base[MB], a_hi[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int arow = (wm * MB + mb) * 32 + l31;
        a_base[mb] = arow * PK + (((half ^ arow) & 1) * 4);
        a_hi[mb] = arow & 14;
    }
    const int brow = wn * 32 + l31;
    const int b_base = A_FLOATS + brow * PK + (((half ^ brow) & 1) * 4), b_hi = brow & 14;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;

    if (c_lo < c_hi) stage(c_lo, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Operand quads one step ahead (two register sets; the step loop is unrolled), carried from chunk to chunk.  The chunk's barrier
    // sits in front of its LAST step's MFMAs -- that step's quads are in registers by then, nobody reads this buffer any more --
    // and the first quads of chunk + 1 are requested right behind it, under those MFMAs, instead of in front of the next chunk's
    // first MFMA (one workgroup per CU: nothing else covers that LDS round trip, which all eight waves start at once).
    f32x4 bq[2], aq[2][MB];
    auto quads = [&](const float* st, int j, int set) {
        bq[set] = *reinterpret_cast<const f32x4*>(st + b_base + ((2 * j) ^ b_hi) * 4);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) aq[set][mb] = *reinterpret_cast<const f32x4*>(st + a_base[mb] + ((2 * j) ^ a_hi[mb]) * 4);
    };
    quads(lds, 0, 0);
    int buf = 0;
    for (int chunk = c_lo; chunk < c_hi; ++chunk, buf ^= 1) {
        if (chunk + 1 < c_hi) stage(chunk + 1, buf ^ 1);
        const float* st = lds + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cb = j & 1;
            if (j < 7) quads(st, j + 1, cb ^ 1);
            else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of chunk + 1 ...
                __syncthreads();                                    // ... and everyone's
                quads(lds + (buf ^ 1) * STAGE, 0, 0);
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cb][mb].x, bq[cb].x, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cb][mb].y, bq[cb].y, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cb][mb].z, bq[cb].z, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[cb][mb].w, bq[cb].w, acc[mb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // quads of step j + 1, then the MFMAs of step j: in program order
        }
    }

    float* out = (p.jobs ? p.outj[blockIdx.z] : p.out) + (int64_t)blockIdx.y * p.slab_stride;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + (wm * MB + mb) * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
            out[(int64_t)co * p.Cin + ci0 + wn * 32 + l31] = acc[mb][i];
        }
}

// dW[i] = sum over slices, in a fixed order (8 interleaved partial sums, then their fixed tree): deterministic.  One float
// per lane and 8 independent loads in flight per trip: the sum is latency bound otherwise (a float4 per lane walking 64
// slices one after the other took longer than the GEMM for the 128 -> 512 layers).
struct PwgDwTab { float* dw[PWG_JOBS_MAX]; };

__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slab,
                                                                   int n_slices, int64_t n, int64_t slab_job_stride = 0,
                                                                   PwgDwTab tab = PwgDwTab{}) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (slab_job_stride) {   // grid.y layers of one shape
        dw = tab.dw[blockIdx.y];
        slab += (int64_t)blockIdx.y * slab_job_stride;
    }
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= n_slices; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += slab[(int64_t)(k + u) * n + i];
    }
    for (int u = 0; k < n_slices; ++k, ++u) s[u] += slab[(int64_t)k * n + i];
    dw[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

template <int WM, int WN, int MB>
int launch_pwg(PwgParams& p, int tiles, int slices, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (32 * WM * MB + 32 * WN) * PK * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_wgrad_f32_kernel<WM, WN, MB>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return sis_fail("sis_conv1x1_wgrad_f32: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    SIS_OCC_REPORT((conv1x1_wgrad_f32_kernel<WM, WN, MB>), 512, lds);
    hipLaunchKernelGGL((conv1x1_wgrad_f32_kernel<WM, WN, MB>), dim3(tiles, slices, p.jobs ? p.jobs : 1), dim3(512), lds, st, p);
    SIS_CHECK_LAUNCH("sis_conv1x1_wgrad_f32");
    return 0;
}

// Tile shape (TM x TN) for (cout, cin): 128 x 128 (two stacked tiles per wave: 8 MFMAs per DMA piece instead of 5.3) where
// both divide and there are at least 8 such tiles (the K slices supply the rest of the parallelism); else 128 x 64,
// 64 x 128 (64 output channels), 256 x 32.
int pwg_tile(int cout, int cin, int* tm, int* tn) {
    if (cout % 128 == 0 && cin % 128 == 0 && (cout / 128) * (cin / 128) >= 8) { *tm = 128; *tn = 128; return 3; }
    if (cout % 128 == 0 && cin % 64 == 0) { *tm = 128; *tn = 64; return 0; }
    if (cout % 64 == 0 && cin % 128 == 0) { *tm = 64; *tn = 128; return 1; }
    if (cout % 256 == 0 && cin % 32 == 0) { *tm = 256; *tn = 32; return 2; }
    return -1;
}

// K slices: about 3 workgroups per compute unit in all, at least 2 chunks each, slabs of at most 32 MB, at most 256
int pwg_slices(int batch, int cin, int cout, int hw, int jobs = 1) {
    int tm, tn;
    pwg_tile(cout, cin, &tm, &tn);
    const int tiles = (cout / tm) * (cin / tn) * jobs, chunks = batch * (hw / PK);   // (the layers of a launch fill the chip together)
    static const int target = getenv("SIS_PWG_TARGET") ? atoi(getenv("SIS_PWG_TARGET")) : 768;   // workgroups a launch is cut for (~3 per CU)
    int slices = (target + tiles - 1) / tiles;
    if (slices > chunks / 2) slices = chunks / 2;
    const int64_t by_bytes = ((int64_t)32 << 20) / ((int64_t)cout * cin * 4);
    if (slices > by_bytes) slices = (int)by_bytes;
    if (slices > 256) slices = 256;
    return slices < 1 ? 1 : slices;
}

}  // namespace

extern "C" int sis_conv1x1_wgrad_f32_supported(int batch, int cin, int cout, int hw) {
    int tm, tn;
    return batch > 0 && hw > 0 && hw % PK == 0 && pwg_tile(cout, cin, &tm, &tn) >= 0 &&
           (int64_t)batch * cout * hw < (1LL << 30) && (int64_t)batch * cin * hw < (1LL << 30) ? 1 : 0;
}

extern "C" int64_t sis_conv1x1_wgrad_f32_workspace(int batch, int cin, int cout, int hw) {
    if (!sis_conv1x1_wgrad_f32_supported(batch, cin, cout, hw)) return 0;
    const int slices = pwg_slices(batch, cin, cout, hw);
    return slices > 1 ? (int64_t)slices * cout * cin * (int64_t)sizeof(float) : 0;
}

static int pwg_jobs(float* const* dw, const float* const* gy, const float* const* x, int n_jobs, int batch, int cin, int cout, int hw,
                    void* workspace, int64_t workspace_bytes, void* stream, const char* who) {
    SIS_REQUIRE(sis_conv1x1_wgrad_f32_supported(batch, cin, cout, hw), "%s: %d x (%d -> %d) on %d pixels not supported", who, batch, cin, cout, hw);
    int tm, tn;
    const int shape = pwg_tile(cout, cin, &tm, &tn);
    PwgParams p;
    p.B = batch; p.Cout = cout; p.Cin = cin; p.HW = hw;
    p.chunks_total = batch * (hw / PK);
    const int tiles = (cout / tm) * (cin / tn);
    int slices = pwg_slices(batch, cin, cout, hw, n_jobs);  // bounded by the workspace the caller brought (every layer its own slabs)
    const int64_t n = (int64_t)cout * cin;
    if (!workspace || (int64_t)slices * n * 4 * n_jobs > workspace_bytes) slices = workspace ? (int)(workspace_bytes / (n * 4 * n_jobs)) : 1;
    if (slices < 1) slices = 1;
    p.chunks_per_slice = (p.chunks_total + slices - 1) / slices;
    slices = (p.chunks_total + p.chunks_per_slice - 1) / p.chunks_per_slice;
    p.slab_stride = slices > 1 ? n : 0;
    p.jobs = n_jobs > 1 ? n_jobs : 0;
    const int64_t job_stride = (int64_t)slices * n;
    PwgDwTab tab = {};
    for (int j = 0; j < n_jobs; ++j) {
        SIS_REQUIRE(dw[j] && gy[j] && x[j], "%s: null pointer in layer %d", who, j);
        SIS_REQUIRE((((uintptr_t)dw[j] | (uintptr_t)gy[j] | (uintptr_t)x[j] | (uintptr_t)workspace) & 15) == 0, "%s: pointers must be 16-byte aligned", who);
        p.gyj[j] = gy[j]; p.xj[j] = x[j]; tab.dw[j] = dw[j];
        p.outj[j] = slices > 1 ? (float*)workspace + j * job_stride : dw[j];
    }
    p.gy = gy[0]; p.x = x[0]; p.out = p.outj[0];
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (shape == 3) rc = launch_pwg<2, 4, 2>(p, tiles, slices, st);
    else if (shape == 0) rc = launch_pwg<4, 2, 1>(p, tiles, slices, st);
    else if (shape == 1) rc = launch_pwg<2, 4, 1>(p, tiles, slices, st);
    else rc = launch_pwg<8, 1, 1>(p, tiles, slices, st);
    if (rc) return rc;
    if (slices > 1) {
        hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3((unsigned)sis_cdiv(n, 256), n_jobs), dim3(256), 0, st, dw[0], (const float*)workspace,
                           slices, n, p.jobs ? job_stride : 0, tab);
        SIS_CHECK_LAUNCH("sis_conv1x1_wgrad_f32 (reduce)");
    }
    sis_kernel_name = "conv1x1_wgrad_f32_kernel";
    return 0;
}

extern "C" int sis_conv1x1_wgrad_f32(float* dw, const float* gy, const float* x, int batch, int cin, int cout, int hw,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    SIS_REQUIRE(dw && gy && x, "sis_conv1x1_wgrad_f32: null pointer");
    return pwg_jobs(&dw, &gy, &x, 1, batch, cin, cout, hw, workspace, workspace_bytes, stream, "sis_conv1x1_wgrad_f32");
}

/* The same for n_jobs layers of ONE shape (EMANet's repeated bottleneck units, queued during the backward): `dw`, `gy`, `x` are
 * HOST arrays of n_jobs device pointers; one tile launch (grid.z = layer) + one reduction launch per <= 16 layers, the K slices
 * planned for the layers' joint tile count (fewer, longer slices per layer: less slab traffic). */
extern "C" int sis_conv1x1_wgrad_f32_multi(float* const* dw, const float* const* gy, const float* const* x, int n_jobs, int batch, int cin,
                                           int cout, int hw, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n_jobs <= 0) return 0;
    SIS_REQUIRE(dw && gy && x, "sis_conv1x1_wgrad_f32_multi: null pointer");
    for (int j0 = 0; j0 < n_jobs; j0 += PWG_JOBS_MAX) {
        const int n = n_jobs - j0 < PWG_JOBS_MAX ? n_jobs - j0 : PWG_JOBS_MAX;
        const int rc = pwg_jobs(dw + j0, gy + j0, x + j0, n, batch, cin, cout, hw, workspace, workspace_bytes, stream, "sis_conv1x1_wgrad_f32_multi");
        if (rc) return rc;
    }
    return 0;
}
