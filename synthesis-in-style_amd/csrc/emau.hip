// EMANet's Expectation-Maximisation Attention Unit on the fp32 matrix cores (BASELINE.json configs[3]).
//
// Reference: networks/ema_net/network.py:219-249 (EMAU.forward), the part between conv1 and conv2, all of it under no_grad:
//     mu = self.mu.repeat(b)                                  [b, c, k]   c = 512 channels, k = 64 bases
//     3 x {  z  = softmax_k( x^T mu )                         [b, n, k]   n = h w = 1024 pixels
//            z_ = z / (1e-6 + sum_n z)
//            mu = l2norm_c( x z_ )  }                         mu / (1e-6 + ||mu||_c)
//     x = relu( mu z^T )                                      [b, c, n]
// torch runs it as 7 batched library GEMMs + ~15 element-wise / reduction launches (1.0 ms of the 28 ms step at B = 16).  Here:
// 7 launches of three kernels, every product on v_mfma_f32_32x32x2_f32 (exact fp32), the two normalisations folded into the
// consumers:
//   emau_e_kernel  (grid n/64 x b)  logits tile [64 k x 64 n] = mu^T x over all channels, softmax over k in LDS, z slice
//                                   -> HBM, partial column sums sum_{n in slice} z[n,k]
//   emau_m_kernel  (grid c/32 x b)  mu_raw[c,k] = (sum_n x[c,n] z[n,k]) / (1e-6 + sum_n z[n,k]) for 32 channels, and the
//                                   partial squared norms sum_{c in slice} mu_raw^2 per base
//   emau_recon_kernel (grid n/64 x b)  mu = mu_raw / (1e-6 + ||mu_raw||)  (returned),  x = relu(mu z^T)
// The E step partitions the pixels (it needs every channel), the M step partitions the channels (it needs every pixel), so no
// partial product ever crosses workgroups: z and mu_raw go through L2 / MALL (4 + 2 MB at B = 16), and the only cross-
// workgroup sums are the 16 partial column sums / squared norms per base, added by each consumer in slice order
// (deterministic, no atomics).  The l2 normalisation of step i is applied by step i + 1 to its logits (a per-base scale).
//
// MFMA operand convention used throughout (as csrc/gen_small_ops.hip): D = mfma(a, b): lane l supplies a = A[row l & 31][kk = l >> 5],
// b = B[kk = l >> 5][col l & 31]; afterwards the lane holds column l & 31, rows (i & 3) + 8 (i >> 2) + 4 (l >> 5), i = 0..15.
#include "sis_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;   // (staging registers: HIP's float4 struct arrays end up in scratch)

constexpr int EK = 64;        // bases (the kernels are written for k = 64: two 32-wide MFMA tiles)
constexpr int E_NS = 64;      // pixels per workgroup of the E and reconstruction kernels
constexpr int E_CC = 64;      // channels per staged chunk (E, reconstruction)
constexpr int M_CS = 32;      // channels per workgroup of the M kernel
constexpr int M_NC = 64;      // pixels per staged chunk and n-half (M)
constexpr int PAD = 65;       // padded row length of LDS images read with the lane index on the ROW axis (conflict-free b32 reads)
constexpr int E_LDS_BYTES = (2 * E_CC * E_NS + 2 * E_CC * EK + EK) * 4;                               // 65 792
constexpr int M_LDS_BYTES = (2 * 2 * M_NC * EK + 2 * 2 * M_CS * PAD + 2 * 16 * 64 + EK) * 4;         // 107 264
static_assert(E_NS * PAD <= 2 * E_CC * E_NS, "the logits image aliases the x stages");

struct EmauParams {
    const float* x;           // [b][c][n]
    const float* mu_in;       // E: bases entering this round, [c][k] (mu_bstride 0: the shared buffer) or [b][c][k] (mu_raw)
    long long mu_bstride;
    float* z;                 // [b][n][k]
    float* mu_raw;            // [b][c][k]
    float* colsum_part;       // [b][n / 64][k]
    float* sq_part;           // [b][c / 32][k]   (E: nullptr in the first round -- the buffer is already normalised)
    float* x_out;             // [b][c][n]
    float* mu_out;            // [b][c][k]
    int c, n;
};

__device__ __forceinline__ int acc_row(int i, int half) { return (i & 3) + 8 * (i >> 2) + 4 * half; }

// scale[k] = 1 / (1e-6 + sqrt(sum over channel slices of sq_part[k])), slices added in order; 1 when sq_part is null
__device__ __forceinline__ float l2_scale(const float* sq_part, int b, int cslices, int k) {
    if (!sq_part) return 1.f;
    const float* p = sq_part + (long long)b * cslices * EK + k;
    float s = 0.f;
    for (int i = 0; i < cslices; ++i) s += p[i * EK];
    return 1.f / (1e-6f + sqrtf(s));
}

// ---- E step -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emau_e_kernel(EmauParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];            // E_LDS_BYTES (above the 64 KB static limit)
    float (*xs)[E_CC][E_NS] = reinterpret_cast<float (*)[E_CC][E_NS]>(lds);                     // [2][64][64]  2 x 16 KB
    float (*ms)[E_CC][EK] = reinterpret_cast<float (*)[E_CC][EK]>(lds + 2 * E_CC * E_NS);       // [2][64][64]  2 x 16 KB
    float* scale_s = lds + 2 * E_CC * E_NS + 2 * E_CC * EK;                                     // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y, n0 = blockIdx.x * E_NS;
    const int nsub = wave & 1, kt = wave >> 1;
    const float* xb = p.x + (long long)b * p.c * p.n + n0;
    const float* mb = p.mu_in + (long long)b * p.mu_bstride;
    if (tid < EK) scale_s[tid] = l2_scale(p.sq_part, b, p.c / M_CS, tid);

    f32x4 xr[4], mr[4];   // one chunk in flight: 64 channels x (64 pixels + 64 bases) = 2048 float4 over 256 threads
    auto load = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, row = i >> 4, q = i & 15;
            xr[j] = *reinterpret_cast<const f32x4*>(xb + (long long)(c0 + row) * p.n + 4 * q);
            mr[j] = *reinterpret_cast<const f32x4*>(mb + (long long)(c0 + row) * EK + 4 * q);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, row = i >> 4, q = i & 15;
            *reinterpret_cast<f32x4*>(&xs[buf][row][4 * q]) = xr[j];
            *reinterpret_cast<f32x4*>(&ms[buf][row][4 * q]) = mr[j];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int chunks = p.c / E_CC;
    load(0);
    store(0);
    __syncthreads();
    auto multiply = [&](int buf) {
        const float* ap = &ms[buf][half][kt * 32 + l31];      // A[row = base][kk = channel]
        const float* bp = &xs[buf][half][nsub * 32 + l31];    // B[kk = channel][col = pixel]
#pragma unroll
        for (int st = 0; st < E_CC / 2; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st * EK], bp[2 * st * E_NS], acc, 0, 0, 0);
    };
    for (int ch = 0; ch + 1 < chunks; ++ch) {   // (no branch around the loads: the staged values stay in registers)
        const int buf = ch & 1;
        load((ch + 1) * E_CC);
        multiply(buf);
        store(buf ^ 1);   // (last read in iteration ch - 1, behind that iteration's barrier)
        __syncthreads();
    }
    multiply((chunks - 1) & 1);
    __syncthreads();
    // logits tile -> LDS as zs[pixel][base] (aliases the x stages: every wave is past its last read of them)
    float* zs = &xs[0][0][0];   // [64][PAD]  (64 * 65 * 4 = 16 640 B <= 32 KB)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = kt * 32 + acc_row(i, half);
        zs[(nsub * 32 + l31) * PAD + k] = acc[i] * scale_s[k];
    }
    __syncthreads();
    // softmax over the 64 bases of a pixel: 4 lanes per pixel, 16 bases each
    {
        const int px = tid >> 2, q = tid & 3;
        float v[16], mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = zs[px * PAD + q * 16 + j]; mx = fmaxf(mx, v[j]); }
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] = expf(v[j] - mx); sum += v[j]; }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const float inv = 1.f / sum;
        float* zg = p.z + ((long long)b * p.n + n0 + px) * EK + q * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) { v[j] *= inv; zs[px * PAD + q * 16 + j] = v[j]; }
#pragma unroll
        for (int j = 0; j < 16; j += 4) *reinterpret_cast<float4*>(zg + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
    }
    __syncthreads();
    if (tid < EK) {   // partial column sum of this slice, pixels in order
        float s = 0.f;
        for (int px = 0; px < E_NS; ++px) s += zs[px * PAD + tid];
        p.colsum_part[((long long)b * (p.n / E_NS) + blockIdx.x) * EK + tid] = s;
    }
}

// ---- M step -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emau_m_kernel(EmauParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];            // M_LDS_BYTES
    float (*zs)[2][M_NC][EK] = reinterpret_cast<float (*)[2][M_NC][EK]>(lds);                   // [stage][n half][pixel][base]  2 x 32 KB
    float (*xs)[2][M_CS][PAD] = reinterpret_cast<float (*)[2][M_CS][PAD]>(lds + 2 * 2 * M_NC * EK);  // [stage][n half][channel][pixel], padded rows
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(lds + 2 * 2 * M_NC * EK + 2 * 2 * M_CS * PAD);   // [k tile][i][lane]: partial tiles of the second n half
    float* inv_s = lds + 2 * 2 * M_NC * EK + 2 * 2 * M_CS * PAD + 2 * 16 * 64;                  // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y, c0 = blockIdx.x * M_CS;
    const int kt = wave & 1, nh = wave >> 1;
    const int nslices = p.n / E_NS, nhalf = p.n / 2;
    const float* xb = p.x + ((long long)b * p.c + c0) * p.n;
    const float* zb = p.z + (long long)b * p.n * EK;
    if (tid < EK) {
        const float* cp = p.colsum_part + (long long)b * nslices * EK + tid;
        float s = 0.f;
        for (int i = 0; i < nslices; ++i) s += cp[i * EK];
        inv_s[tid] = 1.f / (1e-6f + s);
    }
    f32x4 xr[2][2], zr[2][4];   // per n half: 32 channels x 64 pixels = 512 float4, 64 pixels x 64 bases = 1024 float4
    auto load = [&](int nn) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nbase = h * nhalf + nn;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 256 * j, row = i >> 4, q = i & 15;
                xr[h][j] = *reinterpret_cast<const f32x4*>(xb + (long long)row * p.n + nbase + 4 * q);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = tid + 256 * j;
                zr[h][j] = *reinterpret_cast<const f32x4*>(zb + (long long)nbase * EK + 4 * i);
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 256 * j, row = i >> 4, q = i & 15;
                float* d = &xs[buf][h][row][4 * q];
                d[0] = xr[h][j][0]; d[1] = xr[h][j][1]; d[2] = xr[h][j][2]; d[3] = xr[h][j][3];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = tid + 256 * j;
                *reinterpret_cast<f32x4*>(&zs[buf][h][0][0] + 4 * i) = zr[h][j];
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int chunks = nhalf / M_NC;
    load(0);
    store(0);
    __syncthreads();
    auto multiply = [&](int buf) {
        const float* ap = &xs[buf][nh][l31][half];               // A[row = channel][kk = pixel]
        const float* bp = &zs[buf][nh][half][kt * 32 + l31];     // B[kk = pixel][col = base]
#pragma unroll
        for (int st = 0; st < M_NC / 2; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], bp[2 * st * EK], acc, 0, 0, 0);
    };
    for (int ch = 0; ch + 1 < chunks; ++ch) {
        const int buf = ch & 1;
        load((ch + 1) * M_NC);
        multiply(buf);
        store(buf ^ 1);
        __syncthreads();
    }
    multiply((chunks - 1) & 1);
    __syncthreads();
    // the two pixel halves: waves 2, 3 hand their partial tile to waves 0, 1 (first half + second half, a fixed order)
    if (nh == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[kt][i][lane] = acc[i];
    }
    __syncthreads();
    if (nh == 0) {
        const int k = kt * 32 + l31;
        const float inv = inv_s[k];
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = (acc[i] + red[kt][i][lane]) * inv;
            p.mu_raw[((long long)b * p.c + c0 + acc_row(i, half)) * EK + k] = v;
            sq += v * v;
        }
        sq += __shfl_xor(sq, 32);   // the other 16 channels of this base
        if (half == 0) p.sq_part[((long long)b * (p.c / M_CS) + blockIdx.x) * EK + k] = sq;
    }
}

// ---- normalised bases + reconstruction --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emau_recon_kernel(EmauParams p) {
    __shared__ float zs[E_NS][PAD];          // [pixel][base]
    __shared__ float ms[2][E_CC][PAD];       // [stage][channel][base]
    __shared__ float scale_s[EK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y, n0 = blockIdx.x * E_NS;
    const int nsub = wave & 1, ct = wave >> 1;
    const float* mb = p.mu_raw + (long long)b * p.c * EK;
    if (tid < EK) scale_s[tid] = l2_scale(p.sq_part, b, p.c / M_CS, tid);
    {
        const float* zg = p.z + ((long long)b * p.n + n0) * EK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, row = i >> 4, q = i & 15;
            const float4 v = *reinterpret_cast<const float4*>(zg + 4 * i);
            float* d = &zs[row][4 * q];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();   // scale_s, zs
    float4 mr[4];
    auto load = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mr[j] = *reinterpret_cast<const float4*>(mb + (long long)c0 * EK + 4 * (tid + 256 * j));
    };
    auto store = [&](int buf, int c0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, row = i >> 4, q = i & 15;
            const float4 v = make_float4(mr[j].x * scale_s[4 * q], mr[j].y * scale_s[4 * q + 1], mr[j].z * scale_s[4 * q + 2],
                                         mr[j].w * scale_s[4 * q + 3]);
            float* d = &ms[buf][row][4 * q];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            if (blockIdx.x == 0) *reinterpret_cast<float4*>(p.mu_out + ((long long)b * p.c + c0) * EK + 4 * i) = v;   // the bases returned
        }
    };
    const int chunks = p.c / E_CC;
    load(0);
    store(0, 0);
    __syncthreads();
    for (int ch = 0; ch < chunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < chunks) load((ch + 1) * E_CC);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* ap = &ms[buf][ct * 32 + l31][half];      // A[row = channel][kk = base]
        const float* bp = &zs[nsub * 32 + l31][half];         // B[kk = base][col = pixel]
#pragma unroll
        for (int st = 0; st < EK / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], bp[2 * st], acc, 0, 0, 0);
        float* og = p.x_out + ((long long)b * p.c + ch * E_CC + ct * 32) * p.n + n0 + nsub * 32 + l31;
#pragma unroll
        for (int i = 0; i < 16; ++i) og[(long long)acc_row(i, half) * p.n] = fmaxf(acc[i], 0.f);
        if (ch + 1 < chunks) store(buf ^ 1, (ch + 1) * E_CC);
        __syncthreads();
    }
}

}  // namespace

extern "C" int sis_emau_supported(int batch, int channels, int pixels, int bases) {
    return (batch > 0 && bases == EK && channels >= E_CC && channels % E_CC == 0 && pixels >= 2 * M_NC && pixels % (2 * M_NC) == 0) ? 1 : 0;
}

extern "C" int64_t sis_emau_workspace_floats(int batch, int channels, int pixels, int bases) {
    return (int64_t)batch * ((int64_t)pixels * bases + (int64_t)channels * bases + (int64_t)(pixels / E_NS) * bases +
                             (int64_t)(channels / M_CS) * bases);
}

extern "C" int sis_emau_forward(float* x_out, float* mu_out, const float* x, const float* mu0, float* workspace, int batch,
                                int channels, int pixels, int bases, int stages, void* stream) {
    if (batch <= 0) return 0;
    SIS_REQUIRE(x_out && mu_out && x && mu0 && workspace, "sis_emau_forward: null pointer");
    SIS_REQUIRE(sis_emau_supported(batch, channels, pixels, bases),
                "sis_emau_forward: needs %d bases, channels %% %d == 0 and pixels %% %d == 0 (got c=%d n=%d k=%d)", EK, E_CC, 2 * M_NC,
                channels, pixels, bases);
    SIS_REQUIRE(stages >= 1, "sis_emau_forward: at least one EM round");
    SIS_REQUIRE((((uintptr_t)x | (uintptr_t)mu0 | (uintptr_t)x_out | (uintptr_t)mu_out | (uintptr_t)workspace) & 15) == 0,
                "sis_emau_forward: 16-byte alignment");
    SIS_REQUIRE(batch <= 65535, "sis_emau_forward: batch %d", batch);
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&emau_e_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&emau_m_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, M_LDS_BYTES);
        if (e != hipSuccess) return sis_fail("sis_emau_forward: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    EmauParams p;
    p.x = x; p.c = channels; p.n = pixels; p.x_out = x_out; p.mu_out = mu_out;
    p.z = workspace;
    p.mu_raw = p.z + (int64_t)batch * pixels * EK;
    p.colsum_part = p.mu_raw + (int64_t)batch * channels * EK;
    p.sq_part = p.colsum_part + (int64_t)batch * (pixels / E_NS) * EK;
    float* sq = p.sq_part;
    const dim3 grid_n(pixels / E_NS, batch), grid_c(channels / M_CS, batch);
    for (int s = 0; s < stages; ++s) {
        EmauParams q = p;
        if (s == 0) { q.mu_in = mu0; q.mu_bstride = 0; q.sq_part = nullptr; }   // the buffer is l2-normalised already
        else { q.mu_in = p.mu_raw; q.mu_bstride = (long long)channels * EK; q.sq_part = sq; }
        hipLaunchKernelGGL(emau_e_kernel, grid_n, dim3(256), E_LDS_BYTES, st, q);
        SIS_CHECK_LAUNCH("emau_e_kernel");
        hipLaunchKernelGGL(emau_m_kernel, grid_c, dim3(256), M_LDS_BYTES, st, p);
        SIS_CHECK_LAUNCH("emau_m_kernel");
    }
    hipLaunchKernelGGL(emau_recon_kernel, grid_n, dim3(256), 0, st, p);
    SIS_CHECK_LAUNCH("emau_recon_kernel");
    sis_kernel_name = "emau_kernels";
    return 0;
}
