// fp32 pointwise (1x1, stride 1) convolution of the EMANet training step on the fp32 matrix cores, direct on NCHW
// (reference call sites: networks/ema_net/network.py:24,29 Bottleneck conv1 / conv3, :106-107 downsample, :271-289 fc0 / fc1,
// :219-249 EMAU conv1 / conv2), forward and data gradient.
//
//   y[n][m][p] = sum_k A[m][k] * x[n][k][p]          (forward: A = weight [Cout][Cin];  data gradient: A = weight^T)
//
// v_mfma_f32_32x32x2_f32 (exact fp32, one A and one B value per lane): the B operand is a row of the NCHW tensor
// (lane = pixel), the A operand a row of the k-major weight image (lane = output channel) -- both conflict-free
// ds_read_b32.  x rows arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no registers).  The
// weight chunk [32 k][MT m] is a plain row copy by LDS-DMA when the source is k-major (the data gradient reads the
// weight tensor [Cout][Cin] as [k][m] directly); for the forward ([m][k] in memory) each lane loads one float4 of a
// weight row and writes its four values down a column of the k-major image (lanes = consecutive m: conflict-free).
// 32-channel chunks, double-buffered, ONE barrier per 64 MFMAs per wave.
#include <cstdlib>
#include <type_traits>
#include "sis_common.h"

namespace {

#ifndef PW_OCC
#define PW_OCC 4  // waves per SIMD the register allocation aims at (2 workgroups per CU).  6 (three per CU, 80 VGPRs) measured 527 against 541 images/s on EMANet.
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MT_, int NPIX_, int KC_>
struct PwCfg {
    static constexpr int MT = MT_, NPIX = NPIX_, KC = KC_;
    static constexpr int WM = MT == 128 ? (NPIX == 256 ? 2 : 4) : (NPIX == 256 ? 1 : 2);
    static constexpr int WN = 8 / WM;
    static constexpr int MB = MT / 32 / WM, NB = NPIX / 32 / WN;
    static_assert(MB >= 1 && NB >= 1 && MB * WM * 32 == MT && NB * WN * 32 == NPIX, "wave layout");
    static constexpr int A_FLOATS = KC * MT, X_FLOATS = KC * NPIX, STAGE = A_FLOATS + X_FLOATS;
    static constexpr int LDS_BYTES = 2 * STAGE * 4;
    static constexpr int X_PIECES = X_FLOATS * 4 / 1024;   // 1 KiB DMA pieces per chunk
    static constexpr int A_PIECES = A_FLOATS * 4 / 1024;
    static constexpr int A_VEC = (A_FLOATS / 4 + 511) / 512;  // float4 loads per thread per chunk (register-staged weights)
    static constexpr int OCC = 2 * STAGE * 4 <= 52 * 1024 ? PW_OCC : 2;  // waves per SIMD the LDS footprint allows (3 or 1 workgroups per CU)
};

struct PwParams {
    const float* x;     // [N][K][HW]
    const float* a;     // A_KMAJOR: [K][Mtot], else [Mtot][K]
    const float* bias;  // [Mtot] or null
    float* y;           // [N][Mtot][HW]
    const float* accum; // [N][Mtot][HW] added to the result (data gradient + the skip connection's gradient) or null
    int N, K, M, HW, px_tiles;
};

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <typename C, bool A_KMAJOR, bool HAS_BIAS, bool HAS_ACC = false>
__global__ __launch_bounds__(512, C::OCC) void conv1x1_f32_kernel(PwParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave % C::WM, wn = wave / C::WM;
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (id % 8 = XCD group).  The n_m output-channel
    // tiles that share an input tile take CONSECUTIVE slots of ONE group, so the tile is pulled into that XCD's L2 once and
    // re-read there (with the output-channel tile as the slowest grid index it came from beyond L2 once per tile: 16x for
    // 2048 output channels).
    const int n_m = (p.M + C::MT - 1) / C::MT;
    const int slot = blockIdx.x >> 3;
    const int xt = (slot / n_m) * 8 + (blockIdx.x & 7);
    if (xt >= p.N * p.px_tiles) return;
    const int n = xt / p.px_tiles, pt = xt % p.px_tiles;
    const int p0 = pt * C::NPIX, m0 = (slot % n_m) * C::MT;
    const float* xin = p.x + (int64_t)n * p.K * p.HW;
    const int nchunks = p.K / C::KC;

    float4 areg[A_KMAJOR ? 1 : C::A_VEC];

    auto dma_x = [&](int chunk, int stage) {
        float* dst = lds + stage * C::STAGE + C::A_FLOATS;
        for (int piece = wave; piece < C::X_PIECES; piece += 8) {
            const int e = piece * 256 + lane * 4;          // float index inside the [KC][NPIX] chunk
            const int row = e / C::NPIX, col = e % C::NPIX;
            if (p0 + col + 3 < p.HW)
                glds16(xin + (int64_t)(chunk * C::KC + row) * p.HW + p0 + col, dst + piece * 256);
        }
    };
    auto dma_a = [&](int chunk, int stage) {  // k-major source: rows of Mtot floats
        float* dst = lds + stage * C::STAGE;
        for (int piece = wave; piece < C::A_PIECES; piece += 8) {
            const int e = piece * 256 + lane * 4;
            const int row = e / C::MT, col = e % C::MT;
            if (m0 + col + 3 < p.M)
                glds16(p.a + (int64_t)(chunk * C::KC + row) * p.M + m0 + col, dst + piece * 256);
        }
    };
    auto load_a = [&](int chunk) {  // m-major source: thread -> (m = e % MT, k quad = e / MT)
#pragma unroll
        for (int i = 0; i < C::A_VEC; ++i) {
            const int e = tid + i * 512;
            const int m = e % C::MT, kq = e / C::MT;
            areg[i] = (e < C::A_FLOATS / 4 && m0 + m < p.M)
                          ? *reinterpret_cast<const float4*>(p.a + (int64_t)(m0 + m) * p.K + chunk * C::KC + 4 * kq)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_a = [&](int stage) {
        float* dst = lds + stage * C::STAGE;
#pragma unroll
        for (int i = 0; i < C::A_VEC; ++i) {
            const int e = tid + i * 512;
            if (e >= C::A_FLOATS / 4) continue;
            const int m = e % C::MT, kq = e / C::MT;
            dst[(4 * kq + 0) * C::MT + m] = areg[i].x;
            dst[(4 * kq + 1) * C::MT + m] = areg[i].y;
            dst[(4 * kq + 2) * C::MT + m] = areg[i].z;
            dst[(4 * kq + 3) * C::MT + m] = areg[i].w;
        }
    };

    f32x16 acc[C::MB][C::NB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    if constexpr (A_KMAJOR) dma_a(0, 0);
    else { load_a(0); store_a(0); }
    dma_x(0, 0);
    __syncthreads();

    const int a_off = h * C::MT + wm * (C::MB * 32) + r;                       // + kp * 2 * MT + mb * 32
    const int b_off = C::A_FLOATS + h * C::NPIX + wn * (C::NB * 32) + r;       // + kp * 2 * NPIX + nb * 32

    float a[2][C::MB], b[2][C::NB];   // fragments one k-pair ahead, carried from chunk to chunk
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) a[0][mb] = lds[a_off + mb * 32];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) b[0][nb] = lds[b_off + nb * 32];
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        if (more) {
            dma_x(chunk + 1, cur ^ 1);
            if constexpr (A_KMAJOR) dma_a(chunk + 1, cur ^ 1);
            else load_a(chunk + 1);
        }
        const float* st = lds + cur * C::STAGE;
        // Fragments one k-pair ahead: the ds_reads of pair kp + 1 are issued BEFORE the MFMAs of pair kp, so the wait in front
        // of a pair's MFMAs (a counted lgkmcnt) finds its operands already there.  As the compiler scheduled the plain loop,
        // every 4 MFMAs were preceded by `ds_read x2; s_waitcnt lgkmcnt(0)`: a full LDS round trip exposed per 256 MFMA cycles.
        // The chunk's barrier sits in front of its LAST pair's MFMAs (that pair's fragments are in registers by then, the weights
        // of the next chunk are stored): the first fragments of chunk + 1 are requested right behind it, under those MFMAs, instead
        // of in front of the next chunk's first.  (KC / 2 is even: the pair after the last one lands in register set 0 again.)
        static_assert((C::KC / 2) % 2 == 0, "fragment sets alternate per pair and wrap to set 0 at the chunk end");
#pragma unroll
        for (int kp = 0; kp < C::KC / 2; ++kp) {
            const int cb = kp & 1, nx = cb ^ 1;
            if (kp + 1 < C::KC / 2) {
#pragma unroll
                for (int mb = 0; mb < C::MB; ++mb) a[nx][mb] = st[a_off + (kp + 1) * 2 * C::MT + mb * 32];
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) b[nx][nb] = st[b_off + (kp + 1) * 2 * C::NPIX + nb * 32];
            } else {
                if constexpr (!A_KMAJOR) {
                    if (more) store_a(cur ^ 1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of chunk + 1
                __syncthreads();
                const float* sn = lds + (cur ^ 1) * C::STAGE;
#pragma unroll
                for (int mb = 0; mb < C::MB; ++mb) a[nx][mb] = sn[a_off + mb * 32];
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb) b[nx][nb] = sn[b_off + nb * 32];
            }
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][mb], b[cb][nb], acc[mb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // keep the pairs in program order: reads of kp + 1, then MFMAs of kp
        }
    }

    // ---- epilogue.  The bias of this wave's rows is fetched once (not per element), a row's address is the tile's base plus a
    // multiple of the plane stride, and full tiles (the usual case) carry no per-row bounds checks.
    const int mrow0 = m0 + wm * (C::MB * 32) + 4 * h;  // row (mb, i) = mrow0 + mb * 32 + (i & 3) + 8 * (i >> 2)
    // (HAS_BIAS is a template parameter: the 16 * MB bias registers of the biased instance would cost the unbiased one -- nearly
    // every layer of the network -- a workgroup per CU: 112 instead of 84 VGPRs)
    float bv[HAS_BIAS ? C::MB : 1][16];
    if constexpr (HAS_BIAS) {
#pragma unroll
        for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
            for (int i = 0; i < 16; ++i) bv[mb][i] = p.bias[min(mrow0 + mb * 32 + (i & 3) + 8 * (i >> 2), p.M - 1)];  // (rows >= M are not stored)
    }
    float* ybase = p.y + ((int64_t)n * p.M + mrow0) * p.HW + p0 + wn * (C::NB * 32) + r;
    const float* abase = HAS_ACC ? p.accum + ((int64_t)n * p.M + mrow0) * p.HW + p0 + wn * (C::NB * 32) + r : nullptr;
    auto store_tile = [&](auto checked) {
#pragma unroll
        for (int nb = 0; nb < C::NB; ++nb) {
            if (p0 + wn * (C::NB * 32) + nb * 32 + r >= p.HW) continue;
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ro = mb * 32 + (i & 3) + 8 * (i >> 2);
                    if (!decltype(checked)::value || mrow0 + ro < p.M) {
                        float v = HAS_BIAS ? acc[mb][nb][i] + bv[mb][i] : acc[mb][nb][i];
                        if constexpr (HAS_ACC) v += abase[ro * p.HW + nb * 32];
                        ybase[ro * p.HW + nb * 32] = v;
                    }
                }
        }
    };
    if (m0 + C::MT <= p.M) store_tile(std::false_type());
    else store_tile(std::true_type());
}

template <typename C>
int launch_pw(const PwParams& p, int a_kmajor, hipStream_t st, const char* name) {
    typedef void (*kern_t)(PwParams);
    static const kern_t table[3][2] = {{conv1x1_f32_kernel<C, false, false>, conv1x1_f32_kernel<C, false, true>},
                                       {conv1x1_f32_kernel<C, true, false>, conv1x1_f32_kernel<C, true, true>},
                                       {conv1x1_f32_kernel<C, true, false, true>, conv1x1_f32_kernel<C, true, false, true>}};
    static bool attr_set[3][2] = {};
    const int km = p.accum ? 2 : (a_kmajor ? 1 : 0), hb = p.bias ? 1 : 0;
    const kern_t kern = table[km][hb];
    if (!attr_set[km][hb]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return sis_fail("%s: cannot raise the LDS limit: %s", name, hipGetErrorString(e));
        attr_set[km][hb] = true;
    }
    dim3 grid(8 * sis_cdiv(p.M, C::MT) * sis_cdiv((int64_t)p.N * p.px_tiles, 8));
    if (a_kmajor) { SIS_OCC_REPORT((conv1x1_f32_kernel<C, true, false>), 512, C::LDS_BYTES); }
    else { SIS_OCC_REPORT((conv1x1_f32_kernel<C, false, false>), 512, C::LDS_BYTES); }
    hipLaunchKernelGGL(kern, grid, dim3(512), C::LDS_BYTES, st, p);
    SIS_CHECK_LAUNCH(name);
    sis_kernel_name = name;
    return 0;
}

bool pw_ok(int k, int m, int hw) { return k > 0 && m > 0 && hw > 0 && k % 32 == 0 && m % 4 == 0 && hw % 4 == 0; }

template <int KC>
int dispatch_pw(PwParams& p, int batch, int data_gradient, hipStream_t st) {
    // Tile: the largest of 128x256, 128x128, 64x256, 64x128 (output channels x pixels) that still gives one workgroup per CU
    // (narrow layers -- 512 -> 128 channels on 32 x 32 maps -- had 128 tiles of 128 x 128 for 256 CUs), else the smallest.
    static const int mts[4] = {128, 128, 64, 64}, nps[4] = {256, 128, 256, 128};
    int pick = 3;
    for (int i = 0; i < 4; ++i) {
        if (mts[i] == 128 && p.M <= 64) continue;
        static const int min_tiles = getenv("SIS_PW_MIN_TILES") ? atoi(getenv("SIS_PW_MIN_TILES")) : 256;
        if ((int64_t)batch * sis_cdiv(p.HW, nps[i]) * sis_cdiv(p.M, mts[i]) >= min_tiles) { pick = i; break; }
    }
    p.px_tiles = sis_cdiv(p.HW, nps[pick]);
    switch (pick) {
        case 0: return launch_pw<PwCfg<128, 256, KC>>(p, data_gradient, st, KC == 16 ? "conv1x1_f32_kernel<128,256,16>" : "conv1x1_f32_kernel<128,256,32>");
        case 1: return launch_pw<PwCfg<128, 128, KC>>(p, data_gradient, st, KC == 16 ? "conv1x1_f32_kernel<128,128,16>" : "conv1x1_f32_kernel<128,128,32>");
        case 2: return launch_pw<PwCfg<64, 256, KC>>(p, data_gradient, st, KC == 16 ? "conv1x1_f32_kernel<64,256,16>" : "conv1x1_f32_kernel<64,256,32>");
        default: return launch_pw<PwCfg<64, 128, KC>>(p, data_gradient, st, KC == 16 ? "conv1x1_f32_kernel<64,128,16>" : "conv1x1_f32_kernel<64,128,32>");
    }
}

}  // namespace

extern "C" int sis_conv1x1_f32_supported(int cin, int cout, int hw) { return pw_ok(cin, cout, hw) && pw_ok(cout, cin, hw) ? 1 : 0; }

static int conv1x1_impl(float* y, const float* x, const float* weight, const float* bias, const float* accum, int batch, int cin,
                        int cout, int hw, int data_gradient, void* stream) {
    if (batch <= 0) return 0;
    SIS_REQUIRE(y && x && weight, "sis_conv1x1_f32: null pointer");
    SIS_REQUIRE(!accum || (data_gradient && !bias && (((uintptr_t)accum) & 15) == 0), "sis_conv1x1_f32: the accumulate input is for the data gradient (no bias), 16-byte aligned");
    // forward: contraction over cin, A = weight [cout][cin] (m-major).  data gradient: x is dL/dy [batch][cout][hw], the
    // contraction runs over cout and A = the same weight tensor read as [k = cout][m = cin] (k-major); y is dL/dx.
    PwParams p;
    p.x = x; p.a = weight; p.bias = bias; p.y = y; p.accum = accum; p.N = batch; p.HW = hw;
    p.K = data_gradient ? cout : cin;
    p.M = data_gradient ? cin : cout;
    SIS_REQUIRE(pw_ok(p.K, p.M, hw), "sis_conv1x1_f32: %d -> %d channels on %d pixels (K %% 32, M %% 4, pixels %% 4 must be 0)", p.K, p.M, hw);
    SIS_REQUIRE((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y) & 15) == 0, "sis_conv1x1_f32: pointers must be 16-byte aligned");
    SIS_REQUIRE((int64_t)p.K * hw < (1LL << 31) && (int64_t)p.M * hw < (1LL << 31), "sis_conv1x1_f32: planes exceed 2^31 elements");
    hipStream_t st = (hipStream_t)stream;
    // 16-channel chunks: 24 KB per stage, three workgroups per compute unit cover each other's barriers and epilogues;
    // 32-channel chunks (SIS_PW_KC=32): one workgroup per unit, half the barriers
    static const int kc = getenv("SIS_PW_KC") ? atoi(getenv("SIS_PW_KC")) : 16;
    return kc == 32 ? dispatch_pw<32>(p, batch, data_gradient, st) : dispatch_pw<16>(p, batch, data_gradient, st);
}

extern "C" int sis_conv1x1_f32(float* y, const float* x, const float* weight, const float* bias, int batch, int cin, int cout,
                               int hw, int data_gradient, void* stream) {
    return conv1x1_impl(y, x, weight, bias, nullptr, batch, cin, cout, hw, data_gradient, stream);
}

extern "C" int sis_conv1x1_f32_dgrad_add(float* dx, const float* dy, const float* weight, const float* skip_grad, int batch, int cin,
                                         int cout, int hw, void* stream) {
    SIS_REQUIRE(skip_grad, "sis_conv1x1_f32_dgrad_add: null pointer");
    return conv1x1_impl(dx, dy, weight, nullptr, skip_grad, batch, cin, cout, hw, 1, stream);
}
