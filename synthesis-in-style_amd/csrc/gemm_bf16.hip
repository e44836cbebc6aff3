// bf16 GEMM of the TransUNet ViT encoder on the CDNA4 matrix cores, with the element-wise tail of every Linear layer
// fused into the epilogue (BASELINE.json configs[4]: 12 blocks x 4 Linear layers at 8 192 tokens, forward, data
// gradient, weight gradient).
//
// Reference call sites (networks/trans_u_net/vit_seg_modeling.py): Attention.query/key/value/out :60-67,76-96,
// Mlp.fc1 / act_fn / dropout / fc2 / dropout :104-122, Block residual adds :181-189.  torch runs each of them as a
// library GEMM followed by separate bias / GELU / dropout / add / cast kernels; here one launch does
//     C = epilogue(op(A) op(B)),  fp32 accumulation on v_mfma_f32_16x16x32_bf16
// for the three operand layouts a Linear layer's three GEMMs have:
//     NT  forward            y[M,N]   = x[M,K]  W[N,K]^T       both operands K-contiguous ("row operands")
//     NN  data gradient      dx[M,N]  = g[M,K]  W[K,N]         W is K-major
//     TN  weight gradient    dW[M,N]  = g[K,M]^T x[K,N]        both K-major (K = tokens), fp32 output, split over K
//
// Tile: BM x BN x 64, one wave per 64 x 64 sub-tile (4 x 4 MFMA blocks, 64 accumulator registers), 2 LDS stages filled
// by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, range-checked by the buffer descriptor: rows past the
// end of a tensor read as zeros, so M and the TN contraction length need not be tile multiples).  One barrier per
// 64-deep K step: wait for the stage's DMA, barrier, issue the next stage's DMA, then fragments + 32 MFMAs.
// LDS images (the DMA writes linearly, so the swizzle is applied to the per-lane SOURCE address and again on the read):
//   row operand [rows][64 k] = 128-B rows, 16-B chunk c of row r at position c ^ (r & 7): every ds_read_b128 lane group
//     of a fragment (16 rows x 4 chunks) falls on 16 different 16-B slots of the 256-B bank row;
//   K-major operand [64 k][cols] = 2*cols-B rows, chunk c of k-row r at c ^ f(r), f(r) = ((r & 3) << 2) | ((r >> 2) & 3):
//     fragments come out of ds_read_b64_tr_b16 (hardware transpose: 4 k-rows x 16 columns per 16 lanes), whose 32-lane
//     half reads 8 rows x 32 B spread over all 64 banks.
// MFMA orientation: D = Bfrag x Afrag, i.e. the accumulator's lane index is the output ROW m and its 4 registers are 4
// CONSECUTIVE columns n: the epilogue loads / stores 8 or 16 contiguous bytes per lane and block (bias, residual,
// pre-activation), no LDS round trip.
// Workgroup order: id % 8 = XCD group (round-robin dispatch); the column tiles that share an A row tile take consecutive
// slots of one group, so each A tile is fetched into one XCD's L2 once; the weights (<= 4.7 MB) stay L2 / MALL resident.
#include <type_traits>
#include "vit_common.h"
#include "gemm256.h"

namespace {

typedef unsigned short u16;
typedef sis_bf16x8 bf16x8;
typedef sis_bf16x4 bf16x4;
typedef sis_f32x4 f32x4;
typedef __attribute__((address_space(3))) void lds_void;

enum { LAYOUT_NT = 0, LAYOUT_NN = 1, LAYOUT_TN = 2 };

constexpr int GEMM_TAB_MAX = 16;
struct GemmColsumTab { float* db[GEMM_TAB_MAX]; };

struct GemmParams {
    const void* A; const void* B;
    int lda, ldb;                 // elements
    unsigned a_bytes, b_bytes;    // extent of each operand from its base pointer (range check of the DMA)
    int M, N, K;
    void* C; void* C2; int ldc;
    const float* bias; const float* bias1; const float* bias2; int bias_seg;   // bias_seg > 0: columns [k * seg, (k + 1) * seg) take bias k
    const float* resid; const u16* pre;
    const unsigned long long* seed; unsigned site, drop_thr; float drop_scale;   // drop_thr: 16-bit threshold
    int m_tiles, n_tiles, splits, ksteps_per_split;
    long long slab_stride;        // elements between the fp32 partial slabs of a split-K run
    // batches (grid.y): independent problems with their own A / B / C (element strides; 0 = shared operand).  batch_k != 0:
    // the batch index is the split index instead -- every slice contracts ONE batch entry's whole K and the slabs are summed
    // (sum over the images of a convolution's weight gradient)
    int batch_k, grid_batches;
    long long a_bstride, b_bstride, c_bstride;
    // Column sums of the A operand of a TN (weight-gradient) run = the bias gradient of the same Linear layer, computed by EXTRA
    // workgroups of the same launch (block ids >= gemm_blocks): they are dispatched last, into the slots the final, partly
    // filled round of tiles leaves idle, and stream the gradient once more while the tiles multiply.  cs_part [slices][M]
    // partial rows; the slab-reduction launch that follows adds them in slice order (as column_sum.hip does).
    float* cs_part; int gemm_blocks, cs_slices, cs_rows_per_slice;
    // Pointer-table batches (grid.y = ptr_batches > 0): independent problems of ONE shape whose operands are separate allocations --
    // the weight gradients of the encoder's twelve blocks, queued during the backward and multiplied by one launch per Linear
    // shape at its end (no split-K: twelve problems fill the chip by themselves).  cs_part then holds ptr_batches consecutive
    // [cs_slices][M] blocks of partial column sums.
    int ptr_batches;
    const void* tabA[GEMM_TAB_MAX]; const void* tabB[GEMM_TAB_MAX]; void* tabC[GEMM_TAB_MAX];
};

template <int BM_, int BN_, bool AKM_, bool BKM_, int BK_ = 64, int NS_ = 2>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, NS = NS_;   // NS LDS stages: the DMA runs NS - 1 K steps ahead
    static constexpr bool AKM = AKM_, BKM = BKM_;       // operand is K-major in memory
    static constexpr int WNC = BN % 64 == 0 ? 64 : 48;   // columns of a wave's sub-tile: 64, or 48 for the 96-wide tile
    static constexpr int NBK = WNC / 16;                 // 16-column MFMA blocks of a wave along N
    static constexpr int WM = BM / 64, WN = BN / WNC, WAVES = WM * WN, THREADS = WAVES * 64;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;
    static constexpr int PA = (A_PIECES + WAVES - 1) / WAVES, PB = (B_PIECES + WAVES - 1) / WAVES;  // per wave
    static constexpr int LDS = NS * STAGE;
    static constexpr int RROW = BK * 2;                 // bytes of a row-operand row in LDS (128 or 64)
    static constexpr int KSUB = BK / 32;                // MFMA k-steps per stage
    static_assert(BK == 64 || BK == 32, "K step");
    static_assert(!AKM || BM >= 128, "K-major tiles need >= 16 chunks per row");
    static_assert(!BKM || BN >= 128, "K-major tiles need >= 16 chunks per row (the 96-wide tile is for row operands)");
    static_assert(A_PIECES % WAVES == 0 && B_PIECES % WAVES == 0, "pieces must divide over the waves");
    static_assert(WN * WNC == BN && WM * 64 == BM, "wave layout");
    static_assert(LDS <= 160 * 1024, "stages exceed the LDS");
};

// swizzle of a row operand's 16-B chunks: 128-B rows (BK 64): c ^ (r & 7); 64-B rows (BK 32): c ^ ((-(r >> 2)) & 3) -- both
// put the 16 rows x 4 chunks of every ds_read_b128 lane group on 16 different 16-B slots of the 256-B bank row
template <int BK> __device__ __forceinline__ int row_f(int row) { return BK == 64 ? (row & 7) : ((-(row >> 2)) & 3); }

__device__ __forceinline__ int kmaj_f(int krow) { return ((krow & 3) << 2) | ((krow >> 2) & 3); }

template <typename C, int EPI>
__global__ __launch_bounds__(C::THREADS, 2) void gemm_bf16_kernel(GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass has no buffer-resource type: it only needs the launch stub)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (C::AKM && C::BKM && C::THREADS == 256) {
        if (p.cs_part && (int)blockIdx.x >= p.gemm_blocks) {
            // column-sum role (see GemmParams): workgroup (column group of 256, row slice) exactly as column_sum_partial_kernel --
            // lane l of every wave owns columns 4l..4l+3, the 4 waves take rows r, r+4, ..., combined through LDS in wave order
            float* red = reinterpret_cast<float*>(lds);   // [4][256]
            const int cs = (int)blockIdx.x - p.gemm_blocks;
            const int groups = (p.M + 255) / 256;
            const int cg = cs % groups, slice = cs / groups;
            const int c = cg * 256 + 4 * lane;
            const int r_lo = slice * p.cs_rows_per_slice, r_hi = min(p.K, r_lo + p.cs_rows_per_slice);
            const u16* x = (const u16*)(p.ptr_batches ? p.tabA[blockIdx.y] : p.A);
            float* cs_out = p.cs_part + (p.ptr_batches ? (long long)blockIdx.y * p.cs_slices * p.M : 0);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            auto load4 = [&](int r, float* v) {
                const uint2 q = *reinterpret_cast<const uint2*>(x + (long long)r * p.lda + c);
                v[0] = __builtin_bit_cast(float, q.x << 16); v[1] = __builtin_bit_cast(float, q.x & 0xFFFF0000u);
                v[2] = __builtin_bit_cast(float, q.y << 16); v[3] = __builtin_bit_cast(float, q.y & 0xFFFF0000u);
            };
            if (c < p.M) {
                int r = r_lo + wave;
                for (; r + 4 < r_hi; r += 8) {
                    float a[4], b[4];
                    load4(r, a); load4(r + 4, b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += a[e] + b[e];
                }
                for (; r < r_hi; r += 4) {
                    float a[4];
                    load4(r, a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += a[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave * 256 + 4 * lane + e] = acc[e];
            __syncthreads();
            const int cc = cg * 256 + tid;
            if (cc < p.M) cs_out[(long long)slice * p.M + cc] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
            return;
        }
    }
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = wave % C::WM, wn = wave / C::WM;

    // ---- which tile (and K slice) this workgroup computes
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int mt, nt, split = 0;
    if (p.splits == 1) {
        mt = (slot / p.n_tiles) * 8 + xcd;
        nt = slot % p.n_tiles;
        if (mt >= p.m_tiles) return;
    } else {
        int tile;
        if (p.splits >= 8) {
            const int per = p.splits >> 3;
            split = xcd + 8 * (slot % per);
            tile = slot / per;
        } else {
            const int sub = 8 / p.splits;
            split = xcd % p.splits;
            tile = slot * sub + xcd / p.splits;
        }
        if (tile >= p.m_tiles * p.n_tiles) return;
        mt = tile / p.n_tiles;
        nt = tile % p.n_tiles;
    }
    const int m0 = mt * C::BM, n0 = nt * C::BN;
    const int ks_total = (p.K + C::BK - 1) / C::BK;
    const int ks_begin = p.batch_k ? 0 : split * p.ksteps_per_split;
    const int T = min(p.ksteps_per_split, ks_total - ks_begin);   // K steps of this workgroup (>= 1 by construction)

    const long long bidx = p.batch_k ? split : (long long)blockIdx.y;
    const u16* Ab = p.ptr_batches ? (const u16*)p.tabA[blockIdx.y] : (const u16*)p.A + bidx * p.a_bstride;
    const u16* Bb = p.ptr_batches ? (const u16*)p.tabB[blockIdx.y] : (const u16*)p.B + bidx * p.b_bstride;
    void* const Cb = p.ptr_batches ? p.tabC[blockIdx.y] : p.C;
    const long long c_batch = (p.batch_k || p.ptr_batches) ? 0 : (long long)blockIdx.y * p.c_bstride;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(Ab), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(Bb), 0, p.b_bytes, 0x00020000);

    // ---- DMA source offsets of this lane (bytes from the operand base, K step 0), fixed over the loop
    int a_src[C::PA], b_src[C::PB];
#pragma unroll
    for (int i = 0; i < C::PA; ++i) {
        const int q = wave + i * C::WAVES;
        if constexpr (!C::AKM) {
            constexpr int CPR = C::RROW / 16, RPP = 1024 / C::RROW;   // chunks per row, rows per 1 KiB piece
            const int r = RPP * q + lane / CPR;
            const int ch = (lane % CPR) ^ row_f<C::BK>(r);
            a_src[i] = ((m0 + r) * p.lda + ch * 8) * 2;
        } else {
            constexpr int RB = 2 * C::BM;
            const int lin = 1024 * q + 16 * lane;
            const int krow = lin / RB, pos = (lin % RB) >> 4;
            a_src[i] = (krow * p.lda + m0 + ((pos ^ kmaj_f(krow)) << 3)) * 2;
        }
    }
#pragma unroll
    for (int i = 0; i < C::PB; ++i) {
        const int q = wave + i * C::WAVES;
        if constexpr (!C::BKM) {
            constexpr int CPR = C::RROW / 16, RPP = 1024 / C::RROW;
            const int r = RPP * q + lane / CPR;
            const int ch = (lane % CPR) ^ row_f<C::BK>(r);
            b_src[i] = ((n0 + r) * p.ldb + ch * 8) * 2;
        } else {
            constexpr int RB = 2 * C::BN;
            const int lin = 1024 * q + 16 * lane;
            const int krow = lin / RB, pos = (lin % RB) >> 4;
            b_src[i] = (krow * p.ldb + n0 + ((pos ^ kmaj_f(krow)) << 3)) * 2;
        }
    }
    const int a_step = C::AKM ? C::BK * p.lda * 2 : C::RROW;   // bytes per K step
    const int b_step = C::BKM ? C::BK * p.ldb * 2 : C::RROW;

    auto issue = [&](int t, int stage) {
        const int ks = ks_begin + t;
        unsigned char* dst = lds + stage * C::STAGE;
        const int sa = ks * a_step, sb = ks * b_step;
#pragma unroll
        for (int i = 0; i < C::PA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(dst + (wave + i * C::WAVES) * 1024), 16, a_src[i], sa, 0, 0);
#pragma unroll
        for (int i = 0; i < C::PB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(dst + C::A_BYTES + (wave + i * C::WAVES) * 1024), 16, b_src[i], sb, 0, 0);
    };

    // ---- fragment read offsets (bytes inside a stage)
    // row operand: block b (16 rows), k-step ks: (base + 16 b + i16) * 128 + (((4 ks + g) ^ (i16 & 7)) << 4)
    // K-major operand: block b (16 columns), k-step ks, half t: (32 ks + 8 g + 4 t + q) * RB + (((2 cb + (p >> 1)) ^ f) << 4) + 8 (p & 1)
    int a_off[C::AKM ? 8 : 2], b_off[C::BKM ? 8 : 2];
    if constexpr (!C::AKM) {
#pragma unroll
        for (int ks = 0; ks < C::KSUB; ++ks) a_off[ks] = (wm * 64 + i16) * C::RROW + (((4 * ks + g) ^ row_f<C::BK>(i16)) << 4);
    } else {
        constexpr int RB = 2 * C::BM;
        const int q = i16 >> 2, pp = i16 & 3;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int krow = 8 * g + 4 * t + q;   // (+ 32 ks: f unchanged, RB * 32 ks added as an immediate)
                const int ch = 2 * (wm * 4 + b) + (pp >> 1);
                a_off[b * 2 + t] = krow * RB + ((ch ^ kmaj_f(krow)) << 4) + 8 * (pp & 1);
            }
    }
    if constexpr (!C::BKM) {
#pragma unroll
        for (int ks = 0; ks < C::KSUB; ++ks) b_off[ks] = C::A_BYTES + (wn * C::WNC + i16) * C::RROW + (((4 * ks + g) ^ row_f<C::BK>(wn * C::WNC + i16)) << 4);
    } else {
        constexpr int RB = 2 * C::BN;
        const int q = i16 >> 2, pp = i16 & 3;
#pragma unroll
        for (int b = 0; b < C::NBK; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int krow = 8 * g + 4 * t + q;
                const int ch = 2 * (wn * C::NBK + b) + (pp >> 1);
                b_off[b * 2 + t] = C::A_BYTES + krow * RB + ((ch ^ kmaj_f(krow)) << 4) + 8 * (pp & 1);
            }
    }

    f32x4 acc[C::NBK][4];   // [n block][m block]
#pragma unroll
    for (int a = 0; a < C::NBK; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto frag = [&](const unsigned char* st, auto kmajor, const int* off, int rb_bytes, int b, int ks) -> bf16x8 {
        if constexpr (!decltype(kmajor)::value) {
            return *reinterpret_cast<const bf16x8*>(st + off[ks] + b * (16 * C::RROW));
        } else {
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(st + off[b * 2] + ks * 32 * rb_bytes));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(st + off[b * 2 + 1] + ks * 32 * rb_bytes));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };

    // ---- main loop: the DMA of K step t + NS - 1 is issued right after the barrier of step t (its stage was read in step
    // t - 1); the wait in front of the barrier leaves the NS - 2 younger stages in flight (counted vmcnt, raw s_barrier)
    constexpr int NS = C::NS, PW = C::PA + C::PB;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < T) issue(s, s);
    int stage = 0;
    // One K step.  All fragment reads of the step are requested up front (the second k-step's land behind the first one's
    // MFMAs instead of each group of MFMAs waiting for the read issued just before it) and the DMA instructions of the next
    // stage are spread between the MFMAs (sched_group_barrier: 1 DMA per 32 / PW MFMAs), so that their issue slots -- an
    // LDS-DMA holds the wave's issue for tens of cycles -- fall under matrix work instead of in front of it.
    auto step = [&](auto with_dma, int t) {
        const unsigned char* st = lds + stage * C::STAGE;
        bf16x8 a[C::KSUB][4], b[C::KSUB][C::NBK];
#pragma unroll
        for (int ks = 0; ks < C::KSUB; ++ks) {
#pragma unroll
            for (int x = 0; x < 4; ++x) a[ks][x] = frag(st, std::integral_constant<bool, C::AKM>(), a_off, 2 * C::BM, x, ks);
#pragma unroll
            for (int x = 0; x < C::NBK; ++x) b[ks][x] = frag(st, std::integral_constant<bool, C::BKM>(), b_off, 2 * C::BN, x, ks);
        }
        if constexpr (decltype(with_dma)::value) issue(t + NS - 1, stage == 0 ? NS - 1 : stage - 1);
#pragma unroll
        for (int ks = 0; ks < C::KSUB; ++ks)
#pragma unroll
            for (int tn = 0; tn < C::NBK; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][tn], a[ks][tm], acc[tn][tm], 0, 0, 0);
        constexpr int READS = C::KSUB * ((C::AKM ? 8 : 4) + (C::BKM ? 2 : 1) * C::NBK);
        constexpr int MFMAS = C::KSUB * 4 * C::NBK;
        __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);   // every ds_read of the step first
        if constexpr (decltype(with_dma)::value) {
            constexpr int PER = (MFMAS / 2) / PW > 0 ? (MFMAS / 2) / PW : 1;   // the DMAs go under the FIRST half of the MFMAs:
#pragma unroll                                                                  // the data then has the rest of the step to land
            for (int i = 0; i < PW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // one LDS-DMA (VMEM read)
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MFMAS - PER * PW, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, MFMAS, 0);
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    };
    auto arrive = [&](int t) {
        const int ahead = min(NS - 2, T - 1 - t);   // K steps after t whose DMA is already in flight
        if (NS >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
        else if (NS >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of step t have landed
        __builtin_amdgcn_s_barrier();                           // ... everybody's; and everybody is done reading step t - 1's stage
    };
    int t = 0;
    for (; t + NS - 1 < T; ++t) {   // steps that still have a stage to prefetch
        arrive(t);
        step(std::true_type(), t);
    }
    for (; t < T; ++t) {
        arrive(t);
        step(std::false_type(), t);
    }

    // ---- epilogue: lane = output row m (per m block), registers = 4 consecutive columns n.  Interior tiles (the usual
    // case) run without bounds checks; per column block all loads (residual / pre-activation) are issued before the
    // arithmetic and the stores.
    SisDropKey key{0u, 0u};
    constexpr bool HAS_BIAS = EPI == SIS_GEMM_EPI_BIAS || EPI == SIS_GEMM_EPI_BIAS_GELU_DROP || EPI == SIS_GEMM_EPI_BIAS_DROP_RESID;
    constexpr bool HAS_DROP = EPI == SIS_GEMM_EPI_BIAS_GELU_DROP || EPI == SIS_GEMM_EPI_BIAS_DROP_RESID;   // (GELU_BWD: the factor it reads carries the mask)
    if constexpr (HAS_DROP)
        if (p.drop_thr) key = sis_drop_key(p.seed, p.site);
    auto epilogue = [&](auto checked_t) {
        constexpr bool CHECKED = decltype(checked_t)::value;
        float4 bq[C::NBK];
#pragma unroll
        for (int tn = 0; tn < C::NBK; ++tn) {
            const int n = n0 + wn * C::WNC + 16 * tn + 4 * g;
            bq[tn] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (HAS_BIAS)
                if (!CHECKED || n < p.N) {
                    const float* bp = p.bias + n;
                    if (p.bias_seg) {   // query | key | value biases stay three parameters (a lane's 4 columns never straddle two)
                        if (n >= 2 * p.bias_seg) bp = p.bias2 + (n - 2 * p.bias_seg);
                        else if (n >= p.bias_seg) bp = p.bias1 + (n - p.bias_seg);
                    }
                    bq[tn] = *reinterpret_cast<const float4*>(bp);
                }
        }
#pragma unroll
        for (int tn = 0; tn < C::NBK; ++tn) {
            const int n = n0 + wn * C::WNC + 16 * tn + 4 * g;
            const bool n_ok = !CHECKED || n < p.N;
            float4 r[4];
            uint2 h[4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const int m = m0 + wm * 64 + 16 * tm + i16;
                const long long at = (long long)(CHECKED ? min(m, p.M - 1) : m) * p.ldc + (CHECKED ? min(n, p.N - 4) : n);
                if constexpr (EPI == SIS_GEMM_EPI_BIAS_DROP_RESID) r[tm] = *reinterpret_cast<const float4*>(p.resid + at);
                if constexpr (EPI == SIS_GEMM_EPI_GELU_BWD) h[tm] = *reinterpret_cast<const uint2*>(p.pre + at);
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                const int m = m0 + wm * 64 + 16 * tm + i16;
                const bool ok = n_ok && (!CHECKED || m < p.M);
                const long long at = c_batch + (long long)m * p.ldc + n;
                float keep[4] = {1.f, 1.f, 1.f, 1.f};
                if constexpr (HAS_DROP)
                    if (p.drop_thr) sis_drop_quad(key, ((unsigned)m * (unsigned)p.N + (unsigned)n) >> 2, p.drop_thr, p.drop_scale, keep);
                float v[4] = {acc[tn][tm][0] + bq[tn].x, acc[tn][tm][1] + bq[tn].y, acc[tn][tm][2] + bq[tn].z, acc[tn][tm][3] + bq[tn].w};
                if constexpr (EPI == SIS_GEMM_EPI_NONE || EPI == SIS_GEMM_EPI_BIAS) {
                    if (ok) *reinterpret_cast<uint2*>((u16*)Cb + at) = make_uint2(sis_pack_bf16x2(v[0], v[1]), sis_pack_bf16x2(v[2], v[3]));
                } else if constexpr (EPI == SIS_GEMM_EPI_BIAS_GELU_DROP) {
                    // dropout(gelu(pre)) and, for the backward, d/d pre of it: gelu'(pre) * the same dropout factor (bf16) -- both from
                    // ONE evaluation of the normal cdf / pdf here, so that the data-gradient GEMM's epilogue is a multiplication
                    // (it recomputed erf, exp and the dropout hash per element: 74 us per launch against 40 for the plain GEMM)
                    float y[4], dd[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float cdf, pdf;
                        sis_gelu_parts(v[e], cdf, pdf);
                        y[e] = v[e] * cdf * keep[e];
                        dd[e] = __builtin_fmaf(v[e], pdf, cdf) * keep[e];
                    }
                    const uint2 hp = make_uint2(sis_pack_bf16x2(dd[0], dd[1]), sis_pack_bf16x2(dd[2], dd[3]));
                    if (ok) {
                        *reinterpret_cast<uint2*>((u16*)p.C2 + at) = hp;
                        *reinterpret_cast<uint2*>((u16*)Cb + at) = make_uint2(sis_pack_bf16x2(y[0], y[1]), sis_pack_bf16x2(y[2], y[3]));
                    }
                } else if constexpr (EPI == SIS_GEMM_EPI_BIAS_DROP_RESID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= keep[e];
                    if (ok) *reinterpret_cast<float4*>((float*)Cb + at) = make_float4(r[tm].x + v[0], r[tm].y + v[1], r[tm].z + v[2], r[tm].w + v[3]);
                } else if constexpr (EPI == SIS_GEMM_EPI_GELU_BWD) {
                    // gradient w.r.t. the pre-activation: acc * (gelu'(pre) * dropout factor), the factor stored by the forward
                    const float d[4] = {sis_bf16_lo(h[tm].x), sis_bf16_hi(h[tm].x), sis_bf16_lo(h[tm].y), sis_bf16_hi(h[tm].y)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= d[e];
                    if (ok) *reinterpret_cast<uint2*>((u16*)Cb + at) = make_uint2(sis_pack_bf16x2(v[0], v[1]), sis_pack_bf16x2(v[2], v[3]));
                } else {  // SIS_GEMM_EPI_F32: fp32 result or partial slab of a split-K run
                    if (ok) *reinterpret_cast<float4*>((float*)Cb + (long long)split * p.slab_stride + at) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    };
    if (m0 + C::BM <= p.M && n0 + C::BN <= p.N) epilogue(std::false_type());
    else epilogue(std::true_type());
#endif
}

// partial slabs of a split-K run -> result, added in slab order (deterministic)
__global__ __launch_bounds__(256) void gemm_slab_reduce_kernel(float* __restrict__ out, const float* __restrict__ slabs, long long quads,
                                                               int splits, long long slab_stride, int reduce_blocks,
                                                               float* __restrict__ cs_out, const float* __restrict__ cs_part, int cs_n,
                                                               int cs_slices) {
    if ((int)blockIdx.x >= reduce_blocks) {
        // second stage of the bias column sums (blocks beyond the slab reduction): 64 columns per workgroup, wave q adds the
        // slices q, q + 4, ... in order, then (s0 + s1) + (s2 + s3) -- column_sum_finish_kernel's order
        __shared__ float red[4][64];
        const int c = ((int)blockIdx.x - reduce_blocks) * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
        float sum = 0.f;
        if (c < cs_n) {
#pragma unroll 8
            for (int k = q; k < cs_slices; k += 4) sum += cs_part[(long long)k * cs_n + c];
        }
        red[q][threadIdx.x & 63] = sum;
        __syncthreads();
        if (q == 0 && c < cs_n) cs_out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        return;
    }
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    float4 s = *reinterpret_cast<const float4*>(slabs + 4 * i);
    for (int k = 1; k < splits; ++k) {
        const float4 q = *reinterpret_cast<const float4*>(slabs + k * slab_stride + 4 * i);
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    *reinterpret_cast<float4*>(out + 4 * i) = s;
}

template <typename C, int EPI>
int launch_gemm(const GemmParams& p, hipStream_t st, const char* name) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<C, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return sis_fail("%s: cannot raise the LDS limit: %s", name, hipGetErrorString(e));
        attr_set = true;
    }
    int groups;
    if (p.splits == 1) groups = p.n_tiles * sis_cdiv(p.m_tiles, 8);
    else if (p.splits >= 8) groups = p.m_tiles * p.n_tiles * (p.splits / 8);
    else groups = sis_cdiv((int64_t)p.m_tiles * p.n_tiles, 8 / p.splits);
    SIS_OCC_REPORT((gemm_bf16_kernel<C, EPI>), C::THREADS, C::LDS);
    GemmParams q = p;
    q.gemm_blocks = 8 * groups;
    int extra = 0;
    if (q.cs_part) {
        if (!(C::AKM && C::BKM && C::THREADS == 256 && !p.batch_k && (p.grid_batches == 1 || p.ptr_batches)))
            return sis_fail("%s: the bias column sums ride with the 4-wave TN tiles only", name);
        extra = ((p.M + 255) / 256) * p.cs_slices;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<C, EPI>), dim3(8 * groups + extra, p.batch_k ? 1 : p.grid_batches), dim3(C::THREADS), C::LDS, st, q);
    SIS_CHECK_LAUNCH(name);
    sis_kernel_name = name;
    return 0;
}

// the epilogues each layout is built with: what the three GEMMs of a Linear layer need
template <int BM, int BN, int BK, int NS>
int dispatch(const GemmParams& p, int layout, int epi, hipStream_t st) {
    if (layout == LAYOUT_NT) {
        typedef GemmCfg<BM, BN, false, false, BK, NS> C;
        switch (epi) {
            case SIS_GEMM_EPI_NONE: return launch_gemm<C, SIS_GEMM_EPI_NONE>(p, st, "gemm_bf16_kernel<NT>");
            case SIS_GEMM_EPI_BIAS: return launch_gemm<C, SIS_GEMM_EPI_BIAS>(p, st, "gemm_bf16_kernel<NT,bias>");
            case SIS_GEMM_EPI_BIAS_GELU_DROP: return launch_gemm<C, SIS_GEMM_EPI_BIAS_GELU_DROP>(p, st, "gemm_bf16_kernel<NT,bias+gelu+dropout>");
            case SIS_GEMM_EPI_BIAS_DROP_RESID: return launch_gemm<C, SIS_GEMM_EPI_BIAS_DROP_RESID>(p, st, "gemm_bf16_kernel<NT,bias+dropout+residual>");
            case SIS_GEMM_EPI_F32: return launch_gemm<C, SIS_GEMM_EPI_F32>(p, st, "gemm_bf16_kernel<NT,f32>");
        }
    } else if (layout == LAYOUT_NN) {
        typedef GemmCfg<BM, BN, false, true, BK, NS> C;
        switch (epi) {
            case SIS_GEMM_EPI_NONE: return launch_gemm<C, SIS_GEMM_EPI_NONE>(p, st, "gemm_bf16_kernel<NN>");
            case SIS_GEMM_EPI_GELU_BWD: return launch_gemm<C, SIS_GEMM_EPI_GELU_BWD>(p, st, "gemm_bf16_kernel<NN,gelu'+dropout>");
        }
    } else {
        typedef GemmCfg<BM, BN, true, true, BK, NS> C;
        if (epi == SIS_GEMM_EPI_F32) return launch_gemm<C, SIS_GEMM_EPI_F32>(p, st, "gemm_bf16_kernel<TN,f32>");
        if (epi == SIS_GEMM_EPI_NONE) return launch_gemm<C, SIS_GEMM_EPI_NONE>(p, st, "gemm_bf16_kernel<TN>");
    }
    return sis_fail("sis_gemm_bf16: epilogue %d is not built for layout %d", epi, layout);
}

template <int BM, int BN, int BK, int NS>
int dispatch_nt_only(const GemmParams& p, int epi, hipStream_t st) {
    typedef GemmCfg<BM, BN, false, false, BK, NS> C;
    switch (epi) {
        case SIS_GEMM_EPI_NONE: return launch_gemm<C, SIS_GEMM_EPI_NONE>(p, st, "gemm_bf16_kernel<NT,128x96>");
        case SIS_GEMM_EPI_BIAS: return launch_gemm<C, SIS_GEMM_EPI_BIAS>(p, st, "gemm_bf16_kernel<NT,128x96,bias>");
        case SIS_GEMM_EPI_BIAS_GELU_DROP: return launch_gemm<C, SIS_GEMM_EPI_BIAS_GELU_DROP>(p, st, "gemm_bf16_kernel<NT,128x96,bias+gelu+dropout>");
        case SIS_GEMM_EPI_BIAS_DROP_RESID: return launch_gemm<C, SIS_GEMM_EPI_BIAS_DROP_RESID>(p, st, "gemm_bf16_kernel<NT,128x96,bias+dropout+residual>");
        case SIS_GEMM_EPI_F32: return launch_gemm<C, SIS_GEMM_EPI_F32>(p, st, "gemm_bf16_kernel<NT,128x96,f32>");
    }
    return sis_fail("sis_gemm_bf16: epilogue %d is not built for the 128 x 96 tile", epi);
}

struct TilePlan { int bm, bn, bk, ns; };
// tile codes of the C ABI: (BM, BN, BK, LDS stages)
constexpr TilePlan TILE_PLANS[] = {{128, 128, 64, 2}, {256, 128, 64, 2}, {128, 256, 64, 2}, {256, 256, 64, 2},
                                   {128, 128, 32, 3}, {128, 128, 32, 4}, {128, 128, 64, 3}, {256, 128, 64, 3}, {128, 96, 64, 2}};
constexpr int N_TILE_PLANS = sizeof(TILE_PLANS) / sizeof(TILE_PLANS[0]);

}  // namespace

extern "C" int64_t sis_gemm_bf16_workspace_bytes(int m, int n, int splits) {
    return splits > 1 ? (int64_t)splits * m * n * 4 : 0;
}

static int gemm_impl(void* c, void* c2, const void* a, const void* b, int layout, int epilogue, int m, int n, int k,
                     int lda, int ldb, int ldc, const float* bias, const float* bias1, const float* bias2, int bias_seg,
                     const float* resid, const void* pre,
                     const void* seed, int site, float drop_p, int splits, void* workspace, int64_t workspace_bytes,
                     int tile, int batches, int batch_k, int64_t a_bstride, int64_t b_bstride, int64_t c_bstride, void* stream,
                     float* colsum_out = nullptr) {
    if (m <= 0 || n <= 0) return 0;
    SIS_REQUIRE(c && a && b, "sis_gemm_bf16: null pointer");
    SIS_REQUIRE(layout >= 0 && layout <= 2, "sis_gemm_bf16: layout %d (0 NT, 1 NN, 2 TN)", layout);
    SIS_REQUIRE(k > 0 && n % 4 == 0 && ldc % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0,
                "sis_gemm_bf16: n, ldc must be multiples of 4 and lda, ldb of 8 (n=%d ldc=%d lda=%d ldb=%d)", n, ldc, lda, ldb);
    SIS_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0, "sis_gemm_bf16: operands must be 16-byte aligned");
    if (layout != LAYOUT_TN) SIS_REQUIRE(k % 64 == 0, "sis_gemm_bf16: k = %d must be a multiple of 64 for the NT / NN layouts", k);
    const bool has_bias = epilogue == SIS_GEMM_EPI_BIAS || epilogue == SIS_GEMM_EPI_BIAS_GELU_DROP || epilogue == SIS_GEMM_EPI_BIAS_DROP_RESID;
    SIS_REQUIRE(!has_bias || bias, "sis_gemm_bf16: the epilogue needs a bias");
    SIS_REQUIRE(epilogue != SIS_GEMM_EPI_BIAS_DROP_RESID || resid, "sis_gemm_bf16: the epilogue needs the residual");
    SIS_REQUIRE(epilogue != SIS_GEMM_EPI_BIAS_GELU_DROP || c2, "sis_gemm_bf16: the epilogue needs the pre-activation output");
    SIS_REQUIRE(epilogue != SIS_GEMM_EPI_GELU_BWD || pre, "sis_gemm_bf16: the epilogue needs the pre-activation");
    SIS_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "sis_gemm_bf16: dropout probability %f", drop_p);
    SIS_REQUIRE(drop_p == 0.f || seed, "sis_gemm_bf16: dropout needs the seed word");
    SIS_REQUIRE((int64_t)m * n < (1LL << 32), "sis_gemm_bf16: more than 2^32 outputs");
    if (splits < 1) splits = 1;
    SIS_REQUIRE(splits == 1 || epilogue == SIS_GEMM_EPI_F32, "sis_gemm_bf16: split-K needs the fp32 epilogue");
    SIS_REQUIRE(splits == 1 || splits == 2 || splits == 4 || splits % 8 == 0, "sis_gemm_bf16: splits must be 1, 2, 4 or a multiple of 8");

    GemmParams p;
    p.ptr_batches = 0;
    p.A = a; p.B = b; p.lda = lda; p.ldb = ldb; p.M = m; p.N = n; p.K = k;
    // operand extents: row operand [rows][k] -> (rows - 1) * ld + k elements; K-major [k][cols] -> (k - 1) * ld + cols
    const int64_t ae = layout == LAYOUT_TN ? (int64_t)(k - 1) * lda + m : (int64_t)(m - 1) * lda + k;
    const int64_t be = layout == LAYOUT_NT ? (int64_t)(n - 1) * ldb + k : (int64_t)(k - 1) * ldb + n;
    SIS_REQUIRE(ae * 2 < (1LL << 31) && be * 2 < (1LL << 31), "sis_gemm_bf16: operands above 2 GiB");
    p.a_bytes = (unsigned)(ae * 2); p.b_bytes = (unsigned)(be * 2);
    SIS_REQUIRE(bias_seg == 0 || (bias_seg % 4 == 0 && bias1 && bias2 && n == 3 * bias_seg), "sis_gemm_bf16: three bias segments of n / 3 columns each");
    p.C = c; p.C2 = c2; p.ldc = ldc; p.bias = bias; p.bias1 = bias1; p.bias2 = bias2; p.bias_seg = bias_seg; p.resid = resid; p.pre = (const u16*)pre;
    p.seed = (const unsigned long long*)seed; p.site = (unsigned)site;
    p.drop_thr = sis_drop_thr16(drop_p);
    p.drop_scale = sis_drop_scale(p.drop_thr);
    SIS_REQUIRE(p.drop_thr == 0 || ((int64_t)n % 4 == 0), "sis_gemm_bf16: dropout quads need n % 4 == 0");
    if (tile >= SIS_GEMM_TILE_256X96 && tile <= SIS_GEMM_TILE_256X288) {
        // 256-row tiles, one 8-wave workgroup per CU (gemm256_bf16.hip): NT layout, no split-K, K a multiple of 64 (>= 128)
        SIS_REQUIRE(sis_gemm256_ok(layout, epilogue, k, splits) && batches == 1 && !batch_k && !colsum_out,
                    "sis_gemm_bf16: tile %d (256 rows) serves the NT layout without split-K, k %% 64 == 0, k >= 128 (layout %d, k %d, splits %d)",
                    tile, layout, k, splits);
        G256Params g;
        g.A = a; g.B = b; g.lda = lda; g.ldb = ldb; g.a_bytes = p.a_bytes; g.b_bytes = p.b_bytes; g.M = m; g.N = n; g.K = k;
        g.C = c; g.C2 = c2; g.ldc = ldc; g.bias = bias; g.bias1 = bias1; g.bias2 = bias2; g.bias_seg = bias_seg; g.resid = resid;
        g.pre = (const u16*)pre; g.seed = p.seed; g.site = p.site; g.drop_thr = p.drop_thr; g.drop_scale = p.drop_scale;
        g.m_tiles = g.n_tiles = 0;
        return sis_gemm256_dispatch(g, tile - SIS_GEMM_TILE_256X96 + 1, epilogue, (hipStream_t)stream);
    }
    SIS_REQUIRE(tile >= 0 && tile < N_TILE_PLANS, "sis_gemm_bf16: tile code %d (0..%d, or %d..%d for the 256-row tiles)", tile,
                N_TILE_PLANS - 1, SIS_GEMM_TILE_256X96, SIS_GEMM_TILE_256X288);
    const int bm = TILE_PLANS[tile].bm, bn = TILE_PLANS[tile].bn, bk = TILE_PLANS[tile].bk;
    p.m_tiles = sis_cdiv(m, bm); p.n_tiles = sis_cdiv(n, bn);
    const int ksteps = sis_cdiv(k, bk);
    if (splits > ksteps && !batch_k) splits = 1;
    // every slice must own at least one K step: with ceil(ksteps / splits) steps per slice the last slices can come out empty
    // (8 slices of 13 steps: 7 x 2 covers them), so the count steps down through the values the workgroup order supports
    // (multiples of 8, then 4, 2, 1) until none is -- any token count is a valid contraction length
    while (!batch_k && splits > 1 && (int64_t)(splits - 1) * sis_cdiv(ksteps, splits) >= ksteps)
        splits = splits > 8 ? splits - 8 : splits / 2;
    p.ksteps_per_split = sis_cdiv(ksteps, splits);
    p.splits = splits; p.slab_stride = (long long)m * ldc;
    p.batch_k = batch_k; p.grid_batches = batches; p.a_bstride = a_bstride; p.b_bstride = b_bstride; p.c_bstride = c_bstride;
    if (batch_k) {   // one slice per batch entry, each over the whole K
        SIS_REQUIRE(epilogue == SIS_GEMM_EPI_F32 && (batches == 1 || batches == 2 || batches == 4 || batches % 8 == 0),
                    "sis_gemm_bf16_batched: summing over %d batch entries needs the fp32 epilogue and 1, 2, 4 or a multiple of 8 entries", batches);
        p.splits = splits = batches;
        p.ksteps_per_split = ksteps;
    }
    hipStream_t st = (hipStream_t)stream;
    float* result = (float*)c;
    p.cs_part = nullptr; p.gemm_blocks = 0; p.cs_slices = 0; p.cs_rows_per_slice = 0;
    if (splits > 1) {
        SIS_REQUIRE(ldc == n, "sis_gemm_bf16: split-K writes a dense result (ldc == n)");
        SIS_REQUIRE(workspace && workspace_bytes >= sis_gemm_bf16_workspace_bytes(m, n, splits), "sis_gemm_bf16: workspace too small");
        p.C = workspace;
    }
    if (colsum_out) {   // bias gradient = column sums of the TN run's A operand, in the same two launches (see GemmParams)
        SIS_REQUIRE(layout == LAYOUT_TN && epilogue == SIS_GEMM_EPI_F32 && splits > 1 && m % 4 == 0 && lda % 4 == 0,
                    "sis_gemm_bf16_wgrad_bias: needs the TN layout with split-K and m, lda multiples of 4");
        const int groups = sis_cdiv(m, 256);
        int slices = sis_cdiv(1024, groups);
        if (slices > 64) slices = 64;
        if (slices > sis_cdiv(k, 8)) slices = sis_cdiv(k, 8);
        if (slices < 1) slices = 1;
        p.cs_rows_per_slice = sis_cdiv(k, slices);
        p.cs_slices = sis_cdiv(k, p.cs_rows_per_slice);
        const int64_t slab_bytes = sis_gemm_bf16_workspace_bytes(m, n, splits);
        SIS_REQUIRE(workspace_bytes >= slab_bytes + (int64_t)p.cs_slices * m * 4, "sis_gemm_bf16_wgrad_bias: workspace too small");
        p.cs_part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + slab_bytes);
    }
    int rc;
switch (tile) {
        case 0: rc = dispatch<128, 128, 64, 2>(p, layout, epilogue, st); break;
        case 1: rc = dispatch<256, 128, 64, 2>(p, layout, epilogue, st); break;
        case 2: rc = dispatch<128, 256, 64, 2>(p, layout, epilogue, st); break;
        case 3: rc = dispatch<256, 256, 64, 2>(p, layout, epilogue, st); break;
        case 4: rc = dispatch<128, 128, 32, 3>(p, layout, epilogue, st); break;
        case 5: rc = dispatch<128, 128, 32, 4>(p, layout, epilogue, st); break;
        case 6: rc = dispatch<128, 128, 64, 3>(p, layout, epilogue, st); break;
        case 7: rc = dispatch<256, 128, 64, 3>(p, layout, epilogue, st); break;
        default:   // 128 x 96: divides N = 768 and 2304 into tile counts that fill 2 workgroups per CU evenly (row operands only)
            SIS_REQUIRE(layout == LAYOUT_NT, "sis_gemm_bf16: the 128 x 96 tile is built for the NT layout");
            rc = dispatch_nt_only<128, 96, 64, 2>(p, epilogue, st);
            break;
    }
    if (rc) return rc;
    if (splits > 1) {
        const long long quads = (long long)m * n / 4;
        const int reduce_blocks = sis_cdiv(quads, 256), cs_blocks = p.cs_part ? sis_cdiv(m, 64) : 0;
        hipLaunchKernelGGL(gemm_slab_reduce_kernel, dim3(reduce_blocks + cs_blocks), dim3(256), 0, st, result, (const float*)workspace,
                           quads, splits, p.slab_stride, reduce_blocks, colsum_out, (const float*)p.cs_part, m, p.cs_slices);
        SIS_CHECK_LAUNCH("gemm_slab_reduce_kernel");
    }
    return 0;
}

// second stage of the bias column sums of a pointer-table run: db[job][c] = sum over the row slices, in slice order
static __global__ __launch_bounds__(256) void gemm_colsum_finish_multi_kernel(GemmColsumTab tab, const float* __restrict__ part, int m, int slices) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= m) return;
    const float* p = part + (long long)blockIdx.y * slices * m + c;
    float s = 0.f;
    for (int k = 0; k < slices; ++k) s += p[(long long)k * m];
    tab.db[blockIdx.y][c] = s;
}

/* Weight AND bias gradients of n_jobs Linear layers of ONE shape, one launch for the products and one for the column sums:
 * dw[j] [m][n] float32 = grad[j]^T x[j], db[j] [m] = column sums of grad[j]; grad[j] bf16 [k][lda >= m], x[j] bf16 [k][ldb >= n].
 * `dw`, `db`, `grad`, `x`: HOST arrays of n_jobs device pointers.  Every problem contracts its whole K in one workgroup per
 * tile (no split-K, no slabs): n_jobs x (m / 128) x (n / 128) tiles fill the chip without it.  workspace:
 * sis_gemm_bf16_wgrad_multi_workspace_bytes(n_jobs, m, k) bytes (partial column sums). */
extern "C" int64_t sis_gemm_bf16_wgrad_multi_workspace_bytes(int n_jobs, int m, int k) {
    const int groups = sis_cdiv(m, 256);
    int slices = sis_cdiv(1024, groups * (n_jobs > 0 ? n_jobs : 1));
    if (slices > 64) slices = 64;
    if (slices > sis_cdiv(k, 8)) slices = sis_cdiv(k, 8);
    if (slices < 1) slices = 1;
    return (int64_t)n_jobs * slices * m * 4;
}

extern "C" int sis_gemm_bf16_wgrad_bias_multi(void* const* dw, float* const* db, const void* const* grad, const void* const* x,
                                              int n_jobs, int m, int n, int k, int lda, int ldb, void* workspace,
                                              int64_t workspace_bytes, int tile, void* stream) {
    if (n_jobs <= 0 || m <= 0 || n <= 0) return 0;
    SIS_REQUIRE(dw && db && grad && x && workspace, "sis_gemm_bf16_wgrad_bias_multi: null pointer");
    SIS_REQUIRE(k > 0 && m % 4 == 0 && n % 4 == 0 && lda % 8 == 0 && ldb % 8 == 0, "sis_gemm_bf16_wgrad_bias_multi: m, n multiples of 4, lda, ldb of 8");
    SIS_REQUIRE(tile == 0 || tile == 4 || tile == 5 || tile == 6, "sis_gemm_bf16_wgrad_bias_multi: a 128 x 128 four-wave tile (0, 4, 5, 6)");
    SIS_REQUIRE((int64_t)m * n < (1LL << 32), "sis_gemm_bf16_wgrad_bias_multi: more than 2^32 outputs");
    const int64_t ae = (int64_t)(k - 1) * lda + m, be = (int64_t)(k - 1) * ldb + n;
    SIS_REQUIRE(ae * 2 < (1LL << 31) && be * 2 < (1LL << 31), "sis_gemm_bf16_wgrad_bias_multi: operands above 2 GiB");
    hipStream_t st = (hipStream_t)stream;
    for (int j0 = 0; j0 < n_jobs; j0 += GEMM_TAB_MAX) {
        const int nj = std::min(GEMM_TAB_MAX, n_jobs - j0);
        GemmParams p = {};
        GemmColsumTab ct = {};
        for (int j = 0; j < nj; ++j) {
            SIS_REQUIRE(dw[j0 + j] && db[j0 + j] && grad[j0 + j] && x[j0 + j], "sis_gemm_bf16_wgrad_bias_multi: null pointer in job %d", j0 + j);
            SIS_REQUIRE((((uintptr_t)dw[j0 + j] | (uintptr_t)grad[j0 + j] | (uintptr_t)x[j0 + j]) & 15) == 0,
                        "sis_gemm_bf16_wgrad_bias_multi: operands must be 16-byte aligned");
            p.tabA[j] = grad[j0 + j]; p.tabB[j] = x[j0 + j]; p.tabC[j] = dw[j0 + j]; ct.db[j] = db[j0 + j];
        }
        p.ptr_batches = nj; p.grid_batches = nj;
        p.A = p.tabA[0]; p.B = p.tabB[0]; p.C = p.tabC[0];
        p.lda = lda; p.ldb = ldb; p.ldc = n; p.M = m; p.N = n; p.K = k;
        p.a_bytes = (unsigned)(ae * 2); p.b_bytes = (unsigned)(be * 2);
        const int bk = TILE_PLANS[tile].bk;
        p.m_tiles = sis_cdiv(m, TILE_PLANS[tile].bm); p.n_tiles = sis_cdiv(n, TILE_PLANS[tile].bn);
        p.splits = 1; p.ksteps_per_split = sis_cdiv(k, bk); p.slab_stride = (long long)m * n;
        // bias column sums by extra workgroups of the same launch (as sis_gemm_bf16_wgrad_bias): ~1024 of them over all jobs
        const int groups = sis_cdiv(m, 256);
        int slices = sis_cdiv(1024, groups * nj);
        if (slices > 64) slices = 64;
        if (slices > sis_cdiv(k, 8)) slices = sis_cdiv(k, 8);
        if (slices < 1) slices = 1;
        p.cs_rows_per_slice = sis_cdiv(k, slices);
        p.cs_slices = sis_cdiv(k, p.cs_rows_per_slice);
        SIS_REQUIRE(workspace_bytes >= (int64_t)nj * p.cs_slices * m * 4, "sis_gemm_bf16_wgrad_bias_multi: workspace too small");
        p.cs_part = (float*)workspace;
        int rc;
        switch (tile) {
            case 0: rc = dispatch<128, 128, 64, 2>(p, LAYOUT_TN, SIS_GEMM_EPI_F32, st); break;
            case 4: rc = dispatch<128, 128, 32, 3>(p, LAYOUT_TN, SIS_GEMM_EPI_F32, st); break;
            case 5: rc = dispatch<128, 128, 32, 4>(p, LAYOUT_TN, SIS_GEMM_EPI_F32, st); break;
            default: rc = dispatch<128, 128, 64, 3>(p, LAYOUT_TN, SIS_GEMM_EPI_F32, st); break;
        }
        if (rc) return rc;
        hipLaunchKernelGGL(gemm_colsum_finish_multi_kernel, dim3(sis_cdiv(m, 256), nj), dim3(256), 0, st, ct, (const float*)workspace, m, p.cs_slices);
        SIS_CHECK_LAUNCH("gemm_colsum_finish_multi_kernel");
    }
    return 0;
}

extern "C" int sis_gemm_bf16(void* c, void* c2, const void* a, const void* b, int layout, int epilogue, int m, int n, int k,
                             int lda, int ldb, int ldc, const float* bias, const float* bias1, const float* bias2, int bias_seg,
                             const float* resid, const void* pre,
                             const void* seed, int site, float drop_p, int splits, void* workspace, int64_t workspace_bytes,
                             int tile, void* stream) {
    return gemm_impl(c, c2, a, b, layout, epilogue, m, n, k, lda, ldb, ldc, bias, bias1, bias2, bias_seg, resid, pre, seed, site, drop_p,
                     splits, workspace, workspace_bytes, tile, 1, 0, 0, 0, 0, stream);
}

extern "C" int sis_gemm_bf16_batched(void* c, const void* a, const void* b, int layout, int epilogue, int m, int n, int k, int lda,
                                     int ldb, int ldc, int batches, int64_t a_batch_stride, int64_t b_batch_stride,
                                     int64_t c_batch_stride, int sum_over_batches, void* workspace, int64_t workspace_bytes, int tile,
                                     void* stream) {
    if (batches <= 0) return 0;
    SIS_REQUIRE(epilogue == SIS_GEMM_EPI_NONE || epilogue == SIS_GEMM_EPI_F32, "sis_gemm_bf16_batched: epilogue %d (plain bf16 or fp32 results only)", epilogue);
    SIS_REQUIRE(batches < 65536, "sis_gemm_bf16_batched: %d batch entries", batches);
    SIS_REQUIRE(a_batch_stride % 8 == 0 && b_batch_stride % 8 == 0 && c_batch_stride % 4 == 0, "sis_gemm_bf16_batched: batch strides must keep 16-byte alignment");
    return gemm_impl(c, nullptr, a, b, layout, epilogue, m, n, k, lda, ldb, ldc, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0.f,
                     1, workspace, workspace_bytes, tile, batches, sum_over_batches ? 1 : 0, a_batch_stride, b_batch_stride, c_batch_stride, stream);
}

/* Weight gradient AND bias gradient of a Linear layer in the two launches of the split-K weight gradient: dw [m][n] float32 =
 * grad^T x (TN layout: grad [k][m], x [k][n], as sis_gemm_bf16 with SIS_GEMM_EPI_F32, splits > 1) and db [m] float32 = column
 * sums of grad, computed by extra workgroups of the GEMM launch (they land in the slots its last, partly filled round of tiles
 * leaves idle) and finished by extra workgroups of the slab reduction; same summation order as sis_column_sum.
 * workspace: sis_gemm_bf16_workspace_bytes(m, n, splits) + 64 * m * 4 bytes.  tile: 0 or 4..6 (the 4-wave tiles). */
extern "C" int sis_gemm_bf16_wgrad_bias(void* dw, float* db, const void* grad, const void* x, int m, int n, int k, int lda, int ldb,
                                        int splits, void* workspace, int64_t workspace_bytes, int tile, void* stream) {
    SIS_REQUIRE(db, "sis_gemm_bf16_wgrad_bias: null pointer");
    return gemm_impl(dw, nullptr, grad, x, LAYOUT_TN, SIS_GEMM_EPI_F32, m, n, k, lda, ldb, n, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                     nullptr, 0, 0.f, splits, workspace, workspace_bytes, tile, 1, 0, 0, 0, 0, stream, db);
}
