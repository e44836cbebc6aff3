// bf16 GEMM of the TransUNet ViT encoder, 256-row tiles: one 8-wave workgroup per compute unit, register-double-buffered
// fragments, four LDS stages filled by LDS-DMA, ONE barrier per 32-deep K step with nothing waited for in front of the MFMAs.
// Serves the NT layout (C = A B^T, both operands K-contiguous): the four forward GEMMs of a transformer block and -- with the
// transposed bf16 weight shadows networks/trans_u_net/vit_encoder.py keeps -- its four data-gradient GEMMs
// (reference call sites: networks/trans_u_net/vit_seg_modeling.py:60-67,76-96,104-122,181-189).
//
// Why a second kernel next to gemm_bf16.hip (128 x 128 tiles, two workgroups per CU): a 128 x 128 x 64 step needs 32 KB of
// operands per 512 MFMA cycles = 64 B/clk/CU, the whole L1 / TA path of a CU (25 % MFMA-busy on the QKV
// shape), and 8 192 x {768, 2304, 3072} outputs leave 128-wide tile counts that quantise badly on 256 CUs.  Here the tile is
// 256 x 96 NP (NP = 1, 2, 3): 8 192 x 768 -> 256 tiles of 256 x 96, 8 192 x 2304 -> 256 tiles of 256 x 288 (ONE full round
// of the chip), 8 192 x 3072 -> 512 tiles of 256 x 192 (two full rounds); operand traffic per MFMA cycle drops to
// 58 / 37 / 30 B/clk.
//
// Workgroup = 8 waves = 4 (rows) x 2 (columns); a wave owns 64 rows x 48 NP columns = 4 x 3 NP blocks of
// v_mfma_f32_16x16x32_bf16 (48 NP accumulator registers, <= 144).  K step = 32: a stage is A [256][32] + B [96 NP][32] bf16 in
// 64-byte rows whose four 16-byte chunks are XOR-swizzled by row (chunk c of row r at c ^ ((-(r >> 2)) & 3): the layout
// gemm_bf16.hip measured conflict-free for ds_read_b128), written linearly by buffer_load ... lds with the swizzle applied
// to the per-lane SOURCE address; rows beyond M / N read as zeros (buffer range check).
//
// Schedule of K step t (two waves per SIMD, both in the same phase):
//     top        the A fragments and the first panel's B fragments of step t are ALREADY in registers (read during step
//                t - 1): the MFMAs start at once after the barrier;
//     under the  the other panels' B fragments of step t (needed 12 MFMAs later), the early fragments of step t + 1 (second
//     MFMAs      register set), and the LDS-DMA of step t + 3 into the stage step t - 1 occupied;
//     bottom     s_waitcnt vmcnt(pieces of one step): step t + 2 has landed, step t + 3 stays in flight; s_barrier.
// Hazards: a stage is overwritten two barriers after its last read was waited for; a stage is read one barrier after the
// wait that retired its DMA (MI355X_MICROARCH.md: nothing else orders a ds_read behind an LDS-DMA).
//
// MFMA orientation as gemm_bf16.hip: D = Bfrag x Afrag, a lane holds output row m (lane & 15) and 4 consecutive columns
// (4 (lane >> 4) ..) per block, so the epilogues load / store 8 or 16 contiguous bytes per lane and block.
#include <type_traits>
#include "gemm256.h"

namespace {

typedef unsigned short u16;
typedef sis_bf16x8 bf16x8;
typedef sis_f32x4 f32x4;
typedef __attribute__((address_space(3))) void lds_void;

template <int NP_>
struct G256Cfg {
    static constexpr int NP = NP_, NB = 3 * NP_, NS = 4;
    static constexpr int BM = 256, BN = 96 * NP_, BK = 32, WAVES = 8, THREADS = 512;
    static constexpr int WNC = 48 * NP_;                         // columns of a wave
    static constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;     // 16, 6 NP
    static constexpr int PA = A_PIECES / WAVES;                  // 2
    static constexpr int PB_MAX = (B_PIECES + WAVES - 1) / WAVES;  // 1, 2, 3
    static constexpr int PB_FULL_WAVES = B_PIECES - (PB_MAX - 1) * WAVES;   // waves whose PB_MAX-th B piece exists: 6, 4, 2
    // Every wave issues PA + PB_MAX DMA instructions per step so that ONE vmcnt count holds for all of them and no branch
    // surrounds a load: a wave without a last B piece aims that instruction out of range (the hardware writes zeros) at a
    // 1 KiB dummy slot behind the stages.
    static constexpr int DUMMY = NS * STAGE;
    static constexpr int LDS = NS * STAGE + 1024;
    static_assert(LDS <= 160 * 1024, "stages exceed the LDS");
};

__device__ __forceinline__ int row_f32(int row) { return (-(row >> 2)) & 3; }

template <typename C, int EPI>
__global__ __launch_bounds__(C::THREADS) void gemm256_kernel(G256Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NB = C::NB, NS = C::NS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = wave & 3, wn = wave >> 2;

    // ---- tile of this workgroup: id % 8 = XCD group; the column tiles that share an A row tile take consecutive slots of one
    // group, so each A tile is fetched into one XCD's L2 once (as gemm_bf16.hip)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mt = (slot / p.n_tiles) * 8 + xcd, nt = slot % p.n_tiles;
    if (mt >= p.m_tiles) return;
    const int m0 = mt * C::BM, n0 = nt * C::BN;
    const int T = p.K / C::BK;   // K steps (host: K % 64 == 0, K >= 128: T even, >= 4)
#ifdef G256_TRACE
#define G256_STAMP(i) do { if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define G256_STAMP(i) do {} while (0)
#endif
    G256_STAMP(0);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, p.b_bytes, 0x00020000);

    // ---- LDS-DMA source offsets of this lane (bytes from the operand base at K step 0); a 1 KiB piece = 16 rows x 64 B
    int a_src[C::PA], b_src[C::PB_MAX];
#pragma unroll
    for (int i = 0; i < C::PA; ++i) {
        const int r = 16 * (wave + i * C::WAVES) + (lane >> 2);
        a_src[i] = ((m0 + r) * p.lda + (((lane & 3) ^ row_f32(r)) << 3)) * 2;
    }
#pragma unroll
    for (int i = 0; i < C::PB_MAX; ++i) {
        const int r = 16 * (wave + i * C::WAVES) + (lane >> 2);
        b_src[i] = ((n0 + r) * p.ldb + (((lane & 3) ^ row_f32(r)) << 3)) * 2;
    }
    const bool b_last = wave < C::PB_FULL_WAVES;   // this wave's PB_MAX-th B piece exists (wave-uniform)
    if (!b_last) b_src[C::PB_MAX - 1] = 0x7FFF0000;   // out of range for every step: zeros, into the dummy slot
    const int last_dst = b_last ? C::A_BYTES + (wave + (C::PB_MAX - 1) * C::WAVES) * 1024 : C::DUMMY;   // (bytes from lds[0] for the dummy)
    auto issue = [&](int t) {
        unsigned char* dst = lds + (t % NS) * C::STAGE;
        const int koff = t * (C::BK * 2);
#pragma unroll
        for (int i = 0; i < C::PA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(dst + (wave + i * C::WAVES) * 1024), 16, a_src[i], koff, 0, 0);
#pragma unroll
        for (int i = 0; i < C::PB_MAX - 1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(dst + C::A_BYTES + (wave + i * C::WAVES) * 1024), 16, b_src[i], koff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)((b_last ? dst : lds) + last_dst), 16, b_src[C::PB_MAX - 1], koff, 0, 0);
    };
    // End of a K step: at most ONE step's DMA pieces of this wave stay in flight (the older step has landed), every LDS read
    // of this wave has returned (the stage read in this step may be overwritten after the barrier), then the barrier.  The two
    // sched_barriers pin the step's MFMAs in front of it: hipcc otherwise sinks register-only MFMAs -- and the lgkmcnt waits of
    // the fragments they consume -- below the barrier, which would let a wave arrive with reads of the old stage outstanding.
    auto end_of_step = [&](auto last_dma_t) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(last_dma_t)::value) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C::PA + C::PB_MAX) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- fragment addresses (bytes inside a stage): block b of an operand = 16 rows = 1 KiB further on
    const int swz = (g ^ row_f32(i16)) << 4;
    const int a_base = (wm * 64 + i16) * 64 + swz;
    const int b_base = C::A_BYTES + (wn * C::WNC + i16) * 64 + swz;

    f32x4 acc[NB][4];   // [n block][m block]
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa0[4], fb0[3], fa1[4], fb1[3];   // the two register sets of the early fragments (A blocks, first panel of B)
    auto read_early = [&](bf16x8* fa, bf16x8* fb, int t) {
        const unsigned char* st = lds + (t % NS) * C::STAGE;
#pragma unroll
        for (int x = 0; x < 4; ++x) fa[x] = *reinterpret_cast<const bf16x8*>(st + a_base + x * 1024);
#pragma unroll
        for (int x = 0; x < 3; ++x) fb[x] = *reinterpret_cast<const bf16x8*>(st + b_base + x * 1024);
    };

    // One K step.  CUR / NXT: register sets; DMA: step t + 3 exists; NEXT: step t + 1 exists.
    auto step = [&](bf16x8* fa, bf16x8* fb, bf16x8* fan, bf16x8* fbn, auto dma_t, auto next_t, int t) {
        constexpr bool DMA = decltype(dma_t)::value, NEXT = decltype(next_t)::value;
        const unsigned char* st = lds + (t % NS) * C::STAGE;
        bf16x8 fl[NB > 3 ? NB - 3 : 1];   // the later panels' B fragments of THIS step
#pragma unroll
        for (int x = 3; x < NB; ++x) fl[x - 3] = *reinterpret_cast<const bf16x8*>(st + b_base + x * 1024);
        if constexpr (NEXT) read_early(fan, fbn, t + 1);
        if constexpr (DMA) issue(t + 3);
#pragma unroll
        for (int tn = 0; tn < 3; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
                acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[tm], acc[tn][tm], 0, 0, 0);
#pragma unroll
        for (int tn = 3; tn < NB; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
                acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[tn - 3], fa[tm], acc[tn][tm], 0, 0, 0);
        // issue order inside the step: the late reads first (their data is needed 12 MFMAs on), then one early read of the next
        // step per MFMA, then the DMA instructions one per two MFMAs (an LDS-DMA holds the wave's issue for tens of cycles: under
        // matrix work, not in front of it), then the remaining MFMAs
        constexpr int MFMAS = 4 * NB, LATE = NB - 3, EARLY = NEXT ? 7 : 0;
        constexpr int PIECES = DMA ? C::PA + C::PB_MAX : 0;
        if constexpr (LATE > 0) __builtin_amdgcn_sched_group_barrier(0x100, LATE, 0);
#pragma unroll
        for (int i = 0; i < EARLY; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        constexpr int LEFT = MFMAS - EARLY;
        constexpr int PER = PIECES > 0 ? (LEFT / 2 / PIECES > 0 ? LEFT / 2 / PIECES : 1) : 0;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, LEFT - PER * PIECES, 0);
    };
    const std::true_type yes;
    const std::false_type no;

    // ---- prologue: three steps in flight, the first two landed, first fragments in registers
    issue(0);
    issue(1);
    issue(2);
    end_of_step(no);
    G256_STAMP(1);
    read_early(fa0, fb0, 0);
    // ---- step 0 (peeled so that the pairs below start on the odd register set and the loop count is even: T is even)
    step(fa0, fb0, fa1, fb1, yes, yes, 0);
    end_of_step(no);
    int t = 1;
    for (; t < T - 3; t += 2) {
        step(fa1, fb1, fa0, fb0, yes, yes, t);
        end_of_step(no);
        step(fa0, fb0, fa1, fb1, yes, yes, t + 1);
        end_of_step(no);
    }
    // ---- the last three steps: nothing left to prefetch (t = T - 3 here)
    step(fa1, fb1, fa0, fb0, no, yes, t);
    end_of_step(yes);
    step(fa0, fb0, fa1, fb1, no, yes, t + 1);   // (no barrier needed any more: nothing writes the LDS from here on)
    step(fa1, fb1, fa0, fb0, no, no, t + 2);
    __builtin_amdgcn_sched_barrier(0);
    G256_STAMP(2);

    // ---- epilogue: lane = output row m (per m block), registers = 4 consecutive columns n (as gemm_bf16.hip)
    SisDropKey key{0u, 0u};
    constexpr bool HAS_BIAS = EPI == SIS_GEMM_EPI_BIAS || EPI == SIS_GEMM_EPI_BIAS_GELU_DROP || EPI == SIS_GEMM_EPI_BIAS_DROP_RESID;
    constexpr bool HAS_DROP = EPI == SIS_GEMM_EPI_BIAS_GELU_DROP || EPI == SIS_GEMM_EPI_BIAS_DROP_RESID;   // (GELU_BWD: the factor it reads carries the mask)
    if constexpr (HAS_DROP)
        if (p.drop_thr) key = sis_drop_key(p.seed, p.site);
    constexpr bool BF16_OUT = EPI != SIS_GEMM_EPI_BIAS_DROP_RESID;
    auto bias_of = [&](int n, bool n_ok) {
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (HAS_BIAS)
            if (n_ok) {
                const float* bp = p.bias + n;
                if (p.bias_seg) {   // query | key | value biases stay three parameters (a lane's 4 columns never straddle two)
                    if (n >= 2 * p.bias_seg) bp = p.bias2 + (n - 2 * p.bias_seg);
                    else if (n >= p.bias_seg) bp = p.bias1 + (n - p.bias_seg);
                }
                bq = *reinterpret_cast<const float4*>(bp);
            }
        return bq;
    };
    // bf16 results of one 16 x 16 block for this lane: o0 -> C (4 columns), o1 -> C2 (the pre-activation of BIAS_GELU_DROP)
    auto block_bf16 = [&](int tn, int tm, int m, int n, const float4& bq, const uint2& hpre, uint2& o0, uint2& o1) {
        float keep[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (HAS_DROP)
            if (p.drop_thr) sis_drop_quad(key, ((unsigned)m * (unsigned)p.N + (unsigned)n) >> 2, p.drop_thr, p.drop_scale, keep);
        float v[4] = {acc[tn][tm][0] + bq.x, acc[tn][tm][1] + bq.y, acc[tn][tm][2] + bq.z, acc[tn][tm][3] + bq.w};
        if constexpr (EPI == SIS_GEMM_EPI_NONE || EPI == SIS_GEMM_EPI_BIAS) {
            o0 = make_uint2(sis_pack_bf16x2(v[0], v[1]), sis_pack_bf16x2(v[2], v[3]));
        } else if constexpr (EPI == SIS_GEMM_EPI_BIAS_GELU_DROP) {
            float y[4], dd[4];   // dropout(gelu(pre)) and gelu'(pre) * the same dropout factor (what the backward multiplies by), one cdf / pdf evaluation
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float cdf, pdf;
                sis_gelu_parts(v[e], cdf, pdf);
                y[e] = v[e] * cdf * keep[e];
                dd[e] = __builtin_fmaf(v[e], pdf, cdf) * keep[e];
            }
            o1 = make_uint2(sis_pack_bf16x2(dd[0], dd[1]), sis_pack_bf16x2(dd[2], dd[3]));
            o0 = make_uint2(sis_pack_bf16x2(y[0], y[1]), sis_pack_bf16x2(y[2], y[3]));
        } else {   // SIS_GEMM_EPI_GELU_BWD: gradient w.r.t. the pre-activation: acc * the factor the forward stored
            const float d[4] = {sis_bf16_lo(hpre.x), sis_bf16_hi(hpre.x), sis_bf16_lo(hpre.y), sis_bf16_hi(hpre.y)};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= d[e];
            o0 = make_uint2(sis_pack_bf16x2(v[0], v[1]), sis_pack_bf16x2(v[2], v[3]));
        }
    };
    auto epilogue = [&](auto checked_t) {
        constexpr bool CHECKED = decltype(checked_t)::value;
        if constexpr (BF16_OUT && !CHECKED) {
            // Whole tiles, bf16 results: 16-byte stores.  A lane holds 4 consecutive columns of a block (8 bytes); the lanes of
            // rows g and g ^ 1 (16 lanes apart) exchange halves of a PAIR of column blocks through v_permlane16_swap, after
            // which an even-g lane holds 8 consecutive columns of the first block and an odd-g lane 8 of the second: half the
            // store instructions, each twice as wide (a workgroup alone on its CU cannot hide its store tail behind another's
            // matrix work: the epilogue's instruction count is wall time here).
            if ((p.ldc & 7) == 0) {
                const int colpair = 4 * (g & 2);   // 0, 0, 8, 8
#pragma unroll
                for (int tn = 0; tn + 1 < NB; tn += 2) {
                    const int na = n0 + wn * C::WNC + 16 * tn + 4 * g, nb = na + 16;
                    const float4 bqa = bias_of(na, true), bqb = bias_of(nb, true);
                    uint2 ha[4], hb[4];
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm) {
                        const long long row = (long long)(m0 + wm * 64 + 16 * tm + i16) * p.ldc;
                        if constexpr (EPI == SIS_GEMM_EPI_GELU_BWD) {
                            ha[tm] = *reinterpret_cast<const uint2*>(p.pre + row + na);
                            hb[tm] = *reinterpret_cast<const uint2*>(p.pre + row + nb);
                        }
                    }
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm) {
                        const int m = m0 + wm * 64 + 16 * tm + i16;
                        uint2 a0, a1, b0, b1;
                        block_bf16(tn, tm, m, na, bqa, ha[tm], a0, a1);
                        block_bf16(tn + 1, tm, m, nb, bqb, hb[tm], b0, b1);
                        const long long at = (long long)m * p.ldc + n0 + wn * C::WNC + 16 * (tn + (g & 1)) + colpair;
                        {
                            const auto sx = __builtin_amdgcn_permlane16_swap(a0.x, b0.x, false, false);
                            const auto sy = __builtin_amdgcn_permlane16_swap(a0.y, b0.y, false, false);
                            *reinterpret_cast<uint4*>((u16*)p.C + at) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        }
                        if constexpr (EPI == SIS_GEMM_EPI_BIAS_GELU_DROP) {
                            const auto sx = __builtin_amdgcn_permlane16_swap(a1.x, b1.x, false, false);
                            const auto sy = __builtin_amdgcn_permlane16_swap(a1.y, b1.y, false, false);
                            *reinterpret_cast<uint4*>((u16*)p.C2 + at) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        }
                    }
                }
                if constexpr (NB & 1) {   // the unpaired last block: 8-byte stores
                    constexpr int tn = NB - 1;
                    const int n = n0 + wn * C::WNC + 16 * tn + 4 * g;
                    const float4 bq = bias_of(n, true);
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm) {
                        const int m = m0 + wm * 64 + 16 * tm + i16;
                        const long long at = (long long)m * p.ldc + n;
                        uint2 hpre = make_uint2(0u, 0u), o0, o1;
                        if constexpr (EPI == SIS_GEMM_EPI_GELU_BWD) hpre = *reinterpret_cast<const uint2*>(p.pre + at);
                        block_bf16(tn, tm, m, n, bq, hpre, o0, o1);
                        *reinterpret_cast<uint2*>((u16*)p.C + at) = o0;
                        if constexpr (EPI == SIS_GEMM_EPI_BIAS_GELU_DROP) *reinterpret_cast<uint2*>((u16*)p.C2 + at) = o1;
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int tn = 0; tn < NB; ++tn) {
            const int n = n0 + wn * C::WNC + 16 * tn + 4 * g;
            const bool n_ok = !CHECKED || n < p.N;
            const float4 bq = bias_of(n, n_ok);
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
                const long long at = (long long)m * p.ldc + n;
                if constexpr (BF16_OUT) {
                    uint2 o0, o1;
                    block_bf16(tn, tm, m, n, bq, h[tm], o0, o1);
                    if (ok) {
                        *reinterpret_cast<uint2*>((u16*)p.C + at) = o0;
                        if constexpr (EPI == SIS_GEMM_EPI_BIAS_GELU_DROP) *reinterpret_cast<uint2*>((u16*)p.C2 + at) = o1;
                    }
                } else {   // SIS_GEMM_EPI_BIAS_DROP_RESID: fp32 residual stream, 16 bytes per lane and block already
                    float keep[4] = {1.f, 1.f, 1.f, 1.f};
                    if (p.drop_thr) sis_drop_quad(key, ((unsigned)m * (unsigned)p.N + (unsigned)n) >> 2, p.drop_thr, p.drop_scale, keep);
                    const float v[4] = {(acc[tn][tm][0] + bq.x) * keep[0], (acc[tn][tm][1] + bq.y) * keep[1],
                                        (acc[tn][tm][2] + bq.z) * keep[2], (acc[tn][tm][3] + bq.w) * keep[3]};
                    if (ok) *reinterpret_cast<float4*>((float*)p.C + at) = make_float4(r[tm].x + v[0], r[tm].y + v[1], r[tm].z + v[2], r[tm].w + v[3]);
                }
            }
        }
    };
    if (m0 + C::BM <= p.M && n0 + C::BN <= p.N) epilogue(std::false_type());
    else epilogue(std::true_type());
#ifdef G256_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G256_STAMP(3);
    if (p.trace && tid == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); p.trace[(size_t)blockIdx.x * 8 + 4] = ((unsigned long long)xcc << 32) | hw; }
#endif
#endif
}

#ifdef G256_TRACE
unsigned long long* g256_trace = nullptr;
#endif

template <typename C, int EPI>
int launch256(const G256Params& p, hipStream_t st, const char* name) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<C, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return sis_fail("%s: cannot raise the LDS limit: %s", name, hipGetErrorString(e));
        attr_set = true;
    }
    SIS_OCC_REPORT((gemm256_kernel<C, EPI>), C::THREADS, C::LDS);
    const int groups = p.n_tiles * sis_cdiv(p.m_tiles, 8);
    hipLaunchKernelGGL((gemm256_kernel<C, EPI>), dim3(8 * groups), dim3(C::THREADS), C::LDS, st, p);
    SIS_CHECK_LAUNCH(name);
    sis_kernel_name = name;
    return 0;
}

template <int NP>
int dispatch256(const G256Params& q, int epi, hipStream_t st) {
    typedef G256Cfg<NP> C;
    G256Params p = q;
#ifdef G256_TRACE
    p.trace = g256_trace;
#endif
    p.m_tiles = sis_cdiv(p.M, C::BM);
    p.n_tiles = sis_cdiv(p.N, C::BN);
    switch (epi) {
        case SIS_GEMM_EPI_NONE: return launch256<C, SIS_GEMM_EPI_NONE>(p, st, "gemm256_kernel<NT>");
        case SIS_GEMM_EPI_BIAS: return launch256<C, SIS_GEMM_EPI_BIAS>(p, st, "gemm256_kernel<NT,bias>");
        case SIS_GEMM_EPI_BIAS_GELU_DROP: return launch256<C, SIS_GEMM_EPI_BIAS_GELU_DROP>(p, st, "gemm256_kernel<NT,bias+gelu+dropout>");
        case SIS_GEMM_EPI_BIAS_DROP_RESID: return launch256<C, SIS_GEMM_EPI_BIAS_DROP_RESID>(p, st, "gemm256_kernel<NT,bias+dropout+residual>");
        case SIS_GEMM_EPI_GELU_BWD: return launch256<C, SIS_GEMM_EPI_GELU_BWD>(p, st, "gemm256_kernel<NT,gelu'+dropout>");
    }
    return sis_fail("sis_gemm_bf16: epilogue %d is not built for the 256-row tiles", epi);
}

}  // namespace

#ifdef G256_TRACE
extern "C" void sis_gemm256_set_trace(void* buffer) { g256_trace = (unsigned long long*)buffer; }
#endif

bool sis_gemm256_ok(int layout, int epilogue, int k, int splits) {
    return layout == 0 && epilogue != SIS_GEMM_EPI_F32 && splits == 1 && k % 64 == 0 && k >= 128;
}

int sis_gemm256_dispatch(const G256Params& p, int np, int epilogue, hipStream_t st) {
    switch (np) {
        case 1: return dispatch256<1>(p, epilogue, st);
        case 2: return dispatch256<2>(p, epilogue, st);
        case 3: return dispatch256<3>(p, epilogue, st);
    }
    return sis_fail("sis_gemm_bf16: 256-row tile with %d panels", np);
}
