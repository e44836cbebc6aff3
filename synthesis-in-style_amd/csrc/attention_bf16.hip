// Fused multi-head self-attention of the TransUNet ViT encoder, forward and backward, bf16 on the CDNA4 matrix cores.
//
// Reference: networks/trans_u_net/vit_seg_modeling.py:76-96 (Attention.forward): scores = q k^T / sqrt(d), softmax over
// the keys, context = probs v, heads merged back into [B, N, hidden]; attention dropout rate 0.0
// (vit_seg_configs.py:16).  torch materialises the [B, 12, N, N] scores / probabilities; here they never leave the
// registers (online softmax forward, recomputation from the saved log-sum-exp backward), and q / k / v are read IN
// PLACE from the fused projection's output [B, N, 3 * hidden] and the context / gradients are written in the layouts
// the neighbouring GEMMs consume ([B, N, hidden] and [B, N, 3 * hidden]): no permute / contiguous / cat kernels.
//
// Head size 64.  One wave owns 32 rows (queries in the forward and dQ kernels, keys in the dK/dV kernel) and walks the
// other axis in 64-row tiles staged by LDS-DMA (double buffered, one barrier per tile).  All products are
// v_mfma_f32_32x32x16_bf16 with the wave's own axis on the LANE (accumulator column), so that
//   * softmax statistics (row max / sum, log-sum-exp, delta) are per-lane scalars,
//   * a probability tile leaves its MFMA already laid out as the B operand of the next product (contraction over the
//     accumulator's row index, k order permuted consistently on the other operand), no LDS round trip.
// LDS tile image: [64 rows][64 d] bf16 = 128-B rows, 16-B chunk c of row r at position c ^ s2(r),
// s2(r) = t ^ ((t & 1) << 2), t = (r >> 1) & 7: conflict-free both for the ds_read_b128 row fragments (A operand, rows
// on the MFMA row index) and for the ds_read_b64_tr_b16 transposed fragments (A operand with d on the row index).
//
//   forward   S^T = K Q^T (keys x queries), p = exp2(c (s - m)), O^T += V^T P^T; writes context and LSE
//   dQ        S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T; also computes delta = rowsum(dO o O)
//   dK/dV     S = Q K^T, dP = dO V^T (queries x keys), dV^T += dO^T P, dK^T += Q^T dS
// The backward is two kernels (7 products instead of the 5 of a one-kernel design) so that no gradient is accumulated
// across workgroups: no atomics, bitwise reproducible.
#include "vit_common.h"

namespace {

typedef unsigned short u16;
typedef sis_bf16x8 bf16x8;
typedef sis_bf16x4 bf16x4;
typedef sis_f32x16 f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int HD = 64;               // head size
constexpr int TILE_BYTES = 64 * 128;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
[[maybe_unused]] constexpr float NEG_BIG = -1e30f;

struct AttnParams {
    const u16* qkv;      // [B, N, 3 * H * 64]: q | k | v, head h at columns h * 64
    const u16* ctx;      // [B, N, H * 64] forward output (backward input)
    const u16* d_ctx;    // [B, N, H * 64] gradient of the context
    u16* out_ctx;        // forward
    u16* d_qkv;          // backward
    float* lse;          // [B, H, N] log-sum-exp of the scaled scores
    float* delta;        // [B, H, N] rowsum(dO o O)
    int B, N, H, blocks; // blocks = ceil(N / 128)
    unsigned qkv_bytes, ctx_bytes;
    float scale;
};

__device__ __forceinline__ int s2(int r) { const int t = (r >> 1) & 7; return t ^ ((t & 1) << 2); }

// value of the lane 32 away (the other half of the row a 32x32 accumulator splits over lanes l and l + 32): one
// ds_bpermute (crossbar only, no LDS memory); it runs once per tile and wave, far from the critical path
__device__ __forceinline__ float other_half(float x) {
    const int lane = threadIdx.x & 63;
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, x)));
}
__device__ __forceinline__ float half_max(float x) { return fmaxf(x, other_half(x)); }
__device__ __forceinline__ float half_sum(float x) { return x + other_half(x); }

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    const u32x4 u = {sis_pack_bf16x2(v[0], v[1]), sis_pack_bf16x2(v[2], v[3]), sis_pack_bf16x2(v[4], v[5]), sis_pack_bf16x2(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, u);
}

// Everything a wave needs to stage tiles and read fragments; `lane` dependent parts are computed once.
struct TileIo {
    int src_off[2];   // DMA: this lane's byte offset inside a 64-row tile source for the wave's two pieces (row * ld * 2 + chunk * 16)
    int roff[4];      // row fragment (32 rows x 16 k): byte offset for k-step ks, add 4096 * (row block)
    int toff[2][2];   // transposed fragment: [d block][half t], add 2048 * (k-step row base / 16)
};

__device__ __forceinline__ void tile_io_init(TileIo& io, int lane, int wave, int ld) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * (wave + 4 * i) + (lane >> 3);
        io.src_off[i] = r * ld * 2 + (((lane & 7) ^ s2(r)) << 4);
    }
    const int r32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) io.roff[ks] = r32 * 128 + (((2 * ks + h) ^ s2(r32)) << 4);
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p = i16 & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 8 * t + 4 * h + q4;                       // (+ a multiple of 16)
            const int chunk = 4 * db + 2 * (g & 1) + (p >> 1);
            io.toff[db][t] = row * 128 + ((chunk ^ s2(row)) << 4) + 8 * (p & 1);
        }
}

// stage one 64-row tile: rows [row0, row0 + 64) of a [.., ld] bf16 matrix at column col0, into `dst` (wave-uniform)
__device__ __forceinline__ void tile_dma(__amdgpu_buffer_rsrc_t rs, unsigned char* dst, const int* src_off, int wave, int base_bytes) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + (wave + 4 * i) * 1024), 16, src_off[i], base_bytes, 0, 0);
}

__device__ __forceinline__ bf16x8 row_frag(const unsigned char* tile, const TileIo& io, int rb, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + rb * 4096 + io.roff[ks]);
}
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, const TileIo& io, int db, int kbase16) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + kbase16 * 2048 + io.toff[db][0]));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + kbase16 * 2048 + io.toff[db][1]));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// B-operand fragments of the wave's own 32 rows (lane = row, 16 k per step) straight from global memory
__device__ __forceinline__ void own_frags(bf16x8* f, const u16* base, int row, int ld, int lane) {
    const u16* p = base + (long long)row * ld + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const bf16x8*>(p + 16 * ks);
}

// write a wave's 32 x 64 result held as X^T (rows d in the registers, own row on the lane) through LDS as whole 128-B rows
__device__ __forceinline__ void store_rows(unsigned char* scratch, const f32x16* acc, float mul, u16* gbase, int ld, int row0, int n_rows,
                                           int lane) {
    const int r32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = 32 * db + 8 * q + 4 * h;   // registers 4q .. 4q+3 hold d .. d+3
            const uint2 v = make_uint2(sis_pack_bf16x2(acc[db][4 * q] * mul, acc[db][4 * q + 1] * mul),
                                       sis_pack_bf16x2(acc[db][4 * q + 2] * mul, acc[db][4 * q + 3] * mul));
            *reinterpret_cast<uint2*>(scratch + r32 * 128 + ((((d >> 3) ^ (r32 & 7))) << 4) + ((d & 4) << 1)) = v;
        }
    // (wave-private scratch: the compiler orders the read-back behind the writes, no barrier needed)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = 8 * it + (lane >> 3), c = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(scratch + r * 128 + ((c ^ (r & 7)) << 4));
        if (row0 + r < n_rows) *reinterpret_cast<uint4*>(gbase + (long long)(row0 + r) * ld + 8 * c) = v;
    }
}

__device__ __forceinline__ bool map_block(const AttnParams& p, int& pair, int& blk) {
    // workgroups of one (batch, head) pair take consecutive slots of one XCD group (id % 8): its K / V (or Q / dO) tiles are
    // fetched into that XCD's L2 once
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    pair = (slot / p.blocks) * 8 + xcd;
    blk = slot % p.blocks;
    return pair < p.B * p.H;
}

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int pair, blk;
    if (!map_block(p, pair, blk)) return;
    const int b = pair / p.H, hd = pair % p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int ldq = 3 * p.H * HD, ldo = p.H * HD;
    const int q0 = blk * 128 + wave * 32;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.qkv), 0, p.qkv_bytes, 0x00020000);
    TileIo io;
    tile_io_init(io, lane, wave, ldq);

    bf16x8 qf[4];
    own_frags(qf, p.qkv + ((long long)b * p.N) * ldq + hd * HD, min(q0 + r32, p.N - 1), ldq, lane);

    const int tiles = (p.N + 63) >> 6;
    const int k_base = ((b * p.N) * ldq + p.H * HD + hd * HD) * 2, v_base = k_base + p.H * HD * 2;
    auto issue = [&](int t, int stage) {
        tile_dma(rs, lds + stage * STAGE_BYTES, io.src_off, wave, k_base + t * 64 * ldq * 2);
        tile_dma(rs, lds + stage * STAGE_BYTES + TILE_BYTES, io.src_off, wave, v_base + t * 64 * ldq * 2);
    };

    const float c = p.scale * 1.4426950408889634f;
    float m_run = NEG_BIG, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { oacc[0][i] = 0.f; oacc[1][i] = 0.f; }

    issue(0, 0);
    for (int t = 0; t < tiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < tiles) issue(t + 1, (t + 1) & 1);
        const unsigned char* kt = lds + (t & 1) * STAGE_BYTES;
        const unsigned char* vt = kt + TILE_BYTES;
        // Every fragment of a phase is requested before the phase's first MFMA (left to itself the compiler reads one fragment,
        // waits for it, multiplies, reads the next: eight exposed LDS latencies per product); the V^T fragments of the second
        // product are requested before the softmax arithmetic, which does not need them, and land underneath it.
        f32x16 s[2];
        bf16x8 kfr[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kfr[kb][ks] = row_frag(kt, io, kb, ks);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][ks], qf[ks], s[kb], 0, 0, 0);
        }
        bf16x8 vfr[2][4];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) vfr[db][kk] = tr_frag(vt, io, db, kk);
        __builtin_amdgcn_sched_barrier(0);
        if (t * 64 + 64 > p.N) {   // last, partial tile: keys past the end do not exist
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (t * 64 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h >= p.N) s[kb][i] = NEG_BIG;
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
        mx = half_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(c * (m_run - m_new));
        const float mc = c * m_new;
        m_run = m_new;
        float sum = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    e[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(c, s[kb][8 * sx + j], -mc));
                    sum += e[j];
                }
                pf[kb][sx] = pack8(e);
            }
        l_run = l_run * alpha + sum;
#pragma unroll
        for (int i = 0; i < 16; ++i) { oacc[0][i] *= alpha; oacc[1][i] *= alpha; }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[db][2 * kb + sx], pf[kb][sx], oacc[db], 0, 0, 0);
    }
    const float l_tot = half_sum(l_run);
    const int q = q0 + r32;
    if (h == 0 && q < p.N) p.lse[((long long)b * p.H + hd) * p.N + q] = m_run * p.scale + __logf(l_tot);
    __builtin_amdgcn_s_barrier();   // every wave is done with the tiles: the stages become the output scratch
    store_rows(lds + wave * 4096, oacc, 1.f / l_tot, p.out_ctx + ((long long)b * p.N) * ldo + hd * HD, ldo, q0, p.N, lane);
#endif
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int pair, blk;
    if (!map_block(p, pair, blk)) return;
    const int b = pair / p.H, hd = pair % p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int ldq = 3 * p.H * HD, ldo = p.H * HD;
    const int q0 = blk * 128 + wave * 32;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.qkv), 0, p.qkv_bytes, 0x00020000);
    TileIo io;
    tile_io_init(io, lane, wave, ldq);

    const int qc = min(q0 + r32, p.N - 1);
    bf16x8 qf[4], gf[4];
    own_frags(qf, p.qkv + ((long long)b * p.N) * ldq + hd * HD, qc, ldq, lane);
    own_frags(gf, p.d_ctx + ((long long)b * p.N) * ldo + hd * HD, qc, ldo, lane);
    float dl = 0.f;
    {
        bf16x8 of[4];
        own_frags(of, p.ctx + ((long long)b * p.N) * ldo + hd * HD, qc, ldo, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) dl = __builtin_fmaf((float)gf[ks][j], (float)of[ks][j], dl);
    }
    dl = half_sum(dl);
    const long long stat = ((long long)b * p.H + hd) * p.N;
    if (h == 0 && q0 + r32 < p.N) p.delta[stat + q0 + r32] = dl;
    const float c = p.scale * 1.4426950408889634f;
    const float s_init = -p.lse[stat + qc] / p.scale;   // exp2(c * (s - lse / scale)) = exp(scale * s - lse)

    const int tiles = (p.N + 63) >> 6;
    const int k_base = ((b * p.N) * ldq + p.H * HD + hd * HD) * 2, v_base = k_base + p.H * HD * 2;
    auto issue = [&](int t, int stage) {
        tile_dma(rs, lds + stage * STAGE_BYTES, io.src_off, wave, k_base + t * 64 * ldq * 2);
        tile_dma(rs, lds + stage * STAGE_BYTES + TILE_BYTES, io.src_off, wave, v_base + t * 64 * ldq * 2);
    };
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }

    issue(0, 0);
    for (int t = 0; t < tiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < tiles) issue(t + 1, (t + 1) & 1);
        const unsigned char* kt = lds + (t & 1) * STAGE_BYTES;
        const unsigned char* vt = kt + TILE_BYTES;
        bf16x8 dsf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = s_init; dp[i] = -dl; }
            bf16x8 kfr[4], vfr[4];   // all eight row fragments of this key block before its first MFMA
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { kfr[ks] = row_frag(kt, io, kb, ks); vfr[ks] = row_frag(vt, io, kb, ks); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[ks], s, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ks], gf[ks], dp, 0, 0, 0);
            const bool partial = t * 64 + 64 > p.N;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * sx + j;
                    float pr = __builtin_amdgcn_exp2f(c * s[i]);
                    if (partial && t * 64 + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h >= p.N) pr = 0.f;
                    e[j] = pr * dp[i];
                }
                dsf[kb][sx] = pack8(e);
            }
        }
        bf16x8 ktr[2][4];   // K^T fragments of the third product, all requested first
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) ktr[db][kk] = tr_frag(kt, io, db, kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktr[db][2 * kb + sx], dsf[kb][sx], dq[db], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
    store_rows(lds + wave * 4096, dq, p.scale, p.d_qkv + ((long long)b * p.N) * ldq + hd * HD, ldq, q0, p.N, lane);
#endif
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int pair, blk;
    if (!map_block(p, pair, blk)) return;
    const int b = pair / p.H, hd = pair % p.H;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int ldq = 3 * p.H * HD, ldo = p.H * HD;
    const int k0 = blk * 128 + wave * 32;
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.qkv), 0, p.qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.d_ctx), 0, p.ctx_bytes, 0x00020000);
    TileIo io, io_o;
    tile_io_init(io, lane, wave, ldq);
    tile_io_init(io_o, lane, wave, ldo);   // (only the DMA offsets differ: same LDS image)

    const int kc = min(k0 + r32, p.N - 1);
    bf16x8 kf[4], vf[4];
    own_frags(kf, p.qkv + ((long long)b * p.N) * ldq + p.H * HD + hd * HD, kc, ldq, lane);
    own_frags(vf, p.qkv + ((long long)b * p.N) * ldq + 2 * p.H * HD + hd * HD, kc, ldq, lane);
    const long long stat = ((long long)b * p.H + hd) * p.N;
    const float c = p.scale * 1.4426950408889634f, inv_scale = 1.f / p.scale;

    const int tiles = (p.N + 63) >> 6;
    const int q_base = ((b * p.N) * ldq + hd * HD) * 2, g_base = ((b * p.N) * ldo + hd * HD) * 2;
    auto issue = [&](int t, int stage) {
        tile_dma(rsq, lds + stage * STAGE_BYTES, io.src_off, wave, q_base + t * 64 * ldq * 2);
        tile_dma(rso, lds + stage * STAGE_BYTES + TILE_BYTES, io_o.src_off, wave, g_base + t * 64 * ldo * 2);
    };
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }

    issue(0, 0);
    for (int t = 0; t < tiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < tiles) issue(t + 1, (t + 1) & 1);
        const unsigned char* qt = lds + (t & 1) * STAGE_BYTES;
        const unsigned char* gt = qt + TILE_BYTES;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // accumulators start at the row constants: -lse / scale and -delta of the query each register row belongs to
            f32x16 s, dp;
            const int qrow = t * 64 + 32 * qb + 4 * h;   // register i: query qrow + (i & 3) + 8 * (i >> 2)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 l4 = make_float4(0.f, 0.f, 0.f, 0.f), d4 = l4;
                const int qq = qrow + 8 * g4;
                if ((p.N & 3) == 0 && qq + 3 < p.N) {
                    l4 = *reinterpret_cast<const float4*>(p.lse + stat + qq);
                    d4 = *reinterpret_cast<const float4*>(p.delta + stat + qq);
                } else {
                    float* lp = reinterpret_cast<float*>(&l4); float* dp4 = reinterpret_cast<float*>(&d4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (qq + e < p.N) { lp[e] = p.lse[stat + qq + e]; dp4[e] = p.delta[stat + qq + e]; }
                }
                s[4 * g4] = -l4.x * inv_scale; s[4 * g4 + 1] = -l4.y * inv_scale; s[4 * g4 + 2] = -l4.z * inv_scale; s[4 * g4 + 3] = -l4.w * inv_scale;
                dp[4 * g4] = -d4.x; dp[4 * g4 + 1] = -d4.y; dp[4 * g4 + 2] = -d4.z; dp[4 * g4 + 3] = -d4.w;
            }
            bf16x8 qfr[4], gfr[4];   // all eight row fragments of this query block before its first MFMA
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { qfr[ks] = row_frag(qt, io, qb, ks); gfr[ks] = row_frag(gt, io, qb, ks); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[ks], kf[ks], s, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[ks], vf[ks], dp, 0, 0, 0);
            bf16x8 gtr[2][2], qtr[2][2];   // the transposed fragments of the two gradient products: requested before the
#pragma unroll                                // exponentials, which do not need them
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) { gtr[db][sx] = tr_frag(gt, io, db, 2 * qb + sx); qtr[db][sx] = tr_frag(qt, io, db, 2 * qb + sx); }
            __builtin_amdgcn_sched_barrier(0);
            const bool partial = t * 64 + 64 > p.N;
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                float e[8], f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = 8 * sx + j;
                    float pr = __builtin_amdgcn_exp2f(c * s[i]);
                    if (partial && qrow + (i & 3) + 8 * (i >> 2) >= p.N) pr = 0.f;   // queries past the end do not exist
                    e[j] = pr;
                    f[j] = pr * dp[i];
                }
                pf[sx] = pack8(e);
                dsf[sx] = pack8(f);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gtr[db][sx], pf[sx], dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtr[db][sx], dsf[sx], dk[db], 0, 0, 0);
                }
        }
    }
    __builtin_amdgcn_s_barrier();
    u16* gk = p.d_qkv + ((long long)b * p.N) * ldq + p.H * HD + hd * HD;
    store_rows(lds + wave * 4096, dk, p.scale, gk, ldq, k0, p.N, lane);
    store_rows(lds + 16384 + wave * 4096, dv, 1.f, gk + p.H * HD, ldq, k0, p.N, lane);
#endif
}

int attn_params(AttnParams& p, const void* qkv, int batch, int n, int heads, const char* who) {
    if (!(qkv && batch > 0 && n > 0 && heads > 0)) return sis_fail("%s: null pointer or empty shape", who);
    const int64_t qe = (int64_t)batch * n * 3 * heads * HD;
    if (qe * 2 >= (1LL << 31)) return sis_fail("%s: the fused projection exceeds 2 GiB", who);
    if (((uintptr_t)qkv) & 15) return sis_fail("%s: pointers must be 16-byte aligned", who);
    p.qkv = (const u16*)qkv; p.B = batch; p.N = n; p.H = heads; p.blocks = sis_cdiv(n, 128);
    p.qkv_bytes = (unsigned)(qe * 2); p.ctx_bytes = (unsigned)(qe * 2 / 3);
    p.scale = 0.125f;   // 1 / sqrt(64)
    p.ctx = p.d_ctx = nullptr; p.out_ctx = p.d_qkv = nullptr; p.lse = p.delta = nullptr;
    return 0;
}

template <typename K>
int attn_launch(K kernel, const AttnParams& p, hipStream_t st, const char* name) {
    static_assert(LDS_BYTES <= 64 * 1024, "dynamic LDS above 64 KiB needs hipFuncSetAttribute");
    const int grid = 8 * p.blocks * sis_cdiv(p.B * p.H, 8);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), LDS_BYTES, st, p);
    SIS_CHECK_LAUNCH(name);
    return 0;
}

}  // namespace

extern "C" int sis_attention_fwd(void* ctx, float* lse, const void* qkv, int batch, int n, int heads, void* stream) {
    AttnParams p;
    if (int rc = attn_params(p, qkv, batch, n, heads, "sis_attention_fwd")) return rc;
    SIS_REQUIRE(ctx && lse, "sis_attention_fwd: null pointer");
    p.out_ctx = (u16*)ctx; p.lse = lse;
    return attn_launch(attn_fwd_kernel, p, (hipStream_t)stream, "attn_fwd_kernel");
}

extern "C" int sis_attention_bwd(void* d_qkv, float* delta, const void* d_ctx, const void* qkv, const void* ctx, const float* lse,
                                 int batch, int n, int heads, void* stream) {
    AttnParams p;
    if (int rc = attn_params(p, qkv, batch, n, heads, "sis_attention_bwd")) return rc;
    SIS_REQUIRE(d_qkv && delta && d_ctx && ctx && lse, "sis_attention_bwd: null pointer");
    SIS_REQUIRE(((((uintptr_t)d_ctx) | ((uintptr_t)ctx) | ((uintptr_t)d_qkv)) & 15) == 0, "sis_attention_bwd: pointers must be 16-byte aligned");
    p.d_qkv = (u16*)d_qkv; p.delta = delta; p.d_ctx = (const u16*)d_ctx; p.ctx = (const u16*)ctx; p.lse = const_cast<float*>(lse);
    if (int rc = attn_launch(attn_bwd_dq_kernel, p, (hipStream_t)stream, "attn_bwd_dq_kernel")) return rc;   // writes delta
    return attn_launch(attn_bwd_dkv_kernel, p, (hipStream_t)stream, "attn_bwd_dkv_kernel");
}
