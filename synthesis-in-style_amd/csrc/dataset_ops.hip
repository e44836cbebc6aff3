// "Next" rows of SURVEY.md §8(f): what sits immediately downstream of Generator.forward in the dataset loop.
//
//  * sis_kmeans_assign   nearest k-means centre per pixel of an activation map:
//                        labels[b,h,w] = argmin_k sum_c (x[b,c,h,w] - centre[k,c])^2
//                        (segmentation/gan_local_edit/factor_catalog.py:47-62 of the reference, which ships the
//                        activations to the CPU and materialises an N x K x C tensor there).  Direct
//                        (x - c)^2 form in fp32 with the reference's own association of the adds (see the kernel):
//                        label maps are bit-exact against the fp32 oracle.  Two passes: a fast one (2 VALU operations per
//                        term, free order) labels every pixel whose argmin provably does not depend on the association
//                        of the adds, the exact-order one revisits the few workgroups with an undecided pixel.
//  * sis_make_image_u8   float image in [-1,1] (NCHW) -> uint8 NHWC, the conversion in front of the PNG writer
//                        (create_dataset_for_segmentation.py:135; third-party make_image: clamp, add 1, div 2, mul 255,
//                        truncating cast; every step a separate fp32 operation as in the oracle: bytes are bit-exact).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "sis_common.h"

namespace {

typedef float km_f32x2 __attribute__((ext_vector_type(2)));
typedef float km_f32x4 __attribute__((ext_vector_type(4)));
// Packed fp32 forms of the three separately rounded operations of a k-means term, two pixels per instruction (IEEE add / mul
// per half: the same bits as the scalar instructions; inline assembly, so no contraction either).  The centre is one half
// of a register pair as ds_read_b128 delivers it, broadcast to both pixels by op_sel.
template <int HI>
__device__ __forceinline__ km_f32x2 km_sub_bcast(km_f32x2 x, km_f32x2 cpair) {  // x - cpair[HI]
    km_f32x2 r;
    if constexpr (HI == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(cpair));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(cpair));
    return r;
}
__device__ __forceinline__ km_f32x2 km_sq(km_f32x2 d) {
    km_f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %1" : "=v"(r) : "v"(d));
    return r;
}
__device__ __forceinline__ km_f32x2 km_add(km_f32x2 a, km_f32x2 b) {
    km_f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Summation order (documented, bit-exact contract).  The reference evaluates ``((A - B) ** 2.0).sum(dim=-1)`` with
// torch's CPU reduction (factor_catalog.py:55-59), whose result for near-ties depends on the association of the fp32
// adds.  This kernel reproduces, add for add, the association of ATen's inner-dimension float sum
// (aten/src/ATen/native/cpu/SumKernel.cpp: vectorized_inner_sum -> row_sum -> multi_row_sum; 8-float vectors under both
// the AVX2 and the AVX-512 dispatch), restated in oracle/kmeans_ref.py::predict_ordered and pinned there against
// torch's own sum:
//   * channel c < 8*(C/8) is element l = c % 8 of vector i = c / 8; vector i feeds partial sum k = i % 4 in row
//     r = i / 4, for i < 4*(C/32).  Per (l, k): running sum over the rows in order starting from 0; after every 16th
//     row the running sum is added to a second-level sum and reset; p[k] = running + second level;
//   * lane total = (((p[0] + vectors i >= 4*(C/32) in order) + p[1]) + p[2]) + p[3];
//   * result = (((0 + scalar tail terms c >= 8*(C/8) in order) + lane 0) + lane 1) ... + lane 7;
//   * every term is round(round(x - c) * round(x - c)): subtract, multiply and add are separate IEEE operations
//     (no FMA contraction: #pragma clang fp contract(off)); ties go to the lowest centre index (torch.argmin).
// 8 <= C < 8192 here (below 8 ATen takes its scalar path: kmeans_small_c_kernel; at 8192 a third cascade level starts).  The activations are
// still read exactly once, one channel plane row per step.
template <int KMAX, int VEC, bool CASCADE>
__global__ __launch_bounds__(256) void kmeans_assign_kernel(int64_t* __restrict__ labels, const float* __restrict__ x,
                                                            const float* __restrict__ centres, int C, int HW, int K,
                                                            int groups, int refine, const int* __restrict__ open_count,
                                                            const int* __restrict__ open_list) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float cen[];  // [C][KMAX]
    int b = blockIdx.x / groups;
    const int g = blockIdx.x % groups;
    int n_open = 0;
    if (refine == 3) {   // second pass behind kmeans_fast_kernel, list form: lane i takes the i-th pixel it left open (VEC == 1)
        n_open = *open_count;
        if ((int)blockIdx.x * 256 >= n_open) return;
    } else if (refine) {   // block form: only workgroups / waves with a pixel the fast pass could not decide (label -1)
        const int pix0 = (g * 256 + threadIdx.x) * VEC;
        bool open = false;
        for (int v = 0; v < VEC; ++v) open = open || (pix0 + v < HW && labels[(int64_t)b * HW + pix0 + v] < 0);
        if (!__syncthreads_or(open)) return;
        refine = __any(open) ? 1 : 2;   // 2: this wave has nothing to do (it still helps staging the centres)
    }
    for (int e = threadIdx.x; e < C * KMAX; e += 256) {
        const int c = e / KMAX, k = e - c * KMAX;
        cen[e] = k < K ? centres[(int64_t)k * C + c] : 0.f;
    }
    __syncthreads();
    int pix = (g * 256 + threadIdx.x) * VEC;
    if (refine == 3) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= n_open) return;
        const int q = open_list[i];
        b = q / HW; pix = q - b * HW;
    }
    if (pix >= HW || refine == 2) return;
    constexpr int NA1 = CASCADE ? KMAX : 1;
    float fin[KMAX][VEC], lt[KMAX][VEC], acc0[KMAX][VEC], acc1[NA1][VEC];
    const float* xb = x + (int64_t)b * C * HW + pix;
    auto accumulate = [&](int c, float (&acc)[KMAX][VEC]) {  // acc[k][v] += (x[c] - centre[k][c])^2, three roundings
        float xv[VEC];
        if constexpr (VEC == 2) {
            const float2 t = *reinterpret_cast<const float2*>(xb + (int64_t)c * HW);
            xv[0] = t.x; xv[1] = t.y;
        } else {
            xv[0] = xb[(int64_t)c * HW];
        }
        const float* cc = cen + c * KMAX;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float diff = xv[v] - cc[k];
                acc[k][v] = acc[k][v] + diff * diff;
            }
    };
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) fin[k][v] = 0.f;
    const int nvec = C >> 3, rows = nvec >> 2;
    // Fast path (C a multiple of 128, no second cascade level: the generator's 128- and 512-channel keys): the SAME sequence of
    // channels and adds, but the loads of eight steps are requested while the previous eight multiply (the plain loop below
    // waits for every channel's load before its 3 * K VALU operations: with 2-3 waves per SIMD the kernel sat at a third of
    // the VALU rate that bounds it).  A lane position l has 4 * rows steps p = part * rows + r (channel ((4 r + part) * 8) + l),
    // taken in groups of eight; two register sets, the set parity a compile-time constant (rows / 2 groups per l: even).
    const bool fast = !CASCADE && (C & 127) == 0;
    if (fast) {
        float xs[2][8][VEC];
        km_f32x2 a2[KMAX], l2[KMAX], f2[KMAX];  // (VEC == 2: the accumulators as register pairs)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { a2[k] = km_f32x2{0.f, 0.f}; l2[k] = a2[k]; f2[k] = a2[k]; }
        constexpr bool PK = VEC == 2 && KMAX % 4 == 0;
        const int ng = rows >> 1;  // groups of eight steps per lane position
        auto chan = [&](int l, int p) { const int part = p / rows, r = p - part * rows; return (((r << 2) + part) << 3) + l; };
        auto request = [&](auto setc, int l, int g) {
            constexpr int S = decltype(setc)::value;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = chan(l, g * 8 + u);
                if constexpr (VEC == 2) {
                    const float2 t = *reinterpret_cast<const float2*>(xb + (int64_t)c * HW);
                    xs[S][u][0] = t.x; xs[S][u][1] = t.y;
                } else {
                    xs[S][u][0] = xb[(int64_t)c * HW];
                }
            }
        };
        auto group = [&](auto setc, int l, int g) {
            constexpr int S = decltype(setc)::value;
            // the next group (past the end: the last one again, unused) -- requested before this group's arithmetic
            int ln = l, gn = g + 1;
            if (gn == ng) { gn = 0; ++ln; }
            if (ln == 8) { ln = 7; gn = ng - 1; }
            request(std::integral_constant<int, S ^ 1>(), ln, gn);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = g * 8 + u;
                const int part = p / rows, r = p - part * rows;
                if (r == 0) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        if constexpr (PK) a2[k] = km_f32x2{0.f, 0.f};
                        else
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc0[k][v] = 0.f;
                    }
                }
                const float* cc = cen + ((((r << 2) + part) << 3) + l) * KMAX;
                if constexpr (VEC == 2 && KMAX % 4 == 0) {  // two pixels per instruction
                    const km_f32x2 xp = {xs[S][u][0], xs[S][u][1]};
#pragma unroll
                    for (int k4 = 0; k4 < KMAX; k4 += 4) {
                        const km_f32x4 c4 = *reinterpret_cast<const km_f32x4*>(cc + k4);
                        const km_f32x2 c01 = {c4.x, c4.y}, c23 = {c4.z, c4.w};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const km_f32x2 cp = kk < 2 ? c01 : c23;
                            const km_f32x2 d = (kk & 1) ? km_sub_bcast<1>(xp, cp) : km_sub_bcast<0>(xp, cp);
                            a2[k4 + kk] = km_add(a2[k4 + kk], km_sq(d));
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const float diff = xs[S][u][v] - cc[k];
                            acc0[k][v] = acc0[k][v] + diff * diff;
                        }
                }
                if (r == rows - 1) {  // end of the run of `part`: p[part] joins the lane total, the lane total the result
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        if constexpr (PK) {
                            l2[k] = part == 0 ? a2[k] : km_add(l2[k], a2[k]);
                            if (part == 3) f2[k] = km_add(f2[k], l2[k]);
                        } else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                lt[k][v] = part == 0 ? acc0[k][v] : lt[k][v] + acc0[k][v];
                                if (part == 3) fin[k][v] = fin[k][v] + lt[k][v];
                            }
                        }
                    }
                }
            }
        };
        const std::integral_constant<int, 0> set0;
        const std::integral_constant<int, 1> set1;
        request(set0, 0, 0);
        for (int l = 0; l < 8; ++l)
            for (int g = 0; g < ng; g += 2) { group(set0, l, g); group(set1, l, g + 1); }
        if constexpr (PK) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) { fin[k][0] = f2[k].x; fin[k][VEC - 1] = f2[k].y; }
        }
    }
    for (int c = nvec << 3; c < C && !fast; ++c) accumulate(c, fin);  // scalar tail first
    for (int l = 0; l < 8 && !fast; ++l) {
        for (int part = 0; part < 4; ++part) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    acc0[k][v] = 0.f;
                    if constexpr (CASCADE) acc1[k][v] = 0.f;
                }
            for (int r = 0; r < rows; ++r) {
                accumulate((((r << 2) + part) << 3) + l, acc0);
                if constexpr (CASCADE) {
                    if ((r & 15) == 15) {
#pragma unroll
                        for (int k = 0; k < KMAX; ++k)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) { acc1[k][v] = acc1[k][v] + acc0[k][v]; acc0[k][v] = 0.f; }
                    }
                }
            }
            if constexpr (CASCADE) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc0[k][v] = acc0[k][v] + acc1[k][v];
            }
            if (part == 0) {
                for (int i = rows << 2; i < nvec; ++i) accumulate((i << 3) + l, acc0);  // left-over vectors join p[0]
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) lt[k][v] = acc0[k][v];
            } else {
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) lt[k][v] = lt[k][v] + acc0[k][v];
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) fin[k][v] = fin[k][v] + lt[k][v];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float best = fin[0][v];
        int arg = 0;
#pragma unroll
        for (int k = 1; k < KMAX; ++k)
            if (k < K && fin[k][v] < best) { best = fin[k][v]; arg = k; }
        labels[(int64_t)b * HW + pix + v] = arg;
    }
}

// First pass for the common shapes (two pixels per lane, K <= KMAX, C % 16 == 0): the same distances with one subtraction
// and ONE fused multiply-add per term (2 VALU operations instead of the 3 separately rounded ones, no prescribed order),
// accumulated in runs of 16 channels that are then added to a total -- every term passes through <= 16 + C / 16 additions, so
//     |d_fast - D| <= (16 + C/16 + 3) u D,      |d_exact - D| <= (C/32 + 32) u D      (u = 2^-24, D the real distance:
// sums of non-negative terms, standard recursive-summation bound per level; the exact-order kernel's cascade is no deeper
// than C/32 + 27).  A pixel whose two smallest fast distances differ by more than `thr` * the second one (thr = 3 x the sum
// of the two bounds) therefore has the same argmin under the exact association; it gets its label here.  Every other pixel
// (near-ties, duplicate centres, NaN) gets -1 (and, given a workspace, an entry in the list of open pixels) and is decided
// in exact order, add for add as before: the label map stays bit-exact.  Unit-variance data, 24 random centres: 0.05 % (128
// channels) to 0.4 % (512) of the pixels stay open -- at least one in most 512-pixel workgroups, hence the list: the block
// form of the refinement (no workspace) would redo almost everything.
__device__ __forceinline__ km_f32x2 km_fma_sq(km_f32x2 d, km_f32x2 acc) {   // acc + d * d, one rounding per half
    km_f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(r) : "v"(d), "v"(acc));
    return r;
}

template <int KMAX>
__global__ __launch_bounds__(256) void kmeans_fast_kernel(int64_t* __restrict__ labels, const float* __restrict__ x,
                                                          const float* __restrict__ centres, int C, int HW, int K, int groups,
                                                          float thr, int* __restrict__ open_count, int* __restrict__ open_list) {
    static_assert(KMAX % 4 == 0, "centres are read four at a time");
    extern __shared__ __attribute__((aligned(16))) float cen[];  // [C][KMAX]
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    for (int e = threadIdx.x; e < C * KMAX; e += 256) {
        const int c = e / KMAX, k = e - c * KMAX;
        cen[e] = k < K ? centres[(int64_t)k * C + c] : 0.f;
    }
    __syncthreads();
    const int pix = (g * 256 + threadIdx.x) * 2;
    if (pix >= HW) return;
    const float* xb = x + (int64_t)b * C * HW + pix;
    km_f32x2 run[KMAX], tot[KMAX];   // (four pixels per lane -- a channel's centres read once per 96 operations -- measured slower: 400 vs 300 us)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { run[k] = km_f32x2{0.f, 0.f}; tot[k] = run[k]; }
    float2 xs[2][16];
    auto request = [&](auto setc, int c0) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int u = 0; u < 16; ++u) xs[S][u] = *reinterpret_cast<const float2*>(xb + (int64_t)(c0 + u) * HW);
    };
    auto block = [&](auto setc, int c0) {
        constexpr int S = decltype(setc)::value;
        request(std::integral_constant<int, S ^ 1>(), c0 + 16 < C ? c0 + 16 : c0);   // (past the end: this block again, unused)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const km_f32x2 xp = {xs[S][u].x, xs[S][u].y};
            const float* cc = cen + (c0 + u) * KMAX;
#pragma unroll
            for (int k4 = 0; k4 < KMAX; k4 += 4) {
                const km_f32x4 c4 = *reinterpret_cast<const km_f32x4*>(cc + k4);
                const km_f32x2 c01 = {c4.x, c4.y}, c23 = {c4.z, c4.w};
                run[k4 + 0] = km_fma_sq(km_sub_bcast<0>(xp, c01), run[k4 + 0]);
                run[k4 + 1] = km_fma_sq(km_sub_bcast<1>(xp, c01), run[k4 + 1]);
                run[k4 + 2] = km_fma_sq(km_sub_bcast<0>(xp, c23), run[k4 + 2]);
                run[k4 + 3] = km_fma_sq(km_sub_bcast<1>(xp, c23), run[k4 + 3]);
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { tot[k] = km_add(tot[k], run[k]); run[k] = km_f32x2{0.f, 0.f}; }
    };
    request(std::integral_constant<int, 0>(), 0);
    for (int c0 = 0; c0 < C; c0 += 32) {   // C % 16 == 0 (host-checked)
        block(std::integral_constant<int, 0>(), c0);
        if (c0 + 16 < C) block(std::integral_constant<int, 1>(), c0 + 16);
        else break;
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        float d1 = v == 0 ? tot[0].x : tot[0].y, d2 = __builtin_inff();
        int arg = 0;
#pragma unroll
        for (int k = 1; k < KMAX; ++k) {
            if (k < K) {
                const float d = v == 0 ? tot[k].x : tot[k].y;
                if (d < d1) { d2 = d1; d1 = d; arg = k; }
                else if (!(d >= d2)) d2 = d;   // (a NaN lands in d2: the pixel stays open)
            }
        }
        const bool sure = K == 1 || (d2 - d1 > thr * d2);   // false for near-ties, exact ties and NaN
        if (pix + v < HW) {
            labels[(int64_t)b * HW + pix + v] = sure ? arg : -1;
            if (!sure && open_list) open_list[atomicAdd(open_count, 1)] = b * HW + pix + v;   // (order irrelevant: one pixel per entry)
        }
    }
}


// First pass on the matrix cores (the shapes of the generator's catalogued layers: HW % 128 == 0, C % 16 == 0, K <= 32).
// argmin_k |x - c_k|^2 = argmin_k (|c_k|^2 - 2 x.c_k): the dot products are a GEMM  S[k][pixel] = sum_c centres[k][c] x[c][pixel]
// whose B operand is the NCHW activation as it lies in memory -- v_mfma_f32_32x32x2_f32 (exact fp32 products and sums) takes
// B[k = channel][n = pixel] one float per lane, so a lane's 16-byte load of FOUR consecutive pixels of channel c0 + (lane >> 5)
// feeds four 32-pixel column tiles (pixel 4 n + j of the wave's 128 in tile j).  The VALU pass above spends 2 K operations per
// element (28 TF/s, 2.4 TB/s on 2.7 GB of activations per batch of 32); this one is bound by the read of x.
// Decision rule: with S computed by a chain of C/2 MFMAs every product passes through at most C additions, so
//     |d_fast_k - (D_k - |x|^2)| <= (C + 3) u (|x|^2 + 2 |c_k|^2)      (2 sum|x c| <= |x|^2 + |c_k|^2;  |c_k|^2 summed in fp32)
// and the exact-order kernel's |d_exact - D| <= (C/32 + 32) u D <= (C/32 + 32) u 2 (|x|^2 + |c_k|^2).  A pixel whose two smallest
// fast values differ by more than e_x |x|^2 + e_c max|c|^2 (twice the sum of both bounds, x 1.5) has the same argmin in exact
// order and gets its label here; every other pixel (near-ties, NaN) is listed for kmeans_refine_kernel as before.
typedef float km_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void kmeans_mfma_kernel(int64_t* __restrict__ labels, const float* __restrict__ x,
                                                          const float* __restrict__ centres, int C, int HW, int K, int tiles,
                                                          float e_x, float e_c, int* __restrict__ open_count,
                                                          int* __restrict__ open_list) {
    extern __shared__ __attribute__((aligned(16))) float cen[];  // [C][32] centres (zero beyond K), [8][32] partial / [32] squared norms
    float* ccp = cen + C * 32;
    float* ccl = ccp + 8 * 32;
    // Workgroups are persistent over the wave tiles (128 pixels each): the centres are staged and their norms summed once per
    // workgroup, not once per 512 pixels (at 128 channels that prologue was a quarter of a workgroup's time).
    for (int e = threadIdx.x; e < C * 32; e += 256) {
        const int c = e >> 5, k = e & 31;
        cen[e] = k < K ? centres[(int64_t)k * C + c] : 0.f;
    }
    __syncthreads();
    {   // |c_k|^2: eight interleaved partial sums per centre, added in order
        const int k = threadIdx.x & 31, part = threadIdx.x >> 5;
        float s = 0.f;
        for (int c = part; c < C; c += 8) s = fmaf(cen[c * 32 + k], cen[c * 32 + k], s);
        ccp[part * 32 + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.f;
        for (int p = 0; p < 8; ++p) s += ccp[p * 32 + threadIdx.x];
        ccl[threadIdx.x] = s;
    }
    __syncthreads();
    float ccmax = 0.f;
    for (int k = 0; k < K; ++k) ccmax = fmaxf(ccmax, ccl[k]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const float* ab = cen + half * 32 + l31;
    const int tiles_per_sample = HW >> 7;   // HW % 128 == 0: a wave tile lies inside one sample
    constexpr int U = 16;   // MFMA steps (2 channels each) per register set: 16 KB of x in flight per wave
    for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
        const int b = t / tiles_per_sample, pix0 = (t - b * tiles_per_sample) << 7;
        const float* xb = x + ((int64_t)b * C + half) * HW + pix0 + 4 * l31;
        km_f32x16 acc[4];
        float xx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        km_f32x4 xs[2][U];
        auto request = [&](auto setc, int c0) {
            constexpr int S = decltype(setc)::value;
#pragma unroll
            for (int s = 0; s < U; ++s) xs[S][s] = *reinterpret_cast<const km_f32x4*>(xb + (int64_t)(c0 + 2 * s) * HW);
        };
        auto block = [&](auto setc, int c0) {
            constexpr int S = decltype(setc)::value;
            request(std::integral_constant<int, S ^ 1>(), c0 + 2 * U < C ? c0 + 2 * U : c0);   // (past the end: this block again, unused)
#pragma unroll
            for (int s = 0; s < U; ++s) {
                const float a = ab[(c0 + 2 * s) * 32];
                const km_f32x4 v = xs[S][s];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v.w, acc[3], 0, 0, 0);
                xx[0] = fmaf(v.x, v.x, xx[0]); xx[1] = fmaf(v.y, v.y, xx[1]); xx[2] = fmaf(v.z, v.z, xx[2]); xx[3] = fmaf(v.w, v.w, xx[3]);
            }
        };
        request(std::integral_constant<int, 0>(), 0);
        for (int c0 = 0; c0 < C; c0 += 4 * U) {   // C % 64 == 0 (host-checked)
            block(std::integral_constant<int, 0>(), c0);
            block(std::integral_constant<int, 1>(), c0 + 2 * U);
        }
        int64_t lab[4];
        bool open[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xxj = xx[j] + __shfl_xor(xx[j], 32, 64);
            float d1 = __builtin_inff(), d2 = __builtin_inff();
            int arg = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = (i & 3) + 8 * (i >> 2) + 4 * half;   // this register's centre
                if (k < K) {
                    const float d = ccl[k] - 2.f * acc[j][i];
                    if (d < d1) { d2 = d1; d1 = d; arg = k; }
                    else if (!(d >= d2)) d2 = d;   // (a NaN lands in d2: the pixel stays open)
                }
            }
            const float p1 = __shfl_xor(d1, 32, 64), p2 = __shfl_xor(d2, 32, 64);
            const int pa = __shfl_xor(arg, 32, 64);
            float best, second;
            int narg;
            if (p1 < d1) { best = p1; narg = pa; second = d1; if (!(p2 >= second)) second = p2; }
            else { best = d1; narg = arg; second = p1; if (!(d2 >= second)) second = d2; }
            const bool sure = K == 1 || (second - best > e_x * xxj + e_c * ccmax);   // false for near-ties, exact ties and NaN
            lab[j] = sure ? narg : -1;
            open[j] = !sure;
        }
        if (half == 0) {
            int64_t* out = labels + (int64_t)b * HW + pix0 + 4 * l31;
            typedef long long km_i64x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<km_i64x2*>(out) = km_i64x2{lab[0], lab[1]};
            *reinterpret_cast<km_i64x2*>(out + 2) = km_i64x2{lab[2], lab[3]};
            if (open_list) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (open[j]) open_list[atomicAdd(open_count, 1)] = b * HW + pix0 + 4 * l31 + j;   // (order irrelevant: one pixel per entry)
            }
        }
    }
}

// Exact-order distances of the LISTED pixels (those kmeans_fast_kernel left open), 32 lanes per pixel: in the documented order
// a pixel's sum is 8 x 4 independent running sums p[l][part] over the rows r (channel 32 r + 8 part + l), combined as
// ((p0 + p1) + p2) + p3 per l and then over l = 0..7 in order -- so lane 8 part + l keeps one running sum per centre (C / 32
// terms each, the same three roundings per term) and the combination is 11 shuffled adds per centre.  One lane per pixel, as
// the kernel above would do it, is a serial chain of 3 K C operations: 60 us for a single open pixel at 512 channels.
// C % 32 == 0, no second cascade level (C <= 512): the shapes the fast pass takes; same adds in the same order as above.
template <int KMAX>
__global__ __launch_bounds__(256) void kmeans_refine_kernel(int64_t* __restrict__ labels, const float* __restrict__ x,
                                                            const float* __restrict__ centres, int C, int HW, int K,
                                                            const int* __restrict__ open_count, const int* __restrict__ open_list) {
#pragma clang fp contract(off)
    // Latency, not work, is what this pass costs (a few hundred to a few thousand pixels per layer): a workgroup with pixels to
    // visit copies the centres to LDS once (16-byte loads, all in flight together), and a pixel's C / 32 activation loads are all
    // requested before the first subtraction -- two memory round trips per trip instead of one per row of 25 dependent loads
    // (measured 31-113 us per layer before, with at most one trip per workgroup).
    extern __shared__ __attribute__((aligned(16))) float cl[];   // [K][C]
    const int n_open = *open_count;
    if ((int)blockIdx.x * 8 >= n_open) return;
    for (int e = threadIdx.x * 4; e < K * C; e += 1024) *reinterpret_cast<float4*>(cl + e) = *reinterpret_cast<const float4*>(centres + e);
    __syncthreads();
    constexpr int RMAX = 16;   // rows of 32 channels held in registers (C <= 512: the shapes the first pass takes)
    const int rows = C >> 5;
    for (int i = blockIdx.x * 8 + (threadIdx.x >> 5); i < n_open; i += gridDim.x * 8) {   // 8 pixels per workgroup and trip
    const int q = open_list[i];
    const int b = q / HW, pix = q - b * HW;
    const int ln = threadIdx.x & 31;                      // = 8 part + l: channel 32 r + ln in row r
    const float* xp = x + (int64_t)b * C * HW + pix;
    float xr[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) xr[r] = r < rows ? xp[(int64_t)((r << 5) + ln) * HW] : 0.f;
    float acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        if (r < rows) {
            const int c = (r << 5) + ln;
            const float xv = xr[r];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    const float diff = xv - cl[k * C + c];
                    acc[k] = acc[k] + diff * diff;
                }
            }
        }
    }
    float best = 0.f;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            // lane total of position l (in lanes 0..7 of the pixel's 32): ((p0 + p1) + p2) + p3
            float t = acc[k];
            t = t + __shfl(acc[k], (threadIdx.x & 32) + 8 + (ln & 7), 64);
            t = t + __shfl(acc[k], (threadIdx.x & 32) + 16 + (ln & 7), 64);
            t = t + __shfl(acc[k], (threadIdx.x & 32) + 24 + (ln & 7), 64);
            float fin = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) fin = fin + __shfl(t, (threadIdx.x & 32) + l, 64);
            if (k == 0 || fin < best) { best = fin; arg = k; }
        }
    }
    if (ln == 0) labels[q] = arg;
    }
}

// Fewer than 8 channels: ATen sums the row with its scalar path (SumKernel.cpp scalar_inner_sum -> row_sum): four
// interleaved partial sums p[j] = 0 + term_j (j < 4 when C >= 4), the remaining terms join p[0] in order, result
// ((p0 + p1) + p2) + p3 (oracle/kmeans_ref.py::_row_sum).  One lane per pixel, centres straight from global memory.
__global__ __launch_bounds__(256) void kmeans_small_c_kernel(int64_t* __restrict__ labels, const float* __restrict__ x,
                                                             const float* __restrict__ centres, int C, int HW, int K, int64_t total) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / HW, pix = i - b * HW;
    float xv[7];
    for (int c = 0; c < C; ++c) xv[c] = x[(b * C + c) * HW + pix];
    const int lead = C >= 4 ? 4 : 0;
    float best = 0.f;
    int arg = 0;
    for (int k = 0; k < K; ++k) {
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < lead; ++c) { const float d = xv[c] - centres[(int64_t)k * C + c]; p[c] = p[c] + d * d; }
        for (int c = lead; c < C; ++c) { const float d = xv[c] - centres[(int64_t)k * C + c]; p[0] = p[0] + d * d; }
        const float dist = ((p[0] + p[1]) + p[2]) + p[3];
        if (k == 0 || dist < best) { best = dist; arg = k; }
    }
    labels[i] = arg;
}

__global__ __launch_bounds__(256) void make_image_kernel(uint8_t* __restrict__ out, const float* __restrict__ x,
                                                         int C, int HW, int64_t total) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one lane = one pixel, writes C bytes (NHWC)
    if (i >= total) return;
    const int64_t b = i / HW, p = i - b * HW;
    for (int c = 0; c < C; ++c) {
        float v = x[(b * C + c) * HW + p];
        v = fminf(fmaxf(v, -1.f), 1.f);  // clamp(min=-1, max=1)
        v = v + 1.f;                     // .add(1)   each step rounded to fp32 on its own, in the oracle's order
        v = v / 2.f;                     // .div(2)
        v = v * 255.f;                   // .mul(255)
        out[i * C + c] = (uint8_t)v;     // truncating cast, as tensor.type(torch.uint8)
    }
}

// RGB fast path: four consecutive pixels per lane -- three float4 loads (one per channel plane), twelve bytes as three
// 4-byte stores (the general kernel writes single bytes at stride 3: 0.3 TB/s).  Same per-element arithmetic.
__global__ __launch_bounds__(256) void make_image_rgb4_kernel(uint32_t* __restrict__ out, const float* __restrict__ x, int HW,
                                                              int64_t quads) {
#pragma clang fp contract(off)
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // pixel quad: pixels 4q .. 4q+3 of the flattened [B][HW]
    if (q >= quads) return;
    const int64_t i = 4 * q, b = i / HW, p = i - b * HW;
    float4 ch[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) ch[c] = *reinterpret_cast<const float4*>(x + (b * 3 + c) * HW + p);
    uint8_t by[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float in[4] = {ch[c].x, ch[c].y, ch[c].z, ch[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = fminf(fmaxf(in[e], -1.f), 1.f);
            v = v + 1.f;
            v = v / 2.f;
            v = v * 255.f;
            by[e * 3 + c] = (uint8_t)v;
        }
    }
#pragma unroll
    for (int w = 0; w < 3; ++w)
        out[3 * q + w] = (uint32_t)by[4 * w] | ((uint32_t)by[4 * w + 1] << 8) | ((uint32_t)by[4 * w + 2] << 16) | ((uint32_t)by[4 * w + 3] << 24);
}

constexpr int KM_LDS_MAX = 160 * 1024 - 256;   // dynamic LDS of the k-means kernels (the refine mode's block-wide OR keeps a static word)

template <int KMAX, int VEC, bool CASCADE>
int launch_kmeans(int64_t* labels, const float* x, const float* centres, int batch, int C, int HW, int K, hipStream_t st, int refine = 0,
                  const int* open_count = nullptr, const int* open_list = nullptr) {
    const int groups = sis_cdiv(HW, 256 * VEC);
    const size_t lds = (size_t)C * KMAX * sizeof(float);
    SIS_REQUIRE(lds <= KM_LDS_MAX, "sis_kmeans_assign: %d channels x %d centres do not fit in LDS", C, KMAX);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kmeans_assign_kernel<KMAX, VEC, CASCADE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, KM_LDS_MAX);
        if (e != hipSuccess) return sis_fail("sis_kmeans_assign: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int blocks = refine == 3 ? sis_cdiv((int64_t)batch * HW, 256) : batch * groups;
    hipLaunchKernelGGL((kmeans_assign_kernel<KMAX, VEC, CASCADE>), dim3(blocks), dim3(256), lds, st, labels, x, centres, C,
                       HW, K, groups, refine, open_count, open_list);
    SIS_CHECK_LAUNCH("kmeans_assign_kernel");
    return 0;
}

template <int KMAX>
int launch_kmeans_fast(int64_t* labels, const float* x, const float* centres, int batch, int C, int HW, int K, float thr, int* open_count,
                       int* open_list, hipStream_t st) {
    const int groups = sis_cdiv(HW, 512);
    const size_t lds = (size_t)C * KMAX * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kmeans_fast_kernel<KMAX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, KM_LDS_MAX);
        if (e != hipSuccess) return sis_fail("sis_kmeans_assign: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((kmeans_fast_kernel<KMAX>), dim3(batch * groups), dim3(256), lds, st, labels, x, centres, C, HW, K, groups, thr,
                       open_count, open_list);
    SIS_CHECK_LAUNCH("kmeans_fast_kernel");
    sis_kernel_name = "kmeans_fast_kernel";
    return 0;
}

}  // namespace

extern "C" int64_t sis_kmeans_workspace_ints(int batch, int hw) { return (int64_t)batch * hw + 1; }

static int kmeans_assign_impl(int64_t* labels, const float* x, const float* centres, int batch, int channels, int hw, int n_centres,
                              int* workspace, int64_t workspace_ints, void* stream);

extern "C" int sis_kmeans_assign(int64_t* labels, const float* x, const float* centres, int batch, int channels,
                                 int hw, int n_centres, void* stream) {
    return kmeans_assign_impl(labels, x, centres, batch, channels, hw, n_centres, nullptr, 0, stream);
}

extern "C" int sis_kmeans_assign_ws(int64_t* labels, const float* x, const float* centres, int batch, int channels, int hw,
                                    int n_centres, int* workspace, int64_t workspace_ints, void* stream) {
    return kmeans_assign_impl(labels, x, centres, batch, channels, hw, n_centres, workspace, workspace_ints, stream);
}

static int kmeans_assign_impl(int64_t* labels, const float* x, const float* centres, int batch, int channels, int hw, int n_centres,
                              int* workspace, int64_t workspace_ints, void* stream) {
    if (batch <= 0 || hw <= 0) return 0;
    SIS_REQUIRE(labels && x && centres, "sis_kmeans_assign: null pointer");
    SIS_REQUIRE(n_centres >= 1 && n_centres <= 64, "sis_kmeans_assign: %d centres outside 1..64", n_centres);
    SIS_REQUIRE(channels >= 1, "sis_kmeans_assign: no channels");
    hipStream_t st = (hipStream_t)stream;
    if (channels < 8) {
        const int64_t total = (int64_t)batch * hw;
        hipLaunchKernelGGL(kmeans_small_c_kernel, dim3(sis_cdiv(total, 256)), dim3(256), 0, st, labels, x, centres, channels, hw, n_centres, total);
        SIS_CHECK_LAUNCH("kmeans_small_c_kernel");
        return 0;
    }
    SIS_REQUIRE(channels < 8192, "sis_kmeans_assign: %d channels (the documented summation order covers C < 8192)", channels);
    const bool vec = hw % 2 == 0 && (((uintptr_t)x) & 7) == 0;
    const bool cascade = (channels >> 5) > 16;  // more than 16 rows of 4 vectors: the second-level sums carry a rounding
    // two passes where the fast kernel applies: it decides every pixel whose argmin cannot depend on the association of the adds,
    // the exact-order kernel below then only visits the workgroups that still hold a -1 (SIS_KMEANS_FAST=0: exact order only)
    const char* fast_env = getenv("SIS_KMEANS_FAST");
    int refine = 0;
    if (!(fast_env && fast_env[0] == '0') && vec && !cascade && channels % 16 == 0 && n_centres <= 32 &&
        (size_t)channels * 32 * sizeof(float) <= (size_t)KM_LDS_MAX) {
        const float u = 5.9604645e-8f;   // 2^-24
        const float thr = 3.f * ((16.f + channels / 16.f + 3.f) + (channels / 32.f + 32.f)) * u;   // 2 (b_fast + b_exact), x 1.5
        // with a workspace ([0]: counter, [1..]: pixel ids) the open pixels are listed and revisited one per lane
        const bool listed = workspace && workspace_ints >= (int64_t)batch * hw + 1 && (int64_t)batch * hw < ((int64_t)1 << 31);
        int* count = listed ? workspace : nullptr;
        int* list = listed ? workspace + 1 : nullptr;
        if (listed && hipMemsetAsync(count, 0, sizeof(int), st) != hipSuccess) return sis_fail("sis_kmeans_assign: cannot clear the counter");
        int rc;
        // matrix-core first pass where it applies (SIS_KMEANS_MFMA=0: the VALU first pass); it needs the list for its open pixels
        const char* mfma_env = getenv("SIS_KMEANS_MFMA");   // (read per call, like SIS_KMEANS_FAST)
        const bool mfma_on = !(mfma_env && mfma_env[0] == '0');
        if (mfma_on && listed && hw % 128 == 0 && channels % 64 == 0 && (((uintptr_t)x) & 15) == 0 &&
            (size_t)(channels + 9) * 32 * sizeof(float) <= (size_t)KM_LDS_MAX) {
            const float b_fast = 1.01f * (channels + 3.f), b_exact = 2.f * (channels / 32.f + 32.f);
            const float e_x = 1.5f * 2.f * (b_fast + b_exact) * u, e_c = 1.5f * 2.f * (2.f * b_fast + b_exact) * u;
            static bool mfma_attr = false;
            if (!mfma_attr) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&kmeans_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        KM_LDS_MAX) != hipSuccess)
                    return sis_fail("sis_kmeans_assign: cannot raise the LDS limit");
                mfma_attr = true;
            }
            const int64_t tiles = (int64_t)batch * (hw / 128);   // wave tiles of 128 pixels; workgroups persistent over them
            SIS_REQUIRE(tiles < ((int64_t)1 << 31), "sis_kmeans_assign: too many pixels");
            // (SIS_KMEANS_MAX_WG: fewer persistent workgroups leave compute units to whatever runs beside the pass; read per call)
            const char* wg_env = getenv("SIS_KMEANS_MAX_WG");
            const int max_wg = wg_env && atoi(wg_env) > 0 ? atoi(wg_env) : 512;
            const int blocks = (int)std::min<int64_t>(sis_cdiv(tiles, 4), max_wg);
            hipLaunchKernelGGL(kmeans_mfma_kernel, dim3(blocks), dim3(256), (size_t)(channels + 9) * 32 * sizeof(float), st, labels, x,
                               centres, channels, hw, n_centres, (int)tiles, e_x, e_c, count, list);
            SIS_CHECK_LAUNCH("kmeans_mfma_kernel");
            sis_kernel_name = "kmeans_mfma_kernel";
            rc = 0;
        } else
        if (n_centres <= 8) rc = launch_kmeans_fast<8>(labels, x, centres, batch, channels, hw, n_centres, thr, count, list, st);
        else if (n_centres <= 16) rc = launch_kmeans_fast<16>(labels, x, centres, batch, channels, hw, n_centres, thr, count, list, st);
        else if (n_centres <= 24) rc = launch_kmeans_fast<24>(labels, x, centres, batch, channels, hw, n_centres, thr, count, list, st);
        else rc = launch_kmeans_fast<32>(labels, x, centres, batch, channels, hw, n_centres, thr, count, list, st);
        if (rc) return rc;
        refine = listed ? 3 : 1;
        if (listed && channels % 32 == 0 && (((uintptr_t)centres) & 15) == 0) {   // 32 lanes per open pixel, centres in LDS
            const int blocks = (int)std::min<int64_t>(sis_cdiv((int64_t)batch * hw, 8), 1024);   // (the list is short: 262 144 empty workgroups took 170 us)
            const size_t cl_bytes = (size_t)n_centres * channels * sizeof(float);   // <= 32 x 512 floats
#define KM_LIST(KM)                                                                                                              \
    if (n_centres <= KM) {                                                                                                       \
        static bool attr = false;                                                                                                \
        if (!attr) {                                                                                                             \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&kmeans_refine_kernel<KM>),                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, KM_LDS_MAX) != hipSuccess)                       \
                return sis_fail("sis_kmeans_assign: cannot raise the LDS limit");                                                \
            attr = true;                                                                                                         \
        }                                                                                                                        \
        hipLaunchKernelGGL(kmeans_refine_kernel<KM>, dim3(blocks), dim3(256), cl_bytes, st, labels, x, centres, channels, hw,    \
                           n_centres, (const int*)count, (const int*)list);                                                      \
        SIS_CHECK_LAUNCH("kmeans_refine_kernel");                                                                                \
        return 0;                                                                                                                \
    }
            KM_LIST(8) KM_LIST(16) KM_LIST(24) KM_LIST(32)
#undef KM_LIST
        }
        if (listed) {   // one lane per open pixel
#define KM_LIST(KM) if (n_centres <= KM) return launch_kmeans<KM, 1, false>(labels, x, centres, batch, channels, hw, n_centres, st, 3, count, list);
            KM_LIST(8) KM_LIST(16) KM_LIST(24) KM_LIST(32)
#undef KM_LIST
        }
    }
    // accumulators for the centre count rounded up to a multiple of 8 (the reference config's 24 centres: no padded work)
#define KM_CASE(KM)                                                                                                      \
    if (n_centres <= KM) {                                                                                               \
        if (cascade) return launch_kmeans<KM, 1, true>(labels, x, centres, batch, channels, hw, n_centres, st);           \
        return vec ? launch_kmeans<KM, 2, false>(labels, x, centres, batch, channels, hw, n_centres, st, refine)          \
                   : launch_kmeans<KM, 1, false>(labels, x, centres, batch, channels, hw, n_centres, st);                 \
    }
    KM_CASE(8) KM_CASE(16) KM_CASE(24) KM_CASE(32) KM_CASE(64)
#undef KM_CASE
    return sis_fail("sis_kmeans_assign: unreachable");
}

extern "C" int sis_make_image_u8(uint8_t* out, const float* x, int batch, int channels, int hw, void* stream) {
    const int64_t total = (int64_t)batch * hw;
    if (total <= 0) return 0;
    SIS_REQUIRE(out && x, "sis_make_image_u8: null pointer");
    if (channels == 3 && hw % 4 == 0 && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)out) & 3) == 0)) {
        hipLaunchKernelGGL(make_image_rgb4_kernel, dim3(sis_cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, (uint32_t*)out, x, hw,
                           total / 4);
    } else {
        hipLaunchKernelGGL(make_image_kernel, dim3(sis_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, x, channels, hw,
                           total);
    }
    SIS_CHECK_LAUNCH("make_image_kernel");
    return 0;
}
