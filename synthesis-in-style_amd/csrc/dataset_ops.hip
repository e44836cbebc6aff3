// "Next" rows of SURVEY.md §8(f): what sits immediately downstream of Generator.forward in the dataset loop.
//
//  * sis_kmeans_assign   nearest k-means centre per pixel of an activation map:
//                        labels[b,h,w] = argmin_k sum_c (x[b,c,h,w] - centre[k,c])^2
//                        (segmentation/gan_local_edit/factor_catalog.py:47-62 of the reference, which ships the
//                        activations to the CPU and materialises an N x K x C tensor there).  Direct
//                        (x - c)^2 form in fp32 -- not the |x|^2 - 2xc + |c|^2 expansion, whose cancellation would
//                        move near-tie labels -- ties go to the lowest index like torch.argmin.  Reads the
//                        activation exactly once: HBM 4*C bytes per pixel, VALU 3*K flops per pixel-channel.
//  * sis_make_image_u8   float image in [-1,1] (NCHW) -> uint8 NHWC, the conversion in front of the PNG writer
//                        (create_dataset_for_segmentation.py:135; third-party make_image: clamp, (x+1)/2*255,
//                        truncating cast -- rounding unpinned by the reference).
#include "sis_common.h"

namespace {

template <int KMAX, int VEC>
__global__ __launch_bounds__(256) void kmeans_assign_kernel(int64_t* __restrict__ labels, const float* __restrict__ x,
                                                            const float* __restrict__ centres, int C, int HW, int K,
                                                            int groups) {
    extern __shared__ __attribute__((aligned(16))) float cen[];  // [C][KMAX]
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    for (int e = threadIdx.x; e < C * KMAX; e += 256) {
        const int c = e / KMAX, k = e - c * KMAX;
        cen[e] = k < K ? centres[(int64_t)k * C + c] : 0.f;
    }
    __syncthreads();
    const int pix = (g * 256 + threadIdx.x) * VEC;
    if (pix >= HW) return;
    // pixel pairs as 2-wide vectors: the subtract and the fused multiply-add issue as packed fp32 instructions
    // (v_pk_add_f32 / v_pk_fma_f32: two pixels per lane and instruction), same per-element arithmetic as the scalar form
    constexpr int NP = VEC >= 2 ? VEC / 2 : 1;
    typedef float pkf __attribute__((ext_vector_type(VEC >= 2 ? 2 : 1)));
    pkf d[KMAX][NP];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int v = 0; v < NP; ++v) d[k][v] = (pkf)(0.f);
    const float* xb = x + (int64_t)b * C * HW + pix;
    for (int c = 0; c < C; ++c) {
        pkf xv[NP];
        if constexpr (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(xb + (int64_t)c * HW);
            xv[0][0] = t.x; xv[0][1] = t.y; xv[1][0] = t.z; xv[1][1] = t.w;
        } else if constexpr (VEC == 2) {
            const float2 t = *reinterpret_cast<const float2*>(xb + (int64_t)c * HW);
            xv[0][0] = t.x; xv[0][1] = t.y;
        } else {
            xv[0][0] = xb[(int64_t)c * HW];
        }
        const float* cc = cen + c * KMAX;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const pkf ck = (pkf)(cc[k]);
#pragma unroll
            for (int v = 0; v < NP; ++v) {
                const pkf diff = xv[v] - ck;
                d[k][v] = __builtin_elementwise_fma(diff, diff, d[k][v]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        float best = d[0][v / 2][v % 2];
        int arg = 0;
#pragma unroll
        for (int k = 1; k < KMAX; ++k)
            if (k < K && d[k][v / 2][v % 2] < best) { best = d[k][v / 2][v % 2]; arg = k; }
        labels[(int64_t)b * HW + pix + v] = arg;
    }
}

__global__ __launch_bounds__(256) void make_image_kernel(uint8_t* __restrict__ out, const float* __restrict__ x,
                                                         int C, int HW, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one lane = one pixel, writes C bytes (NHWC)
    if (i >= total) return;
    const int64_t b = i / HW, p = i - b * HW;
    for (int c = 0; c < C; ++c) {
        float v = x[(b * C + c) * HW + p];
        v = fminf(fmaxf(v, -1.f), 1.f);
        v = (v + 1.f) / 2.f * 255.f;
        out[i * C + c] = (uint8_t)v;  // truncating cast, as tensor.type(torch.uint8)
    }
}

template <int KMAX, int VEC>
int launch_kmeans(int64_t* labels, const float* x, const float* centres, int batch, int C, int HW, int K, hipStream_t st) {
    const int groups = sis_cdiv(HW, 256 * VEC);
    const size_t lds = (size_t)C * KMAX * sizeof(float);
    SIS_REQUIRE(lds <= 160 * 1024, "sis_kmeans_assign: %d channels x %d centres do not fit in LDS", C, KMAX);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kmeans_assign_kernel<KMAX, VEC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return sis_fail("sis_kmeans_assign: cannot raise the LDS limit: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((kmeans_assign_kernel<KMAX, VEC>), dim3(batch * groups), dim3(256), lds, st, labels, x, centres, C,
                       HW, K, groups);
    SIS_CHECK_LAUNCH("kmeans_assign_kernel");
    return 0;
}

}  // namespace

extern "C" int sis_kmeans_assign(int64_t* labels, const float* x, const float* centres, int batch, int channels,
                                 int hw, int n_centres, void* stream) {
    if (batch <= 0 || hw <= 0) return 0;
    SIS_REQUIRE(labels && x && centres, "sis_kmeans_assign: null pointer");
    SIS_REQUIRE(n_centres >= 1 && n_centres <= 64, "sis_kmeans_assign: %d centres outside 1..64", n_centres);
    SIS_REQUIRE(channels >= 1, "sis_kmeans_assign: no channels");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = hw % 4 == 0 && (((uintptr_t)x) & 15) == 0;
    // accumulators for the centre count rounded up to a multiple of 8 (the reference config's 24 centres: no padded work)
#define KM_CASE(KM)                                                                                                      \
    if (n_centres <= KM) return vec ? launch_kmeans<KM, 4>(labels, x, centres, batch, channels, hw, n_centres, st)        \
                                    : launch_kmeans<KM, 1>(labels, x, centres, batch, channels, hw, n_centres, st);
    KM_CASE(8) KM_CASE(16) KM_CASE(24) KM_CASE(32)
#undef KM_CASE
    return hw % 2 == 0 && (((uintptr_t)x) & 7) == 0 ? launch_kmeans<64, 2>(labels, x, centres, batch, channels, hw, n_centres, st)
                                                    : launch_kmeans<64, 1>(labels, x, centres, batch, channels, hw, n_centres, st);
}

extern "C" int sis_make_image_u8(uint8_t* out, const float* x, int batch, int channels, int hw, void* stream) {
    const int64_t total = (int64_t)batch * hw;
    if (total <= 0) return 0;
    SIS_REQUIRE(out && x, "sis_make_image_u8: null pointer");
    hipLaunchKernelGGL(make_image_kernel, dim3(sis_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, x, channels, hw,
                       total);
    SIS_CHECK_LAUNCH("make_image_kernel");
    return 0;
}
