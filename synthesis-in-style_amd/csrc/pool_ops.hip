// Max pooling of the two segmentation backbones, forward and backward, NCHW, f32 / f16 / bf16:
//   EMANet's ResNet stem  nn.MaxPool2d(3, 2, 1)            (reference networks/ema_net/network.py:66 via :132 here),
//   TransUNet's root      F.max_pool2d(x, 3, 2, padding 0) (vit_seg_modeling_resnet_skip.py:146).
// Same results as ATen's NCHW kernels bit for bit, ties included (post-ReLU maps are full of equal zeros): the forward
// scans the window row-major and keeps the FIRST maximum (strictly-greater test; a NaN always wins), the backward is a
// gather -- every input pixel adds the gradients of the (at most 2 x 2 for kernel 3 / stride 2) windows whose argmax it
// is -- so it is deterministic and needs no atomics.  The argmax is kept as one byte per output (kh * kernel + kw).
#include "sis_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void max_pool_fwd_kernel(T* __restrict__ out, unsigned char* __restrict__ arg,
                                                           const T* __restrict__ x, int h, int w, int oh, int ow, int ks,
                                                           int stride, int pad, int64_t total) {
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
        const int ox = (int)(i % ow);
        const int64_t row = i / ow;
        const int oy = (int)(row % oh);
        const int64_t plane = row / oh;
        const T* xp = x + plane * h * w;
        const int y0 = oy * stride - pad, x0 = ox * stride - pad;
        float best = -INFINITY;
        int best_k = -1;
        for (int kh = 0; kh < ks; ++kh) {
            const int y = y0 + kh;
            if (y < 0 || y >= h) continue;
            for (int kw = 0; kw < ks; ++kw) {
                const int xx = x0 + kw;
                if (xx < 0 || xx >= w) continue;
                const float v = sis_ld(xp, (int64_t)y * w + xx);
                if (v > best || v != v || best_k < 0) { best = v; best_k = kh * ks + kw; }
            }
        }
        sis_st(out, i, best);
        arg[i] = (unsigned char)best_k;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void max_pool_bwd_kernel(T* __restrict__ dx, const T* __restrict__ dy,
                                                           const unsigned char* __restrict__ arg, int h, int w, int oh, int ow,
                                                           int ks, int stride, int pad, int64_t total) {
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += step) {
        const int xx = (int)(i % w);
        const int64_t row = i / w;
        const int y = (int)(row % h);
        const int64_t plane = row / h;
        // windows that contain (y, xx): oy * stride - pad <= y < oy * stride - pad + ks
        const int oy_lo = max(0, (y + pad - ks + stride) / stride), oy_hi = min(oh - 1, (y + pad) / stride);
        const int ox_lo = max(0, (xx + pad - ks + stride) / stride), ox_hi = min(ow - 1, (xx + pad) / stride);
        float g = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int64_t o = (plane * oh + oy) * ow + ox;
                const int k = (y - (oy * stride - pad)) * ks + (xx - (ox * stride - pad));
                if (arg[o] == k) g += sis_ld(dy, o);
            }
        sis_st(dx, i, g);
    }
}

// The same gather, four consecutive pixels of a row per lane (w % 4 == 0): the <= 2 x 3 windows that cover them are looked
// at once (one argmax byte and one gradient each) and their gradients dealt to the four pixels in the order the one-pixel kernel
// adds them (window rows, then window columns: bitwise the same sums); one 8 / 16-byte store.  The one-pixel form spent ~40
// instructions and up to 8 loads per element: 0.7 TB/s on the 128 x 128 maps of both segmentation networks.
template <typename T>
__global__ __launch_bounds__(256) void max_pool_bwd4_kernel(T* __restrict__ dx, const T* __restrict__ dy,
                                                            const unsigned char* __restrict__ arg, int h, int w, int oh, int ow,
                                                            int ks, int stride, int pad, int64_t total4) {
    const int w4 = w >> 2;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += step) {
        const int x0 = (int)(i % w4) * 4;
        const int64_t row = i / w4;
        const int y = (int)(row % h);
        const int64_t plane = row / h;
        const int oy_lo = max(0, (y + pad - ks + stride) / stride), oy_hi = min(oh - 1, (y + pad) / stride);
        const int ox_lo = max(0, (x0 + pad - ks + stride) / stride), ox_hi = min(ow - 1, (x0 + 3 + pad) / stride);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const int ky = y - (oy * stride - pad);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int64_t o = (plane * oh + oy) * ow + ox;
                const int k = arg[o];
                const int kx = k - ky * ks;                 // the window's maximum sits in this row iff 0 <= kx < ks
                const int e = ox * stride - pad + kx - x0;  // ... at pixel x0 + e
                if (kx >= 0 && kx < ks && e >= 0 && e < 4) {
                    const float v = sis_ld(dy, o);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q == e) g[q] += v;
                }
            }
        }
        T t[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sis_st(t, q, g[q]);
        T* dst = dx + (plane * h + y) * (int64_t)w + x0;
        if constexpr (sizeof(T) == 4) {
            float4 v4;
            __builtin_memcpy(&v4, t, 16);
            *reinterpret_cast<float4*>(dst) = v4;
        } else {
            uint2 v2;
            __builtin_memcpy(&v2, t, 8);
            *reinterpret_cast<uint2*>(dst) = v2;
        }
    }
}

template <typename T>
int launch_pool(void* out, void* arg, const void* x, int64_t planes, int h, int w, int oh, int ow, int ks, int stride, int pad,
                int backward, hipStream_t st) {
    const int64_t total = planes * (backward ? (int64_t)h * w : (int64_t)oh * ow);
    const int blocks = (int)(sis_cdiv(total, 256) < 65536 ? sis_cdiv(total, 256) : 65536);
    if (backward && w % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int64_t total4 = total / 4;
        const int blocks4 = (int)(sis_cdiv(total4, 256) < 65536 ? sis_cdiv(total4, 256) : 65536);
        hipLaunchKernelGGL(max_pool_bwd4_kernel<T>, dim3(blocks4), dim3(256), 0, st, (T*)out, (const T*)x, (const unsigned char*)arg, h,
                           w, oh, ow, ks, stride, pad, total4);
    } else if (backward)
        hipLaunchKernelGGL(max_pool_bwd_kernel<T>, dim3(blocks), dim3(256), 0, st, (T*)out, (const T*)x, (const unsigned char*)arg, h,
                           w, oh, ow, ks, stride, pad, total);
    else
        hipLaunchKernelGGL(max_pool_fwd_kernel<T>, dim3(blocks), dim3(256), 0, st, (T*)out, (unsigned char*)arg, (const T*)x, h, w,
                           oh, ow, ks, stride, pad, total);
    SIS_CHECK_LAUNCH("sis_max_pool2d");
    return 0;
}

}  // namespace

extern "C" int sis_max_pool2d(void* out, unsigned char* argmax, const void* x, int dtype, int64_t planes, int h, int w,
                              int out_h, int out_w, int kernel, int stride, int padding, int backward, void* stream) {
    if (planes == 0) return 0;
    SIS_REQUIRE(out && argmax && x, "sis_max_pool2d: null pointer");
    SIS_REQUIRE(planes > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0, "sis_max_pool2d: non-positive size");
    SIS_REQUIRE(kernel >= 1 && kernel <= 15 && stride >= 1 && padding >= 0 && 2 * padding <= kernel,
                "sis_max_pool2d: kernel %d / stride %d / padding %d not supported", kernel, stride, padding);
    SIS_REQUIRE((out_h - 1) * stride - padding < h && (out_w - 1) * stride - padding < w,
                "sis_max_pool2d: output %d x %d has windows outside the %d x %d input", out_h, out_w, h, w);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case SIS_F32: return launch_pool<float>(out, argmax, x, planes, h, w, out_h, out_w, kernel, stride, padding, backward, st);
        case SIS_F16: return launch_pool<__half>(out, argmax, x, planes, h, w, out_h, out_w, kernel, stride, padding, backward, st);
        case SIS_BF16: return launch_pool<__hip_bfloat16>(out, argmax, x, planes, h, w, out_h, out_w, kernel, stride, padding, backward, st);
        default: return sis_fail("sis_max_pool2d: dtype code %d not supported (f32, f16, bf16)", dtype);
    }
}
