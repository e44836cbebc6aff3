/*
 * ORACLE (test infrastructure, not product).
 *
 * Plain-C, index-level restatement of the reference's two CUDA kernels, written
 * from their indexing rules (not a copy of the code):
 *
 *   upfirdn2d        /root/reference/stylegan_code_finder/networks/stylegan2/op/upfirdn2d_kernel.cu
 *                    :18-26  floor_div
 *                    :71-81  taps are stored flipped in both axes
 *                    :112-129 per-output polyphase walk (mid -> in, first tap, stride = up)
 *                    :167-168 output size
 *   fused_bias_act   .../op/fused_bias_act_kernel.cu:25-47 (element formula),
 *                    :62-71 (bias index = (i / step_b) % size_b)
 *
 * One thread of the CUDA grid = one iteration of the loops below; accumulation
 * is in the tensor's own type, taps visited y-outer / x-inner as the kernel does.
 * Built by oracle/Makefile into oracle/_build/liboracle_ops.so and loaded with
 * ctypes (oracle/c_ops.py).  Used only to cross-check oracle/ops_ref.py and as
 * a second checker for the HIP ops.
 */
#include <stddef.h>

static int floor_div_i(int a, int b) {
    int c = a / b;
    if (c * b > a) c--;
    return c;
}

int oracle_upfirdn2d_out_size(int in, int up, int down, int pad0, int pad1, int k) {
    return (in * up + pad0 + pad1 - k + down) / down;
}

#define DEFINE_UPFIRDN2D(NAME, T)                                                            \
void NAME(T* out, const T* in, const T* kernel, int major, int in_h, int in_w, int minor,    \
          int kh, int kw, int up_x, int up_y, int down_x, int down_y,                         \
          int pad_x0, int pad_x1, int pad_y0, int pad_y1) {                                   \
    const int out_h = oracle_upfirdn2d_out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh);      \
    const int out_w = oracle_upfirdn2d_out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw);      \
    for (int mj = 0; mj < major; ++mj)                                                        \
    for (int oy = 0; oy < out_h; ++oy)                                                        \
    for (int ox = 0; ox < out_w; ++ox)                                                        \
    for (int mn = 0; mn < minor; ++mn) {                                                      \
        /* position of this output on the zero-inserted, padded grid */                       \
        const int mid_x = ox * down_x + up_x - 1 - pad_x0;                                    \
        const int mid_y = oy * down_y + up_y - 1 - pad_y0;                                    \
        const int ix0 = floor_div_i(mid_x, up_x);                                             \
        const int iy0 = floor_div_i(mid_y, up_y);                                             \
        /* first non-zero-insert tap for this phase (index into the FLIPPED taps) */          \
        const int kx0 = (ix0 + 1) * up_x - mid_x - 1;                                         \
        const int ky0 = (iy0 + 1) * up_y - mid_y - 1;                                         \
        T v = (T)0;                                                                           \
        for (int ty = 0; ky0 + ty * up_y < kh; ++ty)                                          \
        for (int tx = 0; kx0 + tx * up_x < kw; ++tx) {                                        \
            const int iy = iy0 + ty, ix = ix0 + tx;                                           \
            const int fy = ky0 + ty * up_y, fx = kx0 + tx * up_x;                             \
            T s = (T)0;                                                                       \
            if (ix >= 0 && iy >= 0 && ix < in_w && iy < in_h)                                 \
                s = in[(((size_t)mj * in_h + iy) * in_w + ix) * minor + mn];                  \
            v += s * kernel[(kh - 1 - fy) * kw + (kw - 1 - fx)];                              \
        }                                                                                     \
        out[(((size_t)mj * out_h + oy) * out_w + ox) * minor + mn] = v;                       \
    }                                                                                         \
}

DEFINE_UPFIRDN2D(oracle_upfirdn2d_f32, float)
DEFINE_UPFIRDN2D(oracle_upfirdn2d_f64, double)

#define DEFINE_FUSED_BIAS_ACT(NAME, T)                                                        \
void NAME(T* out, const T* x, const T* b, const T* ref, long n, int step_b, int size_b,       \
          int use_bias, int use_ref, int act, int grad, T alpha, T scale) {                   \
    for (long i = 0; i < n; ++i) {                                                            \
        T v = x[i];                                                                           \
        if (use_bias) v += b[(i / step_b) % size_b];                                          \
        const T r = use_ref ? ref[i] : (T)0;                                                  \
        T y;                                                                                  \
        switch (act * 10 + grad) {                                                            \
            case 12: case 32: y = (T)0; break;                                                \
            case 30: y = (v > (T)0) ? v : v * alpha; break;                                   \
            case 31: y = (r > (T)0) ? v : v * alpha; break;                                   \
            default: y = v; break; /* 10, 11 and anything unknown: linear */                  \
        }                                                                                     \
        out[i] = y * scale;                                                                   \
    }                                                                                         \
}

DEFINE_FUSED_BIAS_ACT(oracle_fused_bias_act_f32, float)
DEFINE_FUSED_BIAS_ACT(oracle_fused_bias_act_f64, double)
