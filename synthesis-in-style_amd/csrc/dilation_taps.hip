// The [4 Cout, 4 Cin] matrix of a 3x3 convolution whose dilation is half the image side (EMANet's last bottleneck: dilation
// 16, padding 16 on 32 x 32 maps; reference networks/ema_net/network.py:82-86,101-131 builds it as an ordinary dilated
// nn.Conv2d).  Output pixel (u d + p, v d + q), u, v in {0, 1}, only sees the 2 x 2 input pixels (u' d + p, v' d + q): tap
// (ky, kx) = (u' - u + 1, v' - v + 1); the other five taps fall into the zero padding.  The layer is therefore ONE dense
// [4 Cin] -> [4 Cout] map per (p, q) -- a pointwise convolution on the d x d grid with 4/9 of the 9-tap multiplies, run by
// csrc/conv1x1_f32.hip -- and this file gathers its matrix
//     taps[(u, v, co)][(u', v', ci)] = W[co][ci][u' - u + 1][v' - v + 1]
// and the weight gradient back (each tap collects the 1, 2 or 4 blocks it appears in, added in a fixed order).
#include "sis_common.h"

namespace {

__global__ __launch_bounds__(256) void half_dilation_taps_kernel(float* __restrict__ taps, const float* __restrict__ w, int cout, int cin) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = 16LL * cout * cin;
    if (i >= total) return;
    const int col = (int)(i % (4 * cin)), row = (int)(i / (4 * cin));
    const int ci = col % cin, vp = (col / cin) & 1, up = col / (2 * cin);
    const int co = row % cout, v = (row / cout) & 1, u = row / (2 * cout);
    taps[i] = w[(((long long)co * cin + ci) * 3 + (up - u + 1)) * 3 + (vp - v + 1)];
}

__global__ __launch_bounds__(256) void half_dilation_taps_bwd_kernel(float* __restrict__ dw, const float* __restrict__ dtaps, int cout, int cin) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = 9LL * cout * cin;
    if (i >= total) return;
    const int kx = (int)(i % 3), ky = (int)((i / 3) % 3);
    const int ci = (int)((i / 9) % cin), co = (int)(i / (9LL * cin));
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int up = u + ky - 1;
        if (up < 0 || up > 1) continue;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int vp = v + kx - 1;
            if (vp < 0 || vp > 1) continue;
            sum += dtaps[((long long)((u * 2 + v) * cout + co)) * (4 * cin) + (up * 2 + vp) * cin + ci];
        }
    }
    dw[i] = sum;
}

}  // namespace

extern "C" int sis_half_dilation_taps(float* taps, const float* weight, int cout, int cin, void* stream) {
    if (cout <= 0 || cin <= 0) return 0;
    SIS_REQUIRE(taps && weight, "sis_half_dilation_taps: null pointer");
    const long long total = 16LL * cout * cin;
    hipLaunchKernelGGL(half_dilation_taps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, taps, weight, cout, cin);
    SIS_CHECK_LAUNCH("half_dilation_taps_kernel");
    return 0;
}

extern "C" int sis_half_dilation_taps_bwd(float* dweight, const float* dtaps, int cout, int cin, void* stream) {
    if (cout <= 0 || cin <= 0) return 0;
    SIS_REQUIRE(dweight && dtaps, "sis_half_dilation_taps_bwd: null pointer");
    const long long total = 9LL * cout * cin;
    hipLaunchKernelGGL(half_dilation_taps_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dweight, dtaps, cout, cin);
    SIS_CHECK_LAUNCH("half_dilation_taps_bwd_kernel");
    return 0;
}
