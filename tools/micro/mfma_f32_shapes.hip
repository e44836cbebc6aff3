// Which fp32 MFMA shape sustains more FLOP/s on MI355X under load (random operands, registers only)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
// Each wave keeps 64 accumulator registers busy: 4 chains of 32x32x2 or 16 chains of 16x16x4; one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void loop_kernel(float* out, const float* in, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xffff]; b[i] = in[(t * 16 + 8 + i) & 0xffff]; }
    if (SHAPE == 32) {
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + c) & 7], b[k], acc[c], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[c][i];
        out[t] = s;
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + c) & 7], b[(k + (c >> 2)) & 7], acc[c], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) s += acc[c][i];
        out[t] = s;
    }
}

int main() {
    const int blocks = 256 * 2, iters = 20000;
    float *in, *out;
    hipMalloc(&in, 65536 * 4);
    hipMalloc(&out, blocks * 512 * 4);
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
            for (int shape : {32, 16}) {
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(threads), 0, 0, out, in, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                // per wave per iteration: 32 MFMAs x 4096 flop (32x32x2) or 64 MFMAs x 2048 flop (16x16x4) = 131072 flop
                const double flop = (double)blocks * (threads / 64) * iters * 131072.0;
                printf("threads/block %d  shape %dx%d  %.2f ms  %.1f TF/s\n", threads, shape, shape, ms, flop / ms / 1e9);
            }
        }
    }
    return 0;
}
