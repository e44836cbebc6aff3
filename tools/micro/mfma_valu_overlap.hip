// How much VALU / LDS work rides along with a stream of fp32 MFMAs on one SIMD of MI355X?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o gpurun_out/mfma_valu_overlap && gpurun_out/mfma_valu_overlap
// same<N>:  every wave runs  { v_mfma_f32_32x32x2_f32 ; N x v_fma_f32 }  (4 accumulator chains), 1 or 2 waves per SIMD
// split<N>: waves 0-3 run MFMAs only, waves 4-7 (their SIMD partners) run  N x v_fma_f32  per loop trip and no MFMA
// Reported: shader-clock ticks (s_memtime) per MFMA / per VALU instruction, median over workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VFMA(t) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(a), "v"(b))

template <int N>
__device__ __forceinline__ void valu_block(float (&t)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < N; ++i) VFMA(t[i & 7]);
}

template <int N, bool LDS>
__device__ __forceinline__ void side_block(float (&t)[8], float a, float b, const float* l) {
    if (LDS) {
#pragma unroll
        for (int i = 0; i < N; ++i) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)l), "n"(i * 256)); t[i & 7] = v; }
    } else {
        valu_block<N>(t, a, b);
    }
}

// MODE 0: same wave interleaves.  MODE 1: split roles (waves >= 4 do only the side work).
template <int N, int MODE, bool LDS>
__global__ __launch_bounds__(512) void k(unsigned* ticks, float* sink, const float* in, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, wave = tid >> 6;
    lds[tid] = in[tid];
    lds[tid + 512] = in[tid + 512];
    __syncthreads();
    float a = in[tid], b = in[tid + 64];
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float t[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const float* l = lds + (tid & 63);
    const bool side_only = MODE == 1 && wave >= 4;
    __syncthreads();
    const unsigned t0 = (unsigned)__builtin_readcyclecounter();
    if (!side_only) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                MFMA(acc[c]);
                if (MODE == 0) side_block<N, LDS>(t, a, b, l);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) side_block<N, LDS>(t, a, b, l);
        }
    }
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned t1 = (unsigned)__builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[c][i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += t[i];
    sink[blockIdx.x * blockDim.x + tid] = s;
    if ((tid & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int N, int MODE, bool LDS>
void run(const char* name, int threads, unsigned* dt, float* sink, const float* in) {
    const int blocks = 256, iters = 2000;
    hipMemset(dt, 0, blocks * 8 * 4);
    hipLaunchKernelGGL((k<N, MODE, LDS>), dim3(blocks), dim3(threads), 0, 0, dt, sink, in, iters);
    hipLaunchKernelGGL((k<N, MODE, LDS>), dim3(blocks), dim3(threads), 0, 0, dt, sink, in, iters);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 8);
    hipMemcpy(h.data(), dt, blocks * 8 * 4, hipMemcpyDeviceToHost);
    std::vector<double> lo, hi;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < threads / 64; ++w) (w < 4 ? lo : hi).push_back(h[b * 8 + w]);
    std::sort(lo.begin(), lo.end());
    std::sort(hi.begin(), hi.end());
    const double per = 4.0 * iters;
    printf("%-34s waves0-3: %7.1f ticks per trip", name, lo[lo.size() / 2] / per);
    if (!hi.empty()) printf("   waves4-7: %7.1f ticks per trip", hi[hi.size() / 2] / per);
    printf("\n");
}

int main() {
    unsigned* dt; float *sink, *in;
    hipMalloc(&dt, 256 * 8 * 4); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&in, 4096 * 4);
    std::vector<float> hin(4096);
    for (auto& v : hin) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, hin.data(), 4096 * 4, hipMemcpyHostToDevice);
    printf("trip = 1 MFMA (+ N side instructions in `same`), or N side instructions (side-only waves)\n");
    run<0, 0, false>("same N=0, 1 wave/SIMD", 256, dt, sink, in);
    run<4, 0, false>("same N=4 VALU, 1 wave/SIMD", 256, dt, sink, in);
    run<8, 0, false>("same N=8 VALU, 1 wave/SIMD", 256, dt, sink, in);
    run<12, 0, false>("same N=12 VALU, 1 wave/SIMD", 256, dt, sink, in);
    run<16, 0, false>("same N=16 VALU, 1 wave/SIMD", 256, dt, sink, in);
    run<0, 0, false>("same N=0, 2 waves/SIMD", 512, dt, sink, in);
    run<4, 0, false>("same N=4 VALU, 2 waves/SIMD", 512, dt, sink, in);
    run<8, 0, false>("same N=8 VALU, 2 waves/SIMD", 512, dt, sink, in);
    run<16, 0, false>("same N=16 VALU, 2 waves/SIMD", 512, dt, sink, in);
    run<24, 0, false>("same N=24 VALU, 2 waves/SIMD", 512, dt, sink, in);
    run<8, 1, false>("split N=8 VALU (side-only partner)", 512, dt, sink, in);
    run<16, 1, false>("split N=16 VALU (side-only partner)", 512, dt, sink, in);
    run<4, 0, true>("same N=4 ds_read, 1 wave/SIMD", 256, dt, sink, in);
    run<4, 0, true>("same N=4 ds_read, 2 waves/SIMD", 512, dt, sink, in);
    run<8, 1, true>("split N=8 ds_read (side-only)", 512, dt, sink, in);
    // side work alone for reference: MODE 1 with 512 threads measures waves 4-7 against MFMA partners; alone = no MFMA:
    return 0;
}
