// Does an in-flight LDS-DMA (buffer_load ... lds) of a wave delay that wave's LATER ds_read of an unrelated LDS address?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma_order.hip -o synthesis-in-style_amd/lib/lds_dma_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
template <int MODE>  // 0: ds_read only; 1: DMA (cold lines) then ds_read; 2: DMA then s_waitcnt vmcnt(0) (DMA latency itself)
__global__ void k(const float* g, unsigned* ticks, float* sink, int stride_floats) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, 0x7FFFFFFF, 0x00020000);
    float acc = 0.f;
    unsigned total = 0;
    for (int it = 0; it < 64; ++it) {
        const unsigned voff = (unsigned)(threadIdx.x & 63) * 16u;
        const unsigned soff = (unsigned)((blockIdx.x * 64 + it) * stride_floats) * 4u;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
        const unsigned t0 = (unsigned)__builtin_readcyclecounter();
        if (MODE >= 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 1024), 16, voff, soff, 0, 0);
        float v = 0.f;
        if (MODE <= 1) {
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(threadIdx.x & 63) * 4u) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned t1 = (unsigned)__builtin_readcyclecounter();
        total += t1 - t0;
        acc += v;
    }
    asm volatile("s_waitcnt vmcnt(0)");
    sink[blockIdx.x * 64 + threadIdx.x] = acc + lds[1024 + threadIdx.x];
    if (threadIdx.x == 0) ticks[blockIdx.x] = total / 64;
}
template <int MODE>
void run(const char* name, const float* g, unsigned* dt, float* sink, int stride) {
    const int blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, g, dt, sink, stride);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks);
    hipMemcpy(h.data(), dt, blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-52s median %u ticks\n", name, h[blocks / 2]);
}
int main() {
    float *g, *sink; unsigned* dt;
    const size_t n = (size_t)256 * 64 * 4096;  // 256 MB of floats: every DMA touches cold lines
    hipMalloc(&g, n * 4); hipMalloc(&sink, 256 * 64 * 4); hipMalloc(&dt, 256 * 4);
    hipMemset(g, 0, n * 4);
    run<0>("ds_read + wait, no DMA in flight", g, dt, sink, 4096);
    run<1>("cold LDS-DMA issued, then ds_read + wait", g, dt, sink, 4096);
    run<2>("cold LDS-DMA issued, then s_waitcnt vmcnt(0)", g, dt, sink, 4096);
    run<1>("same-line (warm) LDS-DMA, then ds_read + wait", g, dt, sink, 0);
    run<2>("same-line (warm) LDS-DMA, then s_waitcnt vmcnt(0)", g, dt, sink, 0);
    return 0;
}
