// What does `buffer_load_dwordx4 ... lds` leave in LDS for lanes whose offset is out of range?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/buffer_lds_oob.hip -o synthesis-in-style_amd/lib/buffer_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* g, float* out, unsigned bytes, unsigned soff) {
    __shared__ __attribute__((aligned(16))) float lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -7.f;  // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, bytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x % 3 == 1) voff = 0x80000000u;            // far out of range
    if (threadIdx.x == 62) voff = bytes - 8;                 // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    float *g, *o;
    const int n = 4096;
    hipMalloc(&g, n * 4); hipMalloc(&o, 512 * 4);
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, 2048u, 1024u);  // records: first 512 floats; soffset 256 floats
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
    for (int l = 60; l < 64; ++l) printf("lane %d: %g %g %g %g\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
    printf("second copy lane 1: %g, lane 2: %g\n", r[256 + 4], r[256 + 8]);
    return 0;
}
