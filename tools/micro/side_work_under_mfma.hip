// Which part of the Winograd input transform is slow while the SIMD partner streams MFMAs?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/side_work_under_mfma.hip -o synthesis-in-style_amd/lib/side_work_under_mfma
// Waves 0-3 (one per SIMD): MFMA stream [3 x ds_read2st64_b32, wait, 2 x v_mfma_f32_32x32x2_f32] (or idle).
// Waves 4-7 (their SIMD partners): `reps` side blocks of variant V; reported: ticks per side block.
//   V0 full transform (12 ds_read_b64, ~52 VALU, 8 ds_write2st64_b32)   V1 LDS reads + waits only
//   V2 VALU only (52 dependent-ish ops)                                   V3 LDS writes only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V>
__device__ __forceinline__ float side_block(float* X, float* Vb, float sv, float carry) {
    float d[4][4];
    if (V == 0 || V == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 a0 = *reinterpret_cast<volatile f32x2*>(X + r * 24);
            const f32x2 a1 = *reinterpret_cast<volatile f32x2*>(X + r * 24 + 2);
            const f32x2 a2 = *reinterpret_cast<volatile f32x2*>(X + r * 24 + 4);
            d[r][0] = a0.y * sv; d[r][1] = a1.x * sv; d[r][2] = a1.y * sv; d[r][3] = a2.x * sv;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) d[r][c] = carry * (float)(r * 4 + c + 1);
    }
    if (V == 1) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) s += d[r][0] + d[r][3];
        return s;
    }
    float tt[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        tt[0][c] = d[0][c] - d[2][c];
        tt[1][c] = d[1][c] + d[2][c];
        tt[2][c] = d[2][c] - d[1][c];
        tt[3][c] = d[1][c] - d[3][c];
    }
    float o[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r * 4 + 0] = tt[r][0] - tt[r][2];
        o[r * 4 + 1] = tt[r][1] + tt[r][2];
        o[r * 4 + 2] = tt[r][2] - tt[r][1];
        o[r * 4 + 3] = tt[r][1] - tt[r][3];
    }
    if (V == 0 || V == 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<volatile float*>(Vb + i * 512) = o[i];
        return o[0];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += o[i];
    return s;
}

template <int V, bool MFMA_ON>
__global__ __launch_bounds__(512) void k(unsigned* ticks, float* sink, const float* in, int chunks, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [0,16K) floats operands, [16K, 24K) V image, [24K, ..) X
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 36 * 1024; i += 512) lds[i] = in[i & 4095];
    __syncthreads();
    float res = 0.f;
    const unsigned t0 = (unsigned)__builtin_readcyclecounter();
    if (wave < 4) {
        if (MFMA_ON) {
            f32x16 acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
            const float* U = lds + lane, *Vv = lds + 8192 + lane;
            for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
                for (int cp = 0; cp < 4; ++cp)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const volatile float* u = U + cp * 2048 + i * 256, *v = Vv + cp * 128 + i * 2048;
                        acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[0], v[0], acc[2 * i], 0, 0, 0);
                        acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[64], v[512], acc[2 * i + 1], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) res += acc[c][i];
        }
    } else {
        float* X = lds + 24 * 1024 + (wave - 4) * 432 + (lane >> 3) * 48 + (lane & 7) * 2;
        float* Vb = lds + 16 * 1024 + (wave - 4) * 64 + lane;
        float carry = in[lane];
        for (int r = 0; r < reps; ++r) carry = side_block<V>(X, Vb, 1.0001f, carry) * 1e-3f + 1.f;
        res = carry;
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned t1 = (unsigned)__builtin_readcyclecounter();
    sink[blockIdx.x * 512 + tid] = res;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int V, bool M>
void run(const char* name, unsigned* dt, float* sink, const float* in, int chunks, int reps) {
    const int blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((k<V, M>), dim3(blocks), dim3(512), 150 * 1024, 0, dt, sink, in, chunks, reps);
    hipLaunchKernelGGL((k<V, M>), dim3(blocks), dim3(512), 150 * 1024, 0, dt, sink, in, chunks, reps);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 8);
    hipMemcpy(h.data(), dt, blocks * 8 * 4, hipMemcpyDeviceToHost);
    std::vector<double> m, s;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? m : s).push_back(h[b * 8 + w]);
    std::sort(m.begin(), m.end()); std::sort(s.begin(), s.end());
    printf("%-40s side block %8.1f ticks    MFMA wave %7.1f ticks per MFMA\n", name, s[s.size() / 2] / reps,
           M ? m[m.size() / 2] / (32.0 * chunks) : 0.0);
}

int main() {
    unsigned* dt; float *sink, *in;
    hipMalloc(&dt, 256 * 8 * 4); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&in, 4096 * 4);
    std::vector<float> hin(4096);
    for (auto& v : hin) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, hin.data(), 4096 * 4, hipMemcpyHostToDevice);
    // MFMA partner runs 600 chunks (~1.3 M ticks); side waves run `reps` blocks that end well before that
    run<0, false>("V0 full transform, partner idle", dt, sink, in, 0, 300);
    run<0, true>("V0 full transform, partner MFMA", dt, sink, in, 600, 300);
    run<1, false>("V1 LDS reads, partner idle", dt, sink, in, 0, 300);
    run<1, true>("V1 LDS reads, partner MFMA", dt, sink, in, 600, 300);
    run<2, false>("V2 VALU only, partner idle", dt, sink, in, 0, 300);
    run<2, true>("V2 VALU only, partner MFMA", dt, sink, in, 600, 300);
    run<3, false>("V3 LDS writes, partner idle", dt, sink, in, 0, 300);
    run<3, true>("V3 LDS writes, partner MFMA", dt, sink, in, 600, 300);
    return 0;
}
