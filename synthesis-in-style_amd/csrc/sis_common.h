// Shared host-side helpers for libsis_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sis_hip.h"

extern thread_local char sis_err_buf[512];
extern thread_local const char* sis_kernel_name;  // device kernel the last sis_modconv2d* call dispatched to

static inline int sis_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(sis_err_buf, sizeof(sis_err_buf), fmt, ap);
    va_end(ap);
    return 1;
}

// Launch-error check: no host sync, just the launch status.
// Development aid (SIS_OCC=1): what the runtime says fits on a CU for this launch -- a kernel's dynamic LDS size or register
// count can silently halve the workgroups per CU its design assumes (the transposed modconv ran one per CU for a round).
#define SIS_OCC_REPORT(kernel, threads, lds_bytes)                                                                      \
    do {                                                                                                                \
        static const bool occ_on_ = getenv("SIS_OCC") != nullptr;                                                       \
        if (occ_on_) {                                                                                                  \
            int nb_ = -1;                                                                                               \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, kernel, (int)(threads), (size_t)(lds_bytes));      \
            fprintf(stderr, "[occ] %s: %d threads, %zu B LDS -> %d workgroups per CU\n", #kernel, (int)(threads), (size_t)(lds_bytes), nb_); \
        }                                                                                                               \
    } while (0)

#define SIS_CHECK_LAUNCH(name)                                                                   \
    do {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                       \
        if (e_ != hipSuccess) return sis_fail("%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

#define SIS_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) return sis_fail(__VA_ARGS__); \
    } while (0)

static inline int sis_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Device-side scalar conversion helpers (compute type for f16/bf16 is float, as the
// reference kernels' scalar_t arithmetic on half promotes through float on the FMA path).
template <typename T> struct sis_acc { typedef T type; };
template <> struct sis_acc<__half> { typedef float type; };
template <> struct sis_acc<__hip_bfloat16> { typedef float type; };

template <typename T> __device__ __forceinline__ typename sis_acc<T>::type sis_ld(const T* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float sis_ld<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <> __device__ __forceinline__ float sis_ld<__hip_bfloat16>(const __hip_bfloat16* p, int64_t i) { return __bfloat162float(p[i]); }

template <typename T> __device__ __forceinline__ void sis_st(T* p, int64_t i, typename sis_acc<T>::type v) { p[i] = v; }
template <> __device__ __forceinline__ void sis_st<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }
template <> __device__ __forceinline__ void sis_st<__hip_bfloat16>(__hip_bfloat16* p, int64_t i, float v) { p[i] = __float2bfloat16(v); }
