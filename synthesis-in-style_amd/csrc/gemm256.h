// Interface between sis_gemm_bf16 (gemm_bf16.hip: argument checks, tile codes) and the 256-row tiles of gemm256_bf16.hip.
#pragma once
#include "vit_common.h"

struct G256Params {
    const void* A; const void* B;   // A [M][K], B [N][K] bf16, both K-contiguous (the NT layout)
    int lda, ldb;                   // elements
    unsigned a_bytes, b_bytes;      // extent of each operand from its base pointer (range check of the LDS-DMA)
    int M, N, K;
    void* C; void* C2; int ldc;
    const float* bias; const float* bias1; const float* bias2; int bias_seg;
    const float* resid; const unsigned short* pre;
    const unsigned long long* seed; unsigned site, drop_thr; float drop_scale;
    int m_tiles, n_tiles;
#ifdef G256_TRACE   // development builds: per-workgroup cycle stamps (tools/trace_gemm256.py)
    unsigned long long* trace;
#endif
};

// np = 48-column panels per wave (1, 2, 3: workgroup tiles 256 x 96 / 192 / 288); epilogue: SIS_GEMM_EPI_* except F32
int sis_gemm256_dispatch(const G256Params& p, int np, int epilogue, hipStream_t st);
bool sis_gemm256_ok(int layout, int epilogue, int k, int splits);
