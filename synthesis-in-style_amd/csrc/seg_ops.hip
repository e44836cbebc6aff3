// Segmentation-training side kernels (fp32): the HBM-bound ends of the EMANet / TransUNet training step.
//
//  * sis_upsample_ce_fwd/bwd  bilinear upsample (align_corners=True) + log-softmax + NLL(ignore) + per-sample
//                             mean in one pass each way: EMANet.forward's tail (network.py:305-311) and
//                             CrossEntropyLoss2d (network.py:319-327).  The [B,C,H,W] full-resolution logits are
//                             never materialised; the backward is a gather (one lane per low-resolution cell),
//                             so it is deterministic (no float atomics).
//  * sis_sgd_momentum         torch.optim.SGD(momentum, weight_decay) over ALL parameter tensors of an optimizer
//                             in ONE launch (training_builder/ema_net_train_builder.py:27-48: three groups with
//                             their own lr / weight decay), driven by a device-resident chunk table.
//  * sis_ema_update           emau.mu <- m*mu + (1-m)*mean_b(mu_b)  (updater/segmentation_updater.py:56-66).
#include "sis_common.h"

namespace {

// PyTorch's upsample_bilinear2d source-index rule for align_corners=True: src = dst * (in-1)/(out-1),
// i0 = (int)src, i1 = i0 + (i0 < in-1), lambda = src - i0.
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_index(int dst, float scale, int in_size) {
    const float src = scale * (float)dst;
    Lerp r;
    r.i0 = (int)src;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

constexpr int CE_MAXC = 32;

struct CeParams {
    int B, C, h, w, H, W;
    float sy, sx;
    long ignore;
    int blocks_per_sample;
};

__device__ __forceinline__ void interp_logits(const float* __restrict__ xb, const CeParams& p, const Lerp& ly,
                                              const Lerp& lx, float* v) {
    const int hw = p.h * p.w;
    for (int c = 0; c < p.C; ++c) {
        const float* xc = xb + (int64_t)c * hw;
        v[c] = ly.l0 * (lx.l0 * xc[ly.i0 * p.w + lx.i0] + lx.l1 * xc[ly.i0 * p.w + lx.i1]) +
               ly.l1 * (lx.l0 * xc[ly.i1 * p.w + lx.i0] + lx.l1 * xc[ly.i1 * p.w + lx.i1]);
    }
}

// partial[b][blk] = sum over this block's pixels of -log_softmax(pred)[label]
__global__ __launch_bounds__(256) void ce_fwd_kernel(float* __restrict__ partial, const float* __restrict__ x,
                                                     const long* __restrict__ labels, CeParams p) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int npix = p.H * p.W;
    const float* xb = x + (int64_t)b * p.C * p.h * p.w;
    const long* lb = labels + (int64_t)b * npix;
    float acc = 0.f;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < npix; pix += gridDim.x * 256) {
        const long lab = lb[pix];
        if (lab == p.ignore) continue;
        const int y = pix / p.W, xx = pix - y * p.W;
        float v[CE_MAXC];
        interp_logits(xb, p, lerp_index(y, p.sy, p.h), lerp_index(xx, p.sx, p.w), v);
        float m = v[0];
        for (int c = 1; c < p.C; ++c) m = fmaxf(m, v[c]);
        float s = 0.f;
        for (int c = 0; c < p.C; ++c) s += expf(v[c] - m);
        acc += (m + logf(s)) - v[(int)lab];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void ce_finish_kernel(float* __restrict__ loss, const float* __restrict__ partial,
                                                       int nblk, float inv_npix) {
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) acc += partial[(int64_t)b * nblk + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) loss[b] = acc * inv_npix;
}

// grad_x[b,c,i,j] = grad_loss[b]/(H*W) * sum over pixels whose bilinear footprint touches (i,j) of
//                   weight(i,j) * (softmax_c - [c == label]); 16 lanes per low-resolution cell, all channels: lane s takes the
// footprint rows y_lo + s, y_lo + s + 16, ... and the 16 partial sums are added by a fixed shuffle tree (deterministic).  (One
// lane per cell walked its ~18 x 18 pixels alone: 16 384 lanes on the whole chip, 0.36 ms of EMANet's step.)
constexpr int CE_BWD_SUB = 16;
__global__ __launch_bounds__(256) void ce_bwd_kernel(float* __restrict__ gx, const float* __restrict__ gloss,
                                                     const float* __restrict__ x, const long* __restrict__ labels,
                                                     CeParams p) {
    const int sub = threadIdx.x & (CE_BWD_SUB - 1);
    const int cell_raw = blockIdx.x * (256 / CE_BWD_SUB) + threadIdx.x / CE_BWD_SUB;
    const int b = blockIdx.y;
    const bool live = cell_raw < p.h * p.w;
    const int cell = live ? cell_raw : p.h * p.w - 1;   // (idle lanes shadow the last cell: they take part in the shuffles)
    const int i = cell / p.w, j = cell - i * p.w;
    const int npix = p.H * p.W;
    const float* xb = x + (int64_t)b * p.C * p.h * p.w;
    const long* lb = labels + (int64_t)b * npix;
    // pixels y with floor(sy*y) in {i-1, i}: y in (ceil((i-1)/sy), floor((i+1)/sy))
    const float inv_sy = p.sy > 0.f ? 1.f / p.sy : 0.f, inv_sx = p.sx > 0.f ? 1.f / p.sx : 0.f;
    int y_lo = p.sy > 0.f ? (int)floorf((float)(i - 1) * inv_sy) - 1 : 0;
    int y_hi = p.sy > 0.f ? (int)ceilf((float)(i + 1) * inv_sy) + 1 : p.H - 1;
    int x_lo = p.sx > 0.f ? (int)floorf((float)(j - 1) * inv_sx) - 1 : 0;
    int x_hi = p.sx > 0.f ? (int)ceilf((float)(j + 1) * inv_sx) + 1 : p.W - 1;
    y_lo = max(y_lo, 0); x_lo = max(x_lo, 0); y_hi = min(y_hi, p.H - 1); x_hi = min(x_hi, p.W - 1);
    float acc[CE_MAXC];
    for (int c = 0; c < p.C; ++c) acc[c] = 0.f;
    for (int y = y_lo + sub; y <= y_hi; y += CE_BWD_SUB) {
        const Lerp ly = lerp_index(y, p.sy, p.h);
        float wy = 0.f;
        if (ly.i0 == i) wy += ly.l0;
        if (ly.i1 == i) wy += ly.l1;
        if (wy == 0.f) continue;
        for (int xx = x_lo; xx <= x_hi; ++xx) {
            const Lerp lx = lerp_index(xx, p.sx, p.w);
            float wx = 0.f;
            if (lx.i0 == j) wx += lx.l0;
            if (lx.i1 == j) wx += lx.l1;
            if (wx == 0.f) continue;
            const long lab = lb[y * p.W + xx];
            if (lab == p.ignore) continue;
            float v[CE_MAXC];
            interp_logits(xb, p, ly, lx, v);
            float m = v[0];
            for (int c = 1; c < p.C; ++c) m = fmaxf(m, v[c]);
            float s = 0.f;
            for (int c = 0; c < p.C; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
            const float wgt = wy * wx, inv = 1.f / s;
            for (int c = 0; c < p.C; ++c) acc[c] += wgt * (v[c] * inv - (c == (int)lab ? 1.f : 0.f));
        }
    }
    for (int c = 0; c < p.C; ++c)
#pragma unroll
        for (int o = CE_BWD_SUB / 2; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    if (!live || sub != 0) return;
    const float g = gloss[b] / (float)npix;
    for (int c = 0; c < p.C; ++c) gx[((int64_t)b * p.C + c) * p.h * p.w + cell] = acc[c] * g;
}

// ---- multi-tensor SGD ----------------------------------------------------------------------
// table: one row of 5 int64 per chunk: {param ptr, grad ptr, momentum-buffer ptr, count | group << 48, bf16 shadow ptr | 0}
// The optional shadow is a bfloat16 copy of the parameter kept current by this very launch: layers that compute in bf16
// (TransUNet's encoder Linear layers under autocast) read it instead of casting the fp32 master weight every forward.
constexpr int SGD_CHUNK = 65536;
struct SgdGroups { float lr[4], wd[4]; float momentum; int first; };

__global__ __launch_bounds__(256) void sgd_kernel(const int64_t* __restrict__ table, SgdGroups g) {
    const int64_t* row = table + (int64_t)blockIdx.x * 5;
    float* __restrict__ p = reinterpret_cast<float*>(row[0]);
    const float* __restrict__ gr = reinterpret_cast<const float*>(row[1]);
    float* __restrict__ buf = reinterpret_cast<float*>(row[2]);
    const int n = (int)(row[3] & 0xffffffffll), grp = (int)(row[3] >> 48);
    __hip_bfloat16* __restrict__ shadow = reinterpret_cast<__hip_bfloat16*>(row[4]);
    const float lr = g.lr[grp], wd = g.wd[grp];
    for (int i = threadIdx.x; i < n; i += 256) {
        const float pv = p[i];
        float d = gr[i];
        if (wd != 0.f) d += wd * pv;
        const float bv = g.first ? d : g.momentum * buf[i] + d;
        buf[i] = bv;
        const float pn = pv - lr * bv;
        p[i] = pn;
        if (shadow) shadow[i] = __float2bfloat16(pn);
    }
}

// Same update with the hyper-parameters read from device memory: hyper = {lr[4], wd[4], momentum}.  Nothing but
// pointers in the kernel arguments, so a captured launch (hipGraph) follows the LR schedule on replay.
__global__ __launch_bounds__(256) void sgd_dev_kernel(const int64_t* __restrict__ table, const float* __restrict__ hyper) {
    const int64_t* row = table + (int64_t)blockIdx.x * 5;
    float* __restrict__ p = reinterpret_cast<float*>(row[0]);
    const float* __restrict__ gr = reinterpret_cast<const float*>(row[1]);
    float* __restrict__ buf = reinterpret_cast<float*>(row[2]);
    const int n = (int)(row[3] & 0xffffffffll), grp = (int)(row[3] >> 48);
    __hip_bfloat16* __restrict__ shadow = reinterpret_cast<__hip_bfloat16*>(row[4]);
    const float lr = hyper[grp], wd = hyper[4 + grp], momentum = hyper[8];
    for (int i = threadIdx.x; i < n; i += 256) {
        const float pv = p[i];
        float d = gr[i];
        if (wd != 0.f) d += wd * pv;
        const float bv = momentum * buf[i] + d;
        buf[i] = bv;
        const float pn = pv - lr * bv;
        p[i] = pn;
        if (shadow) shadow[i] = __float2bfloat16(pn);
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ mu, const float* __restrict__ mu_b, float mom,
                                                  float one_minus, int batch, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m = 0.f;
    for (int b = 0; b < batch; ++b) m += mu_b[(int64_t)b * n + i];
    m /= (float)batch;
    float v = mu[i];
    v *= mom;
    v += m * one_minus;
    mu[i] = v;
}

int ce_setup(CeParams& p, int batch, int classes, int h, int w, int H, int W, int64_t ignore) {
    SIS_REQUIRE(classes >= 1 && classes <= CE_MAXC, "upsample_ce: %d classes outside 1..%d", classes, CE_MAXC);
    SIS_REQUIRE(batch > 0 && h > 0 && w > 0 && H > 0 && W > 0, "upsample_ce: non-positive size");
    p.B = batch; p.C = classes; p.h = h; p.w = w; p.H = H; p.W = W; p.ignore = (long)ignore;
    p.sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    p.sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    p.blocks_per_sample = sis_cdiv((int64_t)H * W, 256 * 4);
    if (p.blocks_per_sample > 256) p.blocks_per_sample = 256;
    return 0;
}

}  // namespace

extern "C" int sis_upsample_ce_workspace(int batch, int out_h, int out_w) {
    int bps = sis_cdiv((int64_t)out_h * out_w, 256 * 4);
    if (bps > 256) bps = 256;
    return batch * bps;  // floats
}

extern "C" int sis_upsample_ce_fwd(float* loss, float* workspace, const float* logits, const int64_t* labels, int batch,
                                   int classes, int h, int w, int out_h, int out_w, int64_t ignore_index, void* stream) {
    SIS_REQUIRE(loss && workspace && logits && labels, "sis_upsample_ce_fwd: null pointer");
    CeParams p;
    if (ce_setup(p, batch, classes, h, w, out_h, out_w, ignore_index)) return 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(p.blocks_per_sample, batch), dim3(256), 0, st, workspace, logits,
                       (const long*)labels, p);
    SIS_CHECK_LAUNCH("ce_fwd_kernel");
    hipLaunchKernelGGL(ce_finish_kernel, dim3(batch), dim3(64), 0, st, loss, workspace, p.blocks_per_sample,
                       1.f / ((float)out_h * (float)out_w));
    SIS_CHECK_LAUNCH("ce_finish_kernel");
    return 0;
}

extern "C" int sis_upsample_ce_bwd(float* grad_logits, const float* grad_loss, const float* logits,
                                   const int64_t* labels, int batch, int classes, int h, int w, int out_h, int out_w,
                                   int64_t ignore_index, void* stream) {
    SIS_REQUIRE(grad_logits && grad_loss && logits && labels, "sis_upsample_ce_bwd: null pointer");
    CeParams p;
    if (ce_setup(p, batch, classes, h, w, out_h, out_w, ignore_index)) return 1;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(sis_cdiv(h * w, 256 / CE_BWD_SUB), batch), dim3(256), 0, (hipStream_t)stream, grad_logits,
                       grad_loss, logits, (const long*)labels, p);
    SIS_CHECK_LAUNCH("ce_bwd_kernel");
    return 0;
}

extern "C" int sis_sgd_chunk_elems(void) { return SGD_CHUNK; }

extern "C" int sis_sgd_momentum(const int64_t* table, int n_chunks, const float* lr, const float* weight_decay,
                                int n_groups, float momentum, int first_step, void* stream) {
    if (n_chunks <= 0) return 0;
    SIS_REQUIRE(table && lr && weight_decay, "sis_sgd_momentum: null pointer");
    SIS_REQUIRE(n_groups >= 1 && n_groups <= 4, "sis_sgd_momentum: %d parameter groups outside 1..4", n_groups);
    SgdGroups g;
    for (int i = 0; i < 4; ++i) { g.lr[i] = i < n_groups ? lr[i] : 0.f; g.wd[i] = i < n_groups ? weight_decay[i] : 0.f; }
    g.momentum = momentum; g.first = first_step;
    hipLaunchKernelGGL(sgd_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, g);
    SIS_CHECK_LAUNCH("sgd_kernel");
    return 0;
}

extern "C" int sis_sgd_momentum_dev(const int64_t* table, int n_chunks, const float* hyper, void* stream) {
    if (n_chunks <= 0) return 0;
    SIS_REQUIRE(table && hyper, "sis_sgd_momentum_dev: null pointer");
    hipLaunchKernelGGL(sgd_dev_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, hyper);
    SIS_CHECK_LAUNCH("sgd_dev_kernel");
    return 0;
}

extern "C" int sis_ema_update(float* mu, const float* mu_batch, float momentum, float one_minus_momentum, int batch,
                              int numel, void* stream) {
    if (numel <= 0 || batch <= 0) return 0;
    SIS_REQUIRE(mu && mu_batch, "sis_ema_update: null pointer");
    hipLaunchKernelGGL(ema_kernel, dim3(sis_cdiv(numel, 256)), dim3(256), 0, (hipStream_t)stream, mu, mu_batch, momentum,
                       one_minus_momentum, batch, numel);
    SIS_CHECK_LAUNCH("ema_kernel");
    return 0;
}
