// Modulated convolution on the CDNA4 matrix cores, fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain; peak 157.3 TFLOP/s on MI355X).
//
// Replaces the composite of networks/stylegan2/model.py:237-278 (per-sample weight
// materialisation + cuDNN grouped conv with groups = batch).  Formulation here:
//
//     y[b,co,.] = dscale[b,co] * sum_{ci,tap} W[co,ci,tap] * ( s[b,ci] * x[b,ci,. + tap] )
//
// i.e. ONE shared weight tensor for the whole batch (prepacked [ci][tap][co], co contiguous),
// the style applied to the input tile while it is staged into LDS, demodulation / noise / bias /
// leaky-ReLU applied to the accumulators in the epilogue.  Direct NCHW convolution, no im2col:
//
//   GEMM view     M = Cout, N = pixels (of several samples for tiny layers), K = Cin * taps
//   A operand     W  -> LDS [ci][tap][co]        lane l reads co = l&31 of channel 2cp + (l>>5)
//   B operand     x*s-> LDS [ci][tile+halo]      lane l reads pixel l&31 of channel 2cp + (l>>5),
//                                                shifted by the tap offset (32 consecutive floats:
//                                                bank-conflict free ds_read_b32)
//   C/D           col = lane&31 = pixel, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = co, so every
//                 store instruction writes two 128-byte pixel runs.
//
// MODE 0: stride-1 "same" convolution (ks 3 or 1).  Workgroup = 4 waves = 128 co x 256 pixels,
//         each wave 64 co x 128 pixels = 2x4 accumulator tiles (128 VGPRs).
// MODE 1: stride-2 transposed convolution (ks 3) in gather form over the 4 output phases
//         T[2h+a, 2w+b], a,b in {0,1}: 4,2,2,1 taps, K = Cin*9 in total, so the FLOPs are those of
//         the zero-insertion-free transposed conv (1/4 of a dense conv at output resolution).
//         Workgroup = 64 co x 128 positions, each wave 64 co x 32 positions x 4 phases.
//         Positions run over (H+1) x (W+1); the odd row / column / corner are handled by extra
//         tile classes (1 x tw, th x 1, 1 x 1 shaped tiles) instead of padding every tile.
//
// Occupancy plan: 2 workgroups per CU (<= 256 VGPRs, ~48 KB LDS each); while one stages its next
// K chunk the other one's MFMAs keep the matrix pipe busy (f32 MFMA issues one 32x32x2 per 64
// cycles per SIMD, so LDS reads -- 12-22 ds_read_b32 per 18-24 MFMAs -- are nowhere near a limit).
#include "modconv_common.h"

namespace {

constexpr int MAX_CLS = MC_MAX_CLS;
constexpr int CC = 8;   // input channels per K chunk
constexpr int XI = MC_XI;

template <int MODE, int KS>
struct Cfg : ConvCfg<MODE, KS> {
    static constexpr int WFLOATS = CC * ConvCfg<MODE, KS>::NTAPS * ConvCfg<MODE, KS>::MBLK;
};

template <int MODE, int KS>
__global__ __launch_bounds__(256, 2) void modconv_mfma_kernel(const ConvParams p) {
    typedef Cfg<MODE, KS> C;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;               // [CC][NTAPS][MBLK]
    float* Xl = lds + C::WFLOATS;  // [CC][xt]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = MODE == 0 ? (wave >> 1) : 0;
    const int wn = MODE == 0 ? (wave & 1) : wave;

    // ---- block -> (output-channel block, tile class, tile) ; oc block slowest so that all CUs
    // stream the same weight slice (<= 2.4 MB, L2 resident) at the same time.
    int pt = blockIdx.x % p.npos_tiles;
    const int o0 = (blockIdx.x / p.npos_tiles) * C::MBLK;
    int ci_cls = 0;
#pragma unroll
    for (int c = 1; c < MAX_CLS; ++c)
        if (c < p.ncls && pt >= p.cls[c].first_block) ci_cls = c;
    const TileClass tc = p.cls[ci_cls];
    pt -= tc.first_block;
    const int twi = pt % tc.ntw; pt /= tc.ntw;
    const int thi = pt % tc.nth;
    const int bt = pt / tc.nth;
    const int thl = tc.th_log2, twl = tc.tw_log2;
    const int th = 1 << thl, tw = 1 << twl;
    const int b0 = bt * tc.nb, h0 = tc.h0 + (thi << thl), w0 = tc.w0 + (twi << twl);
    const int eh = th + C::EXT, ew = tw + C::EXT;
    const int xt = tc.xt;
    const int HW = p.H * p.W;

    // ---- per-lane staging descriptors for the input tile (same for every K chunk)
    int st_goff[XI], st_soff[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int idx = tid + 256 * i;
        st_goff[i] = -1; st_soff[i] = 0;
        if (idx < xt) {
            const int n = idx / (eh * ew), rem = idx - n * (eh * ew);
            const int r = rem / ew, c = rem - r * ew;
            const int b = b0 + n, h = h0 - C::PAD_LO + r, w = w0 - C::PAD_LO + c;
            if (b < p.B && h >= 0 && h < p.H && w >= 0 && w < p.W) {
                st_goff[i] = b * p.Cin * HW + h * p.W + w;
                st_soff[i] = b * p.Cin;
            }
        }
    }

    // ---- per-lane operand offsets
    int xo[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) {
        const int pp = (wn * C::NT + t) * 32 + l31;
        const int n = pp >> (thl + twl), rem = pp & ((1 << (thl + twl)) - 1);
        const int r = rem >> twl, c = rem & (tw - 1);
        xo[t] = n * eh * ew + r * ew + c + half * xt;
    }
    const int aoff = half * C::NTAPS * C::MBLK + wm * 64 + l31;

    f32x16 acc[C::MT][C::NACC];
#pragma unroll
    for (int m = 0; m < C::MT; ++m)
#pragma unroll
        for (int a = 0; a < C::NACC; ++a)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][a][j] = 0.f;

    for (int ci0 = 0; ci0 < p.Cin; ci0 += CC) {
        __syncthreads();  // everyone is done reading the previous chunk
        // ---- stage weights: rows (ci, tap) of MBLK contiguous floats
        constexpr int WV4 = C::WFLOATS / 4;
        if (p.cout_vec4) {
#pragma unroll
            for (int i = 0; i < (WV4 + 255) / 256; ++i) {
                const int e = tid + 256 * i;
                if (e < WV4) {
                    const int row = e / (C::MBLK / 4), q = e - row * (C::MBLK / 4);
                    const int ci = ci0 + row / C::NTAPS, tap = row % C::NTAPS;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ci < p.Cin && o0 + q * 4 < p.Cout)
                        v = *reinterpret_cast<const float4*>(p.wpk + ((int64_t)ci * C::NTAPS + tap) * p.Cout + o0 + q * 4);
                    *reinterpret_cast<float4*>(Wl + row * C::MBLK + q * 4) = v;
                }
            }
        } else {
            for (int e = tid; e < C::WFLOATS; e += 256) {
                const int row = e / C::MBLK, q = e - row * C::MBLK;
                const int ci = ci0 + row / C::NTAPS, tap = row % C::NTAPS;
                float v = 0.f;
                if (ci < p.Cin && o0 + q < p.Cout) v = p.wpk[((int64_t)ci * C::NTAPS + tap) * p.Cout + o0 + q];
                Wl[e] = v;
            }
        }
        // ---- stage the style-modulated input tile (zero padded), 4 channels at a time
#pragma unroll
        for (int j0 = 0; j0 < CC; j0 += 4) {
            float xv[4][XI];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    xv[j][i] = 0.f;
                    const int ci = ci0 + j0 + j;
                    if (st_goff[i] >= 0 && ci < p.Cin) xv[j][i] = p.x[st_goff[i] + ci * HW] * p.s[st_soff[i] + ci];
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < XI; ++i)
                    if (tid + 256 * i < xt) Xl[(j0 + j) * xt + tid + 256 * i] = xv[j][i];
        }
        __syncthreads();

        // ---- MFMA over this chunk: K = CC * NTAPS in steps of 2 channels
        if (MODE == 0) {
#pragma unroll 1
            for (int tap = 0; tap < C::NTAPS; ++tap) {
                const int toff = (tap / KS) * ew + (tap % KS);
#pragma unroll
                for (int cp = 0; cp < CC / 2; ++cp) {
                    float a[C::MT], bv[C::NT];
#pragma unroll
                    for (int m = 0; m < C::MT; ++m) a[m] = Wl[(2 * cp * C::NTAPS + tap) * C::MBLK + aoff + m * 32];
#pragma unroll
                    for (int t = 0; t < C::NT; ++t) bv[t] = Xl[2 * cp * xt + xo[t] + toff];
#pragma unroll
                    for (int m = 0; m < C::MT; ++m)
#pragma unroll
                        for (int t = 0; t < C::NT; ++t)
                            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[t], acc[m][t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int cp = 0; cp < CC / 2; ++cp) {
                const float* xb = Xl + 2 * cp * xt + xo[0];
                const float x_ul = xb[0], x_u = xb[1], x_l = xb[ew], x_c = xb[ew + 1];
#pragma unroll
                for (int m = 0; m < C::MT; ++m) {
                    const float* wb = Wl + 2 * cp * C::NTAPS * C::MBLK + aoff + m * 32;
                    float a[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) a[t] = wb[t * C::MBLK];
                    // phase (0,0): taps (0,0) (2,0) (0,2) (2,2)
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], x_c, acc[m][0], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[6], x_u, acc[m][0], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], x_l, acc[m][0], 0, 0, 0);
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8], x_ul, acc[m][0], 0, 0, 0);
                    // phase (0,1): taps (0,1) (2,1)
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], x_c, acc[m][1], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[7], x_u, acc[m][1], 0, 0, 0);
                    // phase (1,0): taps (1,0) (1,2)
                    acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], x_c, acc[m][2], 0, 0, 0);
                    acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[5], x_l, acc[m][2], 0, 0, 0);
                    // phase (1,1): tap (1,1)
                    acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4], x_c, acc[m][3], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue
    if (MODE == 0) {
        float nw = 0.f;
        if (p.fuse && p.noise) nw = p.noise_w[0];
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const int pp = (wn * C::NT + t) * 32 + l31;
            const int n = pp >> (thl + twl), rem = pp & ((1 << (thl + twl)) - 1);
            const int b = b0 + n, h = h0 + (rem >> twl), w = w0 + (rem & (tw - 1));
            if (n >= tc.nb || b >= p.B || h >= p.H || w >= p.W) continue;
            float nz = 0.f;
            if (p.fuse && p.noise) nz = nw * p.noise[(int64_t)b * p.noise_bstride + h * p.W + w];
            float* ob = p.out + (int64_t)b * p.Cout * HW + h * p.W + w;
            const float* db = p.dscale + (int64_t)b * p.Cout;
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int co = o0 + wm * 64 + m * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
                    if (co < p.Cout) {
                        float v = acc[m][t][j] * db[co];
                        if (p.fuse) {
                            v += nz;
                            if (p.bias) v += p.bias[co];
                            v = (v > 0.f ? v : v * 0.2f) * 1.4142135623730951f;
                        }
                        ob[(int64_t)co * HW] = v;
                    }
                }
        }
    } else {
        const int pp = wn * 32 + l31;
        const int n = pp >> (thl + twl), rem = pp & ((1 << (thl + twl)) - 1);
        const int b = b0 + n, h = h0 + (rem >> twl), w = w0 + (rem & (tw - 1));
        if (n < tc.nb && b < p.B && h < tc.h1 && w < tc.w1) {
            const int OHW = p.OH * p.ORS;
            float* ob = p.out + (int64_t)b * p.Cout * OHW;
            const float* db = p.dscale + (int64_t)b * p.Cout;
#pragma unroll
            for (int m = 0; m < C::MT; ++m)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int co = o0 + m * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
                    if (co < p.Cout) {
                        const float d = db[co];
                        float* oc = ob + (int64_t)co * OHW;
#pragma unroll
                        for (int ph = 0; ph < 4; ++ph) {
                            const int oy = 2 * h + (ph >> 1), ox = 2 * w + (ph & 1);
                            if (oy < p.OH && ox < p.OW) oc[oy * p.ORS + ox] = acc[m][ph][j] * d;
                        }
                    }
                }
        }
    }
}

template <int MODE, int KS>
int launch(ConvParams& p, hipStream_t st) {
    typedef Cfg<MODE, KS> C;
    int xt_max = 0;
    for (int c = 0; c < p.ncls; ++c) xt_max = p.cls[c].xt > xt_max ? p.cls[c].xt : xt_max;
    SIS_REQUIRE(xt_max <= 256 * XI, "modconv: internal tile too large (%d)", xt_max);
    const size_t lds = (size_t)(C::WFLOATS + CC * xt_max) * sizeof(float);
    const int64_t blocks = (int64_t)p.npos_tiles * sis_cdiv(p.Cout, C::MBLK);
    SIS_REQUIRE(blocks > 0 && blocks < ((int64_t)1 << 31), "modconv: bad grid");
    sis_kernel_name = MODE == 1 ? "modconv_mfma_kernel<1, 3>" : KS == 3 ? "modconv_mfma_kernel<0, 3>" : "modconv_mfma_kernel<0, 1>";
    hipLaunchKernelGGL((modconv_mfma_kernel<MODE, KS>), dim3((unsigned)blocks), dim3(256), lds, st, p);
    SIS_CHECK_LAUNCH("modconv_mfma_kernel");
    return 0;
}

int check_common(const char* name, const void* out, const void* x, const void* wpk, const void* s, const void* dscale,
                 int batch, int cin, int cout, int h, int w, int oh, int ow) {
    SIS_REQUIRE(out && x && wpk && s && dscale, "%s: null pointer", name);
    SIS_REQUIRE(batch > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "%s: non-positive size", name);
    SIS_REQUIRE((int64_t)batch * cin * h * w < ((int64_t)1 << 31) && (int64_t)batch * cout * oh * ow < ((int64_t)1 << 31),
                "%s: tensor too large for 32-bit plane offsets", name);
    return 0;
}

}  // namespace

static void init_params(ConvParams& p) {
    p.slab = nullptr; p.ksplit = 1; p.npos_tiles = 0; p.ncls = 0; p.nb_max = 0;
}

extern "C" int sis_modconv2d(float* out, const float* x, const float* wpk, const float* s, const float* dscale,
                             const float* noise, int64_t noise_batch_stride, const float* noise_weight,
                             const float* bias, int batch, int cin, int cout, int h, int w, int ksize, int fuse_act,
                             const float* wino_u, void* workspace, int64_t workspace_bytes, void* stream) {
    if (batch == 0) return 0;
    if (check_common("sis_modconv2d", out, x, wpk, s, dscale, batch, cin, cout, h, w, h, w)) return 1;
    SIS_REQUIRE(ksize == 1 || ksize == 3, "sis_modconv2d: kernel size %d not supported (1 or 3)", ksize);
    if (noise) SIS_REQUIRE(noise_weight, "sis_modconv2d: noise given without noise_weight");
    ConvParams p;
    init_params(p);
    p.x = x; p.wpk = wpk; p.s = s; p.dscale = dscale; p.noise = noise; p.noise_w = noise_weight; p.bias = bias;
    p.out = out; p.noise_bstride = noise_batch_stride;
    p.B = batch; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w; p.OH = h; p.OW = w; p.ORS = w; p.fuse = fuse_act != 0;
    p.kchunk = cin;
    p.cout_vec4 = (cout % 4 == 0) && (((uintptr_t)wpk & 15) == 0);
    mc_add_class(p, 256, ksize - 1, batch, 0, h, 0, w, 32, 16);
    if (wino_u && ksize == 3 && w % 4 == 0 && h % 2 == 0) {
        // Winograd tiles: 16x16-pixel regions (conflict-free patch reads), rows staged as 16-byte aligned supersets
        ConvParams pw = p;
        pw.npos_tiles = 0; pw.ncls = 0; pw.nb_max = 0;
        mc_add_class(pw, 256, 2, batch, 0, h, 0, w, 16, 16);
        TileClass& tcw = pw.cls[0];
        tcw.xt = tcw.nb * ((1 << tcw.th_log2) + 2) * ((1 << tcw.tw_log2) + 8);
        pw.wpk = wino_u;
        const int rcw = modconv_wino_launch(pw, (hipStream_t)stream, workspace, workspace_bytes);
        if (rcw >= 0) return rcw;
    }
    const int rc = modconv_v2_launch(p, 0, ksize, (hipStream_t)stream, workspace, workspace_bytes);
    if (rc >= 0) return rc;
    p.ksplit = 1; p.kchunk = cin; p.slab = nullptr;
    p.npos_tiles = 0; p.ncls = 0; p.nb_max = 0;
    mc_add_class(p, 256, ksize - 1, batch, 0, h, 0, w, 32, 16);  // exact-halo tiles for the register-staged kernel
    if (ksize == 3) return launch<0, 3>(p, (hipStream_t)stream);
    return launch<0, 1>(p, (hipStream_t)stream);
}

// Plain (unmodulated) 3x3 convolution, stride 1, padding 1, on the Winograd kernel: unit style, unit demodulation,
// no epilogue.  Used for the segmentation networks' 3x3 layers (forward, and data gradient with adjoint weights).
static int conv3x3_plan(ConvParams& p, float* out, const float* x, const float* u, int batch, int cin, int cout, int h,
                        int w) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return -1;
    if ((int64_t)batch * (cin > cout ? cin : cout) * h * w >= ((int64_t)1 << 31)) return -1;
    if (w % 4 != 0 || h % 2 != 0 || cin % 8 != 0 || cout % 4 != 0) return -1;
    init_params(p);
    p.x = x; p.wpk = u; p.s = nullptr; p.dscale = nullptr; p.noise = nullptr; p.noise_w = nullptr; p.bias = nullptr;
    p.out = out; p.noise_bstride = 0;
    p.B = batch; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w; p.OH = h; p.OW = w; p.ORS = w; p.fuse = 0;
    p.kchunk = cin;
    p.cout_vec4 = (((uintptr_t)u & 15) == 0);
    mc_add_class(p, 256, 2, batch, 0, h, 0, w, 16, 16);
    TileClass& tc = p.cls[0];
    tc.xt = tc.nb * ((1 << tc.th_log2) + 2) * ((1 << tc.tw_log2) + 8);
    return 0;
}

extern "C" int sis_conv3x3_eligible(int batch, int cin, int cout, int h, int w) {
    ConvParams p;
    if (conv3x3_plan(p, nullptr, nullptr, nullptr, batch, cin, cout, h, w) < 0) return 0;
    return modconv_wino_launch(p, nullptr, nullptr, 0, true) == 0 ? 1 : 0;
}

extern "C" int sis_conv3x3(float* out, const float* x, const float* u, int batch, int cin, int cout, int h, int w,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    if (batch == 0) return 0;
    SIS_REQUIRE(out && x && u, "sis_conv3x3: null pointer");
    ConvParams p;
    SIS_REQUIRE(conv3x3_plan(p, out, x, u, batch, cin, cout, h, w) == 0,
                "sis_conv3x3: needs W %% 4 == 0, H %% 2 == 0, Cin %% 8 == 0, Cout %% 4 == 0 and < 2^31 elements "
                "(got %d x %dx%d, %d -> %d)", batch, h, w, cin, cout);
    const int rc = modconv_wino_launch(p, (hipStream_t)stream, workspace, workspace_bytes);
    SIS_REQUIRE(rc >= 0, "sis_conv3x3: shape %dx%d, %d -> %d not eligible for the Winograd kernel", h, w, cin, cout);
    return rc;
}

extern "C" int sis_modconv2d_up(float* t, const float* x, const float* wpk, const float* s, const float* dscale,
                                int batch, int cin, int cout, int h, int w, int t_row_stride, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    if (batch == 0) return 0;
    const int oh = 2 * h + 1, ow = 2 * w + 1;
    if (t_row_stride <= 0) t_row_stride = ow;
    SIS_REQUIRE(t_row_stride >= ow, "sis_modconv2d_up: row stride %d smaller than 2W+1 = %d", t_row_stride, ow);
    if (check_common("sis_modconv2d_up", t, x, wpk, s, dscale, batch, cin, cout, h, w, oh, t_row_stride)) return 1;
    ConvParams p;
    init_params(p);
    p.x = x; p.wpk = wpk; p.s = s; p.dscale = dscale; p.noise = nullptr; p.noise_w = nullptr; p.bias = nullptr;
    p.out = t; p.noise_bstride = 0;
    p.B = batch; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w; p.OH = oh; p.OW = ow; p.ORS = t_row_stride; p.fuse = 0;
    p.kchunk = cin;
    p.cout_vec4 = (cout % 4 == 0) && (((uintptr_t)wpk & 15) == 0);
    int mblk, npos;
    modconv_v2_tile(1, &mblk, &npos);
    for (int pass = 0; pass < 2; ++pass) {
        const int np = pass == 0 ? npos : 128;
        p.npos_tiles = 0; p.ncls = 0; p.nb_max = 0;
        mc_add_class(p, np, 1, batch, 0, h, 0, w, 32, 8);          // interior positions
        mc_add_class(p, np, 1, batch, h, h + 1, 0, w, np, h >= 16 ? 4 : 8);  // last row (T[2H, 0..2W-1]); 16 x 16: 8 style rows here made the grid's LDS 86 KB, one workgroup per CU
        // last col (T[0..2H-1, 2W]): one position per row, each staged as a padded 8-float row, and the corner: capped at
        // 64 rows / 4 samples per tile.  The grid's LDS size follows its LARGEST class: uncapped, the last column's staging
        // area (2 x 65 x 8 floats per channel) and the corner's 8 style rows made it 120 KB -- one workgroup per CU for
        // every tile of the layer (measured with hipOccupancyMaxActiveBlocksPerMultiprocessor; 74-80 KB now: two per CU).
        // (4x4 and 8x8 layers keep 8 samples per edge tile: they are split-K / launch bound, more tiles cost them 30-50 %.)
        const int edge_nb = h >= 16 ? 4 : 8;
        mc_add_class(p, np < 64 ? np : 64, 1, batch, 0, h, w, w + 1, 1, edge_nb);
        mc_add_class(p, np, 1, batch, h, h + 1, w, w + 1, 1, edge_nb);   // corner    (T[2H, 2W])
        if (pass == 0) {
            const int rc = modconv_v2_launch(p, 1, 3, (hipStream_t)stream, workspace, workspace_bytes);
            if (rc >= 0) return rc;
        }
    }
    p.ksplit = 1; p.kchunk = cin; p.slab = nullptr;
    return launch<1, 3>(p, (hipStream_t)stream);
}
