// Shared between the two modulated-convolution kernels (modconv_mfma.hip: register-staged, any shape;
// modconv_mfma2.hip: LDS-DMA double-buffered fast path).
#pragma once
#include "sis_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MC_MAX_CLS = 4;
constexpr int MC_XI = 3;  // staged x elements per lane per channel (covers xt <= 768)

// A "tile class" is a family of equally shaped position tiles.  The stride-1 conv needs one; the
// transposed conv needs four because its positions run over (H+1) x (W+1): interior, last row,
// last column and corner get their own tile shapes instead of padding every tile.
struct TileClass {
    int th_log2, tw_log2, nb;  // tile = nb samples x 2^th_log2 rows x 2^tw_log2 cols
    int h0, w0, h1, w1;        // region of positions this class owns: [h0,h1) x [w0,w1)
    int nth, ntw;              // tiles per sample
    int first_block;           // first position-tile index of this class
    int xt;                    // floats per channel of the staged input tile
};

struct ConvParams {
    const float* x; const float* wpk; const float* s; const float* dscale;
    const float* noise; const float* noise_w; const float* bias;
    float* out;
    float* slab;               // split-K partial sums [ksplit][B][Cout][OH][OW] or nullptr
    int64_t noise_bstride;
    int B, Cin, Cout, H, W, OH, OW;
    int ORS;                   // output row stride in floats (>= OW)
    int fuse;
    int npos_tiles, ncls;
    int cout_vec4;
    int ksplit, kchunk;        // split-K: blockIdx.y = k slice, kchunk input channels per slice
    int nb_max;
    TileClass cls[MC_MAX_CLS];
};

template <int MODE, int KS>
struct ConvCfg {
    static constexpr int NTAPS = KS * KS;
    static constexpr int MBLK = MODE == 0 ? 128 : 64;
    static constexpr int NPOS = MODE == 0 ? 256 : 128;
    static constexpr int MT = 2;
    static constexpr int NT = MODE == 0 ? 4 : 1;
    static constexpr int NACC = 4;
    static constexpr int PAD_LO = MODE == 0 ? KS / 2 : 1;
    static constexpr int EXT = MODE == 0 ? KS - 1 : 1;
};

static inline int mc_ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Adds a tile class covering positions [h0,h1) x [w0,w1) with tiles of at most npos positions.
static inline void mc_add_class(ConvParams& p, int npos, int ext, int batch, int h0, int h1, int w0, int w1,
                                int tw_cap, int nb_cap) {
    const int hh = h1 - h0, ww = w1 - w0;
    if (hh <= 0 || ww <= 0) return;
    TileClass& c = p.cls[p.ncls];
    int twl = mc_ilog2_ceil(ww);
    const int capl = mc_ilog2_ceil(tw_cap);
    if (twl > capl) twl = capl;
    const int npl = mc_ilog2_ceil(npos);
    int thl = mc_ilog2_ceil(hh);
    if (thl > npl - twl) thl = npl - twl;
    c.th_log2 = thl; c.tw_log2 = twl; c.nb = npos >> (thl + twl);
    if (c.nb > nb_cap) c.nb = nb_cap;
    if (c.nb > batch) c.nb = batch;
    c.h0 = h0; c.w0 = w0; c.h1 = h1; c.w1 = w1;
    c.nth = sis_cdiv(hh, 1 << thl); c.ntw = sis_cdiv(ww, 1 << twl);
    c.first_block = p.npos_tiles;
    c.xt = c.nb * ((1 << thl) + ext) * ((1 << twl) + ext);
    p.npos_tiles += c.nth * c.ntw * sis_cdiv(batch, c.nb);
    if (c.nb > p.nb_max) p.nb_max = c.nb;
    p.ncls++;
}

// Fast paths stage the input tile with 16-byte LDS-DMA: rows become the 16-byte aligned superset of the halo'd
// tile columns (left pad 4 instead of 1, right edge rounded up to a multiple of 4).  W % 4 == 0 makes every
// aligned float4 lie entirely inside or entirely outside the image.
static inline int mc_padded_ew(int tw, int pad_lo, int ext) { return (pad_lo ? 4 : 0) + ((tw + ext - pad_lo + 3) & ~3); }
static inline void mc_set_padded_xt(ConvParams& p, int pad_lo, int ext) {
    for (int c = 0; c < p.ncls; ++c) {
        TileClass& tc = p.cls[c];
        tc.xt = tc.nb * ((1 << tc.th_log2) + ext) * mc_padded_ew(1 << tc.tw_log2, pad_lo, ext);
    }
}

// modconv_mfma2.hip: returns 0 on success, 1 on error, -1 when the shape is not eligible for the fast path.
int modconv_v2_launch(ConvParams& p, int mode, int ks, hipStream_t st, void* workspace, int64_t workspace_bytes);
// modconv_wino.hip: Winograd F(2x2,3x3) path for stride-1 3x3 layers (p.wpk must then hold the transformed weights).
int modconv_wino_launch(ConvParams& p, hipStream_t st, void* workspace, int64_t workspace_bytes, bool plan_only = false);
// Block tile of the fast path for `mode` (positions per tile depend on the selected wave layout).
int modconv_v2_tile(int mode, int* mblk, int* npos);
