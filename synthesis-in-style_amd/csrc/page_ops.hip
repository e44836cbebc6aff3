// Patch-wise inference of a page image, device-resident:
//  * sis_crop_patches_u8   segmentation/analysis_segmenter.py:115-130 -- PIL crop (zero padding outside the image),
//                          ToTensor (u8 / 255) and Normalize(0.5, 0.5) for a whole grid of patches in one pass;
//  * sis_assemble_max      :147-167 -- element-wise maximum over the patches that cover a pixel, as a gather (one
//                          lane per page pixel walks the patch grid: deterministic, no atomics, no -inf fill pass),
//                          optionally with the label map of networks/base_segmenter.py:59-62 (first maximal class).
// Both are HBM-bound single passes: 1 byte read + 4 bytes written per element / 4 bytes read per covering patch
// element + 4 written.
#include "sis_common.h"

namespace {

constexpr int PG_MAXC = 16;

struct PatchGrid {
    const int* xs; const int* ys;  // left / top of the patch columns / rows (device arrays, ascending)
    int nx, ny, patch, height, width, channels;
};

__global__ __launch_bounds__(256) void crop_patches_kernel(float* __restrict__ out, const uint8_t* __restrict__ image,
                                                           PatchGrid g) {
    const int n = blockIdx.z, py = blockIdx.y;
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= g.patch) return;
    const int top = g.ys[n / g.nx], left = g.xs[n % g.nx];
    const int y = top + py, x = left + px;
    const bool inside = y < g.height && x < g.width;
    const uint8_t* src = image + ((int64_t)y * g.width + x) * g.channels;
    float* dst = out + (int64_t)n * g.channels * g.patch * g.patch + (int64_t)py * g.patch + px;
    for (int c = 0; c < g.channels; ++c) {
        const float t = (inside ? (float)src[c] : 0.f) / 255.0f;  // ToTensor
        dst[(int64_t)c * g.patch * g.patch] = (t - 0.5f) / 0.5f;  // Normalize(0.5, 0.5)
    }
}

__global__ __launch_bounds__(256) void assemble_max_kernel(float* __restrict__ out, uint8_t* __restrict__ labels,
                                                           const float* __restrict__ pred, PatchGrid g) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.width) return;
    float m[PG_MAXC];
#pragma unroll
    for (int c = 0; c < PG_MAXC; ++c) m[c] = -INFINITY;
    const int64_t plane = (int64_t)g.patch * g.patch;
    for (int yi = 0; yi < g.ny; ++yi) {
        const int top = g.ys[yi];
        if (y < top || y >= top + g.patch) continue;
        for (int xi = 0; xi < g.nx; ++xi) {
            const int left = g.xs[xi];
            if (x < left || x >= left + g.patch) continue;
            const float* p = pred + (int64_t)(yi * g.nx + xi) * g.channels * plane + (int64_t)(y - top) * g.patch + (x - left);
#pragma unroll
            for (int c = 0; c < PG_MAXC; ++c)
                if (c < g.channels) m[c] = fmaxf(m[c], p[c * plane]);
        }
    }
    int best = 0;
#pragma unroll
    for (int c = 0; c < PG_MAXC; ++c)
        if (c < g.channels) {
            out[((int64_t)c * g.height + y) * g.width + x] = m[c];
            if (m[c] > m[best]) best = c;
        }
    if (labels) labels[(int64_t)y * g.width + x] = (uint8_t)best;
}

int check_grid(const char* who, const PatchGrid& g) {
    SIS_REQUIRE(g.xs && g.ys, "%s: null patch grid", who);
    SIS_REQUIRE(g.nx > 0 && g.ny > 0 && g.patch > 0 && g.height > 0 && g.width > 0, "%s: non-positive size", who);
    SIS_REQUIRE(g.channels >= 1 && g.channels <= PG_MAXC, "%s: %d channels outside 1..%d", who, g.channels, PG_MAXC);
    SIS_REQUIRE((int64_t)g.nx * g.ny <= 65535, "%s: more than 65535 patches", who);
    return 0;
}

}  // namespace

extern "C" int sis_crop_patches_u8(float* out, const uint8_t* image, const int* xs, const int* ys, int nx, int ny,
                                   int height, int width, int channels, int patch, void* stream) {
    PatchGrid g{xs, ys, nx, ny, patch, height, width, channels};
    if (check_grid("sis_crop_patches_u8", g)) return 1;
    SIS_REQUIRE(out && image, "sis_crop_patches_u8: null pointer");
    SIS_REQUIRE(patch <= 65535, "sis_crop_patches_u8: patch too large");
    hipLaunchKernelGGL(crop_patches_kernel, dim3(sis_cdiv(patch, 256), patch, nx * ny), dim3(256), 0, (hipStream_t)stream,
                       out, image, g);
    SIS_CHECK_LAUNCH("crop_patches_kernel");
    return 0;
}

extern "C" int sis_assemble_max(float* out, uint8_t* labels, const float* pred, const int* xs, const int* ys, int nx,
                                int ny, int classes, int height, int width, int patch, void* stream) {
    PatchGrid g{xs, ys, nx, ny, patch, height, width, classes};
    if (check_grid("sis_assemble_max", g)) return 1;
    SIS_REQUIRE(out && pred, "sis_assemble_max: null pointer");
    SIS_REQUIRE(height <= 65535, "sis_assemble_max: image too tall");
    hipLaunchKernelGGL(assemble_max_kernel, dim3(sis_cdiv(width, 256), height), dim3(256), 0, (hipStream_t)stream, out,
                       labels, pred, g);
    SIS_CHECK_LAUNCH("assemble_max_kernel");
    return 0;
}
