// edges.hip — the reference's edge extractor on the GPU (C ABI: include/rsreg.h, "edge features").
//
// Reference: src/edge_extractor.hpp:7-39.  extract_edge_features() runs integral-image normals and
// pcl::OrganizedEdgeFromRGBNormals with all five edge types on, but returns only label_indices[4]
// (:36-38), the points labelled EDGELABEL_RGB_CANNY.  That label comes from
// OrganizedEdgeFromRGB::extractEdges alone: pcl::Edge::detectEdgeCanny (thresholds 40 / 100) on the
// gray image float((r + g + b) / 3) of the organized cloud -- the normals, the depth discontinuities
// and the curvature edges never reach the returned cloud, so they are not computed here.
//
// Stages (float operations un-fused and in PCL's order, file built with -ffp-contract=off):  gray + 3x3 Gaussian
// (clamped borders)  ->  Sobel x / y, magnitude, direction class  ->  non-maximum suppression on the interior  ->
// hysteresis as connected components (8-neighbourhood union-find; a component is kept if it holds a pixel >= the high
// threshold: the set PCL's recursive tracing reaches, independent of its visiting order)  ->  ordered compaction of the
// edge points' records.  Four launches: k_edge_tile (everything up to the components inside a 32 x 32 tile, through
// LDS), k_edge_cc_borders (the joins across tile edges), k_edge_cc_roots, k_edge_compact (flag, scan and gather).
#include <cstring>
#include <string.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "records.hpp"
#include "compact.hpp"

using namespace rsreg;

struct rsreg_cloud;
extern "C" {
const void *rsreg_cloud_device_ptr(const rsreg_cloud *c);
int rsreg_cloud_info(const rsreg_cloud *c, size_t *n, size_t *stride, uint32_t *width, uint32_t *height, int *is_dense);
int rsreg_cloud_adopt_(rsreg_cloud *c, DevBuf *buf, size_t n, size_t stride, uint32_t width, uint32_t height, int is_dense);   // cloud.hip
const rsreg_ctx *rsreg_cloud_ctx_(const rsreg_cloud *c);   // cloud.hip: the context a handle belongs to
}

namespace {

struct Kernel3 {
    float k[9];
};

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

#ifdef RSREG_DIAG   // the three image stages as launches of their own: only for the stage-by-stage dump (RSREG_EDGE_DUMP); the product runs k_edge_tile
// gray image, then pcl::Convolution with the 3x3 Gaussian: correlation, borders clamped, float sum over kernel rows then columns
__global__ __launch_bounds__(kBlock) void k_edge_smooth(const char *rec, size_t stride, int w, int h, Kernel3 kg, float *sm)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= w * h) return;
    const int i = p / w, j = p - i * w;
    float s = 0.0f;
#pragma unroll
    for (int kr = 0; kr < 3; ++kr)
#pragma unroll
        for (int kc = 0; kc < 3; ++kc) {
            const int r = clampi(i + kr - 1, h - 1), c = clampi(j + kc - 1, w - 1);
            const uint32_t col = *reinterpret_cast<const uint32_t *>(rec + (size_t)(r * w + c) * stride + 16);   // b g r a
            const float g = (float)(((int)((col >> 16) & 255u) + (int)((col >> 8) & 255u) + (int)(col & 255u)) / 3);
            s = __fadd_rn(s, __fmul_rn(kg.k[kr * 3 + kc], g));
        }
    sm[p] = s;
}

// Sobel x / y (same convolution), magnitude, direction discretised like pcl::Edge::discretizeAngles
// (class 0 / 1 / 2 / 3 = 0 / 45 / 90 / 135 degrees, 255 = none of them: a NaN direction)
__global__ __launch_bounds__(kBlock) void k_edge_sobel(const float *sm, int w, int h, float *mag, uint8_t *dir)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= w * h) return;
    const int i = p / w, j = p - i * w;
    const float kx[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1}, ky[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
    float gx = 0.0f, gy = 0.0f;
#pragma unroll
    for (int kr = 0; kr < 3; ++kr)
#pragma unroll
        for (int kc = 0; kc < 3; ++kc) {
            const float v = sm[clampi(i + kr - 1, h - 1) * w + clampi(j + kc - 1, w - 1)];
            gx = __fadd_rn(gx, __fmul_rn(kx[kr * 3 + kc], v));
            gy = __fadd_rn(gy, __fmul_rn(ky[kr * 3 + kc], v));
        }
    mag[p] = sqrtf(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)));   // (sqrtf is correctly rounded under hipcc's default; __fsqrt_rn is the 1-ulp native one)
    // the angle as the correctly rounded float of atan2 (PCL's atan2f is libm-dependent in its last ulp: the
    // double-precision result rounded once is what a correctly rounded atan2f returns, on every platform)
    const float angle = __fmul_rn((float)atan2((double)gy, (double)gx), 57.29578f);   // pcl::rad2deg(float)
    uint8_t d = 255;
    if (((angle <= 22.5f) && (angle >= -22.5f)) || (angle >= 157.5f) || (angle <= -157.5f)) d = 0;
    else if (((angle > 22.5f) && (angle < 67.5f)) || ((angle < -112.5f) && (angle > -157.5f))) d = 1;
    else if (((angle >= 67.5f) && (angle <= 112.5f)) || ((angle <= -67.5f) && (angle >= -112.5f))) d = 2;
    else if (((angle > 112.5f) && (angle < 157.5f)) || ((angle < -22.5f) && (angle > -67.5f))) d = 3;
    dir[p] = d;
}
#endif

__device__ __forceinline__ int cc_find(const int *label, int i)
{
    int l = label[i];
    while (l != i) {
        i = l;
        l = label[i];
    }
    return i;
}

__device__ __forceinline__ void cc_union(int *label, int a, int b)
{
    bool done = false;
    while (!done) {
        a = cc_find(label, a);
        b = cc_find(label, b);
        if (a < b) {
            const int old = atomicMin(&label[b], a);
            done = old == b;
            b = old;
        } else if (b < a) {
            const int old = atomicMin(&label[a], b);
            done = old == a;
            a = old;
        } else {
            done = true;
        }
    }
}

// 8-neighbourhood components of the kept pixels (the hysteresis of the Canny edges: a component with one pixel at or above
// the high threshold is kept whole).  Every pixel joins its W, NW, N and NE neighbours -- in two steps: first inside tiles
// of 32 x 32 pixels, in LDS (a union is a chain of dependent finds and an atomic minimum: a few hundred cycles each in LDS,
// microseconds each in HBM -- one kernel doing all of them in HBM took 56 us for a 640 x 480 frame), then only the joins
// that cross a tile's edge, one pixel in sixteen, on the labels in HBM.  A root is the smallest pixel index of its
// component either way.
constexpr int kCcTile = 32;

__device__ __forceinline__ int cc_find_lds(const int *label, int i)
{
    int l = label[i];
    while (l != i) {
        i = l;
        l = label[i];
    }
    return i;
}

__device__ __forceinline__ void cc_union_lds(int *label, int a, int b)
{
    bool done = false;
    while (!done) {
        a = cc_find_lds(label, a);
        b = cc_find_lds(label, b);
        if (a < b) {
            const int old = atomicMin(&label[b], a);
            done = old == b;
            b = old;
        } else if (b < a) {
            const int old = atomicMin(&label[a], b);
            done = old == a;
            a = old;
        } else {
            done = true;
        }
    }
}

// Gray image, 3x3 Gaussian, Sobel, direction classes, non-maximum suppression and the components inside the tile -- one
// workgroup per tile of 32 x 32 pixels, one pixel per thread, the stages handed on through LDS with the halo each needs
// (gray 38 x 38, smoothed 36 x 36, magnitude and direction 34 x 34).  The same operations in the same order as
// pcl::Convolution / pcl::Edge (see the stage kernels above, which the dump build still has): a value outside the image
// is the value at the clamped coordinate, stage by stage, so an LDS entry stands for "the pixel its coordinate clamps to".
// Four launches, three image-sized round trips through HBM and a separate label pass less than the staged form.
// mx[p]: the kept magnitude or 0; label[p] = the smallest pixel index of p's component inside its tile, -1 if not kept;
// strong[p] = 0 (k_edge_cc_roots marks the roots).  Workgroup 0 also clears `clear_words` words for k_edge_compact.
constexpr int kEtG = kCcTile + 6, kEtS = kCcTile + 4, kEtM = kCcTile + 2;
__global__ __launch_bounds__(kCcTile * kCcTile) void k_edge_tile(const char *rec, size_t stride, int w, int h, int tiles_x, Kernel3 kg, float t_low,
                                                                 float *mx, int *label, uint32_t *strong, uint32_t *clear, uint32_t clear_words)
{
    __shared__ float s_g[kEtG][kEtG], s_s[kEtS][kEtS], s_m[kEtM][kEtM];
    __shared__ uint8_t s_d[kEtM][kEtM];
    __shared__ int lab[kCcTile * kCcTile];   // local index ly * 32 + lx: ordered like the pixel index inside the tile
    const int t = threadIdx.x;
    if (blockIdx.x == 0)
        for (uint32_t k = (uint32_t)t; k < clear_words; k += kCcTile * kCcTile) clear[k] = 0u;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x0 = tx * kCcTile, y0 = ty * kCcTile;
    for (int e = t; e < kEtG * kEtG; e += kCcTile * kCcTile) {
        const int r = e / kEtG, c = e - r * kEtG;
        const int gi = clampi(y0 - 3 + r, h - 1), gj = clampi(x0 - 3 + c, w - 1);
        const uint32_t col = *reinterpret_cast<const uint32_t *>(rec + (size_t)(gi * w + gj) * stride + 16);   // b g r a
        s_g[r][c] = (float)(((int)((col >> 16) & 255u) + (int)((col >> 8) & 255u) + (int)(col & 255u)) / 3);
    }
    __syncthreads();
    for (int e = t; e < kEtS * kEtS; e += kCcTile * kCcTile) {
        const int r = e / kEtS, c = e - r * kEtS;
        const int ci = clampi(y0 - 2 + r, h - 1), cj = clampi(x0 - 2 + c, w - 1);   // the pixel this entry stands for
        float s = 0.0f;
#pragma unroll
        for (int kr = 0; kr < 3; ++kr)
#pragma unroll
            for (int kc = 0; kc < 3; ++kc)
                s = __fadd_rn(s, __fmul_rn(kg.k[kr * 3 + kc], s_g[clampi(ci + kr - 1, h - 1) - (y0 - 3)][clampi(cj + kc - 1, w - 1) - (x0 - 3)]));
        s_s[r][c] = s;
    }
    __syncthreads();
    for (int e = t; e < kEtM * kEtM; e += kCcTile * kCcTile) {
        const int r = e / kEtM, c = e - r * kEtM;
        const int ci = clampi(y0 - 1 + r, h - 1), cj = clampi(x0 - 1 + c, w - 1);
        const float kx[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1}, ky[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
        float gx = 0.0f, gy = 0.0f;
#pragma unroll
        for (int kr = 0; kr < 3; ++kr)
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) {
                const float v = s_s[clampi(ci + kr - 1, h - 1) - (y0 - 2)][clampi(cj + kc - 1, w - 1) - (x0 - 2)];
                gx = __fadd_rn(gx, __fmul_rn(kx[kr * 3 + kc], v));
                gy = __fadd_rn(gy, __fmul_rn(ky[kr * 3 + kc], v));
            }
        s_m[r][c] = sqrtf(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)));   // (sqrtf is correctly rounded under hipcc's default; __fsqrt_rn is the 1-ulp native one)
        // the angle as the correctly rounded float of atan2 (PCL's atan2f is libm-dependent in its last ulp: the double-precision
        // result rounded once is what a correctly rounded atan2f returns, on every platform), in degrees like pcl::rad2deg(float);
        // classes as pcl::Edge::discretizeAngles: 0 / 1 / 2 / 3 = 0 / 45 / 90 / 135 degrees, 255 = none of them (a NaN direction)
        const float angle = __fmul_rn((float)atan2((double)gy, (double)gx), 57.29578f);
        uint8_t d = 255;
        if (((angle <= 22.5f) && (angle >= -22.5f)) || (angle >= 157.5f) || (angle <= -157.5f)) d = 0;
        else if (((angle > 22.5f) && (angle < 67.5f)) || ((angle < -112.5f) && (angle > -157.5f))) d = 1;
        else if (((angle >= 67.5f) && (angle <= 112.5f)) || ((angle <= -67.5f) && (angle >= -112.5f))) d = 2;
        else if (((angle > 112.5f) && (angle < 157.5f)) || ((angle < -22.5f) && (angle > -67.5f))) d = 3;
        s_d[r][c] = d;
    }
    __syncthreads();
    const int lx = t % kCcTile, ly = t / kCcTile, x = x0 + lx, y = y0 + ly;
    const bool in_image = x < w && y < h;
    float out = 0.0f;
    if (in_image && y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {   // pcl::Edge::suppressNonMaxima: interior pixels only
        const float m = s_m[ly + 1][lx + 1];
        const uint8_t d = s_d[ly + 1][lx + 1];
        if (!(m < t_low) && d != 255) {
            // the two neighbours along the gradient: (j-1, j+1) / (i-1 j-1, i+1 j+1) / (i-1, i+1) / (i-1 j+1, i+1 j-1)
            const int dy = d == 0 ? 0 : 1, dx = d == 0 ? 1 : (d == 1 ? 1 : (d == 2 ? 0 : -1));
            if (m >= s_m[ly + 1 - dy][lx + 1 - dx] && m >= s_m[ly + 1 + dy][lx + 1 + dx]) out = m;
        }
    }
    const bool kept = out != 0.0f;
    lab[t] = kept ? t : -1;
    __syncthreads();
    if (kept) {   // (a kept pixel's entry only ever moves to a smaller kept index: never negative)
        if (lx > 0 && lab[t - 1] >= 0) cc_union_lds(lab, t, t - 1);
        if (ly > 0) {
            if (lx > 0 && lab[t - kCcTile - 1] >= 0) cc_union_lds(lab, t, t - kCcTile - 1);
            if (lab[t - kCcTile] >= 0) cc_union_lds(lab, t, t - kCcTile);
            if (lx < kCcTile - 1 && lab[t - kCcTile + 1] >= 0) cc_union_lds(lab, t, t - kCcTile + 1);
        }
    }
    __syncthreads();
    if (in_image) {
        const int p = y * w + x;
        int l = -1;
        if (kept) {
            const int r = cc_find_lds(lab, t);
            l = (y0 + r / kCcTile) * w + x0 + r % kCcTile;
        }
        mx[p] = out;
        label[p] = l;
        strong[p] = 0u;
    }
}

// The joins across tile edges: the pixels of a tile's first row, first column and last column (the NE neighbour of those
// lies in the next tile) -- 94 of a tile's 1 024, one thread each.
constexpr int kCcEdge = 3 * kCcTile - 2;
__global__ __launch_bounds__(kBlock) void k_edge_cc_borders(const float *mx, int w, int h, int tiles_x, int n_tiles, int *label)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_tiles * kCcEdge) return;
    const int tile = e / kCcEdge, k = e - tile * kCcEdge;
    int lx, ly;
    if (k < kCcTile) { lx = k; ly = 0; }                                              // first row
    else if (k < 2 * kCcTile - 1) { lx = 0; ly = k - kCcTile + 1; }                   // first column, rows 1..31
    else { lx = kCcTile - 1; ly = k - (2 * kCcTile - 1) + 1; }                        // last column, rows 1..31
    const int j = (tile % tiles_x) * kCcTile + lx, i = (tile / tiles_x) * kCcTile + ly;
    if (i >= h || j >= w) return;
    const int p = i * w + j;
    if (mx[p] == 0.0f) return;
    if (lx == 0 && j > 0 && mx[p - 1] != 0.0f) cc_union(label, p, p - 1);
    if (i > 0) {
        if ((lx == 0 || ly == 0) && j > 0 && mx[p - w - 1] != 0.0f) cc_union(label, p, p - w - 1);
        if (ly == 0 && mx[p - w] != 0.0f) cc_union(label, p, p - w);
        if ((lx == kCcTile - 1 || ly == 0) && j < w - 1 && mx[p - w + 1] != 0.0f) cc_union(label, p, p - w + 1);
    }
}

// root of every kept pixel; a root whose component holds a pixel at or above the high threshold is marked strong
__global__ __launch_bounds__(kBlock) void k_edge_cc_roots(const float *mx, int n, float t_high, int *label, uint32_t *strong)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || mx[p] == 0.0f) return;
    const int r = cc_find(label, p);
    if (!(mx[p] < t_high)) strong[r] = 1u;
    __builtin_nontemporal_store(r, &label[p]);   // (a pixel's label only ever points at a smaller index of its own component)
}

// pcl::copyPointCloud(cloud, indices, out): the records of the pixels whose component is strong, in index order -- flag,
// scan and gather in one launch (compact.hpp: a workgroup flags 4 096 pixels, scans them, learns what lies in front of it
// through the decoupled look-back and copies its records); the last workgroup leaves the number of edge points in a
// pinned host word (no copy queued).  `state` / `ticket`: zero at the start (k_edge_tile clears them).
__global__ __launch_bounds__(kCompactBlock) void k_edge_compact(const char *rec, size_t stride, int n, const float *mx, const int *label,
                                                                const uint32_t *strong, char *out, int32_t *indices, unsigned long long *state,
                                                                uint32_t *ticket, uint32_t *host_count)
{
    __shared__ uint32_t s_bid;
    __shared__ unsigned long long s_wave[kCompactBlock / 64];
    __shared__ unsigned long long s_excl;
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;
    const int i0 = (int)((bid * kCompactBlock + threadIdx.x) * kCompactItems);
    uint32_t f[kCompactItems];
    unsigned long long mine = 0;
#pragma unroll
    for (int j = 0; j < (int)kCompactItems; ++j) {
        const int p = i0 + j;
        f[j] = 0u;
        if (p < n && mx[p] != 0.0f) f[j] = strong[cc_find(label, p)];
        mine += f[j];
    }
    unsigned long long total;
    unsigned long long run = compact_block_scan(mine, s_wave, total);
    const unsigned long long before = compact_lookback(state, bid, total, &s_excl);
    run += before;
    if (bid == gridDim.x - 1 && threadIdx.x == 0) *host_count = (uint32_t)(before + total);
#pragma unroll
    for (int j = 0; j < (int)kCompactItems; ++j) {
        if (!f[j]) continue;
        const int p = i0 + j;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(rec + (size_t)p * stride);
        uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)run * stride);
        for (uint32_t k = 0; k < stride / 4; ++k) dst[k] = src[k];
        if (indices) indices[run] = p;
        ++run;
    }
}

}  // namespace

namespace rsreg {

// records in HBM -> edge points in the scratch set's `out` (+ their indices in its vals_alt); one host sync.
// side_set >= 0: that scratch set of the context and its stream (rsreg_cloud_edge_features_async: the features of the next
// frame beside the alignment of this one, queued by the side worker's thread); -1: the main set, the main stream.
int edge_features_device(rsreg_ctx *ctx, const char *d_rec, size_t stride, uint32_t width, uint32_t height, uint32_t *n_out, int side_set)
{
    *n_out = 0;
    const size_t n = (size_t)width * height;
    if (n == 0) return RSREG_OK;
    if (n > 0x3fffffffull) return fail(ctx, RSREG_ERR_INVALID_ARG, "image too large");
    const bool side = side_set >= 0;
    rsreg_ctx::SideSet &ss = ctx->side_sets[side ? side_set : 0];
    hipStream_t st = side ? ss.stream : ctx->stream;
    DevBuf &b_keys_alt = side ? ss.keys_alt : ctx->d_keys_alt, &b_vals = side ? ss.vals : ctx->d_vals, &b_flags = side ? ss.flags : ctx->d_flags,
           &b_vals_alt = side ? ss.vals_alt : ctx->d_vals_alt, &b_out = side ? ss.out : ctx->d_vox_out;
    PinnedBuf &b_host = side ? ss.host : ctx->h_sums;
    const int w = (int)width, h = (int)height, N = (int)n;
    const CompactPlan cplan = compact_plan(n, 0);   // the look-back words and the ticket of the compaction at the end
    RSREG_HIP(ctx, b_keys_alt.reserve(n * 8));   // maxima | labels
    RSREG_HIP(ctx, b_vals.reserve(n * 4));       // strong flags per root
    RSREG_HIP(ctx, b_flags.reserve((size_t)cplan.end * 4 + 16));
    RSREG_HIP(ctx, b_vals_alt.reserve(n * 4));   // indices of the edge points
    RSREG_HIP(ctx, b_out.reserve(n * stride));
    RSREG_HIP(ctx, b_host.reserve(2048));
    float *mx = b_keys_alt.as<float>();
    int *label = reinterpret_cast<int *>(mx + n);
    uint32_t *strong = b_vals.as<uint32_t>();
    // pcl::kernel::gaussianKernel(size 3, sigma 1): exp of -(i^2 + j^2) / (2 sigma^2) as a float, normalised by the float
    // sum.  (PCL calls expf at run time -- libm-dependent in the last ulp; here the correctly rounded float: exp in
    // double, rounded once, the same on every platform and in the checker.)
    Kernel3 kg;
    float sum = 0.0f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int iks = i - 1, jks = j - 1;
            kg.k[i * 3 + j] = (float)std::exp(-(double)(iks * iks + jks * jks) / 2.0);
            sum += kg.k[i * 3 + j];
        }
    for (int i = 0; i < 9; ++i) kg.k[i] /= sum;
    const float t_low = 40.0f, t_high = 100.0f;   // OrganizedEdgeFromRGB: th_rgb_canny_low_ / _high_ (never changed by the reference)
    const uint32_t nb = div_up((uint32_t)N, kBlock);
    const int tiles_x = (w + kCcTile - 1) / kCcTile, tiles_y = (h + kCcTile - 1) / kCcTile;
    uint32_t *scr = b_flags.as<uint32_t>();   // (the compaction's words: cleared by the tile kernel)
    uint32_t *hb = b_host.as<uint32_t>() + 200;
    k_edge_tile<<<(uint32_t)(tiles_x * tiles_y), kCcTile * kCcTile, 0, st>>>(d_rec, stride, w, h, tiles_x, kg, t_low, mx, label, strong, scr, cplan.end);
    RSREG_HIP(ctx, hipGetLastError());
    k_edge_cc_borders<<<(uint32_t)((tiles_x * tiles_y * kCcEdge + kBlock - 1) / kBlock), kBlock, 0, st>>>(mx, w, h, tiles_x, tiles_x * tiles_y, label);
    RSREG_HIP(ctx, hipGetLastError());
    k_edge_cc_roots<<<nb, kBlock, 0, st>>>(mx, N, t_high, label, strong);
    RSREG_HIP(ctx, hipGetLastError());
    k_edge_compact<<<cplan.blocks, kCompactBlock, 0, st>>>(d_rec, stride, N, mx, label, strong, b_out.as<char>(), b_vals_alt.as<int32_t>(),
                                                           reinterpret_cast<unsigned long long *>(scr + cplan.off_state), scr + cplan.off_ticket, hb);
    RSREG_HIP(ctx, hipGetLastError());
#ifdef RSREG_DIAG
    if (const char *dump = rsreg::tunables().edge_dump) {   // diagnostic builds: the stage images, for a stage-by-stage comparison
        DevBuf &b_keys = side ? ss.keys : ctx->d_keys, &b_dir = side ? ss.cent : ctx->d_brick;
        RSREG_HIP(ctx, b_keys.reserve(n * 8));       // smoothed | magnitude
        RSREG_HIP(ctx, b_dir.reserve(n + 16));       // direction classes
        float *sm = b_keys.as<float>(), *mag = sm + n;
        uint8_t *dir = b_dir.as<uint8_t>();
        k_edge_smooth<<<nb, kBlock, 0, st>>>(d_rec, stride, w, h, kg, sm);
        k_edge_sobel<<<nb, kBlock, 0, st>>>(sm, w, h, mag, dir);
        std::vector<float> hbuf(n * 3);
        std::vector<uint8_t> hdir(n);
        (void)hipMemcpyAsync(hbuf.data(), sm, n * 4, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(hbuf.data() + n, mag, n * 4, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(hbuf.data() + 2 * n, mx, n * 4, hipMemcpyDeviceToHost, st);
        (void)hipMemcpyAsync(hdir.data(), dir, n, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        if (FILE *f = std::fopen(dump, "wb")) {
            std::fwrite(hbuf.data(), 4, hbuf.size(), f);
            std::fwrite(hdir.data(), 1, n, f);
            std::fclose(f);
        }
    }
#endif
    RSREG_HIP(ctx, hipStreamSynchronize(st));
    *n_out = hb[0];
    return RSREG_OK;
}

}  // namespace rsreg

extern "C" {

// extract_edge_features(cloud) for an organized cloud handed over on the host
int rsreg_extract_edge_features(rsreg_ctx *ctx, const void *points, uint32_t width, uint32_t height, size_t stride, void *out,
                                int32_t *indices_out, size_t *n_out)
{
    const size_t n = (size_t)width * height;
    if (!ctx || !n_out || (n && (!points || !out)) || stride < 20 || (stride & 3)) return RSREG_ERR_INVALID_ARG;
    *n_out = 0;
    if (n == 0) return RSREG_OK;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, ctx->d_vox_in.reserve(n * stride));
    RSREG_HIP(ctx, ctx->h_stage.reserve(n * stride));
    {
        char *stage = ctx->h_stage.as<char>();
        const char *src = static_cast<const char *>(points);
        host_parallel_for(n, [=](size_t lo, size_t hi) { rsreg::stream_copy(stage + lo * stride, src + lo * stride, (hi - lo) * stride); });
    }
    RSREG_HIP(ctx, hipMemcpyAsync(ctx->d_vox_in.ptr, ctx->h_stage.ptr, n * stride, hipMemcpyHostToDevice, ctx->stream));
    uint32_t ne = 0;
    int rc = rsreg::edge_features_device(ctx, ctx->d_vox_in.as<char>(), stride, width, height, &ne, -1);
    if (rc || ne == 0) return rc;
    RSREG_HIP(ctx, hipMemcpyAsync(ctx->h_stage.ptr, ctx->d_vox_out.ptr, (size_t)ne * stride, hipMemcpyDeviceToHost, ctx->stream));
    if (indices_out) RSREG_HIP(ctx, hipMemcpyAsync(indices_out, ctx->d_vals_alt.ptr, (size_t)ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->h_stage.ptr, (size_t)ne * stride);
    *n_out = ne;
    return RSREG_OK;
}

// the same on a cloud resident in HBM; out: width = number of edge points, height = 1, is_dense as the input's
// (pcl::copyPointCloud (cloud, indices, out))
int rsreg_cloud_edge_features(rsreg_ctx *ctx, const rsreg_cloud *in, rsreg_cloud *out)
{
    if (!ctx || !in || !out) return RSREG_ERR_INVALID_ARG;
    // like every other rsreg_cloud_* operation: both handles belong to THIS context (its stream, scratch and device)
    if (rsreg_cloud_ctx_(in) != ctx || rsreg_cloud_ctx_(out) != ctx) return RSREG_ERR_INVALID_ARG;
    size_t n = 0, stride = 0;
    uint32_t w = 0, h = 0;
    int dense = 0;
    int rc = rsreg_cloud_info(in, &n, &stride, &w, &h, &dense);
    if (rc) return rc;
    if ((size_t)w * h != n || stride < 20) return fail(ctx, RSREG_ERR_INVALID_ARG, "edge extraction needs an organized XYZRGB cloud");
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t ne = 0;
    rc = rsreg::edge_features_device(ctx, static_cast<const char *>(rsreg_cloud_device_ptr(in)), stride, w, h, &ne, -1);
    if (rc) return rc;
    return rsreg_cloud_adopt_(out, &ctx->d_vox_out, ne, stride, ne, 1, dense);
}

}  // extern "C"
