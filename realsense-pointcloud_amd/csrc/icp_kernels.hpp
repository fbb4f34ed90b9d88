// icp_kernels.hpp — device code of the ICP path (gfx950, wave64).  Included by icp.hip only.
//
// Kernels (each cites the PCL step it replaces; reference call sites in include/rsreg.h):
//   grid build   k_bbox / k_cell_keys / k_flag_runs / k_scatter_sorted / k_table_insert
//                  = KdTreeFLANN build in Registration::initCompute
//   k_init_source     input_transformed = guess * input               (App. A.2 prologue)
//   k_nn_search       CorrespondenceEstimation::determineCorrespondences (App. A.1, A.7a)
//   k_cov_reduce + k_final_reduce   the sums Eigen::umeyama needs     (App. A.3)
//   k_transform       ICP transformCloud in place                     (App. A.2)
//   k_icp_fused       transform + NN + reject + sums in one pass
// Float arithmetic that decides a nearest neighbour or a gate is written with explicit
// un-fused operations in a fixed order (the file is also built with -ffp-contract=off):
//   d2 = ((dx*dx + dy*dy) + dz*dz)            FLANN L2_Simple<float>
//   x' = ((m00*x + m01*y) + m02*z) + m03      PCL transformCloud
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "rsreg_ctx.hpp"

namespace rsreg {

constexpr int kBlock = 256;            // 4 waves of 64
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr float kRingSafety = 0.96875f;  // ring r proves distances up to r*cell*kRingSafety

struct GridDev {
    float ox, oy, oz, inv_cell, cell;
    int dx, dy, dz;
    uint32_t mask;
    int max_ring;
    const CellEntry *table;
    const float4 *pts;
};

struct Mat34 {  // rows of the 3x4 part of a column-major Mat4f, passed by value to kernels
    float r0[4], r1[4], r2[4];
};

inline Mat34 to_mat34(const Mat4f &T)
{
    Mat34 m;
    for (int c = 0; c < 4; ++c) { m.r0[c] = T(0, c); m.r1[c] = T(1, c); m.r2[c] = T(2, c); }
    return m;
}

__device__ __forceinline__ float3 xform(const Mat34 &m, float x, float y, float z)
{
    float ox = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m.r0[0], x), __fmul_rn(m.r0[1], y)), __fmul_rn(m.r0[2], z)), m.r0[3]);
    float oy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m.r1[0], x), __fmul_rn(m.r1[1], y)), __fmul_rn(m.r1[2], z)), m.r1[3]);
    float oz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m.r2[0], x), __fmul_rn(m.r2[1], y)), __fmul_rn(m.r2[2], z)), m.r2[3]);
    return make_float3(ox, oy, oz);
}

__device__ __forceinline__ float l2_simple(float qx, float qy, float qz, float tx, float ty, float tz)
{
    const float dx = __fsub_rn(qx, tx), dy = __fsub_rn(qy, ty), dz = __fsub_rn(qz, tz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return isfinite(x) && isfinite(y) && isfinite(z);
}

__device__ __forceinline__ int cell_coord(float p, float origin, float inv_cell)
{
    // same expression in the build and in every query: a point and a query with equal
    // coordinates always land in the same cell
    float v = floorf(__fmul_rn(__fsub_rn(p, origin), inv_cell));
    v = fminf(fmaxf(v, -4.0f), 70000.0f);
    return (int)v;
}

__device__ __forceinline__ unsigned long long pack_cell(int x, int y, int z)
{
    return ((unsigned long long)(unsigned)z << 32) | ((unsigned long long)(unsigned)y << 16) |
           (unsigned long long)(unsigned)x;
}

__device__ __forceinline__ uint32_t hash_cell(unsigned long long k)
{
    k ^= k >> 29;
    k *= 0xbf58476d1ce4e5b9ull;
    k ^= k >> 32;
    return (uint32_t)k;
}

__device__ __forceinline__ uint32_t hash_xyz16(float x, float y, float z)
{
    // +0.0f folds -0 into +0 so that value-equal points hash alike
    uint32_t a = __float_as_uint(x + 0.0f), b = __float_as_uint(y + 0.0f), c = __float_as_uint(z + 0.0f);
    uint32_t h = a * 0x9e3779b1u;
    h = (h ^ (h >> 15)) + b * 0x85ebca77u;
    h = (h ^ (h >> 13)) + c * 0xc2b2ae3du;
    h ^= h >> 16;
    return h & 0xffffu;
}

// order-preserving float <-> uint map for atomicMin/atomicMax on floats
__device__ __forceinline__ uint32_t float_ordered(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float ordered_float(uint32_t u)
{
    uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

__device__ __forceinline__ const float *rec_xyz(const char *base, size_t stride, size_t i)
{
    return reinterpret_cast<const float *>(base + i * stride);
}

// ------------------------------------------------------------------------------ grid build
// bbox[0..2] = min (ordered uint), bbox[3..5] = max, bbox[6] = number of finite points
__global__ __launch_bounds__(kBlock) void k_bbox(const char *pts, size_t stride, uint32_t n, uint32_t *bbox)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    uint32_t cnt = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *p = rec_xyz(pts, stride, i);
        const float x = p[0], y = p[1], z = p[2];
        if (finite3(x, y, z)) {
            mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
            mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
            ++cnt;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], off));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off));
        }
        cnt += __shfl_down(cnt, off);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        for (int k = 0; k < 3; ++k) {
            atomicMin(&bbox[k], float_ordered(mn[k]));
            atomicMax(&bbox[3 + k], float_ordered(mx[k]));
        }
        atomicAdd(&bbox[6], cnt);
    }
}

// sort key: [cell z:16 | y:16 | x:16 | hash16(xyz)]; non-finite points sort to the very end
__global__ __launch_bounds__(kBlock) void k_cell_keys(const char *pts, size_t stride, uint32_t n, GridDev g,
                                                      unsigned long long *keys, uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = rec_xyz(pts, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    unsigned long long key = kEmptyKey;
    if (finite3(x, y, z)) {
        const int cx = cell_coord(x, g.ox, g.inv_cell), cy = cell_coord(y, g.oy, g.inv_cell),
                  cz = cell_coord(z, g.oz, g.inv_cell);
        key = (pack_cell(cx, cy, cz) << 16) | hash_xyz16(x, y, z);
    }
    keys[i] = key;
    vals[i] = i;
}

// keep[i]: not a value-equal duplicate of its predecessor in the same (cell, hash) run;
// cstart[i]: first point of a cell
__global__ __launch_bounds__(kBlock) void k_flag_runs(const unsigned long long *keys, const uint32_t *vals,
                                                      const char *pts, size_t stride, uint32_t nfin,
                                                      uint32_t *keep, uint32_t *cstart)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    const unsigned long long k = keys[i];
    uint32_t kp = 1, cs = 1;
    if (i > 0) {
        const unsigned long long kprev = keys[i - 1];
        cs = (k >> 16) != (kprev >> 16);
        if (k == kprev) {
            const float *a = rec_xyz(pts, stride, vals[i]);
            const float *b = rec_xyz(pts, stride, vals[i - 1]);
            if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) kp = 0;
        }
    }
    keep[i] = kp;
    cstart[i] = cs;
}

// pos = exclusive scan of keep, cid = exclusive scan of cstart
__global__ __launch_bounds__(kBlock) void k_scatter_sorted(const unsigned long long *keys, const uint32_t *vals,
                                                           const char *pts, size_t stride, uint32_t nfin,
                                                           const uint32_t *keep, const uint32_t *cstart,
                                                           const uint32_t *pos, const uint32_t *cid,
                                                           float4 *sorted, unsigned long long *cellkey,
                                                           uint32_t *cellpos, uint32_t *counts /*[2]*/)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    const uint32_t v = vals[i];
    if (keep[i]) {
        const float *p = rec_xyz(pts, stride, v);
        sorted[pos[i]] = make_float4(p[0], p[1], p[2], __uint_as_float(v));
    }
    if (cstart[i]) {
        cellkey[cid[i]] = keys[i] >> 16;
        cellpos[cid[i]] = pos[i];
    }
    if (i == nfin - 1) {
        const uint32_t nu = pos[i] + keep[i], nc = cid[i] + cstart[i];
        counts[0] = nu;
        counts[1] = nc;
        cellpos[nc] = nu;  // sentinel
    }
}

__global__ __launch_bounds__(kBlock) void k_table_insert(const unsigned long long *cellkey, const uint32_t *cellpos,
                                                         uint32_t ncells, CellEntry *table, uint32_t mask,
                                                         uint32_t *max_count)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    const unsigned long long key = cellkey[c];
    const uint32_t start = cellpos[c], count = cellpos[c + 1] - start;
    uint32_t slot = hash_cell(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&table[slot].key, kEmptyKey, key);
        if (prev == kEmptyKey) {
            table[slot].start = start;
            table[slot].count = count;
            break;
        }
        slot = (slot + 1) & mask;  // keys are unique, the table is at most half full
    }
    atomicMax(max_count, count);
}

// ------------------------------------------------------------------------------ source
// src = {x,y,z,valid}; cur = guess * src (or a copy when the guess is the identity)
__global__ __launch_bounds__(kBlock) void k_init_source(const char *raw, size_t stride, uint32_t n, Mat34 guess,
                                                        int apply_guess, float4 *src, float4 *cur)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = rec_xyz(raw, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    const bool ok = finite3(x, y, z);
    const float4 s = make_float4(x, y, z, ok ? 1.0f : 0.0f);
    if (src) src[i] = s;
    float4 c = s;
    if (ok && apply_guess) {
        const float3 t = xform(guess, x, y, z);
        c = make_float4(t.x, t.y, t.z, 1.0f);
    }
    cur[i] = c;
}

__global__ __launch_bounds__(kBlock) void k_restart_source(const float4 *src, uint32_t n, Mat34 guess, int apply_guess,
                                                           float4 *cur)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = src[i];
    if (s.w != 0.0f && apply_guess) {
        const float3 t = xform(guess, s.x, s.y, s.z);
        s = make_float4(t.x, t.y, t.z, 1.0f);
    }
    cur[i] = s;
}

__global__ __launch_bounds__(kBlock) void k_transform(float4 *cur, uint32_t n, Mat34 T)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = cur[i];
    if (s.w != 0.0f) {
        const float3 t = xform(T, s.x, s.y, s.z);
        cur[i] = make_float4(t.x, t.y, t.z, 1.0f);
    }
}

// out = final * src as packed float3 (the aligned cloud icp.align() hands back)
__global__ __launch_bounds__(kBlock) void k_apply_final(const float4 *src, uint32_t n, Mat34 T, float *out_xyz)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 s = src[i];
    float3 t = make_float3(s.x, s.y, s.z);
    if (s.w != 0.0f) t = xform(T, s.x, s.y, s.z);
    out_xyz[3 * i] = t.x;
    out_xyz[3 * i + 1] = t.y;
    out_xyz[3 * i + 2] = t.z;
}

// ------------------------------------------------------------------------------ NN search
struct Best {
    float d2;
    uint32_t oidx;  // original target index of the current best (tie-break: lowest wins)
    int pos;        // position in the sorted target array
};

__device__ __forceinline__ void scan_cell(const GridDev &g, int x, int y, int z, float qx, float qy, float qz, Best &b)
{
    const unsigned long long key = pack_cell(x, y, z);
    uint32_t slot = hash_cell(key) & g.mask;
    for (;;) {
        const CellEntry e = g.table[slot];
        if (e.key == key) {
            for (uint32_t p = e.start, end = e.start + e.count; p < end; ++p) {
                const float4 t = g.pts[p];
                const float d = l2_simple(qx, qy, qz, t.x, t.y, t.z);
                const uint32_t oi = __float_as_uint(t.w);
                if (d < b.d2 || (d == b.d2 && oi < b.oidx)) {
                    b.d2 = d;
                    b.oidx = oi;
                    b.pos = (int)p;
                }
            }
            return;
        }
        if (e.key == kEmptyKey) return;
        slot = (slot + 1) & g.mask;
    }
}

// Exact nearest neighbour by ring expansion over the uniform grid.  After ring r every
// target whose cell is within Chebyshev distance r of the query's cell has been seen, so
// anything unseen is farther than r*cell; the search stops when the best distance is inside
// that bound or when the rings cover the correspondence gate.
__device__ __forceinline__ Best nn_query(const GridDev &g, float qx, float qy, float qz)
{
    Best b{FLT_MAX, 0xffffffffu, -1};
    const int cx = min(max(cell_coord(qx, g.ox, g.inv_cell), -1), g.dx);
    const int cy = min(max(cell_coord(qy, g.oy, g.inv_cell), -1), g.dy);
    const int cz = min(max(cell_coord(qz, g.oz, g.inv_cell), -1), g.dz);
    for (int r = 0; r <= g.max_ring; ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= g.dz) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= g.dy) continue;
                const bool face = (abs(dz) == r) || (abs(dy) == r);
                const int step = face ? 1 : max(2 * r, 1);
                for (int x = cx - r; x <= cx + r; x += step)
                    if (x >= 0 && x < g.dx) scan_cell(g, x, y, z, qx, qy, qz, b);
            }
        }
        const float bound = (float)r * g.cell * kRingSafety;
        if (b.d2 <= bound * bound) break;
    }
    return b;
}

__global__ __launch_bounds__(kBlock) void k_nn_search(const float4 *cur, uint32_t n, GridDev g, double gate2,
                                                      int *corr_pos, float *corr_d2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = cur[i];
    int pos = -1;
    float d2 = 0.0f;
    if (q.w != 0.0f) {
        const Best b = nn_query(g, q.x, q.y, q.z);
        if (b.pos >= 0 && !((double)b.d2 > gate2)) {  // PCL: if (distance > max_dist_sqr) continue;
            pos = b.pos;
            d2 = b.d2;
        }
    }
    corr_pos[i] = pos;
    corr_d2[i] = d2;
}

// ------------------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// accumulate one accepted pair (p = source, q = target) into the 17 sums
__device__ __forceinline__ void accum_pair(double *a, float px, float py, float pz, float qx, float qy, float qz, float d2)
{
    const double P[3] = {px, py, pz}, Q[3] = {qx, qy, qz};
    a[0] += 1.0;
    for (int k = 0; k < 3; ++k) { a[1 + k] += P[k]; a[4 + k] += Q[k]; }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a[7 + r * 3 + c] += Q[r] * P[c];
    a[16] += (double)d2;
}

// block-level reduction of per-thread sums into partials[blockIdx][17], fixed order
__device__ __forceinline__ void block_reduce_store(double *a, double *partials)
{
    __shared__ double sh[kBlock / 64][RSREG_NUM_SUMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) {
        const double v = wave_sum(a[k]);
        if (lane == 0) sh[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < RSREG_NUM_SUMS) {
        double v = sh[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) v += sh[w][threadIdx.x];
        partials[(size_t)blockIdx.x * RSREG_NUM_SUMS + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(kBlock) void k_cov_reduce(const float4 *cur, const int *corr_pos, const float *corr_d2,
                                                       const float4 *tgt, uint32_t n, double *partials)
{
    double a[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) a[k] = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int pos = corr_pos[i];
        if (pos >= 0) {
            const float4 p = cur[i];
            const float4 q = tgt[pos];
            accum_pair(a, p.x, p.y, p.z, q.x, q.y, q.z, corr_d2[i]);
        }
    }
    block_reduce_store(a, partials);
}

// one block: wave w reduces sums k = w, w+4, ... over all partial slabs in a fixed order
__global__ __launch_bounds__(kBlock) void k_final_reduce(const double *partials, uint32_t nblocks, double *sums)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < RSREG_NUM_SUMS; k += kBlock / 64) {
        double v = 0.0;
        for (uint32_t b = lane; b < nblocks; b += 64) v += partials[(size_t)b * RSREG_NUM_SUMS + k];
        v = wave_sum(v);
        if (lane == 0) sums[k] = v;
    }
}

// One ICP iteration in one pass: apply the previous increment, search, gate, accumulate.
// Writes the transformed source back (next iteration starts from it, like PCL's in-place
// transformCloud) and, when asked, the correspondences.
__global__ __launch_bounds__(kBlock) void k_icp_fused(float4 *cur, uint32_t n, Mat34 T, int apply_t, GridDev g,
                                                      double gate2, int *corr_pos, float *corr_d2, double *partials)
{
    double a[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) a[k] = 0.0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 q = cur[i];
        int pos = -1;
        float d2 = 0.0f;
        if (q.w != 0.0f) {
            if (apply_t) {
                const float3 t = xform(T, q.x, q.y, q.z);
                q = make_float4(t.x, t.y, t.z, 1.0f);
                cur[i] = q;
            }
            const Best b = nn_query(g, q.x, q.y, q.z);
            if (b.pos >= 0 && !((double)b.d2 > gate2)) {
                pos = b.pos;
                d2 = b.d2;
                const float4 t = g.pts[b.pos];
                accum_pair(a, q.x, q.y, q.z, t.x, t.y, t.z, d2);
            }
        }
        if (corr_pos) { corr_pos[i] = pos; corr_d2[i] = d2; }
    }
    block_reduce_store(a, partials);
}

// corr position in the sorted array -> index in the caller's target array
__global__ __launch_bounds__(kBlock) void k_corr_to_index(const int *corr_pos, const float4 *tgt, uint32_t n, int *index_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int pos = corr_pos[i];
    index_out[i] = pos >= 0 ? (int)__float_as_uint(tgt[pos].w) : -1;
}

}  // namespace rsreg
