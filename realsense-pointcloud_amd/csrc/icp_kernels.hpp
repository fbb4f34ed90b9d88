// icp_kernels.hpp — device code of the ICP path (gfx950, wave64).  Included by icp.hip only.
//
// Kernels (each cites the PCL step it replaces; reference call sites in include/rsreg.h):
//   grid build   k_bbox / k_cell_keys / k_flag_runs / k_scatter_sorted / k_brick_insert
//                  = KdTreeFLANN build in Registration::initCompute
//   k_source_keys / k_gather_source   spatial ordering of the source (engine-internal)
//   k_restart_source  input_transformed = guess * input               (App. A.2 prologue)
//   k_nn_search       CorrespondenceEstimation::determineCorrespondences (App. A.1, A.7a)
//   k_cov_reduce + k_final_reduce   the sums Eigen::umeyama needs     (App. A.3)
//   k_transform       ICP transformCloud in place                     (App. A.2)
//   k_icp_fused       transform + NN + reject + sums in one pass
// Float arithmetic that decides a nearest neighbour or a gate is written with explicit
// un-fused operations in a fixed order (the file is also built with -ffp-contract=off):
//   d2 = ((dx*dx + dy*dy) + dz*dz)            FLANN L2_Simple<float>
//   x' = ((m00*x + m01*y) + m02*z) + m03      PCL transformCloud
//
// Target index: uniform grid of cells of edge `cell`; 4x4x4 cells form a
// brick.  Points are sorted brick-major, then by cell inside the brick (x fastest), exact
// duplicates dropped.  A hash table maps a brick coordinate to {64-bit occupancy mask, id of
// its first occupied cell}; cellpos[id] is the first sorted point of an occupied cell.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "rsreg_ctx.hpp"
#include "records.hpp"
#include "osort.hpp"
#include "compact.hpp"

namespace rsreg {

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr float kCellMargin = 0.03f;   // slack (in cells) on every geometric lower bound: covers
                                       // the float rounding of the point -> cell assignment

struct GridDev {
    float ox, oy, oz, inv_cell, cell;
    int dx, dy, dz;          // grid extent in cells
    uint32_t bmask;          // brick hash table has bmask + 1 slots
    int max_ring;            // rings (in cells) that cover the correspondence gate
    float prune2;            // squared search radius as float, rounded up (+inf: unbounded)
    const BrickEntry *bricks;
    const uint32_t *cellpos;
    const float4 *pts;
};

// State of the device-resident loop (RSREG_PIPELINE_DEVICE_LOOP): what update_from_sums keeps
// on the host, kept in HBM so that iteration k+1 can be queued before iteration k has run.
struct IcpDevState {
    Mat34 t_inc;                 // increment the next pass applies to the source
    int apply;                   // 0 until the first solve
    int iterations;              // completed solves
    int stopped;                 // an iteration had < 3 correspondences: frozen from there on
    int pad;
    Mat4f final_t;
    double sums_last[RSREG_NUM_SUMS];
    double cur_mse;
    unsigned long long ncorr;
    double svd_v[9];             // V of the previous solve: where the next Jacobi SVD starts
};

__device__ __forceinline__ float l2_simple(float qx, float qy, float qz, float tx, float ty, float tz)
{
    const float dx = __fsub_rn(qx, tx), dy = __fsub_rn(qy, ty), dz = __fsub_rn(qz, tz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// position in cell units relative to the grid origin; the SAME expression feeds the build
// and every query, so equal coordinates always land in the same cell
__device__ __forceinline__ float cell_pos(float p, float origin, float inv_cell)
{
    return __fmul_rn(__fsub_rn(p, origin), inv_cell);
}

__device__ __forceinline__ int cell_coord(float p, float origin, float inv_cell)
{
    float v = floorf(cell_pos(p, origin, inv_cell));
    v = fminf(fmaxf(v, -4.0f), 70000.0f);
    return (int)v;
}

// brick-major key of a cell: [bz:14 | by:14 | bx:14 | lz:2 | ly:2 | lx:2]
__device__ __forceinline__ unsigned long long cell_key(int x, int y, int z)
{
    const unsigned long long bx = (unsigned)x >> 2, by = (unsigned)y >> 2, bz = (unsigned)z >> 2;
    const unsigned long long bit = ((unsigned)z & 3u) << 4 | ((unsigned)y & 3u) << 2 | ((unsigned)x & 3u);
    return (((bz << 14 | by) << 14 | bx) << 6) | bit;
}

__device__ __forceinline__ unsigned long long brick_key(int bx, int by, int bz)
{
    return ((unsigned long long)(unsigned)bz << 14 | (unsigned long long)(unsigned)by) << 14 | (unsigned long long)(unsigned)bx;
}

__device__ __forceinline__ uint32_t hash_brick(unsigned long long k)
{
    k ^= k >> 23;
    k *= 0xbf58476d1ce4e5b9ull;
    k ^= k >> 29;
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
using rsreg::ordered_float;   // (the host's way back: rsreg_ctx.hpp)

// ------------------------------------------------------------------------------ grid build
// per block: partial[0..2] = min (ordered uint), [3..5] = max, [6] = number of finite points
__global__ __launch_bounds__(kBlock) void k_bbox(const char *pts, size_t stride, uint32_t n, uint32_t *partial)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    uint32_t cnt = 0;
    const bool vec = (stride % 16 == 0) && ((reinterpret_cast<size_t>(pts) & 15) == 0);   // PointXYZRGB records: one 16-B load
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float x, y, z;
        if (vec) {
            const float4 v = *reinterpret_cast<const float4 *>(pts + (size_t)i * stride);
            x = v.x; y = v.y; z = v.z;
        } else {
            const float *p = rec_xyz(pts, stride, i);
            x = p[0]; y = p[1]; z = p[2];
        }
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
    __shared__ float smn[kBlock / 64][3], smx[kBlock / 64][3];
    __shared__ uint32_t scnt[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        for (int k = 0; k < 3; ++k) { smn[wave][k] = mn[k]; smx[wave][k] = mx[k]; }
        scnt[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one partial per block: same-address atomics from 1024 blocks cost ~80 us
        for (int w = 1; w < kBlock / 64; ++w) {
            for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], smn[w][k]); mx[k] = fmaxf(mx[k], smx[w][k]); }
            cnt += scnt[w];
        }
        uint32_t *out = partial + (size_t)blockIdx.x * 8;
        for (int k = 0; k < 3; ++k) { out[k] = float_ordered(mn[k]); out[3 + k] = float_ordered(mx[k]); }
        out[6] = cnt;
    }
}

// one block: combine the per-block partials of k_bbox into bbox[0..6]
// (and clears the 16 device-side counter words the builds that follow use)
__global__ __launch_bounds__(kBlock) void k_bbox_final(const uint32_t *partial, uint32_t nblocks, uint32_t *bbox, uint32_t *counters)
{
    if (threadIdx.x < 16) counters[threadIdx.x] = 0u;
    uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0, 0, 0}, cnt = 0;
    for (uint32_t b = threadIdx.x; b < nblocks; b += blockDim.x) {
        const uint32_t *p = partial + (size_t)b * 8;
        if (p[6]) {
            for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], p[k]); mx[k] = max(mx[k], p[3 + k]); }
            cnt += p[6];
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], __shfl_down(mn[k], off)); mx[k] = max(mx[k], __shfl_down(mx[k], off)); }
        cnt += __shfl_down(cnt, off);
    }
    __shared__ uint32_t sm[kBlock / 64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        for (int k = 0; k < 3; ++k) { sm[wave][k] = mn[k]; sm[wave][3 + k] = mx[k]; }
        sm[wave][6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], sm[w][k]); mx[k] = max(mx[k], sm[w][3 + k]); }
            cnt += sm[w][6];
        }
        for (int k = 0; k < 3; ++k) { bbox[k] = mn[k]; bbox[3 + k] = mx[k]; }
        bbox[6] = cnt;
    }
}

// sort key: [cell_key:48 | hash16(xyz)]; non-finite points sort to the very end
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
        key = (cell_key(cx, cy, cz) << 16) | hash_xyz16(x, y, z);
    }
    keys[i] = key;
    vals[i] = i;
}

// keep[i]: not a value-equal duplicate of its predecessor in the same (cell, hash) run;
// cstart[i] / bstart[i]: first point of a cell / of a brick
__global__ __launch_bounds__(kBlock) void k_flag_runs(const unsigned long long *keys, const uint32_t *vals,
                                                      const char *pts, size_t stride, uint32_t nfin,
                                                      uint32_t *keep, uint32_t *cstart, uint32_t *bstart)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    const unsigned long long k = keys[i];
    uint32_t kp = 1, cs = 1, bs = 1;
    if (i > 0) {
        const unsigned long long kprev = keys[i - 1];
        cs = (k >> 16) != (kprev >> 16);
        bs = (k >> 22) != (kprev >> 22);
        if (k == kprev) {
            const float *a = rec_xyz(pts, stride, vals[i]);
            const float *b = rec_xyz(pts, stride, vals[i - 1]);
            if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) kp = 0;
        }
    }
    keep[i] = kp;
    cstart[i] = cs;
    bstart[i] = bs;
}

// pos / cid / bid = exclusive scans of keep / cstart / bstart
__global__ __launch_bounds__(kBlock) void k_scatter_sorted(const unsigned long long *keys, const uint32_t *vals,
                                                           const char *pts, size_t stride, uint32_t nfin,
                                                           const uint32_t *keep, const uint32_t *cstart, const uint32_t *bstart,
                                                           const uint32_t *pos, const uint32_t *cid, const uint32_t *bid,
                                                           float4 *sorted, uint32_t *cellpos, unsigned long long *brickkey,
                                                           unsigned long long *brickmask, uint32_t *brickbase,
                                                           uint32_t *counts /*[4]*/)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    const uint32_t v = vals[i];
    const unsigned long long k = keys[i];
    if (keep[i]) {
        const float *p = rec_xyz(pts, stride, v);
        sorted[pos[i]] = tgt_rec(p[0], p[1], p[2], v);
    }
    if (cstart[i]) {
        cellpos[cid[i]] = pos[i];
        const uint32_t b = bstart[i] ? bid[i] : bid[i] - 1;   // brick this cell belongs to
        atomicOr(&brickmask[b], 1ull << ((k >> 16) & 63));
    }
    if (bstart[i]) {
        brickkey[bid[i]] = k >> 22;
        brickbase[bid[i]] = cid[i];
    }
    if (i == nfin - 1) {
        const uint32_t nu = pos[i] + keep[i], nc = cid[i] + cstart[i], nb = bid[i] + bstart[i];
        counts[0] = nu;
        counts[1] = nc;
        counts[3] = nb;
        cellpos[nc] = nu;  // sentinel
    }
}

__global__ __launch_bounds__(kBlock) void k_brick_insert(const unsigned long long *brickkey, const unsigned long long *brickmask,
                                                         const uint32_t *brickbase, uint32_t nbricks, BrickEntry *table,
                                                         uint32_t mask)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbricks) return;
    const unsigned long long key = brickkey[b];
    uint32_t slot = hash_brick(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&table[slot].key, kEmptyKey, key);
        if (prev == kEmptyKey) {
            table[slot].mask = brickmask[b];
            table[slot].base = brickbase[b];
            break;
        }
        slot = (slot + 1) & mask;  // keys are unique, the table is at most half full
    }
}

__global__ __launch_bounds__(kBlock) void k_max_cell_count(const uint32_t *cellpos, uint32_t ncells, uint32_t *out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v = c < ncells ? cellpos[c + 1] - cellpos[c] : 0;
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_down(v, off));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}

// ------------------------------------------------------------------------------ source
// spread the low 16 bits of v so that there are two zero bits between consecutive bits
__device__ __forceinline__ unsigned long long spread3(unsigned long long v)
{
    v &= 0xffffull;
    v = (v | (v << 16)) & 0x0000ff0000ffull;
    v = (v | (v << 8)) & 0x00f00f00f00full;
    v = (v | (v << 4)) & 0x0c30c30c30c3ull;
    v = (v | (v << 2)) & 0x249249249249ull;
    return v;
}

// Spatial sort key of a source point: Morton (Z-order) code of its cell in the source's own
// grid, so that any run of consecutive points is a compact patch in all three dimensions.
// 32-bit form: the axes carry bits.x / bits.y / bits.z bits (what the extent needs at this cell size); the low
// min(bits) levels interleave like the 64-bit code, longer axes keep contributing alone -- the key has
// bits.x + bits.y + bits.z bits, the invalid key is one bit above.
struct MortonBits {
    int x, y, z;
};

__device__ __forceinline__ uint32_t morton_mixed(uint32_t cx, uint32_t cy, uint32_t cz, MortonBits b)
{
    uint32_t key = 0;
    int out = 0;
#pragma unroll
    for (int l = 0; l < 12; ++l) {
        if (l < b.x) key |= ((cx >> l) & 1u) << out++;
        if (l < b.y) key |= ((cy >> l) & 1u) << out++;
        if (l < b.z) key |= ((cz >> l) & 1u) << out++;
    }
    return key;
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_source_keys(const char *raw, size_t stride, uint32_t n, float ox, float oy,
                                                        float oz, float inv_cell, KeyT invalid_key, MortonBits bits,
                                                        KeyT *keys, uint32_t *vals, uint32_t *sort_scratch, uint32_t sort_scratch_words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (sort_scratch) radix32_clear(sort_scratch, sort_scratch_words, i, gridDim.x * blockDim.x);   // (the state of the sort that follows: radix32.hpp)
    if (i >= n) return;
    const float *p = rec_xyz(raw, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    KeyT key = invalid_key;   // one bit above every Morton code: non-finite points sort last
    if (finite3(x, y, z)) {
        const unsigned cx = (unsigned)cell_coord(x, ox, inv_cell), cy = (unsigned)cell_coord(y, oy, inv_cell), cz = (unsigned)cell_coord(z, oz, inv_cell);
        if (sizeof(KeyT) == 4) key = (KeyT)morton_mixed(min(cx, (1u << bits.x) - 1u), min(cy, (1u << bits.y) - 1u), min(cz, (1u << bits.z) - 1u), bits);
        else key = (KeyT)(spread3(cx) | spread3(cy) << 1 | spread3(cz) << 2);
    }
    keys[i] = key;
    vals[i] = i;
}

// k_source_keys for 32-bit keys that the library's own sort takes next, with that sort's digit histograms counted on the
// way (what k_os_hist would read the keys again for: one launch and a pass over the keys less on the source's chain, which
// is the longer one of a pair since the target's index is built by counting).  `hist`: passes x 256 counts, zero on entry;
// `hist_next`: the set the NEXT load will count into, cleared here (the two take turns: icp.hip, load_source_queue).
constexpr unsigned kSkItems = 4;
__global__ __launch_bounds__(kOsHistBlock) void k_source_keys_hist(const char *raw, size_t stride, uint32_t n, float ox, float oy, float oz, float inv_cell,
                                                                   uint32_t invalid_key, MortonBits bits, uint32_t *keys, uint32_t *vals, uint32_t *sort_scratch,
                                                                   uint32_t sort_scratch_words, uint32_t end_bit, uint32_t passes, uint32_t *hist,
                                                                   uint32_t *hist_next)
{
    __shared__ uint32_t s_h[OsKey<uint32_t>::max_passes * kOsDigits];
    const uint32_t t = blockIdx.x * kOsHistBlock + threadIdx.x, threads = gridDim.x * kOsHistBlock;
    radix32_clear(sort_scratch, sort_scratch_words, t, threads);   // (look-back words and tickets of the passes: nobody reads them before this kernel is done)
    for (uint32_t k = t; k < OsKey<uint32_t>::max_passes * kOsDigits; k += threads) hist_next[k] = 0u;
    for (uint32_t k = threadIdx.x; k < passes * kOsDigits; k += kOsHistBlock) s_h[k] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * kOsHistBlock * kSkItems;
#pragma unroll
    for (uint32_t j = 0; j < kSkItems; ++j) {
        const uint32_t i = base + j * kOsHistBlock + threadIdx.x;
        if (i >= n) break;
        const float *p = rec_xyz(raw, stride, i);
        const float x = p[0], y = p[1], z = p[2];
        uint32_t key = invalid_key;   // one bit above every Morton code: non-finite points sort last
        if (finite3(x, y, z)) {
            const unsigned cx = (unsigned)cell_coord(x, ox, inv_cell), cy = (unsigned)cell_coord(y, oy, inv_cell), cz = (unsigned)cell_coord(z, oz, inv_cell);
            key = (uint32_t)morton_mixed(min(cx, (1u << bits.x) - 1u), min(cy, (1u << bits.y) - 1u), min(cz, (1u << bits.z) - 1u), bits);
        }
        keys[i] = key;
        vals[i] = i;
        for (uint32_t q = 0; q < passes; ++q) {
            const uint32_t bit = q * kOsBits, nb = min(kOsBits, end_bit - bit);
            atomicAdd(&s_h[q * kOsDigits + ((key >> bit) & ((1u << nb) - 1u))], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < passes * kOsDigits; k += kOsHistBlock)
        if (s_h[k]) atomicAdd(&hist[k], s_h[k]);
}

// src[j] = {xyz of original point perm[j], valid}; cur = copy
// keys + keep (both or neither): also flags the sorted point j that is not an exact copy of its predecessor (same Morton
// key, same xyz; invalid points are never merged: they carry weight 0 anyway), from the predecessor's own record
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_gather_source(const char *raw, size_t stride, uint32_t n, const uint32_t *perm,
                                                          float4 *src, float4 *cur, const KeyT *keys = nullptr,
                                                          uint32_t *keep = nullptr)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float *p = rec_xyz(raw, stride, perm ? perm[j] : j);
    const float x = p[0], y = p[1], z = p[2];
    const float4 s = make_float4(x, y, z, finite3(x, y, z) ? 1.0f : 0.0f);
    src[j] = s;
    if (cur) cur[j] = s;
    if (keep) {
        uint32_t kp = 1;
        if (j > 0 && keys[j] == keys[j - 1]) {
            const float *b = rec_xyz(raw, stride, perm ? perm[j - 1] : j - 1);
            if (s.w != 0.0f && finite3(b[0], b[1], b[2]) && x == b[0] && y == b[1] && z == b[2]) kp = 0;
        }
        keep[j] = kp;
    }
}

// pos = exclusive scan of keep: first[u] = sorted index of unique point u, uniq_of[j] = its id, and
// src[u] = {xyz, 1 (0: invalid point)}.  How many copies a point stands for -- its weight in the 17 sums -- is
// first[u + 1] - first[u]: whoever starts an alignment from src (k_restart_source, or the first search launch itself)
// puts it into the working copy, no launch over first[] here.
__global__ __launch_bounds__(kBlock) void k_source_unique(const float4 *src_all, uint32_t n, const uint32_t *keep,
                                                          const uint32_t *pos, uint32_t *first, uint32_t *uniq_of,
                                                          uint32_t *count, uint32_t *host_count, float4 *src)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t kp = keep[j];
    const uint32_t u = kp ? pos[j] : pos[j] - 1;
    uniq_of[j] = u;
    if (kp) {
        first[u] = j;
        src[u] = src_all[j];
    }
    if (j == n - 1) {
        count[0] = u + 1;
        host_count[0] = u + 1;   // (pinned host memory: read at the join, no copy queued)
        first[u + 1] = n;
    }
}

// The whole load of a SMALL source in the caller's order (no Morton order, no merging of exact copies: at <= 65 536 points
// neither buys the search anything -- 50 k raw frame, 36 k edge cloud: the same 30.1 us and 13 us per launch either way,
// profiles/r03_small_sources.txt -- and together they are 17 launches and a round trip to the host):
// src_all[j] = src[j] = {xyz, 1 or 0 (non-finite)}, perm / uniq_of / first = identity, count = n.
// On its way the launch measures the cloud's bounding box and finite count (what the sorted load gets from k_bbox and a round
// trip): a partial per workgroup, the workgroup that finishes last (ticket: zero before the launch, zero again after it) folds
// them and leaves mn[3] | mx[3] (ordered uints) | count | `seq` in pinned host words -- nobody waits for them; the cloud handle
// picks them up once the alignment that followed has been waited for (cloud.hip: harvest_source_box), so that what is made of
// this cloud -- the aligned cloud, the target it is appended to -- starts with a box (rsreg_ctx.hpp: CloudBox).
__global__ __launch_bounds__(kBlock) void k_source_plain(const char *raw, size_t stride, uint32_t n, float4 *src_all, float4 *src,
                                                         uint32_t *perm, uint32_t *uniq_of, uint32_t *first, uint32_t *count, uint32_t *host_count,
                                                         uint32_t *box_partial, uint32_t *box_ticket, uint32_t *host_box, uint32_t seq)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    uint32_t cnt = 0;
    if (j < n) {
        const float *p = rec_xyz(raw, stride, j);
        const float x = p[0], y = p[1], z = p[2];
        const bool fin = finite3(x, y, z);
        const float4 s = make_float4(x, y, z, fin ? 1.0f : 0.0f);
        src_all[j] = s;
        src[j] = s;
        perm[j] = j;
        uniq_of[j] = j;
        first[j] = j;
        if (j == n - 1) {
            first[n] = n;
            count[0] = n;
            host_count[0] = n;   // (pinned host memory: read at the join)
        }
        if (fin) { mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z; cnt = 1; }
    }
    if (!box_partial) return;
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], off));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off));
        }
        cnt += __shfl_down(cnt, off);
    }
    __shared__ float smn[kBlock / 64][3], smx[kBlock / 64][3];
    __shared__ uint32_t scnt[kBlock / 64];
    __shared__ uint32_t s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        for (int k = 0; k < 3; ++k) { smn[wave][k] = mn[k]; smx[wave][k] = mx[k]; }
        scnt[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], smn[w][k]); mx[k] = fmaxf(mx[k], smx[w][k]); }
            cnt += scnt[w];
        }
        uint32_t *out = box_partial + (size_t)blockIdx.x * 8;
        for (int k = 0; k < 3; ++k) {
            __hip_atomic_store(&out[k], float_ordered(mn[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&out[3 + k], float_ordered(mx[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_store(&out[6], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (written through and acknowledged, then counted: icp_dense.hpp's hand-over)
        s_last = __hip_atomic_fetch_add(box_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // the last workgroup: one partial per thread (at most 256 workgroups of 256 records: kPlainSourceMax), folded as above
    uint32_t omn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, omx[3] = {0, 0, 0}, ocnt = 0;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += blockDim.x) {
        const uint32_t *q = box_partial + (size_t)b * 8;
        const uint32_t c = __hip_atomic_load(&q[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c) {
            for (int k = 0; k < 3; ++k) {
                omn[k] = min(omn[k], __hip_atomic_load(&q[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                omx[k] = max(omx[k], __hip_atomic_load(&q[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            ocnt += c;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; ++k) { omn[k] = min(omn[k], __shfl_down(omn[k], off)); omx[k] = max(omx[k], __shfl_down(omx[k], off)); }
        ocnt += __shfl_down(ocnt, off);
    }
    __shared__ uint32_t sm[kBlock / 64][8];
    if (lane == 0) {
        for (int k = 0; k < 3; ++k) { sm[wave][k] = omn[k]; sm[wave][3 + k] = omx[k]; }
        sm[wave][6] = ocnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            for (int k = 0; k < 3; ++k) { omn[k] = min(omn[k], sm[w][k]); omx[k] = max(omx[k], sm[w][3 + k]); }
            ocnt += sm[w][6];
        }
        *box_ticket = 0u;
        for (int k = 0; k < 3; ++k) { host_box[k] = omn[k]; host_box[3 + k] = omx[k]; }
        host_box[6] = ocnt;
        __threadfence_system();
        __hip_atomic_store(&host_box[7], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (the stamp last: the words in front of it are complete)
    }
}

// the weight of distinct source point i in the 17 sums: how many records of the caller's cloud are exact copies of it
__device__ __forceinline__ float source_weight(const uint32_t *first, uint32_t i) { return (float)(first[i + 1] - first[i]); }

// cur = guess * src (or src) with the points' weights; also: no seeds yet
__global__ __launch_bounds__(kBlock) void k_restart_source(const float4 *src, const uint32_t *first, uint32_t n, Mat34 guess, int apply_guess,
                                                           float4 *cur, int *seed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    seed[i] = -1;
    float4 s = src[i];
    if (s.w != 0.0f) s.w = source_weight(first, i);
    if (s.w != 0.0f && apply_guess) {
        const float3 t = xform(guess, s.x, s.y, s.z);
        s = make_float4(t.x, t.y, t.z, s.w);
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
        cur[i] = make_float4(t.x, t.y, t.z, s.w);
    }
}

// out[perm[j]] = T * src[j] as packed float3 (the aligned cloud icp.align() hands back)
__global__ __launch_bounds__(kBlock) void k_apply_final(const float4 *src, uint32_t n, Mat34 T, const uint32_t *perm,
                                                        float *out_xyz)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float4 s = src[j];
    float3 t = make_float3(s.x, s.y, s.z);
    if (s.w != 0.0f) t = xform(T, s.x, s.y, s.z);
    const size_t o = perm ? perm[j] : j;
    out_xyz[3 * o] = t.x;
    out_xyz[3 * o + 1] = t.y;
    out_xyz[3 * o + 2] = t.z;
}

// ------------------------------------------------------------------------------ NN search
struct Best {
    // key = d2 bits << 32 | original target index: unsigned order = (distance, index) order,
    // so "key < best" is "closer, or equally close with a lower index" (d2 >= +0 always)
    unsigned long long key;
    int pos;        // position in the sorted target array
    float d2;
};

__device__ __forceinline__ void consider(Best &b, float d, uint32_t oi, uint32_t p)
{
    const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | oi;
    if (k < b.key) {
        b.key = k;
        b.pos = (int)p;
        b.d2 = d;
    }
}

// 4 independent loads in flight per lane; the clamped tail re-evaluates the last point,
// which cannot change the result
__device__ __forceinline__ void scan_points(const float4 *pts, uint32_t s, uint32_t e, float qx, float qy, float qz, Best &b)
{
    if (s >= e) return;
    const uint32_t last = e - 1;
    for (uint32_t p = s; p < e; p += 4) {
        const uint32_t p1 = min(p + 1, last), p2 = min(p + 2, last), p3 = min(p + 3, last);
        const float4 t0 = pts[p], t1 = pts[p1], t2 = pts[p2], t3 = pts[p3];
        consider(b, l2_simple(qx, qy, qz, t0.x, t0.y, tgt_z(t0)), tgt_idx(t0), p);
        consider(b, l2_simple(qx, qy, qz, t1.x, t1.y, tgt_z(t1)), tgt_idx(t1), p1);
        consider(b, l2_simple(qx, qy, qz, t2.x, t2.y, tgt_z(t2)), tgt_idx(t2), p2);
        consider(b, l2_simple(qx, qy, qz, t3.x, t3.y, tgt_z(t3)), tgt_idx(t3), p3);
    }
}

__device__ __forceinline__ bool brick_lookup(const GridDev &g, int bx, int by, int bz, unsigned long long &mask, uint32_t &base)
{
    const unsigned long long key = brick_key(bx, by, bz);
    uint32_t slot = hash_brick(key) & g.bmask;
    for (;;) {
        const BrickEntry e = g.bricks[slot];
        if (e.key == key) { mask = e.mask; base = e.base; return true; }
        if (e.key == kEmptyKey) return false;
        slot = (slot + 1) & g.bmask;
    }
}

// lower bound (in cells) on the distance along one axis from a query at in-grid position u
// (cell units) to cells lo..hi; kCellMargin absorbs the float rounding of cell assignment
__device__ __forceinline__ float axis_gap(float u, int lo, int hi, float margin = kCellMargin)
{
    const float a = (float)lo - u, b = u - (float)(hi + 1);
    return fmaxf(fmaxf(a, b) - margin, 0.0f);
}

// bits of a 4x4x4 brick whose local coordinates lie in [x0,x1] x [y0,y1] x [z0,z1] (all in 0..3)
__device__ __forceinline__ unsigned long long box_mask(int x0, int x1, int y0, int y1, int z0, int z1)
{
    const unsigned long long xm = ((1u << (x1 + 1)) - (1u << x0)) & 0xfu;
    const unsigned long long row = xm * 0x1111ull;                       // x pattern in all 4 y rows
    const unsigned long long ym = ((1ull << (4 * (y1 + 1))) - (1ull << (4 * y0))) & 0xffffull;
    const unsigned long long plane = (row & ym) * 0x0001000100010001ull; // in all 4 z planes
    const unsigned long long hi = (z1 >= 3) ? ~0ull : ((1ull << (16 * (z1 + 1))) - 1ull);
    const unsigned long long zm = hi & ~((1ull << (16 * z0)) - 1ull);
    return plane & zm;
}

// Visit the occupied cells of one brick selected by `sel`, nearest-bound pruned.
__device__ __forceinline__ void visit_brick(const GridDev &g, unsigned long long sel, uint32_t base, unsigned long long mask,
                                            int bx, int by, int bz, float ux, float uy, float uz, float cell2, float qx,
                                            float qy, float qz, Best &b, float &limit2)
{
    while (sel) {
        const int bit = __ffsll((long long)sel) - 1;
        sel &= sel - 1;
        const int x = (bx << 2) | (bit & 3), y = (by << 2) | ((bit >> 2) & 3), z = (bz << 2) | (bit >> 4);
        const float gx = axis_gap(ux, x, x), gy = axis_gap(uy, y, y), gz = axis_gap(uz, z, z);
        const float lb2 = (gx * gx + gy * gy + gz * gz) * cell2;
        if (lb2 > limit2) continue;
        const uint32_t id = base + __popcll(mask & ((1ull << bit) - 1ull));
        scan_points(g.pts, g.cellpos[id], g.cellpos[id + 1], qx, qy, qz, b);
        limit2 = fminf(limit2, b.d2);
    }
}

struct QueryGeom {   // where a query sits in the grid
    float ux, uy, uz;    // position in cell units relative to the grid origin
    int cx, cy, cz;      // its cell, clamped to [-1, dim]
};

__device__ __forceinline__ QueryGeom query_geom(const GridDev &g, float qx, float qy, float qz)
{
    QueryGeom q;
    q.ux = cell_pos(qx, g.ox, g.inv_cell);
    q.uy = cell_pos(qy, g.oy, g.inv_cell);
    q.uz = cell_pos(qz, g.oz, g.inv_cell);
    q.cx = min(max((int)fminf(fmaxf(floorf(q.ux), -4.0f), 70000.0f), -1), g.dx);
    q.cy = min(max((int)fminf(fmaxf(floorf(q.uy), -4.0f), 70000.0f), -1), g.dy);
    q.cz = min(max((int)fminf(fmaxf(floorf(q.uz), -4.0f), 70000.0f), -1), g.dz);
    return q;
}

// Rings 0 and 1 from global memory: the 3x3x3 cells around the query, at most 8 bricks.
__device__ __forceinline__ void nn_near_global(const GridDev &g, const QueryGeom &qg, float qx, float qy, float qz, Best &b,
                                               float &limit2)
{
    const float ux = qg.ux, uy = qg.uy, uz = qg.uz;
    const int cx = qg.cx, cy = qg.cy, cz = qg.cz;
    const float cell2 = g.cell * g.cell;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dx - 1);
    const int y0 = max(cy - 1, 0), y1 = min(cy + 1, g.dy - 1);
    const int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.dz - 1);
    if (x0 > x1 || y0 > y1 || z0 > z1) return;
    // own cell first, then the rest of its brick, then the other bricks: the nearest
    // point is usually in the own cell and tightens the limit for everything else
    const int ocx = min(max(cx, x0), x1), ocy = min(max(cy, y0), y1), ocz = min(max(cz, z0), z1);
    const int obx = ocx >> 2, oby = ocy >> 2, obz = ocz >> 2;
    const unsigned long long own_bit = 1ull << ((ocz & 3) << 4 | (ocy & 3) << 2 | (ocx & 3));
    for (int pass = 0; pass < 2; ++pass) {
        for (int bz = z0 >> 2; bz <= z1 >> 2; ++bz)
            for (int by = y0 >> 2; by <= y1 >> 2; ++by)
                for (int bx = x0 >> 2; bx <= x1 >> 2; ++bx) {
                    const bool own = (bx == obx && by == oby && bz == obz);
                    if (own != (pass == 0)) continue;
                    const float gx = axis_gap(ux, max(bx << 2, x0), min((bx << 2) + 3, x1));
                    const float gy = axis_gap(uy, max(by << 2, y0), min((by << 2) + 3, y1));
                    const float gz = axis_gap(uz, max(bz << 2, z0), min((bz << 2) + 3, z1));
                    if ((gx * gx + gy * gy + gz * gz) * cell2 > limit2) continue;
                    unsigned long long mask;
                    uint32_t base;
                    if (!brick_lookup(g, bx, by, bz, mask, base)) continue;
                    unsigned long long sel =
                        mask & box_mask(max(x0 - (bx << 2), 0), min(x1 - (bx << 2), 3), max(y0 - (by << 2), 0),
                                        min(y1 - (by << 2), 3), max(z0 - (bz << 2), 0), min(z1 - (bz << 2), 3));
                    if (own && (sel & own_bit)) {
                        visit_brick(g, own_bit, base, mask, bx, by, bz, ux, uy, uz, cell2, qx, qy, qz, b, limit2);
                        sel &= ~own_bit;
                    }
                    visit_brick(g, sel, base, mask, bx, by, bz, ux, uy, uz, cell2, qx, qy, qz, b, limit2);
                }
    }
}

// Farther rings from global memory, only while something unvisited could still be closer.
// Expands in shells of BRICKS around the query's brick (nearest first, so the limit tightens
// early).  After shell rb every unvisited brick is >= rb+1 bricks away along some axis, i.e.
// at least 4*rb cells from the query.  Cells within `inner` rings of the query's cell are
// assumed visited already.
__device__ __forceinline__ void nn_far_global(const GridDev &g, const QueryGeom &qg, float qx, float qy, float qz, Best &b,
                                              float &limit2, int inner = 1)
{
    const float ux = qg.ux, uy = qg.uy, uz = qg.uz;
    const int cx = qg.cx, cy = qg.cy, cz = qg.cz;
    const float cell2 = g.cell * g.cell;
    // after ring `inner` every unvisited cell is at least (inner - margin) cells away along some axis
    const float ring1 = ((float)inner - kCellMargin) * g.cell;
    if (limit2 <= ring1 * ring1 || g.max_ring <= inner) return;
    const int nbx = (g.dx + 3) >> 2, nby = (g.dy + 3) >> 2, nbz = (g.dz + 3) >> 2;
    const int qbx = min(max(cx, 0), g.dx - 1) >> 2, qby = min(max(cy, 0), g.dy - 1) >> 2, qbz = min(max(cz, 0), g.dz - 1) >> 2;
    const int rb_grid = max(max(max(qbx, nbx - 1 - qbx), max(qby, nby - 1 - qby)), max(qbz, nbz - 1 - qbz));
    const int rb_max = min(rb_grid, (g.max_ring + 3) / 4 + 1);
    const int ix0 = max(cx - inner, 0), ix1 = min(cx + inner, g.dx - 1);
    const int iy0 = max(cy - inner, 0), iy1 = min(cy + inner, g.dy - 1);
    const int iz0 = max(cz - inner, 0), iz1 = min(cz + inner, g.dz - 1);
    const bool inner_ok = ix0 <= ix1 && iy0 <= iy1 && iz0 <= iz1;
    const int rb_inner = (inner + 3) / 4 + 1;  // bricks farther than this cannot touch the inner block
    for (int rb = 0; rb <= rb_max; ++rb) {
        for (int dz = -rb; dz <= rb; ++dz) {
            const int bz = qbz + dz;
            if (bz < 0 || bz >= nbz) continue;
            const float gz = axis_gap(uz, bz << 2, (bz << 2) + 3);
            if (gz * gz * cell2 > limit2) continue;
            for (int dy = -rb; dy <= rb; ++dy) {
                const int by = qby + dy;
                if (by < 0 || by >= nby) continue;
                const float gy = axis_gap(uy, by << 2, (by << 2) + 3);
                if ((gy * gy + gz * gz) * cell2 > limit2) continue;
                const bool face = (abs(dz) == rb) || (abs(dy) == rb);
                const int step = face ? 1 : max(2 * rb, 1);
                for (int bx = qbx - rb; bx <= qbx + rb; bx += step) {
                    if (bx < 0 || bx >= nbx) continue;
                    const float gx = axis_gap(ux, bx << 2, (bx << 2) + 3);
                    if ((gx * gx + gy * gy + gz * gz) * cell2 > limit2) continue;
                    unsigned long long mask;
                    uint32_t base;
                    if (!brick_lookup(g, bx, by, bz, mask, base)) continue;
                    unsigned long long sel = mask;
                    if (inner_ok && rb <= rb_inner) {  // drop the visited inner block where it intersects this brick
                        const int jx0 = max(ix0 - (bx << 2), 0), jx1 = min(ix1 - (bx << 2), 3);
                        const int jy0 = max(iy0 - (by << 2), 0), jy1 = min(iy1 - (by << 2), 3);
                        const int jz0 = max(iz0 - (bz << 2), 0), jz1 = min(iz1 - (bz << 2), 3);
                        if (jx0 <= jx1 && jy0 <= jy1 && jz0 <= jz1) sel &= ~box_mask(jx0, jx1, jy0, jy1, jz0, jz1);
                    }
                    visit_brick(g, sel, base, mask, bx, by, bz, ux, uy, uz, cell2, qx, qy, qz, b, limit2);
                }
            }
        }
        const float reach = fmaxf((float)(4 * rb) - kCellMargin, 0.0f) * g.cell;
        if (limit2 <= reach * reach) break;
    }
}

// Exact nearest neighbour within the gate.  Every cell that could hold a point at least as
// close as the current best (or the gate) is visited; cells and bricks are skipped only on a
// conservative geometric lower bound, so the result equals a brute-force scan including the
// lowest-index tie-break.
//
// seed_pos: a target point known from the previous iteration (or -1).  Its distance is a valid
// upper bound on the nearest distance, so the search starts with a tight limit and usually
// ends inside the query's own cell; exactness is unaffected (the seed is just a candidate).
__device__ __forceinline__ Best nn_query(const GridDev &g, float qx, float qy, float qz, int seed_pos = -1)
{
    Best b{~0ull, -1, FLT_MAX};
    if (g.dx <= 0) return b;
    const QueryGeom qg = query_geom(g, qx, qy, qz);
    float limit2 = g.prune2;  // nothing farther than this can be accepted or improve the best
    if (seed_pos >= 0) {
        const float4 t = g.pts[seed_pos];
        consider(b, l2_simple(qx, qy, qz, t.x, t.y, tgt_z(t)), tgt_idx(t), (uint32_t)seed_pos);
        limit2 = fminf(limit2, b.d2);
    }
    nn_near_global(g, qg, qx, qy, qz, b, limit2);
    nn_far_global(g, qg, qx, qy, qz, b, limit2);
    return b;
}

__global__ __launch_bounds__(kBlock) void k_nn_search(const float4 *cur, uint32_t n, GridDev g, double gate2,
                                                      int *corr_pos, float *corr_d2, int *seed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = cur[i];
    int pos = -1;
    float d2 = 0.0f;
    if (q.w != 0.0f) {
        const Best b = nn_query(g, q.x, q.y, q.z, seed ? seed[i] : -1);
        if (seed) seed[i] = b.pos;
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

// accumulate one accepted pair (p = source, q = target) into the 17 sums; w = how many
// identical source points this one stands for (exact duplicates are searched once)
__device__ __forceinline__ void accum_pair(double *a, float px, float py, float pz, float qx, float qy, float qz, float d2,
                                           double W)
{
    const double P[3] = {px, py, pz}, Q[3] = {qx, qy, qz};
    a[0] += W;
    for (int k = 0; k < 3; ++k) { a[1 + k] += W * P[k]; a[4 + k] += W * Q[k]; }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) a[7 + r * 3 + c] += W * (Q[r] * P[c]);
    a[16] += W * (double)d2;
}

constexpr int kTile = 128;         // source points per workgroup in the iteration kernels: 2 waves
constexpr int kTileWaves = kTile / 64;

// Block-level reduction of per-thread sums into partials[k][blockIdx] (transposed slabs: the
// final reduction reads each sum contiguously).  Fixed order, so the result depends only on
// the number of source points, never on the GPU or on timing.
__device__ __forceinline__ void tile_reduce_store(double *a, double *partials, uint32_t nblocks, uint32_t slot);
__device__ __forceinline__ void tile_reduce_store(double *a, double *partials, uint32_t nblocks) { tile_reduce_store(a, partials, nblocks, blockIdx.x); }
__device__ __forceinline__ void tile_reduce_store(double *a, double *partials, uint32_t nblocks, uint32_t slot)
{
    __shared__ double shr[kTileWaves][RSREG_NUM_SUMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 17 sums over the 64 lanes by recursive halving: at each of the six steps a lane keeps half
    // of the sums it still holds and hands the other half to its partner (lane ^ 32, 16, ... 1),
    // so the steps move 9 + 5 + 3 + 2 + 1 + 1 = 21 doubles instead of 17 x 6 = 102 for seventeen
    // separate butterflies.  The tree (who adds what, in which order) is fixed.
    int base = 0, cnt = RSREG_NUM_SUMS;
    double v9[9], v5[5], v3[3], v2[2], v1[1], v0[1];
    {
        double v17[RSREG_NUM_SUMS];
        for (int k = 0; k < RSREG_NUM_SUMS; ++k) v17[k] = a[k];
        halve_sums<RSREG_NUM_SUMS>(v17, v9, lane, 32, base, cnt);
    }
    halve_sums<9>(v9, v5, lane, 16, base, cnt);
    halve_sums<5>(v5, v3, lane, 8, base, cnt);
    halve_sums<3>(v3, v2, lane, 4, base, cnt);
    halve_sums<2>(v2, v1, lane, 2, base, cnt);
    halve_sums<1>(v1, v0, lane, 1, base, cnt);
    if (cnt >= 1) shr[wave][base] = v0[0];   // exactly one lane ends up owning each of the 17 sums
    __syncthreads();
    if (threadIdx.x < RSREG_NUM_SUMS) {
        double v = shr[0][threadIdx.x];
        for (int w = 1; w < kTileWaves; ++w) v += shr[w][threadIdx.x];
        partials[(size_t)threadIdx.x * nblocks + slot] = v;
    }
}

// one source point per thread; block b covers points [b*128, b*128+128): every iteration
// kernel uses the same mapping, so all pipelines add the same numbers in the same order
__global__ __launch_bounds__(kTile) void k_cov_reduce(const float4 *cur, const int *corr_pos, const float *corr_d2,
                                                      const float4 *tgt, uint32_t n, double *partials)
{
    double a[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) a[k] = 0.0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int pos = corr_pos[i];
        if (pos >= 0) {
            const float4 p = cur[i];
            const float4 q = tgt[pos];
            accum_pair(a, p.x, p.y, p.z, q.x, q.y, tgt_z(q), corr_d2[i], p.w);
        }
    }
    tile_reduce_store(a, partials, gridDim.x);
}

// A thread's share of `count` partials (every kBlock-th, ascending), added in that order.  The loads go out sixteen at a
// time before the first is used: one after the other, each waiting for the last (what the plain loop compiles to),
// thirty trips to memory in a row were 10 of the 14 us between two search kernels.
// (kReduceBlock threads a sum.  Round 6 measured 1 024 -- the 7 813 slabs of a 10^6-point launch as eight loads a thread, one round
// in flight, instead of 31 in two rounds: k_final_reduce_solve 8.24 -> 7.85 us and the single pair 3.165 -> 3.15 ms, but a
// workgroup of sixteen waves has to find sixteen free wave slots on ONE CU behind the other pairs' search launches, and the chain with
// three pairs in flight went from 1.05-1.21 to 1.22-1.27 ms per pair on the same box (profiles/r06_experiments/reduce_block.txt):
// 256 stays.  The order of the additions is part of every pipeline's result: staged, fused and device-loop launches all go through
// this function.)
#ifndef RSREG_REDUCE_BLOCK
#define RSREG_REDUCE_BLOCK 256   // (dev: experiment builds, RSREG_CXXFLAGS=-DRSREG_REDUCE_BLOCK=1024)
#endif
constexpr int kReduceBlock = RSREG_REDUCE_BLOCK;
__device__ __forceinline__ double strided_sum(const double *src, uint32_t count)
{
    constexpr int kInFlight = 16;
    double v = 0.0;
    for (uint32_t b = threadIdx.x; b < count; b += kReduceBlock * kInFlight) {
        double x[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) x[j] = b + j * kReduceBlock < count ? src[b + j * kReduceBlock] : 0.0;
#pragma unroll
        for (int j = 0; j < kInFlight; ++j)
            if (b + j * kReduceBlock < count) v += x[j];
    }
    return v;
}

// 17 blocks of kReduceBlock threads: block k adds sum k over all slabs, fixed order
__global__ __launch_bounds__(kReduceBlock) void k_final_reduce(const double *partials, uint32_t nblocks, double *sums)
{
    __shared__ double shf[kReduceBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *src = partials + (size_t)blockIdx.x * nblocks;
    double v = strided_sum(src, nblocks);
    v = wave_sum(v);
    if (lane == 0) shf[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = shf[0];
        for (int w = 1; w < kReduceBlock / 64; ++w) t += shf[w];
        sums[blockIdx.x] = t;
    }
}

// One ICP iteration in one pass from global memory over the brick-hash index:
// apply the previous increment, search, gate, accumulate.  Writes the transformed source back
// (next iteration starts from it, like PCL's in-place transformCloud).
__global__ __launch_bounds__(kTile) void k_icp_fused(float4 *cur, uint32_t n, Mat34 T, int apply_t, GridDev g,
                                                     double gate2, int *corr_pos, float *corr_d2, double *partials, int *seed,
                                                     const IcpDevState *dev)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (dev) {   // device-resident loop: the increment comes from the previous k_icp_solve
        T = dev->t_inc;
        apply_t = dev->apply;
    }
    int pos = -1;
    float d2 = 0.0f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        q = cur[i];
        if (q.w != 0.0f) {
            if (apply_t) {
                const float3 t = xform(T, q.x, q.y, q.z);
                q = make_float4(t.x, t.y, t.z, q.w);
                cur[i] = q;
            }
            const Best b = nn_query(g, q.x, q.y, q.z, seed ? seed[i] : -1);
            if (seed) seed[i] = b.pos;
            if (b.pos >= 0 && !((double)b.d2 > gate2)) {
                pos = b.pos;
                d2 = b.d2;
            }
        }
        if (corr_pos) { corr_pos[i] = pos; corr_d2[i] = d2; }
    }
    // the 17 accumulators only come alive after the search: they do not cost it registers
    double a[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) a[k] = 0.0;
    if (pos >= 0) {
        const float4 t = g.pts[pos];
        accum_pair(a, q.x, q.y, q.z, t.x, t.y, tgt_z(t), d2, q.w);
    }
    tile_reduce_store(a, partials, gridDim.x);
}

// Umeyama from the 17 sums, compose, count -- update_from_sums with fixed-count criteria (one thread)
__device__ __forceinline__ void icp_solve_step(const double *sums, IcpDevState *st)
{
    if (st->stopped) return;
    double s[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) st->sums_last[k] = s[k] = sums[k];
    st->ncorr = (unsigned long long)(s[0] + 0.5);
    if (st->ncorr < 3) {
        st->stopped = 1;
        st->apply = 0;
        return;
    }
    Mat4f t;
    umeyama_from_sums(s, t, st->svd_v);
    st->t_inc = to_mat34(t);
    st->apply = 1;
    st->final_t = mul(t, st->final_t);
    st->iterations++;
    st->cur_mse = s[16] / s[0];
}

__global__ void k_icp_solve(const double *sums, IcpDevState *st)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) icp_solve_step(sums, st);
}

// k_final_reduce and k_icp_solve in one launch (single-GPU device loop): the block that finishes
// last has all 17 sums in front of it and runs the solve; `ticket` returns to 0 for the next launch
//
// What the solve reads of the state the previous launch left -- the stop flag, the V its Jacobi iteration starts from, the
// composed transform, the iteration count -- is requested by thread 0 of EVERY block before the reduce (28 loads that are on
// their way while the partials are added): the last block's solve then starts from registers instead of two more
// dependent round trips behind the ticket.  (Written by the previous launch of this kernel: visible at the kernel boundary.)
struct IcpSolvePrefetch {
    int stopped, iterations;
    Mat4f final_t;
    double svd_v[9];
};

// ---- umeyama_from_sums across the lanes of a wave (round 6).  Same operations on the same operands in the same order as
// host_linalg.hpp's umeyama_from_sums / jacobi_svd3 (host and the one-lane device form): lane r (r = 0, 1, 2) owns ROW r of W = A V
// and of V.  A rotation (p, q) of the one-sided Jacobi iteration touches columns p and q of every row -- each lane updates its
// own row (4 products-and-sums instead of 12 on one lane) -- and needs the three column dot products, whose terms the lanes
// compute one row each and hand round by v_readlane, added in the scalar code's order ((0 + k0) + k1) + k2.  The chain from
// the dot products to (c, s) -- three divisions and two square roots, each a dozen dependent instructions -- is what no lane
// count shortens; every lane runs it on the same numbers.  All 64 lanes of the wave must be active (lanes >= 3 mirror lane r % 3).
__device__ __forceinline__ double lane_bcast(double x, int k)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), k), hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sel3d(int r, double a0, double a1, double a2) { return r == 0 ? a0 : (r == 1 ? a1 : a2); }

__device__ __forceinline__ bool umeyama_wave(const double *sums, Mat4f &T, double *v_warm /* 9, in/out, uniform */)
{
#pragma clang fp contract(off)
    const int r = (int)(threadIdx.x & 63u) % 3;
    const double n = sums[0];
    if (!(n >= 1.0)) return false;
    double mu_p[3], mu_q[3], sigma[9];
    const double inv_n = 1.0 / n;
#pragma unroll
    for (int i = 0; i < 3; ++i) { mu_p[i] = sums[1 + i] * inv_n; mu_q[i] = sums[4 + i] * inv_n; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) sigma[i * 3 + j] = sums[7 + i * 3 + j] * inv_n - mu_q[i] * mu_p[j];
    // W = A V0, V = V0: this lane's rows
    const double a0 = sel3d(r, sigma[0], sigma[3], sigma[6]), a1 = sel3d(r, sigma[1], sigma[4], sigma[7]), a2 = sel3d(r, sigma[2], sigma[5], sigma[8]);
    double W[3], V[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        W[j] = (a0 * v_warm[j] + a1 * v_warm[3 + j]) + a2 * v_warm[6 + j];
        V[j] = sel3d(r, v_warm[j], v_warm[3 + j], v_warm[6 + j]);
    }
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p + 1 < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double kpp = W[p] * W[p], kqq = W[q] * W[q], kpq = W[p] * W[q];
                double app = 0, aqq = 0, apq = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    app += lane_bcast(kpp, k);
                    aqq += lane_bcast(kqq, k);
                    apq += lane_bcast(kpq, k);
                }
                if (apq == 0.0 || apq * apq <= 1e-32 * app * aqq) continue;
                rotated = true;
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                const double wp = W[p], wq = W[q];
                W[p] = c * wp - sn * wq;
                W[q] = sn * wp + c * wq;
                const double vp = V[p], vq = V[q];
                V[p] = c * vp - sn * vq;
                V[q] = sn * vp + c * vq;
            }
        if (!rotated) break;
    }
    double norm[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double sq = W[j] * W[j];
        double nn = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) nn += lane_bcast(sq, k);
        norm[j] = sqrt(nn);
    }
    int o0 = 0, o1 = 1, o2 = 2;
    if (pick3(norm[0], norm[1], norm[2], o1) > pick3(norm[0], norm[1], norm[2], o0)) { const int x = o0; o0 = o1; o1 = x; }
    if (pick3(norm[0], norm[1], norm[2], o2) > pick3(norm[0], norm[1], norm[2], o0)) { const int x = o0; o0 = o2; o2 = x; }
    if (pick3(norm[0], norm[1], norm[2], o2) > pick3(norm[0], norm[1], norm[2], o1)) { const int x = o1; o1 = o2; o2 = x; }
    const double smax = pick3(norm[0], norm[1], norm[2], o0);
    int rank = 0;
    double Ur[3], Vr[3];   // this lane's rows of the ordered U and V
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j = k == 0 ? o0 : (k == 1 ? o1 : o2);
        const double nj = pick3(norm[0], norm[1], norm[2], j);
        Vr[k] = pick3(V[0], V[1], V[2], j);
        if (nj > 0 && nj > 1e-13 * smax) {
            const double inv = 1.0 / nj;
            Ur[k] = pick3(W[0], W[1], W[2], j) * inv;
            rank = k + 1;
        } else {
            Ur[k] = 0.0;
        }
    }
    // everybody gets everything: the ordered U and V, row-major
    double U[9], Vf[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) { U[i * 3 + k] = lane_bcast(Ur[k], i); Vf[i * 3 + k] = lane_bcast(Vr[k], i); }
    if (rank < 3) complete_u3(U, rank);   // (coplanar / collinear matches: the scalar code, on every lane)
#pragma unroll
    for (int i = 0; i < 9; ++i) v_warm[i] = Vf[i];
    double S[3] = {1, 1, 1};
    if (det3(U) * det3(Vf) < 0) S[2] = -1;
    // R = U S V^T and t = mu_q - R mu_p: row r on lane r, then handed round
    double Rr[3];
    const double u0 = sel3d(r, U[0], U[3], U[6]), u1 = sel3d(r, U[1], U[4], U[7]), u2 = sel3d(r, U[2], U[5], U[8]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double v = 0;
        v += u0 * S[0] * Vf[j * 3 + 0];
        v += u1 * S[1] * Vf[j * 3 + 1];
        v += u2 * S[2] * Vf[j * 3 + 2];
        Rr[j] = v;
    }
    const double tr = sel3d(r, mu_q[0], mu_q[1], mu_q[2]) - (Rr[0] * mu_p[0] + Rr[1] * mu_p[1] + Rr[2] * mu_p[2]);
    T = Mat4f::identity();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) T(i, j) = float(lane_bcast(Rr[j], i));
        T(i, 3) = float(lane_bcast(tr, i));
    }
    return true;
}

// the solve of the device-resident loop on a whole wave (all 64 lanes active; lane 0 stores)
__device__ __forceinline__ void icp_solve_step_wave(const double *sums, IcpDevState *st, IcpSolvePrefetch &pf)
{
    if (pf.stopped) return;
    const bool writer = (threadIdx.x & 63u) == 0;
    double s[RSREG_NUM_SUMS];
#pragma unroll
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) s[k] = sums[k];
    if (writer) {
#pragma unroll
        for (int k = 0; k < RSREG_NUM_SUMS; ++k) st->sums_last[k] = s[k];
        st->ncorr = (unsigned long long)(s[0] + 0.5);
    }
    if ((unsigned long long)(s[0] + 0.5) < 3) {
        if (writer) { st->stopped = 1; st->apply = 0; }
        return;
    }
    Mat4f t;
    umeyama_wave(s, t, pf.svd_v);
    const Mat4f f = mul(t, pf.final_t);
    if (writer) {
#pragma unroll
        for (int i = 0; i < 9; ++i) st->svd_v[i] = pf.svd_v[i];
        st->t_inc = to_mat34(t);
        st->apply = 1;
        st->final_t = f;
        st->iterations = pf.iterations + 1;
        st->cur_mse = s[16] / s[0];
    }
}

__device__ __forceinline__ void icp_solve_step_prefetched(const double *sums, IcpDevState *st, IcpSolvePrefetch &pf)
{
    if (pf.stopped) return;
    double s[RSREG_NUM_SUMS];
#pragma unroll
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) st->sums_last[k] = s[k] = sums[k];
    st->ncorr = (unsigned long long)(s[0] + 0.5);
    if ((unsigned long long)(s[0] + 0.5) < 3) {
        st->stopped = 1;
        st->apply = 0;
        return;
    }
    Mat4f t;
    umeyama_from_sums(s, t, pf.svd_v);
#pragma unroll
    for (int i = 0; i < 9; ++i) st->svd_v[i] = pf.svd_v[i];
    st->t_inc = to_mat34(t);
    st->apply = 1;
    st->final_t = mul(t, pf.final_t);
    st->iterations = pf.iterations + 1;
    st->cur_mse = s[16] / s[0];
}

__global__ __launch_bounds__(kReduceBlock) void k_final_reduce_solve(const double *partials, uint32_t nblocks, double *sums,
                                                                     IcpDevState *st, unsigned int *ticket)
{
    __shared__ double shf[kReduceBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t s_last;
    IcpSolvePrefetch pf;
    if (wave == 0) {   // (every lane of the wave that may run the solve: the same words, one request)
        pf.stopped = st->stopped;
        pf.iterations = st->iterations;
        pf.final_t = st->final_t;
#pragma unroll
        for (int i = 0; i < 9; ++i) pf.svd_v[i] = st->svd_v[i];
    }
    const double *src = partials + (size_t)blockIdx.x * nblocks;
    double v = strided_sum(src, nblocks);
    v = wave_sum(v);
    if (lane == 0) shf[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = shf[0];
        for (int w = 1; w < kReduceBlock / 64; ++w) t += shf[w];
        // written through (agent-scope store), acknowledged, then counted: no fence, which would write back and
        // invalidate this XCD's L2
        __hip_atomic_store(&sums[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || wave != 0) return;
    // the workgroup that finished last: its first wave runs the solve, the 3 x 3 Jacobi iteration a row per lane (umeyama_wave)
    double all[RSREG_NUM_SUMS];
#pragma unroll
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) all[k] = __hip_atomic_load(&sums[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) *ticket = 0;
    icp_solve_step_wave(all, st, pf);
}

// ------------------------------------------------------------------------ no index: a handful of queries
// IncrementalICP aligns what a 1 m voxel filter leaves of a frame -- ten or twenty points -- against everything merged so
// far (incremental_icp.hpp:54-59; the leaf is never set, PCL's default applies).  Building an index over millions of
// target points for twenty queries costs fifty times the search: with at most kScanMaxSource distinct source points every
// target point is scored against every query instead.  Same float32 distance in the same operation order, same gate
// test, lowest index among equals (the minimum of (distance, index) keys does not depend on the order they arrive in).
constexpr uint32_t kScanMaxSource = 64;

// (x, y, index, z) records in the caller's order: position == index.  A non-finite point is infinitely far from
// every query.
__global__ __launch_bounds__(kBlock) void k_scan_pack(const char *pts, size_t stride, uint32_t n, float4 *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = rec_xyz(pts, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    out[i] = finite3(x, y, z) ? tgt_rec(x, y, z, i) : tgt_rec(__uint_as_float(0x7f800000u), 0.0f, 0.0f, i);
}

// keys[s] = min over the target of (d2 bits << 32 | index), d2 <= gate_f (the largest float not above the gate)
__global__ __launch_bounds__(kBlock) void k_scan_nn(const float4 *tgt, uint32_t nt, const float4 *cur, uint32_t ns, float gate_f,
                                                    unsigned long long *keys)
{
    __shared__ float4 s_q[kScanMaxSource];
    __shared__ unsigned long long s_best[kScanMaxSource];
    if (threadIdx.x < ns) {
        s_q[threadIdx.x] = cur[threadIdx.x];
        s_best[threadIdx.x] = ~0ull;
    }
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += gridDim.x * blockDim.x) {
        const float4 t = tgt[i];
        for (uint32_t s = 0; s < ns; ++s) {
            const float4 q = s_q[s];
            if (q.w == 0.0f) continue;
            const float dx = __fsub_rn(q.x, t.x), dy = __fsub_rn(q.y, t.y), dz = __fsub_rn(q.z, tgt_z(t));
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            // the workgroup's best so far bounds what is worth an atomic (a stale bound only costs one)
            const uint32_t bound = (uint32_t)(__hip_atomic_load(&s_best[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32);
            if (d <= gate_f && __float_as_uint(d) <= bound)
                atomicMin(&s_best[s], ((unsigned long long)__float_as_uint(d) << 32) | tgt_idx(t));
        }
    }
    __syncthreads();
    if (threadIdx.x < ns && s_best[threadIdx.x] != ~0ull) atomicMin(&keys[threadIdx.x], s_best[threadIdx.x]);
}

__global__ void k_scan_finish(const unsigned long long *keys, const float4 *cur, uint32_t ns, int *corr_pos, float *corr_d2)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    const unsigned long long k = keys[s];
    const bool hit = k != ~0ull && cur[s].w != 0.0f;
    corr_pos[s] = hit ? (int)(uint32_t)k : -1;
    corr_d2[s] = hit ? __uint_as_float((uint32_t)(k >> 32)) : 0.0f;
}

// corr (sorted source order, position in the sorted target) -> caller's order and indices
__global__ __launch_bounds__(kBlock) void k_export_corr(const int *corr_pos, const float *corr_d2, const float4 *tgt,
                                                        const uint32_t *perm, const uint32_t *uniq_of, uint32_t n,
                                                        int *index_out, float *d2_out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;   // sorted position of an ORIGINAL source point
    if (j >= n) return;
    const uint32_t u = uniq_of[j];
    const int pos = corr_pos[u];
    const uint32_t o = perm ? perm[j] : j;
    index_out[o] = pos >= 0 ? (int)tgt_idx(tgt[pos]) : -1;
    d2_out[o] = corr_d2[u];
}

// ------------------------------------------------------------------------ optional correspondence filters
// (rsreg_icp_params.use_reciprocal_correspondences / trim_overlap_ratio; staged pipeline only)

// how many of the copies of distinct source point u take part: all of them once it has a match
__global__ __launch_bounds__(kBlock) void k_corr_weights(const int *corr_pos, const float4 *cur, uint32_t n, uint32_t *cw)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) cw[u] = corr_pos[u] >= 0 ? (uint32_t)cur[u].w : 0u;
}

// the current source in the caller's order (one record per ORIGINAL point): what the reciprocal search indexes
__global__ __launch_bounds__(kBlock) void k_recip_points(const float4 *cur, const uint32_t *perm, const uint32_t *uniq_of, uint32_t n_source,
                                                         float4 *out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_source) out[perm[j]] = cur[uniq_of[j]];
}

// CorrespondenceRejectorTrimmed: sort keys (squared distance of the matched points, unmatched ones last)
// (sort_scratch: the state of the sort that follows, cleared on the way -- radix32.hpp)
__global__ __launch_bounds__(kBlock) void k_trim_keys(const uint32_t *cw, const float *corr_d2, uint32_t n, uint32_t *keys, uint32_t *vals,
                                                      uint32_t *sort_scratch, uint32_t sort_scratch_words)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (sort_scratch) radix32_clear(sort_scratch, sort_scratch_words, u, gridDim.x * blockDim.x);
    if (u >= n) return;
    keys[u] = cw[u] ? __float_as_uint(corr_d2[u]) : 0xffffffffu;   // (squared distances are >= 0: their bits order like the values)
    vals[u] = u;
}

__global__ __launch_bounds__(kBlock) void k_trim_gather(const uint32_t *cw, const uint32_t *order, uint32_t n, uint32_t *ws)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) ws[j] = cw[order[j]];
}

// of the pairs in order of distance, the first floor(ratio * count) are kept; cum = inclusive sums of ws
__global__ __launch_bounds__(kBlock) void k_trim_apply(const uint32_t *order, const uint32_t *ws, const uint32_t *cum, uint32_t n, float ratio,
                                                       uint32_t *cw, int *corr_pos)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t total = cum[n - 1];
    const uint32_t k = (uint32_t)floorf(ratio * (float)total);   // (int)(floor(overlap_ratio_ * float(size))), in float like PCL
    if (k >= total) return;                                       // "number_valid >= size": nothing to trim
    const uint32_t before = cum[j] - ws[j];
    const uint32_t keep = before >= k ? 0u : min(ws[j], k - before);
    const uint32_t u = order[j];
    cw[u] = keep;
    if (!keep) corr_pos[u] = -1;
}

// the sums of k_cov_reduce with the copies-in-play of every pair given explicitly
__global__ __launch_bounds__(kTile) void k_cov_reduce_w(const float4 *cur, const int *corr_pos, const float *corr_d2, const uint32_t *cw,
                                                        const float4 *tgt, uint32_t n, double *partials)
{
    double a[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) a[k] = 0.0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int pos = corr_pos[i];
        const uint32_t w = cw[i];
        if (pos >= 0 && w) {
            const float4 p = cur[i];
            const float4 q = tgt[pos];
            accum_pair(a, p.x, p.y, p.z, q.x, q.y, tgt_z(q), corr_d2[i], (double)w);
        }
    }
    tile_reduce_store(a, partials, gridDim.x);
}

// determineCorrespondences' list with the filters applied: of the copies of a distinct point the first cw[u] (lowest indices) keep their match
__global__ __launch_bounds__(kBlock) void k_export_corr_w(const int *corr_pos, const float *corr_d2, const uint32_t *cw, const float4 *tgt,
                                                          const uint32_t *perm, const uint32_t *uniq_of, const uint32_t *first, uint32_t n,
                                                          int *index_out, float *d2_out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t u = uniq_of[j];
    const int pos = corr_pos[u];
    const uint32_t o = perm ? perm[j] : j;
    const bool kept = pos >= 0 && (j - first[u]) < cw[u];
    index_out[o] = kept ? (int)tgt_idx(tgt[pos]) : -1;
    d2_out[o] = corr_d2[u];
}

}  // namespace rsreg
