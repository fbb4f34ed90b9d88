// icp_dense.hpp — dense-table form of the target index and its search (gfx950, wave64).
//
// When the grid over the target's bounding box has a bounded number of cells (a room-scale
// depth-camera cloud at ~1 cm cells: ~10^7 cells, tens of MB out of 288 GB of HBM), the
// engine keeps `start[cell]` for EVERY cell instead of hashing the occupied ones:
//   * points are sorted by linear cell id (x fastest), so the cells x-1, x, x+1 of one (y, z)
//     row are one contiguous run of points;
//   * a cell lookup is one 8-byte load, an empty cell costs no hash probe;
//   * the table has a one-cell border of empties, so neighbour offsets need no bounds checks.
// Clouds whose box is too large for the table keep the brick-hash index (icp_kernels.hpp).
//
// The search is written as a flat per-lane loop (each iteration either steps to the next
// neighbour cell or scores up to four candidates of the current one) so that a wave runs
// max-over-lanes of the per-lane work, not the product of per-level maxima that nested loops
// with lane-dependent trip counts produce.
//
// Replaces KdTreeFLANN build + CorrespondenceEstimation::determineCorrespondences
// (SURVEY.md App. A.1, A.7a); reference call sites: include/rsreg.h.
#pragma once

#include "icp_kernels.hpp"

namespace rsreg {

struct DenseDev {
    float ox, oy, oz, inv_cell, cell;
    int nx, ny, nz;          // grid extent in cells (without the border)
    int sx, sxy;             // strides of the padded table: nx + 2, (nx + 2) * (ny + 2)
    int max_ring;
    float prune2;
    const uint32_t *start;   // padded: index ((z+1)*(ny+2) + (y+1))*(nx+2) + (x+1), + end sentinel
    const float4 *pts;       // sorted target points followed by 4 far-away sentinels
    uint32_t n_pts;          // sorted target points (without the sentinels)
    uint32_t table_bytes;
};

__device__ __forceinline__ uint32_t dense_cell_id(const DenseDev &g, int x, int y, int z)
{
    return (uint32_t)(((z + 1) * (g.ny + 2) + (y + 1)) * (g.nx + 2) + (x + 1));
}

// sort key: [padded cell id | hash16(xyz)]; non-finite points sort to the very end
__global__ __launch_bounds__(kBlock) void k_dense_keys(const char *pts, size_t stride, uint32_t n, DenseDev g,
                                                       unsigned long long *keys, uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = rec_xyz(pts, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    unsigned long long key = kEmptyKey;
    if (finite3(x, y, z)) {
        const int cx = min(max(cell_coord(x, g.ox, g.inv_cell), 0), g.nx - 1), cy = min(max(cell_coord(y, g.oy, g.inv_cell), 0), g.ny - 1),
                  cz = min(max(cell_coord(z, g.oz, g.inv_cell), 0), g.nz - 1);
        key = ((unsigned long long)dense_cell_id(g, cx, cy, cz) << 16) | hash_xyz16(x, y, z);
    }
    keys[i] = key;
    vals[i] = i;
}

// keep[i]: not a value-equal duplicate of its predecessor in the same (cell, hash) run
__global__ __launch_bounds__(kBlock) void k_dense_flag(const unsigned long long *keys, const uint32_t *vals, const char *pts,
                                                       size_t stride, uint32_t nfin, uint32_t *keep)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    uint32_t kp = 1;
    if (i > 0 && keys[i] == keys[i - 1]) {
        const float *a = rec_xyz(pts, stride, vals[i]);
        const float *b = rec_xyz(pts, stride, vals[i - 1]);
        if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) kp = 0;
    }
    keep[i] = kp;
}

// far-away sentinels behind the last sorted point: a 4-wide candidate read may run past it
__global__ __launch_bounds__(kBlock) void k_dense_fill_sentinels(float4 *sorted, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sorted[i] = make_float4(1e30f, 1e30f, 1e30f, __uint_as_float(0xffffffffu));
}

// histogram of the kept points per cell + the sorted point array; stats[0] occupied cells,
// stats[1] max points per cell
__global__ __launch_bounds__(kBlock) void k_dense_scatter(const unsigned long long *keys, const uint32_t *vals, const char *pts,
                                                          size_t stride, uint32_t nfin, const uint32_t *keep,
                                                          const uint32_t *pos, float4 *sorted, uint32_t *table,
                                                          uint32_t *stats)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    if (keep[i]) {
        const uint32_t v = vals[i];
        const float *p = rec_xyz(pts, stride, v);
        sorted[pos[i]] = make_float4(p[0], p[1], p[2], __uint_as_float(v));
        const uint32_t old = atomicAdd(&table[(uint32_t)(keys[i] >> 16)], 1u);
        if (old == 0) atomicAdd(&stats[0], 1u);
        atomicMax(&stats[1], old + 1);
    }
    if (i == nfin - 1) stats[2] = pos[i] + keep[i];
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sel3(float a, float b, float c, int i) { return i == 0 ? a : (i == 1 ? b : c); }

// running best as one 64-bit key (distance bits << 32 | original index) + the byte offset of
// the winning point; starts at "+inf, no point" so that sentinel points can never win
struct DBest {
    unsigned long long key;
    uint32_t off;
};

__device__ __forceinline__ void dconsider(DBest &b, float qx, float qy, float qz, const u32x4 &t, uint32_t off)
{
    const float d = l2_simple(qx, qy, qz, __uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z));
    const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | t.w;
    if (k < b.key) {
        b.key = k;
        b.off = off;
    }
}

// score 4 consecutive points starting at byte offset `po` (reading past the end of a cell
// only meets more real target points, or the far-away sentinels behind the last one)
__device__ __forceinline__ void dscan4(DBest &b, __amdgpu_buffer_rsrc_t pts, uint32_t po, float qx, float qy, float qz)
{
    const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(pts, po, 0, 0);
    const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 16, 0, 0);
    const u32x4 t2 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 32, 0, 0);
    const u32x4 t3 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 48, 0, 0);
    dconsider(b, qx, qy, qz, t0, po);
    dconsider(b, qx, qy, qz, t1, po + 16);
    dconsider(b, qx, qy, qz, t2, po + 32);
    dconsider(b, qx, qy, qz, t3, po + 48);
}

__device__ __forceinline__ void dscan_range(DBest &b, __amdgpu_buffer_rsrc_t pts, uint32_t po, uint32_t pe, float qx, float qy,
                                            float qz)
{
    for (; po < pe; po += 64) dscan4(b, pts, po, qx, qy, qz);
}

__device__ __forceinline__ float key_d2(unsigned long long key) { return __uint_as_float((uint32_t)(key >> 32)); }

// Exact nearest neighbour within the gate over the dense table (same contract as nn_query).
__device__ __forceinline__ Best nn_query_dense(const DenseDev &g, float qx, float qy, float qz, int seed_pos)
{
    Best out{~0ull, -1, FLT_MAX};
    if (g.nx <= 0) return out;
    // 32-bit offsets into the two arrays (wave-uniform descriptors, one VALU per address)
    const __amdgpu_buffer_rsrc_t pts = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(g.pts), 0, (g.n_pts + 4) * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(g.start), 0, g.table_bytes, 0x00020000);

    const float ux = cell_pos(qx, g.ox, g.inv_cell), uy = cell_pos(qy, g.oy, g.inv_cell), uz = cell_pos(qz, g.oz, g.inv_cell);
    const int cx = min(max((int)fminf(fmaxf(floorf(ux), -4.0f), 70000.0f), 0), g.nx - 1);
    const int cy = min(max((int)fminf(fmaxf(floorf(uy), -4.0f), 70000.0f), 0), g.ny - 1);
    const int cz = min(max((int)fminf(fmaxf(floorf(uz), -4.0f), 70000.0f), 0), g.nz - 1);
    const float cell2 = g.cell * g.cell;
    float limit2 = g.prune2;
    DBest b{0x7f800000ull << 32, 0xffffffffu};
    if (seed_pos >= 0) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(pts, (uint32_t)seed_pos * 16u, 0, 0);
        dconsider(b, qx, qy, qz, t, (uint32_t)seed_pos * 16u);
        limit2 = fminf(limit2, key_d2(b.key));
    }
    const int base = (int)dense_cell_id(g, cx, cy, cz);

    // ---- ring 0: the query's own cell (it usually holds the nearest point)
    {
        const u32x2 se = __builtin_amdgcn_raw_buffer_load_b64(tab, (uint32_t)base * 4u, 0, 0);
        dscan_range(b, pts, se.x * 16u, se.y * 16u, qx, qy, qz);
        limit2 = fminf(limit2, key_d2(b.key));
    }

    // ---- ring 1: only the cells whose box can still hold something closer.  A neighbour at
    // offset (dx,dy,dz) needs every non-zero axis offset's face to be within the limit, so when
    // no face passes (the common case) the whole ring is skipped with six compares.
    const float lim_c = limit2 / cell2;   // limit in squared cell units (conservative enough: the gaps carry the margin)
    float gx0 = axis_gap(ux, cx - 1, cx - 1), gx2 = axis_gap(ux, cx + 1, cx + 1);
    float gy0 = axis_gap(uy, cy - 1, cy - 1), gy2 = axis_gap(uy, cy + 1, cy + 1);
    float gz0 = axis_gap(uz, cz - 1, cz - 1), gz2 = axis_gap(uz, cz + 1, cz + 1);
    gx0 *= gx0; gx2 *= gx2; gy0 *= gy0; gy2 *= gy2; gz0 *= gz0; gz2 *= gz2;
    const float gx1 = 0.0f, gy1 = 0.0f, gz1 = 0.0f;   // own slab on that axis (clamped queries: still a valid lower bound)
    const bool any_face = (gx0 <= lim_c) | (gx2 <= lim_c) | (gy0 <= lim_c) | (gy2 <= lim_c) | (gz0 <= lim_c) | (gz2 <= lim_c);
    if (any_face) {
        uint32_t mask = 0;   // bit j = dz*9 + dy*3 + dx (offsets 0..2), centre excluded
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            if (j == 13) continue;
            const int dz = j / 9, dy = (j / 3) % 3, dx = j % 3;
            const float lb = (dx == 0 ? gx0 : (dx == 1 ? gx1 : gx2)) + (dy == 0 ? gy0 : (dy == 1 ? gy1 : gy2)) +
                             (dz == 0 ? gz0 : (dz == 1 ? gz1 : gz2));
            mask |= (lb <= lim_c) ? (1u << j) : 0u;
        }
        uint32_t po = 0, pe = 0;
        for (;;) {   // flat: each iteration takes the next plausible cell and/or scores 4 candidates
            if (po >= pe) {
                if (!mask) break;
                const int j = __ffs((int)mask) - 1;
                mask &= mask - 1;
                const int dz = j / 9, dy = (j - dz * 9) / 3, dx = j - dz * 9 - dy * 3;
                const float lb2 = (sel3(gx0, gx1, gx2, dx) + sel3(gy0, gy1, gy2, dy) + sel3(gz0, gz1, gz2, dz)) * cell2;
                if (lb2 <= limit2) {   // the limit may have tightened since the mask was built
                    const int idx = base + (dz - 1) * g.sxy + (dy - 1) * g.sx + (dx - 1);
                    const u32x2 se = __builtin_amdgcn_raw_buffer_load_b64(tab, (uint32_t)idx * 4u, 0, 0);
                    po = se.x * 16u;
                    pe = se.y * 16u;
                }
            }
            if (po < pe) {
                dscan4(b, pts, po, qx, qy, qz);
                po += 64;
                limit2 = fminf(limit2, key_d2(b.key));
            }
        }
    }

    // ---- farther rings: row by row (a row's cells cx-r..cx+r are one contiguous run), nearest
    // ring first, until everything unvisited is provably farther than the best
    for (int r = 2; r <= g.max_ring; ++r) {
        const float reach = ((float)(r - 1) - kCellMargin) * g.cell;   // all of ring r-1 is done
        if (limit2 <= reach * reach) break;
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cz + dz;
            if (z < 0 || z >= g.nz) continue;
            const float az = axis_gap(uz, z, z), gz = az * az;
            if (gz * cell2 > limit2) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cy + dy;
                if (y < 0 || y >= g.ny) continue;
                const float ay = axis_gap(uy, y, y), gyz = ay * ay + gz;
                if (gyz * cell2 > limit2) continue;
                const int row = (int)dense_cell_id(g, 0, y, z);
                const bool face = (abs(dz) == r) || (abs(dy) == r);
                if (face) {   // whole row cx-r .. cx+r
                    const int xa = max(cx - r, 0), xb = min(cx + r, g.nx - 1);
                    const uint32_t s = __builtin_amdgcn_raw_buffer_load_b32(tab, (uint32_t)(row + xa) * 4u, 0, 0);
                    const uint32_t e = __builtin_amdgcn_raw_buffer_load_b32(tab, (uint32_t)(row + xb + 1) * 4u, 0, 0);
                    dscan_range(b, pts, s * 16u, e * 16u, qx, qy, qz);
                    limit2 = fminf(limit2, key_d2(b.key));
                } else {      // only the two end cells belong to ring r
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int x = s ? cx + r : cx - r;
                        if (x < 0 || x >= g.nx) continue;
                        const float ax = axis_gap(ux, x, x);
                        if ((ax * ax + gyz) * cell2 > limit2) continue;
                        const u32x2 se = __builtin_amdgcn_raw_buffer_load_b64(tab, (uint32_t)(row + x) * 4u, 0, 0);
                        dscan_range(b, pts, se.x * 16u, se.y * 16u, qx, qy, qz);
                        limit2 = fminf(limit2, key_d2(b.key));
                    }
                }
            }
        }
    }
    if (b.off != 0xffffffffu) {
        out.key = b.key;
        out.pos = (int)(b.off >> 4);
        out.d2 = key_d2(b.key);
    }
    return out;
}

__global__ __launch_bounds__(kBlock) void k_nn_search_dense(const float4 *cur, uint32_t n, DenseDev g, double gate2,
                                                            int *corr_pos, float *corr_d2, int *seed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = cur[i];
    int pos = -1;
    float d2 = 0.0f;
    if (q.w != 0.0f) {
        const Best b = nn_query_dense(g, q.x, q.y, q.z, seed ? seed[i] : -1);
        if (seed) seed[i] = b.pos;
        if (b.pos >= 0 && !((double)b.d2 > gate2)) {  // PCL: if (distance > max_dist_sqr) continue;
            pos = b.pos;
            d2 = b.d2;
        }
    }
    corr_pos[i] = pos;
    corr_d2[i] = d2;
}

// One ICP iteration in one pass over the dense index: apply the previous increment, search,
// gate, accumulate (same contract and summation order as k_icp_fused).
__global__ __launch_bounds__(kTile) void k_icp_fused_dense(float4 *cur, uint32_t n, Mat34 T, int apply_t, DenseDev g,
                                                           double gate2, int *corr_pos, float *corr_d2, double *partials,
                                                           int *seed, unsigned long long *wave_times)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long t_start = 0;
    if (wave_times) t_start = wall_clock64();
    int pos = -1;
    float d2 = 0.0f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        q = cur[i];
        if (q.w != 0.0f) {
            if (apply_t) {
                const float3 t = xform(T, q.x, q.y, q.z);
                q = make_float4(t.x, t.y, t.z, 1.0f);
                cur[i] = q;
            }
            const Best b = nn_query_dense(g, q.x, q.y, q.z, seed ? seed[i] : -1);
            if (seed) seed[i] = b.pos;
            if (b.pos >= 0 && !((double)b.d2 > gate2)) {
                pos = b.pos;
                d2 = b.d2;
            }
        }
        if (corr_pos) { corr_pos[i] = pos; corr_d2[i] = d2; }
    }
    if (wave_times && (threadIdx.x & 63) == 0) {   // diagnostic build of the launch only (RSREG_WAVE_TIMES)
        const uint32_t w = i >> 6;
        wave_times[2 * w] = t_start;
        wave_times[2 * w + 1] = wall_clock64();
    }
    double a[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) a[k] = 0.0;
    if (pos >= 0) {
        const float4 t = g.pts[pos];
        accum_pair(a, q.x, q.y, q.z, t.x, t.y, t.z, d2);
    }
    tile_reduce_store(a, partials, gridDim.x);
}

}  // namespace rsreg
