// icp_tile_kernel.hpp — the LDS-staged form of the ICP iteration kernel (gfx950, wave64).
//
// One workgroup = 256 consecutive source points in the engine's spatial order (a compact
// patch of the cloud).  The workgroup finds the box of grid cells its queries can touch
// (their cells +- 1), copies the target points of those cells from HBM into LDS once, with
// consecutive lanes reading consecutive points of a cell, and every lane then runs its
// pruned 3x3x3 search against LDS instead of chasing brick -> cell -> point through L2.
// Lanes that still need farther rings (rare once the clouds overlap) continue in global
// memory.  Tiles whose box or point count does not fit the LDS budget take the global-memory
// search of icp_kernels.hpp wholesale; both paths return the same exact nearest neighbour.
//
// Replaces CorrespondenceEstimation::determineCorrespondences + the Umeyama sums of
// pcl::IterativeClosestPoint; reference call sites: include/rsreg.h.
#pragma once

#include <climits>

#include "icp_kernels.hpp"

namespace rsreg {

constexpr int kTilePts = 1536;     // staged target points per tile (24 KB)
constexpr int kTileCells = 384;    // staged occupied cells per tile
constexpr int kTileBox = 2048;     // cells of the dense box table
constexpr int kHaloMax = 3;        // rings of cells staged around the queries, at most
constexpr int kTileBricks = 64;    // bricks overlapping the box (one lane each)

struct TileShared {
    float4 pts[kTilePts];
    uint32_t cofs[kTileCells + 2];   // LDS offset of each staged cell (+ end sentinel)
    uint32_t cgs[kTileCells];        // position of the cell's first point in the sorted target
    uint16_t ctab[kTileBox];         // dense box cell -> staged cell id + 1, 0 = empty
    uint16_t own[kTilePts];          // staged point -> staged cell id (for the flat copy)
    uint32_t wmax[kTileWaves];
    // occupied bricks of the box, compacted
    unsigned long long bclip[kTileBricks], bfull[kTileBricks];
    uint32_t bbase[kTileBricks];
    uint32_t bcoff[kTileBricks + 2];
    uint16_t bcoord[kTileBricks];    // local brick coordinate in the box (x | y << 5 | z << 10)
    uint32_t cpos[kTileWaves][2][66];  // cellpos run of the brick a wave is unpacking (2 in flight)
    int box[6];
    uint32_t wsum[kTileWaves];
    uint32_t totals[3];              // staged cells, staged points, occupied bricks
};

__device__ __forceinline__ void scan_points_lds(const float4 *pts, uint32_t s, uint32_t e, uint32_t gpos0, float qx, float qy,
                                                float qz, Best &b)
{
    if (s >= e) return;
    const uint32_t last = e - 1;
    for (uint32_t p = s; p < e; p += 4) {
        const uint32_t p1 = min(p + 1, last), p2 = min(p + 2, last), p3 = min(p + 3, last);
        const float4 t0 = pts[p], t1 = pts[p1], t2 = pts[p2], t3 = pts[p3];
        consider(b, l2_simple(qx, qy, qz, t0.x, t0.y, tgt_z(t0)), tgt_idx(t0), gpos0 + (p - s));
        consider(b, l2_simple(qx, qy, qz, t1.x, t1.y, tgt_z(t1)), tgt_idx(t1), gpos0 + (p1 - s));
        consider(b, l2_simple(qx, qy, qz, t2.x, t2.y, tgt_z(t2)), tgt_idx(t2), gpos0 + (p2 - s));
        consider(b, l2_simple(qx, qy, qz, t3.x, t3.y, tgt_z(t3)), tgt_idx(t3), gpos0 + (p3 - s));
    }
}

__device__ __forceinline__ int wave_min_i(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// apply_t: transform the point by T first (and write it back); write_corr: store the
// correspondences; accumulate: reduce the 17 sums of the accepted pairs into partials.
__global__ __launch_bounds__(kTile) void k_icp_tile(float4 *cur, uint32_t n, Mat34 T, int apply_t, GridDev g, double gate2,
                                                    int *corr_pos, float *corr_d2, int write_corr, double *partials,
                                                    int accumulate, uint32_t *stats, int *seed)
{
    __shared__ TileShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i = blockIdx.x * blockDim.x + tid;

    // ---- A. load (and move) the query, find the tile's cell box
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) q = cur[i];
    const bool valid = (i < n) && q.w != 0.0f && g.dx > 0;
    if ((i < n) && q.w != 0.0f && apply_t) {
        const float3 t = xform(T, q.x, q.y, q.z);
        q = make_float4(t.x, t.y, t.z, q.w);
        cur[i] = q;
    }
    QueryGeom qg{0.f, 0.f, 0.f, 0, 0, 0};
    if (valid) qg = query_geom(g, q.x, q.y, q.z);
    if (tid < 3) { sh.box[tid] = INT_MAX; sh.box[3 + tid] = INT_MIN; }
    __syncthreads();
    {
        const int mnx = wave_min_i(valid ? qg.cx : INT_MAX), mny = wave_min_i(valid ? qg.cy : INT_MAX),
                  mnz = wave_min_i(valid ? qg.cz : INT_MAX);
        const int mxx = wave_max_i(valid ? qg.cx : INT_MIN), mxy = wave_max_i(valid ? qg.cy : INT_MIN),
                  mxz = wave_max_i(valid ? qg.cz : INT_MIN);
        if (lane == 0 && mnx <= mxx) {
            atomicMin(&sh.box[0], mnx); atomicMin(&sh.box[1], mny); atomicMin(&sh.box[2], mnz);
            atomicMax(&sh.box[3], mxx); atomicMax(&sh.box[4], mxy); atomicMax(&sh.box[5], mxz);
        }
    }
    __syncthreads();
    const bool any_valid = sh.box[0] <= sh.box[3];
    // box = the queries' cells +- H rings, clipped to the grid
    const int H = g.halo;
    const int x0 = max(sh.box[0] - H, 0), x1 = min(sh.box[3] + H, g.dx - 1);
    const int y0 = max(sh.box[1] - H, 0), y1 = min(sh.box[4] + H, g.dy - 1);
    const int z0 = max(sh.box[2] - H, 0), z1 = min(sh.box[5] + H, g.dz - 1);
    const int bnx = x1 - x0 + 1, bny = y1 - y0 + 1, bnz = z1 - z0 + 1;
    const bool box_ok = any_valid && bnx > 0 && bny > 0 && bnz > 0;
    const int bx0 = x0 >> 2, by0 = y0 >> 2, bz0 = z0 >> 2;
    const int nbx = (x1 >> 2) - bx0 + 1, nby = (y1 >> 2) - by0 + 1, nbz = (z1 >> 2) - bz0 + 1;
    long long box_cells = 0, nb_ll = 0;
    if (box_ok) { box_cells = (long long)bnx * bny * bnz; nb_ll = (long long)nbx * nby * nbz; }
    bool staged = box_ok && box_cells <= kTileBox && nb_ll <= kTileBricks;
    const int nb = staged ? (int)nb_ll : 0;
    uint32_t n_cells = 0, n_pts = 0, n_occ = 0;

    if (staged) {
        // ---- B. clear the box table; C. one lane per brick of the box: hash lookup, clip to
        // the box, compact the occupied ones (first wave: nb <= 64)
        for (int k = tid; k < (int)box_cells; k += kTile) sh.ctab[k] = 0;
        if (wave == 0) {
            unsigned long long mask = 0, clip = 0;
            uint32_t base = 0;
            int lbx = 0, lby = 0, lbz = 0;
            if (lane < nb) {
                lbx = lane % nbx; lby = (lane / nbx) % nby; lbz = lane / (nbx * nby);
                const int bx = bx0 + lbx, by = by0 + lby, bz = bz0 + lbz;
                if (brick_lookup(g, bx, by, bz, mask, base))
                    clip = mask & box_mask(max(x0 - (bx << 2), 0), min(x1 - (bx << 2), 3), max(y0 - (by << 2), 0),
                                           min(y1 - (by << 2), 3), max(z0 - (bz << 2), 0), min(z1 - (bz << 2), 3));
            }
            const uint32_t nc = __popcll(clip);
            const unsigned long long occ = __ballot(nc != 0);
            const uint32_t slot = __popcll(occ & ((1ull << lane) - 1ull));
            const uint32_t incl = wave_incl_scan(nc);
            if (nc) {
                sh.bclip[slot] = clip;
                sh.bfull[slot] = mask;
                sh.bbase[slot] = base;
                sh.bcoff[slot] = incl - nc;
                sh.bcoord[slot] = (uint16_t)(lbx | lby << 5 | lbz << 10);
            }
            if (lane == 63) { sh.totals[0] = incl; sh.totals[2] = __popcll(occ); }
        }
        __syncthreads();
        n_cells = sh.totals[0];
        n_occ = sh.totals[2];
        staged = n_cells <= kTileCells;
    }
    if (staged) {
        // ---- E. each wave unpacks occupied bricks: one coalesced read of the brick's run of
        // cellpos, then one lane per cell bit writes the cell record and the box table entry.
        // Two bricks per wave are in flight so that their loads overlap.
        for (uint32_t b0 = wave * 2; b0 < n_occ; b0 += kTileWaves * 2) {
            uint32_t v[2] = {0, 0};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t b = b0 + u;
                if (b < n_occ) {
                    const uint32_t nfull = __popcll(sh.bfull[b]);
                    if ((uint32_t)lane <= nfull) v[u] = g.cellpos[sh.bbase[b] + lane];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t b = b0 + u;
                if (b < n_occ) {
                    sh.cpos[wave][u][lane] = v[u];
                    const uint32_t nfull = __popcll(sh.bfull[b]);
                    if (lane == 0 && nfull == 64) sh.cpos[wave][u][64] = g.cellpos[sh.bbase[b] + 64];
                }
            }
            __builtin_amdgcn_wave_barrier();  // same wave wrote and now reads cpos: LDS ops of a wave complete in order
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t b = b0 + u;
                if (b >= n_occ) continue;
                const unsigned long long clip = sh.bclip[b];
                if ((clip >> lane) & 1ull) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const uint32_t cid = sh.bcoff[b] + __popcll(clip & below);
                    const uint32_t r = __popcll(sh.bfull[b] & below);
                    const uint32_t gs = sh.cpos[wave][u][r], ge = sh.cpos[wave][u][r + 1];
                    sh.cgs[cid] = gs;
                    sh.cofs[cid] = ge - gs;  // count for now, offset after the scan
                    const uint32_t bc = sh.bcoord[b];
                    const int x = ((bx0 + (int)(bc & 31)) << 2) | (lane & 3), y = ((by0 + (int)((bc >> 5) & 31)) << 2) | ((lane >> 2) & 3),
                              z = ((bz0 + (int)(bc >> 10)) << 2) | (lane >> 4);
                    sh.ctab[((z - z0) * bny + (y - y0)) * bnx + (x - x0)] = (uint16_t)(cid + 1);
                }
            }
        }
        __syncthreads();
        // ---- F. block exclusive scan of the cell sizes (3 cells per lane)
        {
            const uint32_t i0 = 3 * tid;
            uint32_t c[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) c[u] = (i0 + u) < n_cells ? sh.cofs[i0 + u] : 0;
            const uint32_t tot = c[0] + c[1] + c[2];
            const uint32_t incl = wave_incl_scan(tot);
            if (lane == 63) sh.wsum[wave] = incl;
            __syncthreads();
            uint32_t woff = 0, all = 0;
            for (int w = 0; w < kTileWaves; ++w) { if (w < wave) woff += sh.wsum[w]; all += sh.wsum[w]; }
            uint32_t run = woff + incl - tot;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (i0 + u <= n_cells) sh.cofs[i0 + u] = run;
                run += c[u];
            }
            n_pts = all;
        }
        if (tid == 0) sh.cofs[n_cells] = n_pts;   // (index 3*kTile would be out of the lanes' range)
        staged = n_pts <= kTilePts;
        __syncthreads();
    }
    if (staged) {
        // ---- G. stage the points with a flat, balanced copy.  own[p] = staged cell of staged
        // point p: heads are scattered at each cell's first offset, a running maximum fills
        // the rest (cell ids grow with the offset).
        constexpr int CH = (kTilePts + kTile - 1) / kTile;  // consecutive points per lane in the fill
        for (uint32_t p = tid; p < n_pts; p += kTile) sh.own[p] = 0;
        __syncthreads();
        for (uint32_t c = tid; c < n_cells; c += kTile) sh.own[sh.cofs[c]] = (uint16_t)c;
        __syncthreads();
        {
            const uint32_t p0 = tid * CH;
            uint32_t m = 0;
            for (int k = 0; k < CH; ++k)
                if (p0 + k < n_pts) m = max(m, (uint32_t)sh.own[p0 + k]);
            uint32_t incl = m;   // inclusive running max across the wave
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off);
                if (lane >= off) incl = max(incl, t);
            }
            if (lane == 63) sh.wmax[wave] = incl;
            uint32_t carry = __shfl_up(incl, 1);
            if (lane == 0) carry = 0;
            __syncthreads();
            for (int w = 0; w < wave; ++w) carry = max(carry, sh.wmax[w]);
            for (int k = 0; k < CH; ++k)
                if (p0 + k < n_pts) {
                    carry = max(carry, (uint32_t)sh.own[p0 + k]);
                    sh.own[p0 + k] = (uint16_t)carry;
                }
        }
        __syncthreads();
        for (uint32_t p = tid; p < n_pts; p += 4 * kTile) {
            float4 v[4];
            uint32_t pp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pp[u] = p + u * kTile;
                if (pp[u] < n_pts) {
                    const uint32_t c = sh.own[pp[u]];
                    v[u] = g.pts[sh.cgs[c] + (pp[u] - sh.cofs[c])];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pp[u] < n_pts) sh.pts[pp[u]] = v[u];
        }
        __syncthreads();
    }
    if (stats && tid == 0) {
        atomicAdd(&stats[staged ? 0 : (any_valid ? 1 : 2)], 1u);
        if (!staged && any_valid) {
            const int why = !(box_cells <= kTileBox) ? 3 : !(nb_ll <= kTileBricks) ? 4 : (n_cells > kTileCells) ? 5 : 6;
            atomicAdd(&stats[why], 1u);
            atomicMax(&stats[7], (uint32_t)min(box_cells, 1000000ll));
            atomicMax(&stats[8], n_pts);
        }
    }

    // ---- H. per-lane pruned search
    Best b{~0ull, -1, FLT_MAX};
    if (valid) {
        float limit2 = g.prune2;
        int inner = 1;
        const int seed_pos = seed ? seed[i] : -1;
        if (seed_pos >= 0) {
            const float4 t = g.pts[seed_pos];
            consider(b, l2_simple(q.x, q.y, q.z, t.x, t.y, tgt_z(t)), tgt_idx(t), (uint32_t)seed_pos);
            limit2 = fminf(limit2, b.d2);
        }
        if (staged) {
            const float cell2 = g.cell * g.cell;
            // ring 0 (own cell: it usually holds the nearest point), then rings 1..H, stopping
            // as soon as everything unvisited is provably farther than the best so far
            for (int r = 0; r <= H; ++r) {
                for (int dz = -r; dz <= r; ++dz) {
                    const int z = qg.cz + dz;
                    if (z < z0 || z > z1) continue;
                    const float az = axis_gap(qg.uz, z, z), gz2 = az * az;
                    if (gz2 * cell2 > limit2) continue;
                    for (int dy = -r; dy <= r; ++dy) {
                        const int y = qg.cy + dy;
                        if (y < y0 || y > y1) continue;
                        const float ay = axis_gap(qg.uy, y, y), gyz2 = ay * ay + gz2;
                        if (gyz2 * cell2 > limit2) continue;
                        const bool face = (abs(dz) == r) || (abs(dy) == r);
                        const int step = face ? 1 : max(2 * r, 1);
                        const int row = ((z - z0) * bny + (y - y0)) * bnx - x0;
                        for (int dx = -r; dx <= r; dx += step) {
                            const int x = qg.cx + dx;
                            if (x < x0 || x > x1) continue;
                            const float ax = axis_gap(qg.ux, x, x);
                            if ((ax * ax + gyz2) * cell2 > limit2) continue;
                            const uint32_t cid1 = sh.ctab[row + x];
                            if (!cid1) continue;
                            const uint32_t s = sh.cofs[cid1 - 1], e = sh.cofs[cid1];
                            scan_points_lds(sh.pts, s, e, sh.cgs[cid1 - 1], q.x, q.y, q.z, b);
                            limit2 = fminf(limit2, b.d2);
                        }
                    }
                }
                const float reach = fmaxf((float)r - kCellMargin, 0.0f) * g.cell;
                if (limit2 <= reach * reach) break;
            }
            inner = H;
        } else {
            nn_near_global(g, qg, q.x, q.y, q.z, b, limit2);
        }
        nn_far_global(g, qg, q.x, q.y, q.z, b, limit2, inner);
        if (seed) seed[i] = b.pos;
    }

    // ---- I. gate, outputs, sums
    int pos = -1;
    float d2 = 0.0f;
    if (b.pos >= 0 && !((double)b.d2 > gate2)) {  // PCL: if (distance > max_dist_sqr) continue;
        pos = b.pos;
        d2 = b.d2;
    }
    if (write_corr && i < n) { corr_pos[i] = pos; corr_d2[i] = d2; }
    if (accumulate) {
        double a[RSREG_NUM_SUMS];
        for (int k = 0; k < RSREG_NUM_SUMS; ++k) a[k] = 0.0;
        if (pos >= 0) {
            const float4 t = g.pts[pos];
            accum_pair(a, q.x, q.y, q.z, t.x, t.y, tgt_z(t), d2, q.w);
        }
        tile_reduce_store(a, partials, gridDim.x);
    }
}

}  // namespace rsreg
