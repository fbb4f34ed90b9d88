// icp_rows.hpp — one ICP iteration with the target staged through LDS row by row (gfx950, wave64).
//
// The dense index (icp_dense.hpp) sorts the target by cell id, x fastest, and by x inside a cell: the cells
// x0 .. x1 of one (y, z) row are ONE contiguous, x-sorted run of records.  A tile of 128 Morton-neighbouring
// queries covers a few centimetres; after the first iteration every query carries the match of the previous one, so
// the box that can hold a closer point is known before anything is searched.  The workgroup therefore
//   1. reduces the per-query windows (cells the query's limit reaches) to one region of rows x cells,
//   2. one thread per row finds the row's occupied span (occupancy words) and its record range (two table entries),
//   3. copies all those ranges into LDS with whole-wave coalesced 16-byte loads (each target record once per tile),
//   4. every lane searches the rows of its own window in LDS: a binary search for the query's x in the run, then a
//      walk to both sides that ends where the x distance alone exceeds the best so far.
// A region of more than kRowsCap records is staged in passes; a region of more than kRowsMax rows or kRowsMaxW cells
// (a tile that straddles a jump of the Morton curve, an unbounded gate) is searched from global memory by
// nn_query_dense.  Same candidates, same float operations, same (distance, original index) order as every other
// search of the engine: matches and sums are bit-identical.
//
// Replaces the per-iteration correspondence search + transformation sums of pcl::IterativeClosestPoint::align
// (incremental_icp.hpp:59), like k_icp_fused_dense.
#pragma once

#include "icp_dense.hpp"

namespace rsreg {

constexpr int kRowsMax = 128;    // rows of a region: one thread of the tile each
constexpr int kRowsMaxW = 30;    // cells of a row of the region (their occupancy is one 32-bit mask)
constexpr int kRowsCap = 1024;   // records staged per pass (16 KB)

struct RowsStats {   // RSREG_ROWS_STATS: per launch
    uint32_t tiles, fallback_tiles, passes, records, rows, rows_nonempty, row_visits, candidates, bs_steps;
};

struct RowsShared {
    u32x4 pts[kRowsCap];
    uint32_t off[kRowsMax + 1];   // records of the rows before row r (the region's rows back to back)
    uint32_t gs[kRowsMax];        // first record of row r in the sorted target array
    int box[kTileWaves][6];
    uint32_t wsum[kTileWaves];
};

__device__ __forceinline__ int rows_cell_clamp(float v) { return (int)fminf(fmaxf(floorf(v), -4.0f), 70000.0f); }

// one sorted run [a, e) of LDS records, all in the row whose (y, z) box is at least sqrt(yz2) away
template <bool kStats>
__device__ __forceinline__ void rows_search_run(const u32x4 *lds, uint32_t a, uint32_t e, f32x2 qxy, float qz, float yz2, float x_slack,
                                                DBest &b, float &limit2, uint32_t *n_cand, uint32_t *n_bs)
{
    // first record whose x is not below the query's (the order is exact up to x_slack, which the walk's exits allow for)
    uint32_t lo = a, hi = e;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const float x = __uint_as_float(reinterpret_cast<const uint32_t *>(lds + mid)[0]);
        if (x < qxy.x) lo = mid + 1; else hi = mid;
        if (kStats) ++*n_bs;
    }
    // walk to both sides; a side is left once the x distance alone (plus what the row is away) exceeds the best so far.
    // A side that is done re-reads its last record: scoring a record twice changes nothing.
    uint32_t pl = lo > a ? lo - 1 : a, pr = lo < e ? lo : e - 1;
    bool go_l = lo > a, go_r = lo < e;
    while (go_l || go_r) {
        const u32x4 tl = lds[pl], tr = lds[pr];
        dconsider(b, qxy, qz, tl);
        dconsider(b, qxy, qz, tr);
        if (kStats) *n_cand += (go_l ? 1u : 0u) + (go_r ? 1u : 0u);
        limit2 = min_nn(limit2, b.d);
        const float gl = (qxy.x - __uint_as_float(tl.x)) - x_slack, gr = (__uint_as_float(tr.x) - qxy.x) - x_slack;
        go_l = go_l && pl > a && !(gl > 0.0f && gl * gl + yz2 > limit2);
        go_r = go_r && pr + 1 < e && !(gr > 0.0f && gr * gr + yz2 > limit2);
        pl -= go_l ? 1u : 0u;
        pr += go_r ? 1u : 0u;
    }
}

template <int kFar, bool kStats>
__global__ __launch_bounds__(kTile, 4) void k_icp_fused_rows(float4 *cur, uint32_t n, Mat34 T, int apply_t, DenseDev g, double gate2,
                                                             int *corr_pos, float *corr_d2, double *partials, int *seed,
                                                             const IcpDevState *dev, TileSched sched, RowsStats *stats)
{
    __shared__ RowsShared sh;
    const uint32_t item = sched.items ? sched.items[blockIdx.x] : blockIdx.x;
    if (item == 0xffffffffu) return;
    const uint32_t tile = item & 0xffffffu;   // (the host never splits tiles for this kernel)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t i = tile * kTile + tid;
    if (dev) {
        T = dev->t_inc;
        apply_t = dev->apply;
    }
    unsigned long long t_start = 0;
    if (sched.cost) t_start = wall_clock64();

    // ---- the query, its seed, and the window of cells its limit reaches
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) q = cur[i];
    const bool active = q.w != 0.0f;
    const DRes rs = dense_res(g);
    DQuery dq{};
    DBest b{__uint_as_float(0x7f800000u), 0xffffffffu};
    float limit2 = g.prune2;
    int seed_in = -1;
    int ylo = 0x7fffffff, yhi = -0x7fffffff, zlo = 0x7fffffff, zhi = -0x7fffffff, xlo = 0x7fffffff, xhi = -0x7fffffff;
    if (active) {
        if (apply_t) {
            const float3 t = xform(T, q.x, q.y, q.z);
            q = make_float4(t.x, t.y, t.z, q.w);
            cur[i] = q;
        }
        dq = dense_query(g, q.x, q.y, q.z);
        seed_in = seed ? seed[i] : -1;
        dense_seed(rs, dq, seed_in, b, limit2);
        // nothing farther than sqrt(limit2) matters: in cell units, with the slack of the cell assignment's rounding
        const float rc = sqrtf(limit2) * g.inv_cell + g.margin + 1e-4f;
        ylo = min(max(rows_cell_clamp(dq.uy - rc), 0), g.ny - 1); yhi = min(max(rows_cell_clamp(dq.uy + rc), 0), g.ny - 1);
        zlo = min(max(rows_cell_clamp(dq.uz - rc), 0), g.nz - 1); zhi = min(max(rows_cell_clamp(dq.uz + rc), 0), g.nz - 1);
        xlo = min(max(rows_cell_clamp(dq.ux - rc), 0), g.nx - 1); xhi = min(max(rows_cell_clamp(dq.ux + rc), 0), g.nx - 1);
    }
    // ---- the tile's region
    {
        int v[6] = {ylo, -yhi, zlo, -zhi, xlo, -xhi};   // all minima
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) v[k] = min(v[k], __shfl_xor(v[k], m));
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) sh.box[wave][k] = v[k];
        }
    }
    __syncthreads();
    int Y0 = sh.box[0][0], Y1 = -sh.box[0][1], Z0 = sh.box[0][2], Z1 = -sh.box[0][3], X0 = sh.box[0][4], X1 = -sh.box[0][5];
#pragma unroll
    for (int w = 1; w < kTileWaves; ++w) {
        Y0 = min(Y0, sh.box[w][0]); Y1 = max(Y1, -sh.box[w][1]);
        Z0 = min(Z0, sh.box[w][2]); Z1 = max(Z1, -sh.box[w][3]);
        X0 = min(X0, sh.box[w][4]); X1 = max(X1, -sh.box[w][5]);
    }
    const bool any = Y0 <= Y1;   // (no valid query in the tile: nothing to search)
    const int nyr = any ? Y1 - Y0 + 1 : 0, nzr = any ? Z1 - Z0 + 1 : 0, W = any ? X1 - X0 + 1 : 0;
    const bool staged = any && (long long)nyr * nzr <= kRowsMax && W <= kRowsMaxW;
    const uint32_t R = staged ? (uint32_t)(nyr * nzr) : 0u;
    uint32_t n_visit = 0, n_cand = 0, n_bs = 0;

    if (any && !staged) {
        // ---- a region too large for the directory: this tile is searched from global memory
        if (active) {
            const Best r = nn_query_dense<false, kFar>(g, q.x, q.y, q.z, seed_in);
            if (r.pos >= 0) { b.d = r.d2; b.idx = (uint32_t)r.key; }
            else { b.d = __uint_as_float(0x7f800000u); b.idx = 0xffffffffu; }
        }
    } else if (staged) {
        // ---- the directory: thread r looks row r up
        uint32_t len = 0, gs = 0;
        if (tid < R) {
            const int y = Y0 + (int)(tid % (uint32_t)nyr), z = Z0 + (int)(tid / (uint32_t)nyr);
            const uint32_t id0 = dense_cell_id(g, 0, y, z);   // cell x of this row is id0 + x
            // bits 12..14 of the occupancy word of cell c: do cells c - 1, c, c + 1 of this row hold points
            uint32_t mask = 0;
            const int nw = (W + 2) / 3;
            uint32_t w[10];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b32(rs.nbr, (id0 + (uint32_t)(X0 + 1 + 3 * k)) * 4u, 0, 0);
            if (nw > 4) {
#pragma unroll
                for (int k = 4; k < 10; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b32(rs.nbr, (id0 + (uint32_t)(X0 + 1 + 3 * k)) * 4u, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (k < 4 || nw > 4) mask |= k < nw ? ((w[k] >> 12) & 7u) << (3 * k) : 0u;
            mask &= W >= 32 ? 0xffffffffu : ((1u << W) - 1u);
            if (mask) {
                const int first = __ffs((int)mask) - 1, last = 31 - __clz((int)mask);
                // an occupied cell's entry is its first record, the entry behind it its end (k_dense_scatter)
                const uint32_t s = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (id0 + (uint32_t)(X0 + first)) * 4u, 0, 0);
                const uint32_t e = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (id0 + (uint32_t)(X0 + last + 1)) * 4u, 0, 0);
                gs = s;
                len = e > s ? e - s : 0u;
            }
        }
        uint32_t incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if ((int)lane >= off) incl += v;
        }
        if (lane == 63u) sh.wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, P = 0;
#pragma unroll
        for (int w = 0; w < kTileWaves; ++w) {
            if ((uint32_t)w < wave) before += sh.wsum[w];
            P += sh.wsum[w];
        }
        sh.off[tid] = before + incl - len;
        sh.gs[tid] = gs;
        if (tid == kTile - 1) sh.off[kRowsMax] = P;
        static_assert(kRowsMax == kTile, "one thread per row of the directory");
        __syncthreads();
        if (kStats && tid == 0) {
            atomicAdd(&stats->records, P);
            atomicAdd(&stats->rows, R);
            atomicAdd(&stats->passes, (P + kRowsCap - 1) / kRowsCap);
        }
        if (kStats && len) atomicAdd(&stats->rows_nonempty, 1u);

        const f32x2 qxy = {q.x, q.y};
        const float cell2 = g.cell * g.cell, x_slack = g.x_slack;
        for (uint32_t pass_lo = 0; pass_lo < P; pass_lo += kRowsCap) {
            const uint32_t pass_n = min(P - pass_lo, (uint32_t)kRowsCap);
            if (pass_lo) __syncthreads();   // (everyone has finished with the records of the pass before)
            // ---- stage: record pass_lo + j of the region goes to slot j, 128 consecutive records per trip
            {
                uint32_t r = 0;
                {   // the row of this thread's first record: the last r with off[r] <= flat
                    const uint32_t flat = pass_lo + min(tid, pass_n - 1u);
                    uint32_t lo = 0, hi = kRowsMax;   // off[lo] <= flat < off[hi] (off[kRowsMax] = P > flat)
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (sh.off[mid] <= flat) lo = mid; else hi = mid;
                    }
                    r = lo;
                }
                for (uint32_t j0 = tid; j0 < pass_n; j0 += 4u * kTile) {
                    uint32_t src[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t j = j0 + (uint32_t)k * kTile;
                        src[k] = 0xffffffffu;
                        if (j < pass_n) {
                            const uint32_t flat = pass_lo + j;
                            while (flat >= sh.off[r + 1]) ++r;
                            src[k] = sh.gs[r] + (flat - sh.off[r]);
                        }
                    }
                    u32x4 rec[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (src[k] != 0xffffffffu) rec[k] = __builtin_amdgcn_raw_buffer_load_b128(rs.pts, src[k] * 16u, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (src[k] != 0xffffffffu) sh.pts[j0 + (uint32_t)k * kTile] = rec[k];
                }
            }
            __syncthreads();
            // ---- search: the rows of this lane's window, its own row first
            if (active) {
                int iy = ylo - 1, iz = zlo;
                bool first = true;
                for (;;) {
                    int ry, rz;
                    if (first) {
                        ry = dq.cy; rz = dq.cz; first = false;
                    } else {
                        ++iy;
                        if (iy > yhi) { iy = ylo; ++iz; }
                        if (iz > zhi) break;
                        if (iy == dq.cy && iz == dq.cz) continue;
                        ry = iy; rz = iz;
                    }
                    const uint32_t r = (uint32_t)((rz - Z0) * nyr + (ry - Y0));
                    const uint32_t a = max(sh.off[r], pass_lo), e = min(sh.off[r + 1], pass_lo + pass_n);
                    if (a >= e) continue;
                    const float ay = axis_gap(dq.uy, ry, ry, g.margin), az = axis_gap(dq.uz, rz, rz, g.margin);
                    const float yz2 = (ay * ay + az * az) * cell2;
                    if (yz2 > limit2) continue;
                    if (kStats) ++n_visit;
                    rows_search_run<kStats>(sh.pts, a - pass_lo, e - pass_lo, qxy, q.z, yz2, x_slack, b, limit2, &n_cand, &n_bs);
                }
            }
        }
    }
    if (kStats) {
        if (tid == 0) {
            atomicAdd(&stats->tiles, 1u);
            if (any && !staged) atomicAdd(&stats->fallback_tiles, 1u);
        }
        atomicAdd(&stats->row_visits, n_visit);
        atomicAdd(&stats->candidates, n_cand);
        atomicAdd(&stats->bs_steps, n_bs);
    }

    // ---- the match, the gate, the sums (as k_icp_fused_dense)
    int pos = -1;
    float d2 = 0.0f;
    if (active) {
        const Best r = dense_result(g, b);
        if (seed && r.pos != seed_in) seed[i] = r.pos;
        if (r.pos >= 0 && !((double)r.d2 > gate2)) {
            pos = r.pos;
            d2 = r.d2;
        }
    }
    if (corr_pos && i < n) { corr_pos[i] = pos; corr_d2[i] = d2; }
    if (sched.cost && lane == 0)
        sched.cost[tile * kTileWaves + wave] = (uint32_t)min(wall_clock64() - t_start, 0xffffffffull);
    double a17[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) a17[k] = 0.0;
    if (pos >= 0) {
        const float4 t = g.pts[pos];
        accum_pair(a17, q.x, q.y, q.z, t.x, t.y, tgt_z(t), d2, q.w);
    }
    tile_reduce_store(a17, partials, sched.n_tiles, tile);
}

}  // namespace rsreg
