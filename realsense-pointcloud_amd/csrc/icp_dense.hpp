// icp_dense.hpp — dense-table form of the target index and its search (gfx950, wave64).
//
// When the grid over the target's bounding box has a bounded number of cells (a room-scale
// depth-camera cloud at ~1 cm cells: ~10^7 cells, tens of MB out of 288 GB of HBM), the
// engine keeps `start[cell]` for EVERY cell instead of hashing the occupied ones:
//   * points are sorted by linear cell id (x fastest) and by x inside a cell, so the cells
//     x-1, x, x+1 of one (y, z) row are one contiguous, x-sorted run of points -- a search walks
//     a cell from the end nearer to the query and stops once the x distance alone is too large;
//   * a word per cell says which of its 27 neighbours hold points: empty cells are never opened;
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
    float margin;            // slack (in cells) on every geometric lower bound: covers the float rounding of
                             // cell assignment, which grows with the grid (6e-7 x its largest dimension, >= 0.004)
    float x_slack;           // metres: slack on the x order of a sorted run (sort-key bucket + float rounding of the position)
    const uint32_t *nbr;     // per cell: bit j = dz*9+dy*3+dx (offsets 0..2) set when that neighbour holds points
    const uint32_t *pos_of;  // original index -> position in pts (kept points only)
#ifdef RSREG_DIAG
    uint32_t debug_skip;     // diagnostic builds only (-DRSREG_DIAG; RSREG_DEBUG_SKIP, results WRONG): 1 = no search beyond ring 1, 3 = nothing beyond the own cell
#endif
};

__device__ __forceinline__ uint32_t dense_cell_id(const DenseDev &g, int x, int y, int z)
{
    return (uint32_t)(((z + 1) * (g.ny + 2) + (y + 1)) * (g.nx + 2) + (x + 1));
}

// sort key: [padded cell id | x position inside the cell, `xbits` bits]: a cell's points are sorted
// by x, which lets a search stop inside a cell (dense_walk); non-finite points sort to the very end (all-ones key).
// KeyT = uint32_t whenever cell id and x position fit 32 bits together (any room-scale cloud: 23 + 9 bits at 1 M
// points): the radix sort then moves half the bytes in four passes instead of six.
// (sort_scratch: the state of the radix sort that follows, cleared on the way -- radix32.hpp; occ: the occupancy words
// k_dense_nbr will OR together, cleared on the way too -- as a hipMemsetAsync they were two fill kernels and 8 us of
// launch gaps at the head of every build)
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_dense_keys(const char *pts, size_t stride, uint32_t n, DenseDev g, uint32_t xbits,
                                                       KeyT *keys, uint32_t *vals, uint32_t *sort_scratch, uint32_t sort_scratch_words,
                                                       uint32_t *occ, uint32_t occ_words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (sort_scratch) radix32_clear(sort_scratch, sort_scratch_words, i, gridDim.x * blockDim.x);
    if (occ) radix32_clear(occ, occ_words, i, gridDim.x * blockDim.x);
    if (i >= n) return;
    const float *p = rec_xyz(pts, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    KeyT key = (KeyT)~(KeyT)0;
    if (finite3(x, y, z)) {
        const int cx = min(max(cell_coord(x, g.ox, g.inv_cell), 0), g.nx - 1), cy = min(max(cell_coord(y, g.oy, g.inv_cell), 0), g.ny - 1),
                  cz = min(max(cell_coord(z, g.oz, g.inv_cell), 0), g.nz - 1);
        const float fx = (cell_pos(x, g.ox, g.inv_cell) - (float)cx) * (float)(1u << xbits);
        key = ((KeyT)dense_cell_id(g, cx, cy, cz) << xbits) | (KeyT)min(max((int)fx, 0), (int)(1u << xbits) - 1);
    }
    keys[i] = key;
    vals[i] = i;
}

// flags[i] = keep | cstart << 32.  keep: not a value-equal duplicate of its predecessor in the same (cell, x bucket)
// run; cstart: first point of a cell (always kept).  One 64-bit exclusive scan of the flags gives both running
// counts at once: pos (low word) and cid (high word).
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_dense_flag(const KeyT *keys, const uint32_t *vals, const char *pts,
                                                       size_t stride, uint32_t nfin, uint32_t xbits, unsigned long long *flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    uint32_t kp = 1, cs = 1;
    if (i > 0) {
        const KeyT k = keys[i], kprev = keys[i - 1];
        cs = (k >> xbits) != (kprev >> xbits);
        if (k == kprev) {
            const float *a = rec_xyz(pts, stride, vals[i]);
            const float *b = rec_xyz(pts, stride, vals[i - 1]);
            if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) kp = 0;
        }
    }
    flags[i] = (unsigned long long)kp | (unsigned long long)cs << 32;
}

// the sorted point array, and for every occupied cell (in sorted order) its table slot and the
// position of its first point; scan[i] = exclusive scan of flags: pos | cid << 32.
// stats[0] = occupied cells, stats[2] = kept points.
// table != null (grids searched through the occupancy words only, max_ring <= 4): the two table entries a search can
// ever read of an occupied cell -- its first point and, one slot on, its end (= the first point of the next occupied
// cell in sorted order) -- are written here, and the table is neither cleared nor scanned (stale
// entries of earlier builds lie only where no occupancy bit points).
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_dense_scatter(const KeyT *keys, const uint32_t *vals, const char *pts,
                                                          size_t stride, uint32_t nfin, uint32_t xbits, const unsigned long long *flags,
                                                          const unsigned long long *scan, float4 *sorted, uint32_t *pos_of,
                                                          uint32_t *cellslot, uint32_t *cellpos, uint32_t *stats, uint32_t *host_stats,
                                                          uint32_t *table)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    const unsigned long long f = flags[i], sc = scan[i];
    const uint32_t keep = (uint32_t)f, cstart = (uint32_t)(f >> 32), pos = (uint32_t)sc, cid = (uint32_t)(sc >> 32);
    const uint32_t slot = (uint32_t)(keys[i] >> xbits);
    if (keep) {
        const uint32_t v = vals[i];
        const float *p = rec_xyz(pts, stride, v);
        sorted[pos] = tgt_rec(p[0], p[1], p[2], v);
        pos_of[v] = pos;
    }
    if (cstart) {
        cellslot[cid] = slot;
        cellpos[cid] = pos;
        if (table) {
            table[slot] = pos;
            if (i > 0) table[(uint32_t)(keys[i - 1] >> xbits) + 1u] = pos;   // (the same value when the two slots coincide)
        }
    }
    if (i == nfin - 1) {
        const uint32_t nu = pos + keep, nc = cid + cstart;
        stats[0] = nc;
        stats[2] = nu;
        host_stats[0] = nc;   // (pinned host memory: what the host reads when the build has drained, no copy queued)
        host_stats[2] = nu;
        cellpos[nc] = nu;   // sentinel
        if (table) table[slot + 1u] = nu;
        // far-away points behind the last sorted point: a 4-wide candidate read may run past it
        for (uint32_t k = 0; k < 4; ++k) sorted[nu + k] = tgt_rec(1e30f, 1e30f, 1e30f, 0xffffffffu);
    }
}

// k_dense_flag, the scan of the flags and k_dense_scatter in one launch (compact.hpp): a workgroup flags 4 096 sorted
// records (4 per thread, in order), scans (keep | cstart << 32) in place, looks back for what lies in front of it, and
// scatters exactly as k_dense_scatter does.  `state` / `ticket`: zero at the start (cleared by k_dense_keys on its way).
template <typename KeyT>
__global__ __launch_bounds__(kCompactBlock) void k_dense_compact(const KeyT *keys, const uint32_t *vals, const char *pts, size_t stride, uint32_t nfin,
                                                                 uint32_t xbits, float4 *sorted, uint32_t *pos_of, uint32_t *cellslot, uint32_t *cellpos,
                                                                 uint32_t *stats, uint32_t *host_stats, uint32_t *table, unsigned long long *state,
                                                                 uint32_t *ticket)
{
    __shared__ uint32_t s_bid;
    __shared__ unsigned long long s_wave[kCompactBlock / 64];
    __shared__ unsigned long long s_excl;
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid;
    const uint32_t i0 = (bid * kCompactBlock + threadIdx.x) * kCompactItems;
    KeyT k[kCompactItems + 1];   // k[j + 1] = key of record i0 + j, k[0] = the one in front of them
    uint32_t kp[kCompactItems], cs[kCompactItems], v[kCompactItems];
    float px[kCompactItems], py[kCompactItems], pz[kCompactItems];
    k[0] = i0 > 0 && i0 <= nfin ? keys[i0 - 1] : (KeyT)0;
    // every load a record needs goes out before the scan (the gathers of the points are what takes long, and nothing
    // behind the first store could be moved in front of it by the compiler)
#pragma unroll
    for (uint32_t j = 0; j < kCompactItems; ++j) {
        const uint32_t i = i0 + j;
        k[j + 1] = (KeyT)0;
        v[j] = 0;
        px[j] = py[j] = pz[j] = 0.0f;
        if (i < nfin) {
            k[j + 1] = keys[i];
            v[j] = vals[i];
            const float *p = rec_xyz(pts, stride, v[j]);
            px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
        }
    }
    unsigned long long mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < kCompactItems; ++j) {
        const uint32_t i = i0 + j;
        kp[j] = cs[j] = 0;
        if (i < nfin) {
            kp[j] = cs[j] = 1;
            if (i > 0) {
                cs[j] = (k[j + 1] >> xbits) != (k[j] >> xbits);
                if (k[j + 1] == k[j]) {
                    const float *b = rec_xyz(pts, stride, j > 0 ? v[j - 1] : vals[i - 1]);
                    if (px[j] == b[0] && py[j] == b[1] && pz[j] == b[2]) kp[j] = 0;
                }
            }
            mine += (unsigned long long)kp[j] | (unsigned long long)cs[j] << 32;
        }
    }
    unsigned long long total;
    unsigned long long run = compact_block_scan(mine, s_wave, total);
    run += compact_lookback(state, bid, total, &s_excl);
#pragma unroll
    for (uint32_t j = 0; j < kCompactItems; ++j) {
        const uint32_t i = i0 + j;
        if (i >= nfin) break;
        const uint32_t pos = (uint32_t)run, cid = (uint32_t)(run >> 32);
        const uint32_t slot = (uint32_t)(k[j + 1] >> xbits);
        if (kp[j]) {
            sorted[pos] = tgt_rec(px[j], py[j], pz[j], v[j]);
            pos_of[v[j]] = pos;
        }
        if (cs[j]) {
            cellslot[cid] = slot;
            cellpos[cid] = pos;
            if (table) {
                table[slot] = pos;
                if (i > 0) table[(uint32_t)(k[j] >> xbits) + 1u] = pos;   // (the same value when the two slots coincide)
            }
        }
        if (i == nfin - 1) {
            const uint32_t nu = pos + kp[j], nc = cid + cs[j];
            stats[0] = nc;
            stats[2] = nu;
            host_stats[0] = nc;   // (pinned host memory: what the host reads when the build has drained, no copy queued)
            host_stats[2] = nu;
            cellpos[nc] = nu;   // sentinel
            if (table) table[slot + 1u] = nu;
            for (uint32_t q = 0; q < 4; ++q) sorted[nu + q] = tgt_rec(1e30f, 1e30f, 1e30f, 0xffffffffu);
        }
        run += (unsigned long long)kp[j] | (unsigned long long)cs[j] << 32;
    }
}

// table[slot of occupied cell c] = its point count (the exclusive scan of the table then gives
// the first point of EVERY cell, empty ones included); plain stores, no atomics
__global__ __launch_bounds__(kBlock) void k_dense_counts(const uint32_t *cellslot, const uint32_t *cellpos, const uint32_t *stats,
                                                         uint32_t *table)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= stats[0]) return;
    table[cellslot[c]] = cellpos[c + 1] - cellpos[c];
}

// largest cell population, from the starts of the occupied cells (introspection only: rsreg_icp_grid_info)
__global__ __launch_bounds__(kBlock) void k_dense_max_count(const uint32_t *cellpos, const uint32_t *stats, uint32_t *out)
{
    uint32_t v = 0;
    const uint32_t nc = stats[0];
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) v = max(v, cellpos[c + 1] - cellpos[c]);
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_down(v, off));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out, v);
}

// nbr[n] |= bit(offset of c seen from n) for the 27 cells n around every occupied cell c.
// Bit j = dz * 9 + dy * 3 + dx of a cell's word: its neighbour at (dx - 1, dy - 1, dz - 1) holds points.  The occupied cells
// come in slot order, so the occupied neighbours of c in its own row are the threads next to it, and the three x offsets
// of one (y, z) offset are put together before they are OR-ed in: the word of a cell is written, per (y, z) offset, by
// the thread of that cell if it is occupied, else by the thread of the occupied cell to its left, else by the one to
// its right -- 9 atomics per occupied cell inside a run of occupied cells, 27 for a lone one (it was 27 for all:
// 3.9 M atomic ORs, 27 us, for the 10^6-point target of the bench).
__device__ __forceinline__ void dense_nbr_body(uint32_t bid, const uint32_t *cellslot, const uint32_t *stats, int sx, int sxy, uint32_t *nbr)
{
    const uint32_t c = bid * blockDim.x + threadIdx.x, nc = stats[0];
    if (c >= nc) return;
    const int slot = (int)cellslot[c];
    const int sl1 = c >= 1 ? (int)cellslot[c - 1] : -9, sl2 = c >= 2 ? (int)cellslot[c - 2] : -9;
    const int sr1 = c + 1 < nc ? (int)cellslot[c + 1] : -9, sr2 = c + 2 < nc ? (int)cellslot[c + 2] : -9;
    const bool L = sl1 == slot - 1, L2 = sl1 == slot - 2 || sl2 == slot - 2;    // cells slot - 1, slot - 2 hold points
    const bool R = sr1 == slot + 1, R2 = sr1 == slot + 2 || sr2 == slot + 2;    // cells slot + 1, slot + 2
    // bits of the three x offsets as seen from: the cell itself (its left neighbour, itself, its right neighbour), the empty
    // cell to its right (its left neighbour = this cell, its right neighbour = slot + 2), the empty cell to its left
    const uint32_t own = (L ? 1u : 0u) | 2u | (R ? 4u : 0u), right = 1u | (R2 ? 4u : 0u), left = 4u;
    const bool owns_right = !R, owns_left = !L && !L2;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int dz = r / 3, dy = r % 3;   // the (y, z) offset, as seen from the cell whose word is written
        const int row = slot - ((dz - 1) * sxy + (dy - 1) * sx);
        const int sh = dz * 9 + dy * 3;
        // the words of row - 1, row, row + 1 are neighbours in memory: the one that shares an aligned 8-byte pair with
        // `row` goes out with it in ONE 64-bit atomic (a lone cell -- an edge cloud is lines in space -- 18 atomics instead of 27)
        const uint32_t w_own = own << sh, w_right = owns_right ? right << sh : 0u, w_left = owns_left ? left << sh : 0u;
        if (reinterpret_cast<uintptr_t>(nbr + row) & 4u) {   // (row - 1, row) is the aligned pair
            if (w_left) atomicOr(reinterpret_cast<unsigned long long *>(nbr + row - 1), (unsigned long long)w_left | (unsigned long long)w_own << 32);
            else atomicOr(&nbr[row], w_own);
            if (w_right) atomicOr(&nbr[row + 1], w_right);
        } else {         // (row, row + 1)
            if (w_right) atomicOr(reinterpret_cast<unsigned long long *>(nbr + row), (unsigned long long)w_own | (unsigned long long)w_right << 32);
            else atomicOr(&nbr[row], w_own);
            if (w_left) atomicOr(&nbr[row - 1], w_left);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_dense_nbr(const uint32_t *cellslot, const uint32_t *stats, int sx, int sxy, uint32_t *nbr)
{
    dense_nbr_body(blockIdx.x, cellslot, stats, sx, sxy, nbr);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sel3(float a, float b, float c, int i) { return i == 0 ? a : (i == 1 ? b : c); }

typedef float f32x2 __attribute__((ext_vector_type(2)));


// running best: squared distance and original index (the tie-break) of the winning point -- its
// place in the sorted array is looked up once at the end (pos_of), not carried through every
// comparison; starts at "+inf, no point" so that the far-away sentinel points never win
struct DBest {
    float d;
    uint32_t idx;
};

// FLANN L2_Simple in its own order ((dx^2 + dy^2) + dz^2, nothing fused); x and y go through
// the packed f32 pipe straight out of the loaded register pair; a record is (x, y, index, z), see tgt_rec
__device__ __forceinline__ void dconsider(DBest &b, f32x2 qxy, float qz, const u32x4 &t)
{
    const f32x2 txy = {__uint_as_float(t.x), __uint_as_float(t.y)};
    const f32x2 dxy = qxy - txy;
    const f32x2 sq = dxy * dxy;
    const float dz = __fsub_rn(qz, __uint_as_float(t.w));
    const float d = __fadd_rn(__fadd_rn(sq.x, sq.y), __fmul_rn(dz, dz));
    // (distance, original index) ordered as one 64-bit key: squared distances are >= 0, so their bit patterns
    // order like their values, and the lower index wins among equal distances
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | t.z;
    const unsigned long long bkey = ((unsigned long long)__float_as_uint(b.d) << 32) | b.idx;
    const bool better = key < bkey;
    b.d = better ? d : b.d;
    b.idx = better ? t.z : b.idx;
}

// score 4 consecutive points starting at byte offset `po` (reading past the end of a cell
// only meets more real target points, or the far-away sentinels behind the last one)
__device__ __forceinline__ void dscan4(DBest &b, __amdgpu_buffer_rsrc_t pts, uint32_t po, f32x2 qxy, float qz)
{
    const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(pts, po, 0, 0);
    const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 16, 0, 0);
    const u32x4 t2 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 32, 0, 0);
    const u32x4 t3 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 48, 0, 0);
    dconsider(b, qxy, qz, t0);
    dconsider(b, qxy, qz, t1);
    dconsider(b, qxy, qz, t2);
    dconsider(b, qxy, qz, t3);
}

// A query of a split tile is searched by 2^lg lanes: lane `sub` of them scores the chunks sub, sub + 2^lg, ... of
// every run of points; the lanes' bests are merged at the end (nn_query_dense).  {0, 0}: one lane does it all
// (a compile-time constant on the ordinary path, where none of this costs anything).
struct DSplit {
    uint32_t lg, sub;
};

__device__ __forceinline__ void dscan_range(DBest &b, __amdgpu_buffer_rsrc_t pts, uint32_t po, uint32_t pe, f32x2 qxy, float qz,
                                            DSplit sp)
{
    if (sp.lg != 0) {
        // The lanes of a split query each derive the range [po, pe) of a row from their OWN best so far, so the ranges
        // differ: chunks are therefore owned by ABSOLUTE position (64-byte chunk c belongs to lane c mod 2^lg), not by
        // their place in the range.  A chunk that lies outside its owner's range is never read -- rightly: the owner's
        // best proves that nothing in it can be nearer.  (Reading from the 64-byte boundary below po only meets more
        // real points of the row before.)
        const uint32_t first = po >> 6;
        po = (first + ((sp.sub - first) & ((1u << sp.lg) - 1u))) << 6;
    }
    for (; po < pe; po += 64u << sp.lg) dscan4(b, pts, po, qxy, qz);
}

// A cell's points are sorted by x (to 2^-16 of a cell).  A walk reads a cell 4 points at a time
// from the end that is nearer to the query in x and stops as soon as the x distance alone
// exceeds the best so far: everything behind that point is farther still.
struct DWalk {
    uint32_t cur;       // first byte of the next chunk to read, in either direction
    uint32_t floor64;   // backward: the cell's first byte + |step| (the last chunk of a backward walk starts at the cell's
                        // first point); forward: 0.  next = max(cur, floor64) + step in both directions
    uint32_t step;      // +64 / -64 (mod 2^32), times the lanes a split query is spread over
    int left;           // chunks still to read (0: nothing)
    bool back;
    float yz2;          // what every point of this cell is away from the query in y and z at least, squared
};

__device__ __forceinline__ void dwalk_open(DWalk &w, const u32x2 &se, bool back, float yz2, DSplit sp)
{
    const uint32_t chunks = (se.y - se.x + 3u) >> 2, stride = 64u << sp.lg, first = 64u * (sp.sub + 1u);
    w.yz2 = yz2;
    w.left = chunks > sp.sub ? (int)((chunks - sp.sub + (1u << sp.lg) - 1u) >> sp.lg) : 0;   // chunks sub, sub + 2^lg, ...
    w.back = back;
    w.floor64 = back ? se.x * 16u + stride : 0u;
    w.step = back ? 0u - stride : stride;
    // backward, chunk k starts at max(end - 64 (k + 1), first byte of the cell)
    w.cur = back ? max(se.y * 16u, se.x * 16u + first) - first : se.x * 16u + first - 64u;
}

// running limit = min(limit, best): one v_min (both are squared distances or +inf, never NaN, so the
// canonicalising max pair the compiler puts around fminf is not needed)
__device__ __forceinline__ float min_nn(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// one chunk; x_slack = how far the x order inside a cell can be off (quantisation of the sort key)
__device__ __forceinline__ void dwalk_step(DWalk &w, DBest &b, __amdgpu_buffer_rsrc_t pts, f32x2 qxy, float qz, float x_slack,
                                           float &limit2)
{
    const uint32_t po = w.cur;
    w.cur = max(po, w.floor64) + w.step;
    --w.left;
    const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(pts, po, 0, 0);
    const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 16, 0, 0);
    const u32x4 t2 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 32, 0, 0);
    const u32x4 t3 = __builtin_amdgcn_raw_buffer_load_b128(pts, po + 48, 0, 0);
    dconsider(b, qxy, qz, t0);
    dconsider(b, qxy, qz, t1);
    dconsider(b, qxy, qz, t2);
    dconsider(b, qxy, qz, t3);
    limit2 = min_nn(limit2, b.d);
    // beyond this chunk (in walking direction) every point of the cell is at least `gap` away in x
    const float gap = (w.back ? qxy.x - __uint_as_float(t0.x) : __uint_as_float(t3.x) - qxy.x) - x_slack;
    if (gap > 0.0f && gap * gap + w.yz2 > limit2) w.left = 0;
}

struct DDiag {       // diagnostic launches only: per-lane step counts and two clock stamps
    uint32_t own = 0, r1_cells = 0, r1_scans = 0, far_rows = 0, far_scans = 0;
    unsigned long long t_near = 0, t_far = 0;
};

struct DQuery {      // a query and where it sits in the grid
    float qx, qy, qz;
    float ux, uy, uz;
    int cx, cy, cz;
};

struct DRes {        // the arrays behind 32-bit offsets (wave-uniform descriptors)
    __amdgpu_buffer_rsrc_t pts, tab, nbr;
};

__device__ __forceinline__ DRes dense_res(const DenseDev &g)
{
    DRes r;
    r.pts = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(g.pts), 0, (g.n_pts + 4) * 16, 0x00020000);
    r.tab = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(g.start), 0, g.table_bytes, 0x00020000);
    r.nbr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(g.nbr), 0, g.table_bytes, 0x00020000);
    return r;
}

__device__ __forceinline__ DQuery dense_query(const DenseDev &g, float qx, float qy, float qz)
{
    DQuery q;
    q.qx = qx; q.qy = qy; q.qz = qz;
    q.ux = cell_pos(qx, g.ox, g.inv_cell);
    q.uy = cell_pos(qy, g.oy, g.inv_cell);
    q.uz = cell_pos(qz, g.oz, g.inv_cell);
    q.cx = min(max((int)fminf(fmaxf(floorf(q.ux), -4.0f), 70000.0f), 0), g.nx - 1);
    q.cy = min(max((int)fminf(fmaxf(floorf(q.uy), -4.0f), 70000.0f), 0), g.ny - 1);
    q.cz = min(max((int)fminf(fmaxf(floorf(q.uz), -4.0f), 70000.0f), 0), g.nz - 1);
    return q;
}

__device__ __forceinline__ void dense_seed(const DRes &rs, const DQuery &q, int seed_pos, DBest &b, float &limit2)
{
    if (seed_pos >= 0) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs.pts, (uint32_t)seed_pos * 16u, 0, 0);
        dconsider(b, f32x2{q.qx, q.qy}, q.qz, t);
        limit2 = fminf(limit2, b.d);
    }
}

// what the ring-1 search of a query needs after its own cell has been read
struct DRing1 {
    float gx0, gx2, gy0, gy2, gz0, gz2;   // squared gaps (cell units, margin taken off) to the six neighbouring slabs
    int base;                             // table index of the own cell
    uint32_t mask;                        // bit j = dz*9 + dy*3 + dx: occupied neighbour whose box can hold something closer
    bool right_half;
};

// The occupancy word of a cell's 27-cell neighbourhood read off the cell table itself -- for an index built WITHOUT the words
// (icp.hip: build_dense_counted for a small source; the table of a counting build has an entry for every cell of the padded
// grid, and a cell holds points exactly when the entry behind it is larger): nine 16-byte loads, one per (y, z) row, entries
// x - 1 .. x + 2, instead of one word that cost the build 18 scattered atomics per occupied cell of a sparse cloud.  The
// own cell's range is the middle row's middle pair: that row is read with the first loads of a search, the other eight only
// by the queries that have to open ring 1 at all.
__device__ __forceinline__ uint32_t dense_occ_bits(const u32x4 &t) { return ((t.y > t.x) ? 1u : 0u) | ((t.z > t.y) ? 2u : 0u) | ((t.w > t.z) ? 4u : 0u); }
__device__ __forceinline__ uint32_t dense_occ_from_table(const DenseDev &g, const DRes &rs, int base, uint32_t middle)
{
    uint32_t occ = middle << 12;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        if (r == 4) continue;
        const int dz = r / 3, dy = r % 3;
        const int row = base + (dz - 1) * g.sxy + (dy - 1) * g.sx - 1;   // the entry of the cell at x - 1 of that row
        occ |= dense_occ_bits(__builtin_amdgcn_raw_buffer_load_b128(rs.tab, (uint32_t)row * 4u, 0, 0)) << (3 * r);
    }
    return occ;
}

// ring 0: the seed and the query's own cell (it usually holds the nearest point), then which neighbours have to be opened
// (kOccTab: the occupancy word comes from the table, above)
template <bool kDiag = false, bool kOccTab = false>
__device__ __forceinline__ uint32_t dense_own(const DenseDev &g, const DRes &rs, const DQuery &q, int seed_pos, DBest &b, float &limit2,
                                              DSplit sp, DRing1 &r1, DDiag *dg = nullptr)
{
    const f32x2 qxy = {q.qx, q.qy};
    const float qz = q.qz;
    const float cell2 = g.cell * g.cell;
    const int base = (int)dense_cell_id(g, q.cx, q.cy, q.cz);
    // the three first loads (neighbourhood word, own cell's range, the seed point) go out together:
    // a search is a chain of dependent loads, and every round trip saved shortens the slowest waves
    uint32_t occ;
    u32x2 se;
    if (kOccTab) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs.tab, (uint32_t)(base - 1) * 4u, 0, 0);   // entries x - 1 .. x + 2 of the own row
        dense_seed(rs, q, seed_pos, b, limit2);
        se = u32x2{t.y, t.z};
        occ = dense_occ_bits(t) << 12;   // (the own row's three bits: 12, 13, 14)
    } else {
        occ = __builtin_amdgcn_raw_buffer_load_b32(rs.nbr, (uint32_t)base * 4u, 0, 0);
        se = __builtin_amdgcn_raw_buffer_load_b64(rs.tab, (uint32_t)base * 4u, 0, 0);
        dense_seed(rs, q, seed_pos, b, limit2);
        // (an empty cell's table entry may be left over from an earlier build: k_dense_scatter writes occupied cells only)
        if (!(occ & (1u << 13))) se = u32x2{0u, 0u};
    }
    const float x_slack = g.x_slack;
    const bool right_half = (q.ux - (float)q.cx) >= 0.5f;   // the query sits in the right half of its cell column
    DWalk w;
    dwalk_open(w, se, right_half, 0.0f, sp);
    while (w.left > 0) {
        if (kDiag) ++dg->own;
        dwalk_step(w, b, rs.pts, qxy, qz, x_slack, limit2);
    }
    // ---- ring 1: only the cells whose box can still hold something closer.  A neighbour at
    // offset (dx,dy,dz) needs every non-zero axis offset's face to be within the limit, so when
    // no face passes (the common case) the whole ring is skipped with six compares.
    const float lim_c = limit2 / cell2;   // limit in squared cell units (the gaps carry the safety margin)
    float gx0 = axis_gap(q.ux, q.cx - 1, q.cx - 1, g.margin), gx2 = axis_gap(q.ux, q.cx + 1, q.cx + 1, g.margin);
    float gy0 = axis_gap(q.uy, q.cy - 1, q.cy - 1, g.margin), gy2 = axis_gap(q.uy, q.cy + 1, q.cy + 1, g.margin);
    float gz0 = axis_gap(q.uz, q.cz - 1, q.cz - 1, g.margin), gz2 = axis_gap(q.uz, q.cz + 1, q.cz + 1, g.margin);
    gx0 *= gx0; gx2 *= gx2; gy0 *= gy0; gy2 *= gy2; gz0 *= gz0; gz2 *= gz2;
    const float gx1 = 0.0f, gy1 = 0.0f, gz1 = 0.0f;   // own slab on that axis: 0 is always a valid lower bound
    r1.gx0 = gx0; r1.gx2 = gx2; r1.gy0 = gy0; r1.gy2 = gy2; r1.gz0 = gz0; r1.gz2 = gz2;
    r1.base = base;
    r1.right_half = right_half;
    r1.mask = 0;
    const bool any_face = (gx0 <= lim_c) | (gx2 <= lim_c) | (gy0 <= lim_c) | (gy2 <= lim_c) | (gz0 <= lim_c) | (gz2 <= lim_c);
#ifdef RSREG_DIAG
    if (g.debug_skip & 2u) return occ;
#endif
    if (kOccTab) {
        if (!any_face) return occ;
        occ = dense_occ_from_table(g, rs, base, occ >> 12);
    }
    if (!any_face || !(occ & ~(1u << 13))) return occ;
    uint32_t mask = 0;   // bit j = dz*9 + dy*3 + dx (offsets 0..2), centre excluded
#pragma unroll
    for (int j = 0; j < 27; ++j) {
        if (j == 13) continue;
        const int dz = j / 9, dy = (j / 3) % 3, dx = j % 3;
        const float lb = (dx == 0 ? gx0 : (dx == 1 ? gx1 : gx2)) + (dy == 0 ? gy0 : (dy == 1 ? gy1 : gy2)) +
                         (dz == 0 ? gz0 : (dz == 1 ? gz1 : gz2));
        mask |= (lb <= lim_c) ? (1u << j) : 0u;
    }
    r1.mask = mask & occ;         // empty cells are never opened
    return occ;
}

// ring 1, lane by lane: a flat loop -- a trip moves on to the next plausible cell and/or scores 4 candidates.  The range
// of the cell after the current one is already in flight (nse) while the current one is scored.
template <bool kDiag = false>
__device__ __forceinline__ void dense_ring1_lane(const DenseDev &g, const DRes &rs, const DQuery &q, const DRing1 &r1, uint32_t mask, DBest &b,
                                                 float &limit2, DSplit sp, DDiag *dg = nullptr)
{
    if (!mask) return;
    const f32x2 qxy = {q.qx, q.qy};
    const float qz = q.qz, cell2 = g.cell * g.cell, x_slack = g.x_slack;
    const float gx0 = r1.gx0, gx2 = r1.gx2, gy0 = r1.gy0, gy2 = r1.gy2, gz0 = r1.gz0, gz2 = r1.gz2, gx1 = 0.0f, gy1 = 0.0f, gz1 = 0.0f;
    const int base = r1.base;
    const bool right_half = r1.right_half;
    DWalk w;
    w.left = 0;
    u32x2 nse = {0u, 0u};
    float nlb2 = 0.0f, nyz2 = 0.0f;
    bool nvalid = false, nback = false;
    for (;;) {
        if (w.left <= 0) {
            if (!nvalid && !mask) break;
            if (nvalid && nlb2 <= limit2) dwalk_open(w, nse, nback, nyz2, sp);   // (the limit may have tightened since that range was asked for)
            nvalid = false;
            if (mask) {
                if (kDiag) ++dg->r1_cells;
                const int j = __ffs((int)mask) - 1;
                mask &= mask - 1;
                const int dz = j / 9, dy = (j - dz * 9) / 3, dx = j - dz * 9 - dy * 3;
                const float lb2 = (sel3(gx0, gx1, gx2, dx) + sel3(gy0, gy1, gy2, dy) + sel3(gz0, gz1, gz2, dz)) * cell2;
                if (lb2 <= limit2) {
                    const int idx = base + (dz - 1) * g.sxy + (dy - 1) * g.sx + (dx - 1);
                    nse = __builtin_amdgcn_raw_buffer_load_b64(rs.tab, (uint32_t)idx * 4u, 0, 0);
                    nlb2 = lb2;
                    nyz2 = (sel3(gy0, gy1, gy2, dy) + sel3(gz0, gz1, gz2, dz)) * cell2;
                    nvalid = true;
                    nback = dx == 0 ? true : (dx == 1 ? right_half : false);   // a cell to the left is read from its right end
                }
            }
        }
        if (w.left > 0) {
            if (kDiag) ++dg->r1_scans;
            dwalk_step(w, b, rs.pts, qxy, qz, x_slack, limit2);
        }
    }
}

// rings 0 and 1; returns the occupancy word of the query's 27-cell neighbourhood
template <bool kDiag = false, bool kOccTab = false>
__device__ __forceinline__ uint32_t dense_near(const DenseDev &g, const DRes &rs, const DQuery &q, int seed_pos, DBest &b,
                                               float &limit2, DSplit sp, DDiag *dg = nullptr)
{
    DRing1 r1;
    const uint32_t occ = dense_own<kDiag, kOccTab>(g, rs, q, seed_pos, b, limit2, sp, r1, dg);
    dense_ring1_lane<kDiag>(g, rs, q, r1, r1.mask, b, limit2, sp, dg);
    return occ;
}

// does this query still need rings >= 2 after rings 0-1 ?
__device__ __forceinline__ bool dense_needs_far(const DenseDev &g, float limit2)
{
    const float reach = (1.0f - g.margin) * g.cell;   // ring 1 proves everything up to here
#ifdef RSREG_DIAG
    if (g.debug_skip) return false;
#endif
    return g.max_ring >= 2 && limit2 > reach * reach;
}

// Everything beyond rings 0-1, row by row: the cells of one (y, z) row are one contiguous run of
// points, so a row costs two table loads whatever its x-extent.  Rows are taken by their
// Chebyshev distance rho from the query's row, nearest first, each ONCE: of a row only the
// x-extent the remaining budget (limit - gap_yz) reaches is read, and the search stops as soon
// as every unvisited row is provably farther than the best.  The nine central rows
// (rho <= 1) have had their cells cx-1..cx+1 searched already: only their two outer parts remain.
template <bool kDiag = false>
__device__ __forceinline__ void dense_far_row(const DenseDev &g, const DRes &rs, const DQuery &q, int dy, int dz, bool central,
                                              float inv_cell2, f32x2 qxy, DBest &b, float &limit2, DSplit sp, DDiag *dg)
{
    const int y = q.cy + dy, z = q.cz + dz;
    if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) return;
    const float ay = axis_gap(q.uy, y, y, g.margin), az = axis_gap(q.uz, z, z, g.margin);
    const float rem = limit2 * inv_cell2 - (ay * ay + az * az);   // budget left for the x gap, squared cells
    if (rem < 0.0f) return;
    if (kDiag) ++dg->far_rows;
    const int row = (int)dense_cell_id(g, 0, y, z);
    // cells cx-kl .. cx+kr are the ones whose x gap fits the budget (gap = fx + k - 1 to the left, k - fx to the right)
    const float sr = sqrtf(rem) + g.margin + 1e-4f, fx = q.ux - (float)q.cx;
    const int kl = (int)fminf(fmaxf(sr + 1.0f - fx, 0.0f), (float)g.max_ring), kr = (int)fminf(fmaxf(sr + fx, 0.0f), (float)g.max_ring);
    const int xa = max(q.cx - kl, 0), xb = min(q.cx + kr, g.nx - 1);
    // a row is one x-sorted run of points: it is walked like a cell, from the end nearer to the
    // query, and left once the x distance alone no longer fits the budget
    const float yz2 = (ay * ay + az * az) * (g.cell * g.cell), x_slack = g.x_slack;
    DWalk w;
    if (!central) {
        const uint32_t s = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + xa) * 4u, 0, 0);
        const uint32_t e = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + xb + 1) * 4u, 0, 0);
        if (kDiag) dg->far_scans += (e - s + 3) / 4;
        dscan_range(b, rs.pts, s * 16u, e * 16u, qxy, q.qz, sp);
    } else {
        // [xa, cx-2] and [cx+2, xb]; an empty part reads the same table entry twice
        const int l1 = min(q.cx - 1, xb + 1), r0 = max(q.cx + 2, xa);
        const uint32_t s0 = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + min(xa, l1)) * 4u, 0, 0);
        const uint32_t e0 = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + l1) * 4u, 0, 0);
        const uint32_t s1 = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + min(r0, xb + 1)) * 4u, 0, 0);
        const uint32_t e1 = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + xb + 1) * 4u, 0, 0);
        if (kDiag) dg->far_scans += (e0 - s0 + 3) / 4 + (e1 - s1 + 3) / 4;
        dwalk_open(w, u32x2{s0, e0}, true, yz2, sp);    // the part to the left, from its right end
        while (w.left > 0) dwalk_step(w, b, rs.pts, qxy, q.qz, x_slack, limit2);
        dwalk_open(w, u32x2{s1, e1}, false, yz2, sp);   // the part to the right, from its left end
        while (w.left > 0) dwalk_step(w, b, rs.pts, qxy, q.qz, x_slack, limit2);
    }
    limit2 = fminf(limit2, b.d);
}

template <bool kDiag = false>
__device__ __forceinline__ void dense_far(const DenseDev &g, const DRes &rs, const DQuery &q, DBest &b, float &limit2,
                                          DSplit sp, DDiag *dg = nullptr)
{
    const f32x2 qxy = {q.qx, q.qy};
    const float inv_cell2 = 1.0f / (g.cell * g.cell);
    for (int rho = 0; rho <= g.max_ring; ++rho) {
        if (rho >= 2) {   // a row at distance rho is at least rho - 1 cells away on one axis
            const float reach = ((float)(rho - 1) - g.margin) * g.cell;
            if (limit2 <= reach * reach) break;
        }
        const bool central = rho <= 1;
        for (int dz = -rho; dz <= rho; ++dz) {
            if (dz == -rho || dz == rho) {
                for (int dy = -rho; dy <= rho; ++dy) dense_far_row<kDiag>(g, rs, q, dy, dz, central, inv_cell2, qxy, b, limit2, sp, dg);
            } else {
                dense_far_row<kDiag>(g, rs, q, -rho, dz, central, inv_cell2, qxy, b, limit2, sp, dg);
                dense_far_row<kDiag>(g, rs, q, rho, dz, central, inv_cell2, qxy, b, limit2, sp, dg);
            }
        }
    }
}

// The same search for grids whose gate reaches at most 4 cells (max_ring <= 4: every bounded gate), driven by the
// neighbourhood words instead of the table: the 81 rows around the query are 9 blocks of 3 x 3 rows, and the three
// words at columns cx-3, cx, cx+3 of a block's middle row say which of the block's 9 x 9 cells hold points.
// Empty rows (most rows of a query that has to look this far) cost a few bit operations instead of two dependent
// table loads each, and an occupied row is read only over the span of its occupied cells.
template <bool kDiag = false>
__device__ __forceinline__ void dense_far_row_occ(const DenseDev &g, const DRes &rs, const DQuery &q, int y, int z, bool central,
                                                  uint32_t rowbits, float inv_cell2, f32x2 qxy, DBest &b, float &limit2, DSplit sp,
                                                  DDiag *dg)
{
    const float ay = axis_gap(q.uy, y, y, g.margin), az = axis_gap(q.uz, z, z, g.margin);
    const float rem = limit2 * inv_cell2 - (ay * ay + az * az);   // budget left for the x gap, squared cells
    if (rem < 0.0f) return;
    // rowbits: bit k = cell cx - 4 + k, already cut to the columns the BLOCK's budget reaches (a superset of this row's own
    // extent: a few more candidates read, no square root per row) and without the searched middle of the central rows
    const uint32_t m = rowbits;
    if (kDiag) ++dg->far_rows;
    const int row = (int)dense_cell_id(g, q.cx - 4, y, z);
    const float yz2 = (ay * ay + az * az) * (g.cell * g.cell), x_slack = g.x_slack;
    if (!central) {
        const int xa = __ffs((int)m) - 1, xb = 31 - __clz((int)m);   // first and last occupied cell in reach
        const uint32_t s = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + xa) * 4u, 0, 0);
        const uint32_t e = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + xb + 1) * 4u, 0, 0);
        if (kDiag) dg->far_scans += (e - s + 3) / 4;
        dscan_range(b, rs.pts, s * 16u, e * 16u, qxy, q.qz, sp);
    } else {
        // the part left of the searched middle (bits 0..2) from its right end, the part right of it (bits 6..8) from its left end;
        // an empty part reads the same table entry twice
        const uint32_t ml = m & 7u, mr = m >> 6;
        const int la = ml ? __ffs((int)ml) - 1 : 0, lb = ml ? 32 - __clz((int)ml) : 0;
        const int ra = mr ? 6 + __ffs((int)mr) - 1 : 6, rb = mr ? 6 + 32 - __clz((int)mr) : 6;
        const uint32_t s0 = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + la) * 4u, 0, 0);
        const uint32_t e0 = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + lb) * 4u, 0, 0);
        const uint32_t s1 = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + ra) * 4u, 0, 0);
        const uint32_t e1 = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (uint32_t)(row + rb) * 4u, 0, 0);
        if (kDiag) dg->far_scans += (e0 - s0 + 3) / 4 + (e1 - s1 + 3) / 4;
        DWalk w;
        dwalk_open(w, u32x2{s0, e0}, true, yz2, sp);
        while (w.left > 0) dwalk_step(w, b, rs.pts, qxy, q.qz, x_slack, limit2);
        dwalk_open(w, u32x2{s1, e1}, false, yz2, sp);
        while (w.left > 0) dwalk_step(w, b, rs.pts, qxy, q.qz, x_slack, limit2);
    }
    limit2 = fminf(limit2, b.d);
}

template <bool kDiag = false>
__device__ __forceinline__ void dense_far_blocks(const DenseDev &g, const DRes &rs, const DQuery &q, DBest &b, float &limit2,
                                                 DSplit sp, DDiag *dg = nullptr)
{
    const f32x2 qxy = {q.qx, q.qy};
    const float cell2 = g.cell * g.cell, inv_cell2 = 1.0f / cell2;
    // blocks by distance: the middle one (the nine central rows), its four edge neighbours, the four corners;
    // (by + 1) and (bz + 1) of block k, two bits each
    constexpr uint32_t kBy = 1u | 0u << 2 | 2u << 4 | 1u << 6 | 1u << 8 | 0u << 10 | 2u << 12 | 0u << 14 | 2u << 16;
    constexpr uint32_t kBz = 1u | 1u << 2 | 1u << 4 | 0u << 6 | 2u << 8 | 0u << 10 | 0u << 12 | 2u << 14 | 2u << 16;
#pragma unroll 1
    for (int k = 0; k < 9; ++k) {
        const int by = (int)((kBy >> (2 * k)) & 3u) - 1, bz = (int)((kBz >> (2 * k)) & 3u) - 1;
        const int yc = q.cy + 3 * by, zc = q.cz + 3 * bz;   // the block's middle row (a border row still has a valid word)
        if (yc < -1 || yc > g.ny || zc < -1 || zc > g.nz) continue;
        // nothing in the block's rows yc-1 .. yc+1, zc-1 .. zc+1 can be closer than this
        const float aby = k ? axis_gap(q.uy, yc - 1, yc + 1, g.margin) : 0.0f, abz = k ? axis_gap(q.uz, zc - 1, zc + 1, g.margin) : 0.0f;
        const float rem_b = limit2 * inv_cell2 - (aby * aby + abz * abz);
        if (rem_b < 0.0f) continue;
        const int base = (int)dense_cell_id(g, q.cx, yc, zc);
        // (a column outside the padded table reads as "nothing there": the descriptor's bounds check returns 0)
        uint32_t w0 = __builtin_amdgcn_raw_buffer_load_b32(rs.nbr, q.cx >= 2 ? (uint32_t)(base - 3) * 4u : 0xfffffff0u, 0, 0);
        uint32_t w1 = __builtin_amdgcn_raw_buffer_load_b32(rs.nbr, (uint32_t)base * 4u, 0, 0);
        uint32_t w2 = __builtin_amdgcn_raw_buffer_load_b32(rs.nbr, q.cx + 3 <= g.nx ? (uint32_t)(base + 3) * 4u : 0xfffffff0u, 0, 0);
        {
            // the columns ANY row of the block can reach with the budget left (a superset of each row's own extent, from the
            // block's smallest possible (y, z) gap): occupied cells outside it are dropped from all nine rows at once, so
            // only rows that hold points where they can matter come up below
            const float sr = sqrtf(rem_b) + g.margin + 1e-4f, fx = q.ux - (float)q.cx;
            const int kl = (int)fminf(fmaxf(sr + 1.0f - fx, 0.0f), 4.0f), kr = (int)fminf(fmaxf(sr + fx, 0.0f), 4.0f);
            uint32_t e = (2u << (4 + kr)) - (1u << (4 - kl));
            if (k == 0) e &= ~0x38u;   // the middle three columns of the nine central rows were searched in rings 0-1
            w0 &= ((e & 1u) ? 0x1249249u : 0u) | ((e & 2u) ? 0x2492492u : 0u) | ((e & 4u) ? 0x4924924u : 0u);
            w1 &= ((e & 8u) ? 0x1249249u : 0u) | ((e & 16u) ? 0x2492492u : 0u) | ((e & 32u) ? 0x4924924u : 0u);
            w2 &= ((e & 64u) ? 0x1249249u : 0u) | ((e & 128u) ? 0x2492492u : 0u) | ((e & 256u) ? 0x4924924u : 0u);
        }
        // bit 3 m of t: row m = lz * 3 + ly of the block holds points somewhere in columns cx-4 .. cx+4
        const uint32_t any = w0 | w1 | w2;
        uint32_t t = (any | any >> 1 | any >> 2) & 0x1249249u;
        while (t) {
            const int m3 = __ffs((int)t) - 1;   // = lz * 9 + ly * 3: the shift of the row's three bits in each word
            t &= t - 1;
            const int lz = (m3 >= 9) + (m3 >= 18), ly = ((m3 - 9 * lz) >= 3) + ((m3 - 9 * lz) >= 6);
            const int dy = 3 * by + ly - 1, dz = 3 * bz + lz - 1;
            if (max(abs(dy), abs(dz)) > g.max_ring) continue;
            const uint32_t rowbits = ((w0 >> m3) & 7u) | ((w1 >> m3) & 7u) << 3 | ((w2 >> m3) & 7u) << 6;
            dense_far_row_occ<kDiag>(g, rs, q, q.cy + dy, q.cz + dz, k == 0, rowbits, inv_cell2, qxy, b, limit2, sp, dg);
        }
    }
}

__device__ __forceinline__ Best dense_result(const DenseDev &g, const DBest &b)
{
    Best out{~0ull, -1, FLT_MAX};
    if (b.idx != 0xffffffffu) {
        out.key = ((unsigned long long)__float_as_uint(b.d) << 32) | b.idx;
        out.pos = (int)g.pos_of[b.idx];
        out.d2 = b.d;
    }
    return out;
}

// Exact nearest neighbour within the gate over the dense table (same contract as nn_query).
// kFar: which search beyond ring 1 is compiled in -- 0: both, chosen by the grid (max_ring <= 4: blocks); 1: blocks only
// (the caller knows max_ring <= 4); 2: rows only; 3: none, and the index has no occupancy words (the caller knows max_ring
// <= 1: the gate fits into ring 1; the words are read off the table, dense_occ_from_table)
template <bool kDiag = false, int kFar = 0>
__device__ __forceinline__ Best nn_query_dense(const DenseDev &g, float qx, float qy, float qz, int seed_pos, DDiag *dg = nullptr,
                                               DSplit sp = DSplit{0u, 0u})
{
    if (g.nx <= 0) return Best{~0ull, -1, FLT_MAX};
    const DRes rs = dense_res(g);
    const DQuery q = dense_query(g, qx, qy, qz);
    float limit2 = g.prune2;
    DBest b{__uint_as_float(0x7f800000u), 0xffffffffu};
    dense_near<kDiag, kFar == 3>(g, rs, q, seed_pos, b, limit2, sp, dg);
    if (kDiag) dg->t_near = wall_clock64();
    const bool far = kFar != 3 && dense_needs_far(g, limit2);
    if (far) {
        if (kFar == 1 || (kFar == 0 && g.max_ring <= 4)) dense_far_blocks<kDiag>(g, rs, q, b, limit2, sp, dg);
        else dense_far<kDiag>(g, rs, q, b, limit2, sp, dg);
    }
    if (kDiag) dg->t_far = wall_clock64();
    // the lanes of a split query hold the bests of disjoint parts of the candidate set: the smallest
    // (distance, index) key among them is the query's (every lane of the group ends up with it)
    for (uint32_t m = 1; m < (1u << sp.lg); m <<= 1) {
        const unsigned long long mine = ((unsigned long long)__float_as_uint(b.d) << 32) | b.idx;
        const unsigned long long other = (unsigned long long)__shfl_xor((long long)mine, (int)m);
        if (other < mine) { b.d = __uint_as_float((uint32_t)(other >> 32)); b.idx = (uint32_t)other; }
    }
    return dense_result(g, b);
}

template <int kFar = 0>
__global__ __launch_bounds__(kBlock) void k_nn_search_dense(const float4 *cur, uint32_t n, DenseDev g, double gate2,
                                                            int *corr_pos, float *corr_d2, int *seed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = cur[i];
    int pos = -1;
    float d2 = 0.0f;
    if (q.w != 0.0f) {
        const Best b = nn_query_dense<false, kFar>(g, q.x, q.y, q.z, seed ? seed[i] : -1);
        if (seed) seed[i] = b.pos;
        if (b.pos >= 0 && !((double)b.d2 > gate2)) {  // PCL: if (distance > max_dist_sqr) continue;
            pos = b.pos;
            d2 = b.d2;
        }
    }
    corr_pos[i] = pos;
    corr_d2[i] = d2;
}

// How the workgroups of a fused launch map to tiles (icp.hip: build_schedule).  All null: workgroup b is tile b.
// With a schedule, the tiles that took longest in an earlier iteration come first, and the longest of
// them are split: 2 or 4 workgroups share the tile's 128 queries, 2 or 4 lanes search each query (DSplit),
// and whichever of those workgroups finishes last adds up the tile's 17 sums in the usual order.
struct TileSched {
    const uint32_t *items;   // per workgroup: tile | part << 24 | log2(lanes per query) << 28
    uint32_t *cost;          // per wave of an unsplit tile: how long it ran in this launch (100 MHz ticks), or null
    uint32_t *done;          // per tile: parts finished so far (split tiles; goes back to 0 by itself)
    int *pos;                // per query: where the parts of a split tile leave their matches
    float *d2;
    uint32_t n_tiles;        // slabs of `partials`
    // A schedule carried over from an earlier alignment of this context (icp.hip: launch_fused) was built for another source:
    // workgroups [0, n_items) take its items -- those of tiles this source does not have do nothing --, the workgroups
    // behind them the tiles it did not know, first_extra + 0, 1, ..., unsplit.  (A schedule of this alignment's own: n_items =
    // the grid, no extras.)
    uint32_t n_items, first_extra;
};

// device-wide visible accesses for what the parts of a split tile hand to each other inside one launch
// (they may run on different XCDs, whose L2s are not coherent for ordinary accesses)
__device__ __forceinline__ void coherent_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t coherent_load(const uint32_t *p)
{
    return __hip_atomic_load(const_cast<uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coherent_store(float4 *p, const float4 &v)
{
    auto *u = reinterpret_cast<unsigned long long *>(p);
    __hip_atomic_store(u, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(u + 1, ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 coherent_load(const float4 *p)
{
    auto *u = reinterpret_cast<unsigned long long *>(const_cast<float4 *>(p));
    const unsigned long long a = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)b),
                       __uint_as_float((uint32_t)(b >> 32)));
}

// One part of a split tile: 128 >> lg queries, 2^lg lanes each.  Returns true in the workgroup that finishes
// the tile last (it then adds the tile's sums up from what all parts have left in sched.pos / sched.d2 / cur).
template <int kFar>
__device__ __forceinline__ bool fused_dense_split_part(float4 *cur, const float4 *restart, const uint32_t *first, uint32_t n, const Mat34 &T, int apply_t,
                                                       const DenseDev &g, double gate2, int *seed, const TileSched &sched, uint32_t tile, uint32_t part,
                                                       uint32_t lg)
{
    const DSplit sp{lg, threadIdx.x & ((1u << lg) - 1u)};
    const uint32_t i = tile * kTile + part * (kTile >> lg) + (threadIdx.x >> lg);
    if (i < n) {
        // (restart: the first launch of an alignment under a schedule carried over from an earlier one -- the query is the
        // source point itself under the guess, the working copy is written whatever the point, no seed: as on the ordinary path)
        float4 q = restart ? restart[i] : cur[i];
        int pos = -1;
        float d2 = 0.0f;
        if (q.w != 0.0f) {
            if (restart) q.w = source_weight(first, i);
            if (apply_t) {
                const float3 t = xform(T, q.x, q.y, q.z);
                q = make_float4(t.x, t.y, t.z, q.w);
            }
            const int seed_in = (seed && !restart) ? seed[i] : -1;
            const Best b = nn_query_dense<false, kFar>(g, q.x, q.y, q.z, seed_in, nullptr, sp);
            if (sp.sub == 0) {
                if (apply_t || restart) coherent_store(&cur[i], q);
                if (seed && (restart || b.pos != seed_in)) seed[i] = b.pos;
            }
            if (b.pos >= 0 && !((double)b.d2 > gate2)) {
                pos = b.pos;
                d2 = b.d2;
            }
        } else if (restart && sp.sub == 0) {
            coherent_store(&cur[i], q);
        }
        if (sp.sub == 0) {
            coherent_store(reinterpret_cast<uint32_t *>(sched.pos) + i, (uint32_t)pos);
            coherent_store(reinterpret_cast<uint32_t *>(sched.d2) + i, __float_as_uint(d2));
        }
    }
    __shared__ uint32_t s_last;
    // Every wave's stores must have arrived device-wide BEFORE thread 0 counts this part as done.  The workgroup
    // barrier alone does not wait for them (one CU, one L1: workgroup scope needs no wait), and most of the stores
    // come from other waves than the one that bumps the counter.  The stores are write-through (sc1), so "arrived"
    // is all that is needed: no cache write-back (an agent-scope release fence would add one: 99 -> 160 us per launch).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(&sched.done[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = before + 1u == (1u << lg);
        if (s_last) __hip_atomic_store(&sched.done[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_last != 0;
}

// RSREG_WAVE_TIMES_LIGHT: start, end and hardware slot of this wave (one record per launched wave)
__device__ __forceinline__ void light_stamp(unsigned long long *wave_times, unsigned long long t_start, uint32_t lg, uint32_t item)
{
    unsigned long long *w = wave_times + 16ull * (blockIdx.x * kTileWaves + (threadIdx.x >> 6));
    w[0] = t_start;
    w[4] = wall_clock64();
    w[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4) /* HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13 */ | 1ull << 63;
    w[11] = __builtin_amdgcn_s_getreg((31 << 11) | 20) /* XCC_ID: 3:0 */ | (unsigned long long)lg << 8;
    w[12] = item;
}

// One ICP iteration in one pass over the dense index: apply the previous increment, search,
// gate, accumulate (same contract and summation order as k_icp_fused).  kDiag: the diagnostic
// instantiation (RSREG_WAVE_TIMES) also writes clock stamps and step counts per wave (1), or only the start and
// end stamp and the hardware slot of each wave, at the product kernel's own occupancy (2: RSREG_WAVE_TIMES_LIGHT=1).
template <int kDiag, int kFar>
__global__ __launch_bounds__(kTile, kDiag == 1 ? 4 : 8) void k_icp_fused_dense(float4 *cur, const float4 *restart, const uint32_t *first, uint32_t n, Mat34 T, int apply_t, DenseDev g,
                                                           double gate2, int *corr_pos, float *corr_d2, double *partials,
                                                           int *seed, unsigned long long *wave_times, const IcpDevState *dev,
                                                           TileSched sched)
{
    const uint32_t item = sched.items ? (blockIdx.x < sched.n_items ? sched.items[blockIdx.x] : sched.first_extra + (blockIdx.x - sched.n_items)) : blockIdx.x;
    if (item == 0xffffffffu) return;   // (a workgroup the schedule has nothing for)
    const uint32_t tile = item & 0xffffffu, lg = (item >> 28) & 3u;
    if (tile >= sched.n_tiles) return;   // (a carried schedule's tile that this source does not have)
    const uint32_t i = tile * kTile + threadIdx.x;
    if (dev && !restart) {   // device-resident loop: the increment comes from the previous k_icp_solve
        T = dev->t_inc;
        apply_t = dev->apply;
    }
    unsigned long long t_start = 0, t_search = 0;
    DDiag dg;
    if (kDiag || sched.cost) t_start = wall_clock64();
    constexpr bool kFull = kDiag == 1;
    int pos = -1;
    float d2 = 0.0f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lg != 0) {
        if (!fused_dense_split_part<kFar>(cur, restart, first, n, T, apply_t, g, gate2, seed, sched, tile, (item >> 24) & 15u, lg)) {
            if (kDiag == 2 && (threadIdx.x & 63) == 0) light_stamp(wave_times, t_start, lg, item);
            return;
        }
        if (i < n) {   // the tile is complete: this workgroup adds it up, one query per thread as on the ordinary path
            q = coherent_load(&cur[i]);
            if (q.w != 0.0f) {
                pos = (int)coherent_load(reinterpret_cast<const uint32_t *>(sched.pos) + i);
                d2 = __uint_as_float(coherent_load(reinterpret_cast<const uint32_t *>(sched.d2) + i));
            }
        }
    } else if (i < n) {
        // restart (the first launch of an alignment): the query is the source point itself under the guess (T, apply_t: the
        // host's), the working copy is written whatever the point, and there is no seed -- what k_restart_source did
        q = restart ? restart[i] : cur[i];
        if (q.w != 0.0f) {
            if (restart) q.w = source_weight(first, i);   // (how many records of the caller's cloud this distinct point stands for)
            if (apply_t) {
                const float3 t = xform(T, q.x, q.y, q.z);
                q = make_float4(t.x, t.y, t.z, q.w);
            }
            if (apply_t || restart) cur[i] = q;
            const int seed_in = (seed && !restart) ? seed[i] : -1;
            const Best b = nn_query_dense<kFull, kFar>(g, q.x, q.y, q.z, seed_in, &dg);
            if (seed && (restart || b.pos != seed_in)) seed[i] = b.pos;   // (most matches do not change once the clouds have settled)
            if (b.pos >= 0 && !((double)b.d2 > gate2)) {
                pos = b.pos;
                d2 = b.d2;
            }
        } else if (restart) {
            cur[i] = q;
        }
        if (corr_pos) { corr_pos[i] = pos; corr_d2[i] = d2; }
    }
    if (kFull) t_search = wall_clock64();
    if (sched.cost && lg == 0 && (threadIdx.x & 63) == 0)
        sched.cost[tile * kTileWaves + (threadIdx.x >> 6)] = (uint32_t)min(wall_clock64() - t_start, 0xffffffffull);
    double a[RSREG_NUM_SUMS];
    for (int k = 0; k < RSREG_NUM_SUMS; ++k) a[k] = 0.0;
    if (pos >= 0) {
        const float4 t = g.pts[pos];
        accum_pair(a, q.x, q.y, q.z, t.x, t.y, tgt_z(t), d2, q.w);
    }
    tile_reduce_store(a, partials, sched.n_tiles, tile);
    if (kDiag == 2 && (threadIdx.x & 63) == 0) light_stamp(wave_times, t_start, lg, item);
    if (kFull) {
        // per wave: start, end of rings 0-1 (latest lane), end of far rings, end of search, end,
        // then max-over-lanes | sum-over-lanes (<< 32) of the step counts
        unsigned long long near_end = dg.t_near, far_end = dg.t_far;
        uint32_t mx[5] = {dg.own, dg.r1_cells, dg.r1_scans, dg.far_rows, dg.far_scans}, sm[5];
        for (int k = 0; k < 5; ++k) sm[k] = mx[k];
        for (int off = 32; off > 0; off >>= 1) {
            near_end = max(near_end, (unsigned long long)__shfl_down((long long)near_end, off));
            far_end = max(far_end, (unsigned long long)__shfl_down((long long)far_end, off));
            for (int k = 0; k < 5; ++k) {
                mx[k] = max(mx[k], (uint32_t)__shfl_down((int)mx[k], off));
                sm[k] += (uint32_t)__shfl_down((int)sm[k], off);
            }
        }
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *w = wave_times + 16ull * (i >> 6);
            w[0] = t_start; w[1] = near_end; w[2] = far_end; w[3] = t_search; w[4] = wall_clock64();
            for (int k = 0; k < 5; ++k) w[5 + k] = (unsigned long long)mx[k] | ((unsigned long long)sm[k] << 32);
            for (int k = 10; k < 16; ++k) w[k] = 0;
        }
        // per-lane step counts behind the wave records: own | ring-1 scans << 8 | far rows << 16 | far scans << 24
        if (i < n)
            reinterpret_cast<uint32_t *>(wave_times + 16ull * ((n + 63) / 64))[i] =
                min(dg.own, 255u) | min(dg.r1_scans, 255u) << 8 | min(dg.far_rows, 255u) << 16 | min(dg.far_scans, 255u) << 24;
    }
}

}  // namespace rsreg
