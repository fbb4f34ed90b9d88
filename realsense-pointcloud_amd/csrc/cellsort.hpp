// cellsort.hpp — the target index built as a COUNTING SORT on the dense cell table (gfx950, wave64).
//
// The dense table (icp_dense.hpp) is the histogram of a counting sort: a point's cell is a table slot, so the sorted
// array needs no comparison sort by cell id at all -- count the points of every cell, scan the counts, scatter.  What is
// left to order is the inside of a cell, by x position (dense_walk's early exit wants that, to g.x_slack; the lowest-index
// tie-break is decided by the search's (distance, index) keys, not by the order), and a cell holds six points on average:
//   k_cc_count    every point's cell; the points of a 4 096-point tile are pooled per cell in an LDS hash, so a cell gets
//                 ONE global atomic per tile whatever it holds, and the copies of a tile's lowest-index point of a cell
//                 are dropped on the spot (RealSense writes every invalid pixel as the point (0, 0, 0): 11 % of a frame
//                 in one cell; at most one of them per tile survives); a point's rank inside its cell, in arrival order,
//                 is kept per point;
//   k_cc_scan     one pass over the counts (what lies in front of a workgroup's span comes from per-span totals the
//                 counting kernel has left: no look-back): first record of every cell, the occupied cells in slot
//                 order, and the counts back to zero for the next build;
//   k_cc_scatter  record -> first record of its cell + its rank, in arrival order;
//   k_cc_small    cells of at most 48 records, one lane per record: its place is the number of the cell's records with a
//                 smaller (x position, original index);
//   k_cc_big      crowded cells, one wave per cell: a counting sort of its own over 256 x buckets in LDS.
// followed by k_dense_nbr as before.  It replaces keys + 2 histogram launches + 4 onesweep passes + k_dense_compact of
// the sort-based build (9 launches; still there for grids too large to scan and as RSREG_COUNT_SORT=0), uses no sort of
// any library, and moves every record twice instead of five times.
// Exact copies: the sort-based build drops a record that equals its predecessor in sorted order; this one drops the
// copies of a tile's lowest-index point of a cell and keeps every other record, copies included -- a copy has the
// coordinates of a record with a lower index, loses every tie against it and can be nobody's match either way.
// Replaces KdTreeFLANN's build inside pcl::Registration::initCompute (incremental_icp.hpp:58, icp_edge_based_registration.hpp:79,109).
#pragma once

#include "icp_dense.hpp"

namespace rsreg {

constexpr unsigned kCcBlock = 1024, kCcItems = 4, kCcTile = kCcBlock * kCcItems, kCcHash = 8192;
constexpr uint32_t kCcDropped = 0xffffffffu, kCcEmpty = 0xffffffffu;
constexpr uint32_t kCcSmall = 48;        // cells up to this many records: one lane per record (k_cc_small), beyond: one wave per cell
constexpr uint32_t kCcXBits = 8;         // x buckets per cell the crowded cells are ordered by (and g.x_slack is set for)
constexpr unsigned kCcScanItems = 16;    // table entries per thread of k_cc_scan: 16 384 per workgroup, all workgroups resident at 7 M cells

// slot of a finite point in the padded table (the same clamps as k_dense_keys)
__device__ __forceinline__ uint32_t cc_slot(const DenseDev &g, float x, float y, float z, int *cx_out = nullptr)
{
    const int cx = min(max(cell_coord(x, g.ox, g.inv_cell), 0), g.nx - 1), cy = min(max(cell_coord(y, g.oy, g.inv_cell), 0), g.ny - 1),
              cz = min(max(cell_coord(z, g.oz, g.inv_cell), 0), g.nz - 1);
    if (cx_out) *cx_out = cx;
    return dense_cell_id(g, cx, cy, cz);
}

// x position inside the cell, `xbits` bits: the second part of k_dense_keys' sort key
__device__ __forceinline__ uint32_t cc_xq(const DenseDev &g, float x, int cx, uint32_t xbits)
{
    const float fx = (cell_pos(x, g.ox, g.inv_cell) - (float)cx) * (float)(1u << xbits);
    return (uint32_t)min(max((int)fx, 0), (int)(1u << xbits) - 1);
}

__device__ __forceinline__ uint32_t cc_hash_xyz(float x, float y, float z)
{
    // +0.0f folds -0 into +0 so that value-equal points hash alike
    const uint32_t a = __float_as_uint(x + 0.0f), b = __float_as_uint(y + 0.0f), c = __float_as_uint(z + 0.0f);
    uint32_t h = a * 0x9e3779b1u;
    h = (h ^ (h >> 15)) + b * 0x85ebca77u;
    h = (h ^ (h >> 13)) + c * 0xc2b2ae3du;
    return h ^ (h >> 16);
}

// k_cc_scan's workgroup b covers the table entries of span b, [b * span, (b + 1) * span) with span = kCcChunk * m; what lies
// in front of it is the sum of the totals k_cc_count has left for the spans before it (points | occupied cells << 32): no
// look-back, no ticket.  At most kCcMaxSpans spans (k_cc_count pools a tile's share of them in an LDS array of that size).
constexpr uint32_t kCcScanBlock = 256, kCcChunk = kCcScanBlock * kCcScanItems, kCcMaxSpans = kCcHash;
inline uint32_t cc_span_chunks(size_t slots) { return (uint32_t)std::max<size_t>(1, (slots + (size_t)kCcChunk * kCcMaxSpans - 1) / ((size_t)kCcChunk * kCcMaxSpans)); }
inline uint32_t cc_spans(size_t slots) { const size_t span = (size_t)kCcChunk * cc_span_chunks(slots); return (uint32_t)((slots + span - 1) / span); }

// stats (device words): [0] occupied cells, [1] records in the sorted array, [2] crowded cells (entries of `big`)
__global__ __launch_bounds__(kCcBlock) void k_cc_count(const char *pts, size_t stride, uint32_t n, DenseDev g, uint32_t *gcnt, uint32_t *rank,
                                                       unsigned long long *coarse, uint32_t span, uint32_t *clear_a, uint32_t clear_a_words, uint32_t *clear_b, uint32_t clear_b_words, uint32_t *stats)
{
    __shared__ uint32_t s_key[kCcHash], s_min[kCcHash], s_hash[kCcHash];
    const uint32_t t = threadIdx.x, gt = blockIdx.x * kCcBlock + t, gthreads = gridDim.x * kCcBlock;
    // on the way: the occupancy words k_dense_nbr will OR together, and the coarse totals of the NEXT build
    if (clear_a) radix32_clear(clear_a, clear_a_words, gt, gthreads);
    if (clear_b) radix32_clear(clear_b, clear_b_words, gt, gthreads);
    if (gt == 0) stats[2] = 0u;
    for (uint32_t h = t; h < kCcHash; h += kCcBlock) { s_key[h] = kCcEmpty; s_min[h] = 0xffffffffu; }
    __syncthreads();
    const uint32_t i0 = blockIdx.x * kCcTile + t;
    float px[kCcItems], py[kCcItems], pz[kCcItems];
    uint32_t ent[kCcItems];
#pragma unroll
    for (uint32_t j = 0; j < kCcItems; ++j) {
        const uint32_t i = i0 + j * kCcBlock;
        ent[j] = kCcEmpty;
        px[j] = py[j] = pz[j] = 0.0f;
        if (i < n) {
            const float *p = rec_xyz(pts, stride, i);
            px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
            if (finite3(px[j], py[j], pz[j])) {
                const uint32_t slot = cc_slot(g, px[j], py[j], pz[j]);
                uint32_t h = (slot * 0x9E3779B1u) >> 19;   // 13 bits
                for (;;) {   // (at most 4 096 distinct cells in 8 192 entries: an empty one always comes)
                    const uint32_t prev = atomicCAS(&s_key[h], kCcEmpty, slot);
                    if (prev == kCcEmpty || prev == slot) break;
                    h = (h + 1u) & (kCcHash - 1u);
                }
                ent[j] = h;
                atomicMin(&s_min[h], i);
            }
        }
    }
    __syncthreads();
    // the tile's lowest-index point of every cell stands for its copies in the tile (same coordinates, lower index: it wins
    // every tie against a copy, so a copy can never be anybody's match and is not counted)
    uint32_t rep[kCcItems];
#pragma unroll
    for (uint32_t j = 0; j < kCcItems; ++j) rep[j] = ent[j] != kCcEmpty ? s_min[ent[j]] : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kCcItems; ++j)
        if (ent[j] != kCcEmpty && rep[j] == i0 + j * kCcBlock) {
            s_hash[ent[j]] = cc_hash_xyz(px[j], py[j], pz[j]);
            s_min[ent[j]] = 0u;   // from here on: the tile's count of the cell
        }
    __syncthreads();
    uint32_t local[kCcItems];
#pragma unroll
    for (uint32_t j = 0; j < kCcItems; ++j) {
        local[j] = kCcDropped;
        if (ent[j] != kCcEmpty) {
            const uint32_t i = i0 + j * kCcBlock;
            bool keep = true;
            if (rep[j] != i && s_hash[ent[j]] == cc_hash_xyz(px[j], py[j], pz[j])) {   // (the coordinates themselves only where the hashes agree)
                const float *q = rec_xyz(pts, stride, rep[j]);
                keep = !(q[0] == px[j] && q[1] == py[j] && q[2] == pz[j]);
            }
            if (keep) local[j] = atomicAdd(&s_min[ent[j]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t h = t; h < kCcHash; h += kCcBlock) s_hash[h] = 0u;   // (from here on: the tile's share of every span, cells << 16 | points)
    __syncthreads();
    for (uint32_t h = t; h < kCcHash; h += kCcBlock)
        if (s_key[h] != kCcEmpty) {
            const uint32_t slot = s_key[h], c = s_min[h];
            const uint32_t base = atomicAdd(&gcnt[slot], c);
            s_min[h] = base;   // (from here on: where the tile's share of the cell begins)
            // the span's totals for k_cc_scan: its points, and its cell when this tile is the first to count into it --
            // pooled per tile first (a tile's cells lie in a handful of spans: same-address global atomics queue up)
            atomicAdd(&s_hash[slot / span], c | (base == 0u ? 1u << 16 : 0u));
        }
    __syncthreads();
    for (uint32_t h = t; h < kCcHash; h += kCcBlock) {
        const uint32_t v = s_hash[h];
        if (v) atomicAdd(&coarse[h], (unsigned long long)(v & 0xffffu) | (unsigned long long)(v >> 16) << 32);
    }
#pragma unroll
    for (uint32_t j = 0; j < kCcItems; ++j) {
        const uint32_t i = i0 + j * kCcBlock;
        if (i < n) rank[i] = local[j] != kCcDropped ? s_min[ent[j]] + local[j] : kCcDropped;
    }
}

// One pass over the counts of all `slots` table entries: table[slot] = first record of EVERY cell (so a cell's end is the
// entry behind it, and the row search of unbounded gates finds empty cells' entries valid too), the occupied cells' slots and
// first records in slot order (cellslot / cellpos), the slots of the cells beyond kCcSmall records (`big`, in any order),
// and the counts back to zero.  Everything a workgroup reads and writes of the table and the counts is one contiguous piece.
// coarse[b]: points | occupied cells << 32 of span b (k_cc_count); chunks: kCcChunk-entry pieces per span.
__global__ __launch_bounds__(kCcScanBlock) void k_cc_scan(uint32_t *gcnt, uint32_t slots, uint32_t chunks, const unsigned long long *coarse, uint32_t *table,
                                                          uint32_t *cellslot, uint32_t *cellpos, uint32_t *big, uint32_t *stats, uint32_t *host_stats)
{
    __shared__ unsigned long long s_wave[kCcScanBlock / 64];
    __shared__ unsigned long long s_front[kCcScanBlock / 64];
    const uint32_t bid = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // what lies in front of this span: the totals of the spans before it
    unsigned long long front = 0;
    for (uint32_t b = threadIdx.x; b < bid; b += kCcScanBlock) front += coarse[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) front += (unsigned long long)__shfl_down((long long)front, off);
    if (lane == 0) s_front[wave] = front;
    __syncthreads();
    unsigned long long carry = 0;
#pragma unroll
    for (uint32_t w = 0; w < kCcScanBlock / 64; ++w) carry += s_front[w];
    for (uint32_t ch = 0; ch < chunks; ++ch) {
        const uint32_t i0 = ((bid * chunks + ch) * kCcScanBlock + threadIdx.x) * kCcScanItems;
        if ((bid * chunks + ch) * kCcChunk >= slots) break;   // (the whole workgroup: the last span may be short)
        uint32_t v[kCcScanItems];
#pragma unroll
        for (uint32_t q4 = 0; q4 < kCcScanItems / 4; ++q4) {
            const uint32_t at = i0 + 4u * q4;
            if (at + 4u <= slots) {
                const uint4 q = *reinterpret_cast<const uint4 *>(gcnt + at);   // (the counts start on a 16-byte boundary)
                v[4 * q4] = q.x; v[4 * q4 + 1] = q.y; v[4 * q4 + 2] = q.z; v[4 * q4 + 3] = q.w;
                *reinterpret_cast<uint4 *>(gcnt + at) = make_uint4(0u, 0u, 0u, 0u);
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    v[4 * q4 + j] = at + j < slots ? gcnt[at + j] : 0u;
                    if (at + j < slots) gcnt[at + j] = 0u;
                }
            }
        }
        unsigned long long mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kCcScanItems; ++j) mine += (unsigned long long)v[j] | (unsigned long long)(v[j] != 0u) << 32;
        // exclusive scan over the workgroup, in thread order
        unsigned long long inc = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = (unsigned long long)__shfl_up((long long)inc, off);
            if ((int)lane >= off) inc += o;
        }
        __syncthreads();   // (s_wave of the chunk before has been read by everybody)
        if (lane == 63u) s_wave[wave] = inc;
        __syncthreads();
        unsigned long long before = 0, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < kCcScanBlock / 64; ++w) {
            const unsigned long long tw = s_wave[w];
            if (w < wave) before += tw;
            tot += tw;
        }
        unsigned long long run = carry + before + inc - mine;
        carry += tot;
        uint32_t starts[kCcScanItems];
#pragma unroll
        for (uint32_t j = 0; j < kCcScanItems; ++j) {
            const uint32_t slot = i0 + j;
            const uint32_t pos = (uint32_t)run, cid = (uint32_t)(run >> 32);
            starts[j] = pos;
            if (slot < slots && v[j]) {
                cellslot[cid] = slot;
                cellpos[cid] = pos;
            }
            run += (unsigned long long)v[j] | (unsigned long long)(v[j] != 0u) << 32;
            if (slot == slots - 1u) {
                const uint32_t nrec = (uint32_t)run, nc = (uint32_t)(run >> 32);
                stats[0] = nc;
                stats[1] = nrec;
                host_stats[0] = nc;     // (pinned host memory: read when the build has drained, no copy queued)
                host_stats[1] = nrec;
                cellpos[nc] = nrec;   // sentinel
                table[slots] = nrec;
            }
        }
        // the crowded cells (k_cc_big takes them one wave each) are listed in any order: a wave pools its own and takes one
        // place in the list for all of them (same-address atomics from every thread would queue up behind each other)
        uint32_t nb = 0;
#pragma unroll
        for (uint32_t j = 0; j < kCcScanItems; ++j) nb += (i0 + j < slots && v[j] > kCcSmall) ? 1u : 0u;
        if (__any(nb != 0u)) {
            uint32_t incl = nb;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off);
                if ((int)lane >= off) incl += o;
            }
            uint32_t at_list = 0;
            if (lane == 63u) at_list = atomicAdd(&stats[2], incl);
            at_list = __shfl(at_list, 63) + incl - nb;
#pragma unroll
            for (uint32_t j = 0; j < kCcScanItems; ++j)
                if (i0 + j < slots && v[j] > kCcSmall) big[at_list++] = i0 + j;
        }
#pragma unroll
        for (uint32_t q4 = 0; q4 < kCcScanItems / 4; ++q4) {
            const uint32_t at = i0 + 4u * q4;
            if (at + 4u <= slots) {
                *reinterpret_cast<uint4 *>(table + at) = make_uint4(starts[4 * q4], starts[4 * q4 + 1], starts[4 * q4 + 2], starts[4 * q4 + 3]);
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j)
                    if (at + j < slots) table[at + j] = starts[4 * q4 + j];
            }
        }
    }
}

// record -> first record of its cell + its rank in arrival order
__device__ __forceinline__ void cc_scatter_body(uint32_t bid, const char *pts, size_t stride, uint32_t n, const DenseDev &g, const uint32_t *rank,
                                                const uint32_t *table, float4 *arrived)
{
    const uint32_t i = bid * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rank[i];
    if (r == kCcDropped) return;
    const float *p = rec_xyz(pts, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    arrived[table[cc_slot(g, x, y, z)] + r] = tgt_rec(x, y, z, i);
}

__global__ __launch_bounds__(kBlock) void k_cc_scatter(const char *pts, size_t stride, uint32_t n, DenseDev g, const uint32_t *rank, const uint32_t *table,
                                                       float4 *arrived)
{
    cc_scatter_body(blockIdx.x, pts, stride, n, g, rank, table, arrived);
}

// k_cc_scatter and k_dense_nbr need k_cc_scan's results and nothing of each other: ONE launch, the occupancy words'
// workgroups (the longer job: scattered atomics) in front.  Round 6: the build is count, scan, this, k_cc_small_big -- four
// dependent launches where it had six (RSREG_CC_APART=1: the six).
__global__ __launch_bounds__(kBlock) void k_cc_scatter_nbr(const char *pts, size_t stride, uint32_t n, DenseDev g, const uint32_t *rank, const uint32_t *table,
                                                           float4 *arrived, uint32_t nbr_blocks, const uint32_t *cellslot, const uint32_t *stats, uint32_t *occ)
{
    if (blockIdx.x < nbr_blocks) dense_nbr_body(blockIdx.x, cellslot, stats, g.sx, g.sxy, occ);
    else cc_scatter_body(blockIdx.x - nbr_blocks, pts, stride, n, g, rank, table, arrived);
}

// One lane per record of the arrival-order array, for the cells of at most kCcSmall records: the record belongs behind the
// cell's records with a smaller (x position to 16 bits, original index).  Records of crowded cells are k_cc_big's.
// pos_of: original index -> position.
__device__ __forceinline__ void cc_small_body(uint32_t bid, const float4 *arrived, const DenseDev &g, const uint32_t *table, float4 *sorted, uint32_t *pos_of,
                                              const uint32_t *stats)
{
    const uint32_t p = bid * blockDim.x + threadIdx.x, nrec = stats[1];
    if (p == 0) {
        // far-away points behind the last sorted point: a 4-wide candidate read may run past it
        for (uint32_t k = 0; k < 4; ++k) sorted[nrec + k] = tgt_rec(1e30f, 1e30f, 1e30f, 0xffffffffu);
    }
    if (p >= nrec) return;
    const float4 me = arrived[p];
    const uint32_t idx = tgt_idx(me);
    int cx;
    const uint32_t slot = cc_slot(g, me.x, me.y, tgt_z(me), &cx);
    const uint32_t s = table[slot], e = table[slot + 1u];
    if (e - s > kCcSmall) return;
    const unsigned long long key = (unsigned long long)cc_xq(g, me.x, cx, 16u) << 32 | idx;
    uint32_t before = 0;
    for (uint32_t k = s; k < e; ++k) {
        const float4 o = arrived[k];
        const unsigned long long ko = (unsigned long long)cc_xq(g, o.x, cx, 16u) << 32 | tgt_idx(o);
        before += ko < key ? 1u : 0u;
    }
    sorted[s + before] = me;
    pos_of[idx] = s + before;
}

__global__ __launch_bounds__(kBlock) void k_cc_small(const float4 *arrived, DenseDev g, const uint32_t *table, float4 *sorted, uint32_t *pos_of,
                                                     const uint32_t *stats)
{
    cc_small_body(blockIdx.x, arrived, g, table, sorted, pos_of, stats);
}

// The crowded cells (more than kCcSmall records: `big`, k_cc_scan's list), one wave per cell: count the cell's records per
// x bucket (kCcXBits bits) in LDS, scan the 256 counts, place every record behind the buckets before its own, in arrival
// order inside a bucket.
__device__ __forceinline__ void cc_big_body(uint32_t bid, uint32_t nblocks, const float4 *arrived, const DenseDev &g, const uint32_t *table, const uint32_t *big,
                                            float4 *sorted, uint32_t *pos_of, const uint32_t *stats)
{
    constexpr uint32_t kBins = 1u << kCcXBits;
    __shared__ uint32_t s_bin[kBlock / 64][kBins];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nbig = stats[2];
    uint32_t *bin = s_bin[wave];
    for (uint32_t c = bid * (kBlock / 64) + wave; c < nbig; c += nblocks * (kBlock / 64)) {
        const uint32_t slot = big[c];
        const uint32_t s = table[slot], e = table[slot + 1u];
        // (the slot's x coordinate: the padded table's index is ((z + 1) * (ny + 2) + (y + 1)) * (nx + 2) + (x + 1))
        const int cx = (int)(slot % (uint32_t)(g.nx + 2)) - 1;
        for (uint32_t b = lane; b < kBins; b += 64u) bin[b] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = s + lane; k < e; k += 64u) atomicAdd(&bin[cc_xq(g, arrived[k].x, cx, kCcXBits)], 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // exclusive scan of the 256 counts: four consecutive bins per lane
        uint32_t v[kBins / 64], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < kBins / 64; ++j) { v[j] = bin[lane * (kBins / 64) + j]; sum += v[j]; }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if ((int)lane >= off) incl += o;
        }
        uint32_t at = incl - sum;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t j = 0; j < kBins / 64; ++j) { bin[lane * (kBins / 64) + j] = at; at += v[j]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = s + lane; k < e; k += 64u) {
            const float4 r = arrived[k];
            const uint32_t to = s + atomicAdd(&bin[cc_xq(g, r.x, cx, kCcXBits)], 1u);
            sorted[to] = r;
            pos_of[tgt_idx(r)] = to;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(kBlock) void k_cc_big(const float4 *arrived, DenseDev g, const uint32_t *table, const uint32_t *big, float4 *sorted,
                                                   uint32_t *pos_of, const uint32_t *stats)
{
    cc_big_body(blockIdx.x, gridDim.x, arrived, g, table, big, sorted, pos_of, stats);
}

// The in-cell order of the small cells and of the crowded ones touch different records: ONE launch, the crowded cells'
// workgroups (a wave per cell, the long jobs) in front.
__global__ __launch_bounds__(kBlock) void k_cc_small_big(const float4 *arrived, DenseDev g, const uint32_t *table, const uint32_t *big, float4 *sorted,
                                                         uint32_t *pos_of, const uint32_t *stats, uint32_t big_blocks)
{
    if (blockIdx.x < big_blocks) cc_big_body(blockIdx.x, big_blocks, arrived, g, table, big, sorted, pos_of, stats);
    else cc_small_body(blockIdx.x - big_blocks, arrived, g, table, sorted, pos_of, stats);
}

}  // namespace rsreg
