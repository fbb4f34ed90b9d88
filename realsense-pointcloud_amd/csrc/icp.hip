// icp.hip — ICP half of the C ABI (include/rsreg.h): context, target grid build, the
// iteration loop of pcl::IterativeClosestPoint::computeTransformation, transformPointCloud.
//
// Reference call sites replaced: src/incremental_icp.hpp:46-49,57-63,
// src/icp_edge_based_registration.hpp:42-52,78-79,95,104,108-117,
// src/ndt_edge_based_registration.hpp:47-50,96-105.
#include <cstring>
#include <string.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <new>
#include <thread>
#include <vector>

#include "icp_kernels.hpp"
#include "icp_dense.hpp"
#include "cellsort.hpp"
#include "oscan.hpp"

using namespace rsreg;

namespace {

inline uint32_t reduce_blocks(size_t n) { return (uint32_t)std::max<size_t>((n + kTile - 1) / kTile, 1); }  // depends on n only

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

hipEvent_t take_event(rsreg_ctx *ctx)
{
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[ctx->ev_used++];
}

struct ScopedEvents {  // records start/stop around a launch when profiling is on
    rsreg_ctx *ctx;
    std::vector<std::pair<size_t, size_t>> *list;
    size_t a = 0, b = 0;
    bool on;
    ScopedEvents(rsreg_ctx *c, std::vector<std::pair<size_t, size_t>> *l) : ctx(c), list(l), on(c->profiling)
    {
        if (!on) return;
        hipEvent_t e = take_event(ctx);
        a = ctx->ev_used - 1;
        if (e) (void)hipEventRecord(e, ctx->stream);
    }
    ~ScopedEvents()
    {
        if (!on) return;
        hipEvent_t e = take_event(ctx);
        b = ctx->ev_used - 1;
        if (e) (void)hipEventRecord(e, ctx->stream);
        list->push_back({a, b});
    }
};

double sum_events(rsreg_ctx *ctx, std::vector<std::pair<size_t, size_t>> &list)
{
    double ms = 0;
    for (auto &p : list) {
        float t = 0;
        if (hipEventElapsedTime(&t, ctx->ev_pool[p.first], ctx->ev_pool[p.second]) == hipSuccess) ms += t;
    }
    return ms;
}

GridDev grid_dev(const rsreg_ctx *ctx, double max_dist)
{
    const GridParams &p = ctx->grid;
    GridDev g;
    g.ox = p.origin[0]; g.oy = p.origin[1]; g.oz = p.origin[2];
    g.inv_cell = p.inv_cell; g.cell = p.cell;
    g.dx = p.dims[0]; g.dy = p.dims[1]; g.dz = p.dims[2];
    g.bmask = p.table_mask;
    g.max_ring = p.max_ring;
    // squared search radius as a float that is never below the f64 gate PCL compares with
    const double gate2 = max_dist * max_dist;
    if (!(gate2 < (double)FLT_MAX)) {
        g.prune2 = INFINITY;
    } else {
        float f = (float)gate2;
        if ((double)f < gate2) f = std::nextafter(f, INFINITY);
        g.prune2 = f;
    }
    g.bricks = ctx->d_table.as<BrickEntry>();
    g.cellpos = ctx->d_cellpos.as<uint32_t>();
    g.pts = ctx->d_tgt_sorted.as<float4>();
    return g;
}

// host mirror of cell_coord (same IEEE operations, no contraction)
int host_cell_coord(float p, float origin, float inv_cell)
{
    volatile float d = p - origin;
    volatile float v = d * inv_cell;
    float f = std::floor(v);
    f = std::min(std::max(f, -4.0f), 70000.0f);
    return (int)f;
}

double cell_cap_from_env() { return tunables().cell_cap; }

// bounding box + count of the finite points of a device-resident cloud (one host sync), on the given stream with
// the given scratch: d_misc (64 words; result in the first 16), h_misc (pinned, 16 words), partial (1024 x 8 words)
int device_bbox_on(rsreg_ctx *ctx, hipStream_t st, uint32_t *d_misc, uint32_t *h_misc, uint32_t *partial, const char *d_pts, size_t n,
                   size_t stride, float mn[3], float mx[3], uint32_t *nfin)
{
    if (n == 0) {   // no kernel runs: the empty box
        for (int k = 0; k < 3; ++k) { mn[k] = ordered_float(0xffffffffu); mx[k] = ordered_float(0u); }
        *nfin = 0;
        return RSREG_OK;
    }
    if (n > 0) {
        const uint32_t nb = std::min<uint32_t>(div_up((uint32_t)n, kBlock), 1024);
        k_bbox<<<nb, kBlock, 0, st>>>(d_pts, stride, (uint32_t)n, partial);
        RSREG_HIP(ctx, hipGetLastError());
        k_bbox_final<<<1, kBlock, 0, st>>>(partial, nb, h_misc, d_misc);   // the box straight into the pinned host buffer (no copy queued); clears the counters
        RSREG_HIP(ctx, hipGetLastError());
    }
    RSREG_HIP(ctx, hipStreamSynchronize(st));
    *nfin = h_misc[6];
    for (int k = 0; k < 3; ++k) { mn[k] = ordered_float(h_misc[k]); mx[k] = ordered_float(h_misc[3 + k]); }
    return RSREG_OK;
}

int device_bbox(rsreg_ctx *ctx, const char *d_pts, size_t n, size_t stride, float mn[3], float mx[3], uint32_t *nfin)
{
    RSREG_HIP(ctx, ctx->d_misc.reserve(64 * sizeof(uint32_t)));
    RSREG_HIP(ctx, ctx->h_sums.reserve(64 * 8));
    RSREG_HIP(ctx, ctx->d_comm.reserve(1024 * 8 * sizeof(uint32_t) + 64 * sizeof(double)));
    return device_bbox_on(ctx, ctx->stream, ctx->d_misc.as<uint32_t>(), ctx->h_sums.as<uint32_t>(),
                          reinterpret_cast<uint32_t *>(ctx->d_comm.as<char>() + 64 * sizeof(double)), d_pts, n, stride, mn, mx, nfin);
}

float prune2_of(double max_dist)
{
    // squared search radius as a float that is never below the f64 gate PCL compares with
    const double gate2 = max_dist * max_dist;
    if (!(gate2 < (double)FLT_MAX)) return INFINITY;
    float f = (float)gate2;
    if ((double)f < gate2) f = std::nextafter(f, INFINITY);
    return f;
}

DenseDev dense_dev(const rsreg_ctx *ctx, double max_dist)
{
    const GridParams &p = ctx->grid;
    DenseDev g;
    g.ox = p.origin[0]; g.oy = p.origin[1]; g.oz = p.origin[2];
    g.inv_cell = p.inv_cell; g.cell = p.cell;
    g.nx = p.dims[0]; g.ny = p.dims[1]; g.nz = p.dims[2];
    g.sx = p.dims[0] + 2;
    g.sxy = (p.dims[0] + 2) * (p.dims[1] + 2);
    g.max_ring = p.max_ring;
    g.prune2 = prune2_of(max_dist);
    g.start = ctx->d_dense.as<uint32_t>();
    g.pts = ctx->d_tgt_sorted.as<float4>();
    g.n_pts = p.n_points;
    g.table_bytes = (uint32_t)(((size_t)(p.dims[0] + 2) * (p.dims[1] + 2) * (p.dims[2] + 2) + 2) * 4);
    g.nbr = ctx->d_dense.as<uint32_t>() + g.table_bytes / 4;   // the occupancy words lie right behind the table (one memset clears both)
    g.pos_of = ctx->d_pos_of.as<uint32_t>();
#ifdef RSREG_DIAG   // (diagnostic builds only -- RSREG_CXXFLAGS=-DRSREG_DIAG: the shipped library has no switch that changes a result)
    g.debug_skip = tunables().debug_skip;
#endif
    // positions in cell units carry the rounding of (p - origin) * inv_cell, ~2^-23 of their size
    g.margin = std::min(kCellMargin, std::max(0.004f, 6.0e-7f * (float)std::max(p.dims[0], std::max(p.dims[1], p.dims[2]))));
    // how far the x order of a sorted run can be off: one bucket of the sort key (2^-xbits of a cell), plus the float
    // rounding of (x - ox) * inv_cell, which grows with the grid (ulp of the largest in-grid position)
    g.x_slack = p.cell * (std::ldexp(1.0f, -(p.xbits > 0 ? p.xbits : 16)) + 9.5367431640625e-7f * (float)std::max(p.dims[0], 1));
    return g;
}

long long dense_cell_budget() { return tunables().dense_max_cells; }

// Dense-table index (icp_dense.hpp).  The grid geometry (ctx->grid) is already decided.
// KeyT: the sort key's type -- uint32_t when the cell id and at least 6 bits of x position fit 32 bits (build_dense).
template <typename KeyT>
int build_dense_keyed(rsreg_ctx *ctx, const char *d_pts, size_t n, size_t stride, double max_dist, uint32_t nfin, int id_bits)
{
    hipStream_t st = ctx->stream;
    GridParams &gp = ctx->grid;
    const size_t total = (size_t)(gp.dims[0] + 2) * (gp.dims[1] + 2) * (gp.dims[2] + 2);
    uint32_t *d_misc = ctx->d_misc.as<uint32_t>();
    uint32_t *h_misc = ctx->h_smisc.as<uint32_t>() + 32;   // (+ 8, + 10: pinned words of the counts' own -- an alignment's sums land in h_sums)
    RSREG_HIP(ctx, ctx->d_keys.reserve(n * sizeof(KeyT)));
    RSREG_HIP(ctx, ctx->d_keys_alt.reserve(n * sizeof(KeyT)));
    const DenseDev g = dense_dev(ctx, max_dist);
    auto *keys = ctx->d_keys.as<KeyT>();
    auto *keys2 = ctx->d_keys_alt.as<KeyT>();
    auto *vals = ctx->d_vals.as<uint32_t>();
    auto *vals2 = ctx->d_vals_alt.as<uint32_t>();
    auto *flags = ctx->d_flags.as<unsigned long long>(), *scan = ctx->d_scan.as<unsigned long long>();   // keep | cstart << 32 and its scan
    uint32_t *cellslot = ctx->d_brick.as<uint32_t>(), *cellpos = ctx->d_cellpos.as<uint32_t>();
    uint32_t *table = ctx->d_dense.as<uint32_t>();
    const uint32_t xbits = (uint32_t)gp.xbits;
    // a table that is only read where an occupancy bit points is written by the scatter kernel, occupied cells only: no
    // clear, no scan of (nx+2)(ny+2)(nz+2) entries; the occupancy words behind it are OR-ed together and start from zero
    // (the occupancy words alone are cleared by the keys kernel on its way, unless that would be more than 64 words a thread)
    const bool occ_on_the_way = gp.table_sparse && (total + 2) <= 64ull * div_up((uint32_t)n, kBlock) * kBlock;
    if (!gp.table_sparse) RSREG_HIP(ctx, hipMemsetAsync(table, 0, (total + 2) * 4 * 2, st));
    else if (!occ_on_the_way) RSREG_HIP(ctx, hipMemsetAsync(table + (total + 2), 0, (total + 2) * 4, st));
    const unsigned end_bit = (unsigned)std::min<int>((int)sizeof(KeyT) * 8, (int)xbits + id_bits);
    // the library's own radix sort (osort.hpp; 32- and 64-bit keys alike since round 6), its state cleared by the keys kernel on
    // its way instead of by memsets
    const Radix32Plan plan = radix32_plan<KeyT>(n, 0, end_bit);
    // ... and then flag, scan and scatter are one launch too (compact.hpp), its look-back words cleared with the sort's state
    const bool scan_apart = tunables().scan_apart;
    const bool one_tail = !scan_apart && nfin < 0x7fffffffu;
    const CompactPlan cplan = compact_plan(nfin, plan.words);
    const uint32_t scratch_words = one_tail ? cplan.end : plan.words;
    const size_t sort_bytes = (size_t)scratch_words * 4;
    const size_t scan_bytes = oscan_scratch_bytes<unsigned long long>(nfin), tscan_bytes = oscan_scratch_bytes<uint32_t>(total + 1);
    // (the sort's state, then -- on the paths that scan apart -- the scans' sums behind it)
    const size_t off_scan = (sort_bytes + 255) & ~(size_t)255;
    RSREG_HIP(ctx, ctx->d_tmp.reserve(off_scan + std::max(scan_bytes, tscan_bytes) + 256));
    char *scan_scratch = ctx->d_tmp.as<char>() + off_scan;
    // (an even number of passes ends in the pair it started from: the keys are then written where the result belongs)
    const bool start_in_out = plan.ends_in_first;
    k_dense_keys<KeyT><<<div_up((uint32_t)n, kBlock), kBlock, 0, st>>>(d_pts, stride, (uint32_t)n, g, xbits, start_in_out ? keys2 : keys,
                                                                       start_in_out ? vals2 : vals, ctx->d_tmp.as<uint32_t>(), scratch_words,
                                                                       occ_on_the_way ? table + (total + 2) : nullptr,
                                                                       occ_on_the_way ? (uint32_t)(total + 2) : 0u);
    RSREG_HIP(ctx, hipGetLastError());
    {
        bool in_first = false;
        RSREG_HIP(ctx, radix32_sort_pairs<KeyT>(plan, ctx->d_tmp.as<uint32_t>(), start_in_out ? keys2 : keys, start_in_out ? keys : keys2,
                                                start_in_out ? vals2 : vals, start_in_out ? vals : vals2, n, 0, end_bit, st, &in_first));
        if (in_first != start_in_out) return fail(ctx, RSREG_ERR_STATE, "osort: the sorted pairs are not where they belong");
    }
    const uint32_t nbf = div_up(nfin, kBlock);
    if (one_tail) {
        uint32_t *scr = ctx->d_tmp.as<uint32_t>();
        k_dense_compact<KeyT><<<cplan.blocks, kCompactBlock, 0, st>>>(keys2, vals2, d_pts, stride, nfin, xbits, ctx->d_tgt_sorted.as<float4>(),
                                                                      ctx->d_pos_of.as<uint32_t>(), cellslot, cellpos, d_misc + 8, h_misc + 8,
                                                                      gp.table_sparse ? table : nullptr,
                                                                      reinterpret_cast<unsigned long long *>(scr + cplan.off_state), scr + cplan.off_ticket);
        RSREG_HIP(ctx, hipGetLastError());
    } else {
        k_dense_flag<KeyT><<<nbf, kBlock, 0, st>>>(keys2, vals2, d_pts, stride, nfin, xbits, flags);
        RSREG_HIP(ctx, hipGetLastError());
        RSREG_HIP(ctx, (oscan<unsigned long long>(flags, scan, (size_t)nfin, 0ull, scan_scratch, st)));
        k_dense_scatter<KeyT><<<nbf, kBlock, 0, st>>>(keys2, vals2, d_pts, stride, nfin, xbits, flags, scan, ctx->d_tgt_sorted.as<float4>(),
                                                      ctx->d_pos_of.as<uint32_t>(), cellslot, cellpos, d_misc + 8, h_misc + 8,
                                                      gp.table_sparse ? table : nullptr);
        RSREG_HIP(ctx, hipGetLastError());
    }
    if (!gp.table_sparse) {
        k_dense_counts<<<nbf, kBlock, 0, st>>>(cellslot, cellpos, d_misc + 8, table);
        RSREG_HIP(ctx, hipGetLastError());
        // counts -> first sorted point of every cell (in place), entry [total] = number of points
        RSREG_HIP(ctx, (oscan<uint32_t>(table, table, total + 1, 0u, scan_scratch, st)));
    }
    // occupancy word of every cell's 27-cell neighbourhood: a query never opens an empty cell
    k_dense_nbr<<<nbf, kBlock, 0, st>>>(cellslot, d_misc + 8, g.sx, g.sxy, table + (total + 2));
    RSREG_HIP(ctx, hipGetLastError());
    return RSREG_OK;
}

// The same index by counting (cellsort.hpp): no sort, four launches + k_dense_nbr.  For grids a pass over all table
// entries is cheap for: at most 32 cells per point, or 16 M cells.  RSREG_COUNT_SORT=0: never.
bool count_sort_pays(size_t n, size_t total)
{
    return tunables().count_sort && total <= std::max<size_t>(32 * n, (size_t)16 << 20) && total < (1ull << 31);
}

// sources up to this many points (the one-launch load's limit, kPlainSourceMax below) make their target's counting build leave
// the occupancy words out when the gate fits into ring 1
constexpr size_t kSmallSourceForTable = 65536;

int build_dense_counted(rsreg_ctx *ctx, const char *d_pts, size_t n, size_t stride, double max_dist, uint32_t nfin, bool with_nbr)
{
    hipStream_t st = ctx->stream;
    GridParams &gp = ctx->grid;
    const size_t total = (size_t)(gp.dims[0] + 2) * (gp.dims[1] + 2) * (gp.dims[2] + 2);
    uint32_t *d_misc = ctx->d_misc.as<uint32_t>();
    uint32_t *h_counts = ctx->h_smisc.as<uint32_t>() + 40;   // pinned words of the counts' own
    // the counts are zero between builds (k_cc_scan puts them back); a new or larger buffer, or a build that did not
    // get to its end, starts from a memset
    RSREG_HIP(ctx, ctx->d_cnt.reserve((total + 16) * 4 + 2 * (size_t)kCcMaxSpans * 8));
    if (ctx->cnt_zero_ptr != ctx->d_cnt.ptr || ctx->cnt_zero_cap != ctx->d_cnt.cap || ctx->cnt_dirty) {
        RSREG_HIP(ctx, hipMemsetAsync(ctx->d_cnt.ptr, 0, ctx->d_cnt.cap, st));
        ctx->cnt_zero_ptr = ctx->d_cnt.ptr;
        ctx->cnt_zero_cap = ctx->d_cnt.cap;
    }
    ctx->cnt_dirty = true;
    RSREG_HIP(ctx, ctx->d_arrived.reserve(((size_t)nfin + 8) * sizeof(float4)));
    const DenseDev g = dense_dev(ctx, max_dist);
    uint32_t *cnt = ctx->d_cnt.as<uint32_t>() + 4u * kCcMaxSpans, *rank = ctx->d_vals.as<uint32_t>();   // (behind the two sets of per-span totals)
    uint32_t *cellslot = ctx->d_brick.as<uint32_t>(), *cellpos = ctx->d_cellpos.as<uint32_t>();
    uint32_t *table = ctx->d_dense.as<uint32_t>(), *occ = table + (total + 2);
    uint32_t *big = ctx->d_flags.as<uint32_t>();   // (n * 8 bytes: room for every cell beyond kCcSmall records)
    uint32_t *stats = d_misc + 8;
    // per-span totals of the counts (what k_cc_scan's workgroups start from): two sets used in turn, each cleared by the
    // counting kernel of the build before the one that fills it; they lie in front of the counts (zeroed with them)
    const uint32_t chunks = cc_span_chunks(total), spans = cc_spans(total), span = chunks * kCcChunk;
    auto *coarse_all = ctx->d_cnt.as<unsigned long long>();   // (at the head of the buffer, wherever the table ends)
    unsigned long long *coarse = coarse_all + (size_t)(ctx->cnt_flip ? kCcMaxSpans : 0), *coarse_next = coarse_all + (size_t)(ctx->cnt_flip ? 0 : kCcMaxSpans);
    ctx->cnt_flip = !ctx->cnt_flip;
    const uint32_t count_blocks = div_up((uint32_t)n, kCcTile);
    // the occupancy words are cleared by the counting kernel on its way, unless that would be more than 64 words a thread
    // (with_nbr false: an index without the words -- a small source's searches read them off the table, icp_dense.hpp:
    //  dense_occ_from_table; for a sparse cloud the words were half the build: 18 scattered atomics per occupied cell and a
    //  table's worth of zeroes)
    const bool occ_on_the_way = with_nbr && (total + 2) <= 64ull * count_blocks * kCcBlock;
    if (with_nbr && !occ_on_the_way) RSREG_HIP(ctx, hipMemsetAsync(occ, 0, (total + 2) * 4, st));
    k_cc_count<<<count_blocks, kCcBlock, 0, st>>>(d_pts, stride, (uint32_t)n, g, cnt, rank, coarse, span, occ_on_the_way ? occ : nullptr,
                                                   occ_on_the_way ? (uint32_t)(total + 2) : 0u, reinterpret_cast<uint32_t *>(coarse_next), 2u * kCcMaxSpans, stats);
    RSREG_HIP(ctx, hipGetLastError());
    k_cc_scan<<<spans, kCcScanBlock, 0, st>>>(cnt, (uint32_t)total, chunks, coarse, table, cellslot, cellpos, big, stats, h_counts);
    RSREG_HIP(ctx, hipGetLastError());
    const uint32_t nbf = div_up(nfin, kBlock), scatter_blocks = div_up((uint32_t)n, kBlock);
    // (one wave per crowded cell, the waves of the grid in turn; the grid covers every cell a cloud of nfin points can crowd)
    const uint32_t big_blocks = std::max(1u, std::min(div_up(nfin / (kCcSmall + 1u) + 1u, kBlock / 64), 2048u));
    if (!tunables().cc_apart || !with_nbr) {
        // four dependent launches: scatter + occupancy words side by side (both need the scan only), then the in-cell order of the
        // small and of the crowded cells side by side
        if (with_nbr)
            k_cc_scatter_nbr<<<nbf + scatter_blocks, kBlock, 0, st>>>(d_pts, stride, (uint32_t)n, g, rank, table, ctx->d_arrived.as<float4>(), nbf, cellslot, stats, occ);
        else
            k_cc_scatter<<<scatter_blocks, kBlock, 0, st>>>(d_pts, stride, (uint32_t)n, g, rank, table, ctx->d_arrived.as<float4>());
        RSREG_HIP(ctx, hipGetLastError());
        k_cc_small_big<<<big_blocks + nbf, kBlock, 0, st>>>(ctx->d_arrived.as<float4>(), g, table, big, ctx->d_tgt_sorted.as<float4>(), ctx->d_pos_of.as<uint32_t>(), stats,
                                                            big_blocks);
        RSREG_HIP(ctx, hipGetLastError());
        return RSREG_OK;
    }
    k_cc_scatter<<<scatter_blocks, kBlock, 0, st>>>(d_pts, stride, (uint32_t)n, g, rank, table, ctx->d_arrived.as<float4>());
    RSREG_HIP(ctx, hipGetLastError());
    k_cc_small<<<nbf, kBlock, 0, st>>>(ctx->d_arrived.as<float4>(), g, table, ctx->d_tgt_sorted.as<float4>(), ctx->d_pos_of.as<uint32_t>(), stats);
    RSREG_HIP(ctx, hipGetLastError());
    k_cc_big<<<big_blocks, kBlock, 0, st>>>(ctx->d_arrived.as<float4>(), g, table, big, ctx->d_tgt_sorted.as<float4>(), ctx->d_pos_of.as<uint32_t>(), stats);
    RSREG_HIP(ctx, hipGetLastError());
    // occupancy word of every cell's 27-cell neighbourhood: a query never opens an empty cell
    k_dense_nbr<<<nbf, kBlock, 0, st>>>(cellslot, stats, g.sx, g.sxy, occ);
    RSREG_HIP(ctx, hipGetLastError());
    return RSREG_OK;
}

// The counts of a counting build that was queued and not waited for (build_dense): taken over from their pinned words once the
// stream has got there.  wait: wait for the stream first (somebody asks for the counts); false: the caller has just done so.
int target_counts(rsreg_ctx *ctx, bool wait)
{
    if (!ctx->counts_pending) return RSREG_OK;
    if (wait) RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->counts_pending = false;
    ctx->cnt_dirty = false;   // (k_cc_scan has put the counts back to zero)
    const uint32_t *h_counts = ctx->h_smisc.as<uint32_t>() + 40;
    GridParams &gp = ctx->grid;
    gp.n_cells = h_counts[0];
    gp.n_points = h_counts[1];
    rsreg_grid_info &gi = ctx->grid_info;
    gi.n_unique_points = gp.n_points;
    gi.n_cells = gp.n_cells;
    gi.index_bytes = (uint64_t)(gp.n_points + 4) * sizeof(float4) + (uint64_t)(ctx->counts_total + 1) * 2 * sizeof(uint32_t) + (uint64_t)ctx->counts_n * sizeof(uint32_t);
    return RSREG_OK;
}

int build_dense(rsreg_ctx *ctx, const char *d_pts, size_t n, size_t stride, double max_dist, uint32_t nfin, hipEvent_t ev0,
                hipEvent_t ev1)
{
    hipStream_t st = ctx->stream;
    GridParams &gp = ctx->grid;
    gp.dense = 1;
    const size_t total = (size_t)(gp.dims[0] + 2) * (gp.dims[1] + 2) * (gp.dims[2] + 2);
    RSREG_HIP(ctx, ctx->h_smisc.reserve(64 * sizeof(uint32_t)));
    RSREG_HIP(ctx, ctx->d_vals.reserve(n * 4));
    RSREG_HIP(ctx, ctx->d_vals_alt.reserve(n * 4));
    RSREG_HIP(ctx, ctx->d_flags.reserve(n * 8));
    RSREG_HIP(ctx, ctx->d_scan.reserve(n * 8));
    RSREG_HIP(ctx, ctx->d_cellpos.reserve(((size_t)nfin + 2) * 4));
    RSREG_HIP(ctx, ctx->d_brick.reserve(((size_t)nfin + 2) * 4));
    RSREG_HIP(ctx, ctx->d_dense.reserve((total + 2) * 4 * 2));   // cell starts, then the neighbourhood occupancy words
    RSREG_HIP(ctx, ctx->d_tgt_sorted.reserve(((size_t)nfin + 8) * sizeof(float4)));
    RSREG_HIP(ctx, ctx->d_pos_of.reserve((n + 1) * 4));
    int id_bits = 1;
    while ((1ull << id_bits) <= total) ++id_bits;   // all-ones (non-finite) stays above every valid id
    const bool wide_keys = tunables().keys64, full_table = tunables().full_table;
    const bool narrow = !wide_keys && id_bits <= 26;   // at least 6 bits of x order inside a cell
    gp.xbits = narrow ? std::min(16, 32 - id_bits) : 16;
    // the searches of gates up to four cells go through the occupancy words only (icp_dense.hpp: dense_far_blocks);
    // the row search of wider or unbounded gates reads table entries of empty cells too and needs all of them
    gp.table_sparse = (gp.max_ring <= 4 && !full_table && !tunables().far_rows) ? 1 : 0;
    const uint32_t *h_counts = ctx->h_smisc.as<uint32_t>() + 40;
    const bool counted = count_sort_pays(n, total);
    if (counted) {
        gp.xbits = (int)kCcXBits;   // what the crowded cells are ordered by (g.x_slack follows)
        gp.table_sparse = 0;        // every table entry is written
        // a small source already loaded (the reference sets the source first: incremental_icp.hpp:57-58) and a gate inside
        // ring 1: no occupancy words -- every launch over this index then takes the instantiation that reads them off the table
        const bool small_source = ctx->have_source && ctx->n_source > 0 && ctx->n_source <= kSmallSourceForTable;
        gp.have_nbr = !(small_source && gp.max_ring <= 1 && tunables().nbr_from_table) ? 1 : 0;
        int rc = build_dense_counted(ctx, d_pts, n, stride, max_dist, nfin, gp.have_nbr != 0);
        if (rc) return rc;
        if (ctx->profiling) (void)hipEventRecord(ev1, st);
        // the build is queued, not waited for: the searches bound the record array by the number of finite points (the sorted
        // array holds at most that many, and the four far-away records behind its last one are inside the buffer either
        // way), and the two counts are taken over when somebody asks for them or has waited for the stream (target_counts)
        gp.n_cells = 0;
        gp.n_points = nfin;
        ctx->counts_pending = true;
        ctx->counts_total = total;
        ctx->counts_n = n;
        // ... unless the build's own time is wanted (profiling), or a source load is on its way on the context's other thread:
        // the caller's thread then has a wait in front of it anyway (join_source), and queueing the alignment's launches now
        // only gets in the way of the thread that is queueing the source's (measured: + 7 us per pair at 10^6 points)
        if (ctx->profiling || ctx->src_on_worker) {
            int rcc = target_counts(ctx, true);
            if (rcc) return rcc;
        }
    } else {
        gp.have_nbr = 1;
        int rc = narrow ? build_dense_keyed<uint32_t>(ctx, d_pts, n, stride, max_dist, nfin, id_bits)
                        : build_dense_keyed<unsigned long long>(ctx, d_pts, n, stride, max_dist, nfin, id_bits);
        if (rc) return rc;
        if (ctx->profiling) (void)hipEventRecord(ev1, st);
        RSREG_HIP(ctx, hipStreamSynchronize(st));   // (the scatter kernel left the two counts in pinned words of their own)
        gp.n_cells = h_counts[0];
        gp.n_points = h_counts[2];
    }
    gp.n_bricks = 0;
    rsreg_grid_info &gi = ctx->grid_info;
    for (int k = 0; k < 3; ++k) { gi.origin[k] = gp.origin[k]; gi.dims[k] = gp.dims[k]; }
    gi.cell_size = gp.cell;
    gi.n_unique_points = gp.n_points;   // (a counting build not yet waited for: the bound; target_counts puts the counts in)
    gi.n_cells = gp.n_cells;
    gi.max_points_per_cell = 0;   // dense mode: computed on demand by rsreg_icp_grid_info
    gi.index_kind = 1;
    gi.index_bytes = (uint64_t)(gp.n_points + 4) * sizeof(float4) + (uint64_t)(total + 1) * 2 * sizeof(uint32_t) + (uint64_t)n * sizeof(uint32_t);
    if (ctx->profiling) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) gi.ms_build = ms;
    }
    ctx->have_target = true;
    return RSREG_OK;
}

// Builds the grid from records already in HBM (d_pts/stride); keeps no pointer to them.
int build_grid(rsreg_ctx *ctx, const char *d_pts, size_t n, size_t stride, double max_dist, double refine = 1.0)
{
    hipStream_t st = ctx->stream;
    ctx->have_target = false;
    ctx->counts_pending = false;   // (of a build nobody asked the counts of: this one's take their place; cnt_dirty stays as it is)
    ctx->tgt_cloud_id = 0;
    ctx->n_target_raw = n;
    std::memset(&ctx->grid_info, 0, sizeof(ctx->grid_info));
    std::memset(&ctx->grid, 0, sizeof(ctx->grid));
    ctx->gate_built_for = max_dist;
    if (n > 0xfffffff0ull) return fail(ctx, RSREG_ERR_INVALID_ARG, "target too large");

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->profiling) {
        ctx->ev_used = 0;
        ev0 = take_event(ctx);
        ev1 = take_event(ctx);
        (void)hipEventRecord(ev0, st);
    }
    float mn[3], mx[3];
    uint32_t nfin = 0;
    const rsreg::CloudBox known = ctx->next_tgt_box;   // (of the cloud handle this target comes from, if it has been measured before)
    ctx->next_tgt_box.valid = false;
    ctx->last_tgt_box.valid = false;
    const bool no_box_cache = !tunables().box_cache;
    if (known.valid && !no_box_cache && n > 0) {
        RSREG_HIP(ctx, ctx->d_misc.reserve(64 * sizeof(uint32_t)));
        RSREG_HIP(ctx, ctx->h_sums.reserve(64 * 8));
        RSREG_HIP(ctx, ctx->d_comm.reserve(1024 * 8 * sizeof(uint32_t) + 64 * sizeof(double)));
        for (int k = 0; k < 3; ++k) { mn[k] = known.mn[k]; mx[k] = known.mx[k]; }
        nfin = known.nfin;
    } else {
        int rc = device_bbox(ctx, d_pts, n, stride, mn, mx, &nfin);
        if (rc) return rc;
    }
    for (int k = 0; k < 3; ++k) { ctx->last_tgt_box.mn[k] = mn[k]; ctx->last_tgt_box.mx[k] = mx[k]; }
    ctx->last_tgt_box.nfin = nfin;
    ctx->last_tgt_box.valid = n > 0;
    ctx->last_tgt_box.exact = !(known.valid && !no_box_cache && n > 0) || known.exact;   // (measured just now, or what the handle knew)
    uint32_t *d_misc = ctx->d_misc.as<uint32_t>();
    uint32_t *h_misc = ctx->h_sums.as<uint32_t>();
    const size_t misc_bytes = 16 * sizeof(uint32_t);
    ctx->grid_info.n_target_points = nfin;
    if (nfin == 0) {
        ctx->have_target = true;  // an empty index: every search comes back empty
        ctx->grid.cell = ctx->grid.inv_cell = 1.0f;
        RSREG_HIP(ctx, ctx->d_table.reserve(sizeof(BrickEntry)));
        RSREG_HIP(ctx, hipMemsetAsync(ctx->d_table.ptr, 0xff, sizeof(BrickEntry), st));
        RSREG_HIP(ctx, ctx->d_tgt_sorted.reserve(sizeof(float4)));
        RSREG_HIP(ctx, ctx->d_cellpos.reserve(16));
        return RSREG_OK;
    }

    // ---- cell size: a whole fraction of the gate, no larger than the cap; an unbounded
    // gate searches outward until the grid is exhausted.  `refine` (second pass only) shrinks
    // the cells of a dense cloud towards ~8 points per occupied cell.
    // the cap is tuned on 10^6-point D435i-like frames; sparser clouds (lower resolution of the same
    // scene: sample spacing ~ n^-1/2) do best with somewhat larger cells (swept at 50 k and 300 k points)
    const double cap = cell_cap_from_env() * std::min(3.0, std::max(1.0, std::pow(1.0e6 / std::max<double>(nfin, 1.0e4), 0.28)));
    double extent = 0;
    for (int k = 0; k < 3; ++k) extent = std::max(extent, (double)mx[k] - (double)mn[k]);
    double cell;
    const bool bounded = std::isfinite(max_dist) && max_dist > 0 && max_dist < 0.25 * extent + 1e-3;
    if (bounded) {
        const double padded = max_dist * 1.04;
        int parts = std::max(1, (int)std::ceil(padded / (cap * refine)));
        parts = std::min(parts, 8);   // the far phase walks (2*parts/4+3)^3 bricks for an unmatched query
        cell = padded / parts;
        // a gate below the cap (the reference's 1 cm on sparse edge clouds): cells as large as the cap still need one
        // ring only, and the table (cleared and scanned on every build) shrinks with the cube of the cell
        const bool wide = tunables().wide_cells;
        if (wide && parts == 1 && cap * refine > padded) cell = cap * refine;
    } else {
        cell = cap * refine;
    }
    cell = std::max(cell, extent / 60000.0);  // 16 bits per axis in the cell key
    cell = std::max(cell, 1e-6);
    GridParams &gp = ctx->grid;
    gp.cell = (float)cell;
    gp.inv_cell = 1.0f / gp.cell;
    for (int k = 0; k < 3; ++k) {
        gp.origin[k] = mn[k];
        gp.dims[k] = host_cell_coord(mx[k], gp.origin[k], gp.inv_cell) + 2;
    }
    const int max_dim = std::max(gp.dims[0], std::max(gp.dims[1], gp.dims[2]));
    int max_ring = max_dim + 1;
    if (std::isfinite(max_dist) && max_dist >= 0) {
        const double rings = std::ceil(max_dist / (double)gp.cell + kCellMargin);   // smallest R with (R - margin) * cell >= gate
        if (rings < (double)(max_dim + 1)) max_ring = std::max(1, (int)rings);
    }
    gp.max_ring = max_ring;
    {
        const long long padded = (long long)(gp.dims[0] + 2) * (gp.dims[1] + 2) * (gp.dims[2] + 2);
        // (the dense search addresses points by 32-bit byte offsets: 16 B x 2^28)
        if (padded <= dense_cell_budget() && (unsigned long long)nfin + 4ull < (1ull << 28)) return build_dense(ctx, d_pts, n, stride, max_dist, nfin, ev0, ev1);
    }

    // ---- sort by (brick, cell in brick, xyz hash)
    // (the brick path counts with atomics in the words k_bbox_final clears; the dense table's kernels only store there)
    if (known.valid && !no_box_cache && n > 0) RSREG_HIP(ctx, hipMemsetAsync(ctx->d_misc.ptr, 0, 16 * sizeof(uint32_t), st));
    RSREG_HIP(ctx, ctx->d_keys.reserve(n * 8));
    RSREG_HIP(ctx, ctx->d_keys_alt.reserve(n * 8));
    RSREG_HIP(ctx, ctx->d_vals.reserve(n * 4));
    RSREG_HIP(ctx, ctx->d_vals_alt.reserve(n * 4));
    RSREG_HIP(ctx, ctx->d_flags.reserve(n * 12));
    RSREG_HIP(ctx, ctx->d_scan.reserve(n * 12));
    RSREG_HIP(ctx, ctx->d_cellpos.reserve(((size_t)nfin + 2) * 4));
    RSREG_HIP(ctx, ctx->d_brick.reserve(((size_t)nfin + 2) * 20));
    RSREG_HIP(ctx, ctx->d_tgt_sorted.reserve(((size_t)nfin + 1) * sizeof(float4)));
    GridDev g = grid_dev(ctx, max_dist);
    const uint32_t nb = div_up((uint32_t)n, kBlock);
    auto *keys = ctx->d_keys.as<unsigned long long>();
    auto *keys2 = ctx->d_keys_alt.as<unsigned long long>();
    auto *vals = ctx->d_vals.as<uint32_t>();
    auto *vals2 = ctx->d_vals_alt.as<uint32_t>();
    k_cell_keys<<<nb, kBlock, 0, st>>>(d_pts, stride, (uint32_t)n, g, keys, vals);
    RSREG_HIP(ctx, hipGetLastError());
    // the library's own radix sort over all 64 key bits (osort.hpp), its state cleared by a memset in front; then three scans (oscan.hpp)
    uint32_t *keep = ctx->d_flags.as<uint32_t>(), *cstart = keep + n, *bstart = cstart + n;
    uint32_t *pos = ctx->d_scan.as<uint32_t>(), *cid = pos + n, *bid = cid + n;
    const size_t sort_bytes = (size_t)osort_plan<unsigned long long>(n, 0, 64).words * 4, scan_bytes = oscan_scratch_bytes<uint32_t>(nfin);
    RSREG_HIP(ctx, ctx->d_tmp.reserve(std::max(sort_bytes, scan_bytes) + 256));
    {
        bool in_first = false;
        RSREG_HIP(ctx, osort_pairs_cleared<unsigned long long>(ctx->d_tmp.as<uint32_t>(), keys, keys2, vals, vals2, n, 0, 64, st, &in_first));
        if (in_first) {   // (keys2 / vals2 below: the sorted pairs, wherever the passes have left them)
            std::swap(keys, keys2);
            std::swap(vals, vals2);
        }
    }
    const uint32_t nbf = div_up(nfin, kBlock);
    k_flag_runs<<<nbf, kBlock, 0, st>>>(keys2, vals2, d_pts, stride, nfin, keep, cstart, bstart);
    RSREG_HIP(ctx, hipGetLastError());
    RSREG_HIP(ctx, (oscan<uint32_t>(keep, pos, (size_t)nfin, 0u, ctx->d_tmp.ptr, st)));
    RSREG_HIP(ctx, (oscan<uint32_t>(cstart, cid, (size_t)nfin, 0u, ctx->d_tmp.ptr, st)));
    RSREG_HIP(ctx, (oscan<uint32_t>(bstart, bid, (size_t)nfin, 0u, ctx->d_tmp.ptr, st)));
    // brick staging arrays: key[nfin+1] u64 | mask[nfin+1] u64 | base[nfin+1] u32
    auto *brickkey = ctx->d_brick.as<unsigned long long>();
    auto *brickmask = brickkey + (nfin + 1);
    auto *brickbase = reinterpret_cast<uint32_t *>(brickmask + (nfin + 1));
    RSREG_HIP(ctx, hipMemsetAsync(brickmask, 0, ((size_t)nfin + 1) * 8, st));
    k_scatter_sorted<<<nbf, kBlock, 0, st>>>(keys2, vals2, d_pts, stride, nfin, keep, cstart, bstart, pos, cid, bid,
                                             ctx->d_tgt_sorted.as<float4>(), ctx->d_cellpos.as<uint32_t>(), brickkey, brickmask,
                                             brickbase, d_misc + 8);
    RSREG_HIP(ctx, hipGetLastError());
    RSREG_HIP(ctx, hipMemcpyAsync(h_misc, d_misc, misc_bytes, hipMemcpyDeviceToHost, st));
    RSREG_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t n_unique = h_misc[8], n_cells = h_misc[9], n_bricks = h_misc[11];
    gp.n_points = n_unique;
    gp.n_cells = n_cells;
    gp.n_bricks = n_bricks;

    // ---- hash table of occupied bricks, at most half full
    uint32_t slots = 64;
    while (slots < 2 * n_bricks) slots <<= 1;
    gp.table_mask = slots - 1;
    RSREG_HIP(ctx, ctx->d_table.reserve((size_t)slots * sizeof(BrickEntry)));
    RSREG_HIP(ctx, hipMemsetAsync(ctx->d_table.ptr, 0xff, (size_t)slots * sizeof(BrickEntry), st));
    k_brick_insert<<<div_up(n_bricks, kBlock), kBlock, 0, st>>>(brickkey, brickmask, brickbase, n_bricks,
                                                                ctx->d_table.as<BrickEntry>(), gp.table_mask);
    RSREG_HIP(ctx, hipGetLastError());
    k_max_cell_count<<<div_up(n_cells, kBlock), kBlock, 0, st>>>(ctx->d_cellpos.as<uint32_t>(), n_cells, d_misc + 10);
    RSREG_HIP(ctx, hipGetLastError());
    if (ctx->profiling) (void)hipEventRecord(ev1, st);
    RSREG_HIP(ctx, hipMemcpyAsync(h_misc, d_misc, misc_bytes, hipMemcpyDeviceToHost, st));
    RSREG_HIP(ctx, hipStreamSynchronize(st));

    rsreg_grid_info &gi = ctx->grid_info;
    for (int k = 0; k < 3; ++k) { gi.origin[k] = gp.origin[k]; gi.dims[k] = gp.dims[k]; }
    gi.cell_size = gp.cell;
    gi.n_unique_points = n_unique;
    gi.n_cells = n_cells;
    gi.max_points_per_cell = h_misc[10];
    gi.index_kind = 0;
    gi.index_bytes = (uint64_t)n_unique * sizeof(float4) + (uint64_t)slots * sizeof(BrickEntry) + ((uint64_t)n_cells + 1) * 4;
    if (ctx->profiling) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) gi.ms_build = ms;
    }
    ctx->have_target = true;
    // dense cloud: rebuild once with smaller cells (fewer candidates per query)
    const bool adaptive = tunables().adaptive_cell;
    if (adaptive && refine == 1.0 && n_cells > 0) {
        const double per_cell = (double)n_unique / (double)n_cells;
        if (per_cell > 14.0) {
            const double ms_first = gi.ms_build;
            const double r = std::max(0.2, std::sqrt(8.0 / per_cell));
            const float cell_before = gp.cell;
            ctx->next_tgt_box = ctx->last_tgt_box;   // (the same records: no second measurement)
            int rc2 = build_grid(ctx, d_pts, n, stride, max_dist, r);
            if (rc2) return rc2;
            ctx->grid_info.ms_build += ms_first;
            (void)cell_before;
        }
    }
    return RSREG_OK;
}

// No index at all (icp_kernels.hpp: k_scan_nn): the target is only put into (x, y, index, z) records in the caller's
// order.  Chosen by rsreg_icp_set_target_cloud when the source that is already loaded has a handful of points;
// rsreg_icp_begin builds the real index after all if the alignment turns out to need it (`d_pts` must stay valid
// until then: the contract of cloud handles).
int scan_target(rsreg_ctx *ctx, const char *d_pts, size_t n, size_t stride, double max_dist)
{
    ctx->have_target = false;
    ctx->tgt_cloud_id = 0;
    ctx->n_target_raw = n;
    std::memset(&ctx->grid_info, 0, sizeof(ctx->grid_info));
    std::memset(&ctx->grid, 0, sizeof(ctx->grid));
    ctx->gate_built_for = max_dist;
    if (n > 0x7ffffff0ull) return fail(ctx, RSREG_ERR_INVALID_ARG, "target too large");
    RSREG_HIP(ctx, ctx->d_tgt_sorted.reserve((n + 8) * sizeof(float4)));
    RSREG_HIP(ctx, ctx->d_scan_keys.reserve(kScanMaxSource * 8));
    k_scan_pack<<<div_up((uint32_t)n, kBlock), kBlock, 0, ctx->stream>>>(d_pts, stride, (uint32_t)n, ctx->d_tgt_sorted.as<float4>());
    RSREG_HIP(ctx, hipGetLastError());
    ctx->grid.dense = 2;
    ctx->grid.n_points = (uint32_t)n;
    ctx->scan_raw = d_pts;
    ctx->scan_stride = stride;
    ctx->grid_info.index_kind = 2;
    ctx->grid_info.n_unique_points = n;
    ctx->grid_info.index_bytes = (uint64_t)n * sizeof(float4);
    ctx->have_target = true;
    return RSREG_OK;
}

// The source load runs on its own stream; whoever needs its result (the number of distinct points, the source
// buffers) joins it first.
int join_source(rsreg_ctx *ctx)
{
    if (!ctx->src_pending) return RSREG_OK;
    ctx->src_pending = false;
    ctx->src_on_worker = false;
    const int rcw = ctx->source_enqueued();   // (the load's host side may still be running on the context's worker thread)
    if (rcw) {
        ctx->have_source = false;
        return rcw;
    }
    if (ctx->src_plain) {
        // one launch on the main stream, the caller's order, no merging: the number of queries IS the number of records, and
        // whatever reads the working copy is queued behind the launch on the same stream -- no wait on the caller's thread
        ctx->n_work = (uint32_t)ctx->n_source;
        return RSREG_OK;
    }
    RSREG_HIP(ctx, hipEventSynchronize(ctx->ev_src_done));
    ctx->n_work = ctx->h_smisc.as<uint32_t>()[32];
    return RSREG_OK;
}

// Sources of at most this many points are searched in the caller's order (k_source_plain).  RSREG_SORT_SMALL=1: never.
constexpr size_t kPlainSourceMax = 65536;
bool source_is_small(size_t n)
{
    return n <= tunables().plain_source_max && !tunables().sort_small;
}

// The part of a source load that queues work on stream_src (after one round trip for the bounding box); runs on the
// context's worker thread (rsreg_ctx.hpp: SourceWorker) or, with RSREG_NO_WORKER=1, on the caller's.
// `known`: the box of the cloud handle this source comes from, if the handle has one (by value: ctx->next_src_box belongs
// to the caller's thread alone)
int load_source_queue(rsreg_ctx *ctx, const char *d_raw, size_t n, size_t stride, const rsreg::CloudBox known)
{
    {
        RSREG_HIP(ctx, hipSetDevice(ctx->device));   // (this may be the context's worker thread)
        uint32_t *d_misc = ctx->d_smisc.as<uint32_t>();
        uint32_t *h_misc = ctx->h_smisc.as<uint32_t>();
        if (source_is_small(n)) {
            // one launch, the caller's order -- on the MAIN stream: the streams of a process share a few hardware queues,
            // and a kernel on the source stream can find itself behind a 0.4 ms voxel filter of a side stream
            hipStream_t st = ctx->stream;
            // (its box and finite count land in pinned words 48 .. 55 behind the stamp `seq`; the ticket word lives in a buffer of its
            // own that is zero between launches)
            const bool box_too = tunables().box_cache;
            if (box_too && !ctx->d_plain_ticket.ptr) {
                RSREG_HIP(ctx, ctx->d_plain_ticket.reserve(64));
                RSREG_HIP(ctx, hipMemsetAsync(ctx->d_plain_ticket.ptr, 0, 64, st));
            }
            ctx->plain_box_seq = box_too ? ++ctx->plain_box_counter : 0u;
            k_source_plain<<<div_up((uint32_t)n, kBlock), kBlock, 0, st>>>(d_raw, stride, (uint32_t)n, ctx->d_src_all.as<float4>(), ctx->d_src.as<float4>(),
                                                                           ctx->d_perm.as<uint32_t>(), ctx->d_uniq_of.as<uint32_t>(),
                                                                           ctx->d_first.as<uint32_t>(), d_misc + 12, h_misc + 32,
                                                                           box_too ? d_misc + 64 : nullptr, ctx->d_plain_ticket.as<uint32_t>(), h_misc + 48,
                                                                           ctx->plain_box_seq);
            RSREG_HIP(ctx, hipGetLastError());
            RSREG_HIP(ctx, hipEventRecord(ctx->ev_src_done, st));
            return RSREG_OK;
        }
        hipStream_t st = ctx->stream_src;
        RSREG_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_main, 0));
        float mn[3], mx[3];
        uint32_t nfin = 0;
        const bool no_box_cache = !tunables().box_cache;
        if (known.valid && !no_box_cache) {   // (nothing of the load counts in the words k_bbox_final clears: the kernels below only store there)
            for (int k = 0; k < 3; ++k) { mn[k] = known.mn[k]; mx[k] = known.mx[k]; }
            nfin = known.nfin;
        } else {
            int rc = device_bbox_on(ctx, st, d_misc, h_misc, d_misc + 64, d_raw, n, stride, mn, mx, &nfin);
            if (rc) return rc;
        }
        for (int k = 0; k < 3; ++k) { ctx->last_src_box.mn[k] = mn[k]; ctx->last_src_box.mx[k] = mx[k]; }
        ctx->last_src_box.nfin = nfin;
        ctx->last_src_box.valid = true;
        if (nfin == 0) { mn[0] = mn[1] = mn[2] = 0; mx[0] = mx[1] = mx[2] = 0; }
        double extent = 0;
        for (int k = 0; k < 3; ++k) extent = std::max(extent, (double)mx[k] - (double)mn[k]);
        // fine Morton resolution (a few mm): consecutive points then form compact blobs
        float cell = (float)std::max(cell_cap_from_env() / 8.0, extent / 60000.0);
        // a 32-bit key (31 bits of Morton code + the invalid bit) halves the bytes the radix sort moves and saves it a pass
        // or two: the cell grows (by at most 2x: the search time moves by +- 1.5 % between 1.5 and 3 mm)
        // until the three axes need 31 bits together
        const bool wide_keys = tunables().keys64;
        auto axis_bits_of = [](double ext, double c) {
            int b = 1;
            while (b < 16 && (double)(1u << b) <= ext / c + 2.0) ++b;
            return b;
        };
        MortonBits mb{0, 0, 0};
        bool narrow = false;
        if (!wide_keys) {
            const int morton_bits = tunables().morton_bits;   // (23 + the invalid bit: three digit passes)
            for (double c = cell; c <= (morton_bits < 31 ? 64.0 : 2.0) * (double)cell + 1e-12; c *= 1.05) {
                mb = MortonBits{axis_bits_of((double)mx[0] - (double)mn[0], c), axis_bits_of((double)mx[1] - (double)mn[1], c),
                                axis_bits_of((double)mx[2] - (double)mn[2], c)};
                if (mb.x + mb.y + mb.z <= morton_bits && std::max(mb.x, std::max(mb.y, mb.z)) <= 12) {
                    narrow = true;
                    cell = (float)c;
                    break;
                }
            }
        }
        RSREG_HIP(ctx, ctx->d_skeys.reserve(n * 8));
        RSREG_HIP(ctx, ctx->d_skeys_alt.reserve(n * 8));
        RSREG_HIP(ctx, ctx->d_svals.reserve(n * 4));
        RSREG_HIP(ctx, ctx->d_sflags.reserve(n * 4));
        RSREG_HIP(ctx, ctx->d_sscan.reserve(n * 4));
        auto *vals = ctx->d_svals.as<uint32_t>();
        uint32_t *perm = ctx->d_perm.as<uint32_t>();
        uint32_t *keep = ctx->d_sflags.as<uint32_t>(), *pos = ctx->d_sscan.as<uint32_t>();
        const uint32_t nb = div_up((uint32_t)n, kBlock);
        size_t sort_bytes = 0;
        // (the sort's state at the head of d_stmp, the scan's sums behind it: the scan starts when the sort is done, but its
        // launches are queued before that)
        const size_t scan_bytes = oscan_scratch_bytes<uint32_t>(n);
        if (narrow) {
            auto *keys = ctx->d_skeys.as<uint32_t>();
            auto *keys2 = ctx->d_skeys_alt.as<uint32_t>();
            const unsigned sort_bits = (unsigned)(mb.x + mb.y + mb.z) + 1u;
            constexpr bool own_sort = true;   // (osort.hpp: the sort's state is cleared by the keys kernel, no memsets)
            const Radix32Plan plan = radix32_plan(n, 0, sort_bits);
            sort_bytes = (size_t)plan.words * 4;
            RSREG_HIP(ctx, ctx->d_stmp.reserve(std::max(sort_bytes, scan_bytes) + 256));
            // (an even number of passes ends in the pair it started from: the keys are then written where the result belongs)
            const bool start_in_out = plan.ends_in_first;
            uint32_t *keys_a = start_in_out ? keys2 : keys, *vals_a = start_in_out ? perm : vals;
            uint32_t *keys_b = start_in_out ? keys : keys2, *vals_b = start_in_out ? vals : perm;
            // the library's own sort over more than one workgroup's worth of pairs: its digit histograms are counted by the keys
            // kernel (two sets of counts in turn: the kernel clears the one the NEXT load will use, so no memset is queued;
            // a new buffer, or a load that did not get to its end, starts from one)
            const bool hist_on_the_way = own_sort && n > kOsTile;
            uint32_t *hist_now = nullptr;
            if (hist_on_the_way) {
                constexpr size_t set = OsKey<uint32_t>::max_passes * kOsDigits;
                const bool fresh = !ctx->d_shist.ptr;
                RSREG_HIP(ctx, ctx->d_shist.reserve(2 * set * 4));
                if (fresh || ctx->shist_dirty) RSREG_HIP(ctx, hipMemsetAsync(ctx->d_shist.ptr, 0, 2 * set * 4, st));
                ctx->shist_dirty = true;
                hist_now = ctx->d_shist.as<uint32_t>() + (ctx->shist_flip ? set : 0);
                uint32_t *hist_next = ctx->d_shist.as<uint32_t>() + (ctx->shist_flip ? 0 : set);
                ctx->shist_flip = !ctx->shist_flip;
                k_source_keys_hist<<<div_up((uint32_t)n, kOsHistBlock * kSkItems), kOsHistBlock, 0, st>>>(
                    d_raw, stride, (uint32_t)n, mn[0], mn[1], mn[2], 1.0f / cell, 1u << (sort_bits - 1u), mb, keys_a, vals_a, ctx->d_stmp.as<uint32_t>(), plan.words,
                    sort_bits, plan.passes, hist_now, hist_next);
            } else {
                k_source_keys<uint32_t><<<nb, kBlock, 0, st>>>(d_raw, stride, (uint32_t)n, mn[0], mn[1], mn[2], 1.0f / cell, 1u << (sort_bits - 1u), mb, keys_a, vals_a,
                                                               own_sort ? ctx->d_stmp.as<uint32_t>() : nullptr, own_sort ? plan.words : 0u);
            }
            RSREG_HIP(ctx, hipGetLastError());
            {
                bool in_first = false;
                RSREG_HIP(ctx, radix32_sort_pairs<uint32_t>(plan, ctx->d_stmp.as<uint32_t>(), keys_a, keys_b, vals_a, vals_b, n, 0, sort_bits, st, &in_first, hist_now));
                if (in_first != start_in_out) return fail(ctx, RSREG_ERR_STATE, "osort: the sorted pairs are not where they belong");
            }
            k_gather_source<uint32_t><<<nb, kBlock, 0, st>>>(d_raw, stride, (uint32_t)n, perm, ctx->d_src_all.as<float4>(), nullptr, keys2, keep);
            RSREG_HIP(ctx, hipGetLastError());
        } else {
            auto *keys = ctx->d_skeys.as<unsigned long long>();
            auto *keys2 = ctx->d_skeys_alt.as<unsigned long long>();
            // the sort only has to look at the bits the Morton codes of this extent can set (+ the invalid bit)
            const int axis_bits = axis_bits_of(extent, (double)cell);
            const unsigned sort_bits = 3u * (unsigned)axis_bits + 1u;
            const Radix32Plan plan = radix32_plan<unsigned long long>(n, 0, sort_bits);
            sort_bytes = (size_t)plan.words * 4;
            RSREG_HIP(ctx, ctx->d_stmp.reserve(std::max(sort_bytes, scan_bytes) + 256));
            const bool start_in_out = plan.ends_in_first;   // (as above: the sorted pairs must end in (keys2, perm))
            k_source_keys<unsigned long long><<<nb, kBlock, 0, st>>>(d_raw, stride, (uint32_t)n, mn[0], mn[1], mn[2], 1.0f / cell, 1ull << (3 * axis_bits), mb,
                                                                     start_in_out ? keys2 : keys, start_in_out ? perm : vals, ctx->d_stmp.as<uint32_t>(), plan.words);
            RSREG_HIP(ctx, hipGetLastError());
            {
                bool in_first = false;
                RSREG_HIP(ctx, radix32_sort_pairs<unsigned long long>(plan, ctx->d_stmp.as<uint32_t>(), start_in_out ? keys2 : keys, start_in_out ? keys : keys2,
                                                                      start_in_out ? perm : vals, start_in_out ? vals : perm, n, 0, sort_bits, st, &in_first));
                if (in_first != start_in_out) return fail(ctx, RSREG_ERR_STATE, "osort: the sorted pairs are not where they belong");
            }
            k_gather_source<unsigned long long><<<nb, kBlock, 0, st>>>(d_raw, stride, (uint32_t)n, perm, ctx->d_src_all.as<float4>(), nullptr, keys2, keep);
            RSREG_HIP(ctx, hipGetLastError());
        }
        RSREG_HIP(ctx, (oscan<uint32_t>(keep, pos, n, 0u, ctx->d_stmp.ptr, st)));
        k_source_unique<<<nb, kBlock, 0, st>>>(ctx->d_src_all.as<float4>(), (uint32_t)n, keep, pos, ctx->d_first.as<uint32_t>(),
                                               ctx->d_uniq_of.as<uint32_t>(), d_misc + 12, h_misc + 32,   // the number of distinct points: read at the join
                                               ctx->d_src.as<float4>());
        RSREG_HIP(ctx, hipGetLastError());
        RSREG_HIP(ctx, hipEventRecord(ctx->ev_src_done, st));
        ctx->shist_dirty = false;   // (everything of this load is queued: the counts' two sets are where the next load expects them)
    }
    return RSREG_OK;
}

// Loads the source: orders it spatially (Morton order of a few-mm grid, so the lanes of a wave
// query neighbouring cells) and merges exact copies of a point into one weighted point (the
// RealSense (0,0,0) pixels are ~11 % of a frame: they are searched once, not 10^5 times).
// d_perm: sorted position -> caller's index; d_uniq_of: sorted position -> distinct point.
// Everything after the bounding box (one host sync) is only queued -- on ctx->stream_src, behind whatever the
// main stream holds so far -- and joined by join_source: when the caller sets the source before the target (the
// reference's order, incremental_icp.hpp:57-58) the load runs beside the target's index build.
int load_source(rsreg_ctx *ctx, const char *d_raw, size_t n, size_t stride)
{
    if (n > 0xfffffff0ull) return fail(ctx, RSREG_ERR_INVALID_ARG, "source too large");
    int rcj = join_source(ctx);   // (a load still in flight owns the buffers this one is about to fill)
    if (rcj) return rcj;
    ctx->last_src_box.valid = false;   // (of the load before this one: cloud.hip has taken it over by now)
    ctx->plain_box_seq = 0;
    // the box of the cloud this source comes from, if its handle knows it: taken over -- or dropped -- here, on the caller's
    // thread, after the load before this one has been joined; the worker's job gets a copy
    const rsreg::CloudBox known = ctx->next_src_box;
    ctx->next_src_box.valid = false;
    ctx->prep_join();
    if (!ctx->stream_src) {
        RSREG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_src, hipStreamNonBlocking));
        RSREG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_src_done, hipEventDisableTiming));
        RSREG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
    }
    RSREG_HIP(ctx, ctx->d_src_all.reserve((n + 1) * sizeof(float4)));
    RSREG_HIP(ctx, ctx->d_src.reserve((n + 1) * sizeof(float4)));
    RSREG_HIP(ctx, ctx->d_cur.reserve((n + 1) * sizeof(float4)));
    RSREG_HIP(ctx, ctx->d_corr_pos.reserve((n + 1) * 4));
    RSREG_HIP(ctx, ctx->d_corr_d2.reserve((n + 1) * 4));
    RSREG_HIP(ctx, ctx->d_seed.reserve((n + 1) * 4));
    RSREG_HIP(ctx, ctx->d_perm.reserve((n + 1) * 4));
    RSREG_HIP(ctx, ctx->d_uniq_of.reserve((n + 1) * 4));
    RSREG_HIP(ctx, ctx->d_first.reserve((n + 2) * 4));
    RSREG_HIP(ctx, ctx->d_partials.reserve((size_t)reduce_blocks(n) * RSREG_NUM_SUMS * 8));
    {
        // the tile schedule's arrays (items | wave costs | done counters | ...) lie at offsets that depend on the buffer's capacity
        // only, so that a schedule can be carried over to the next alignment (launch_fused); a new buffer starts from zero
        // (the done counters go back to zero by themselves) and without a schedule
        // (a larger buffer can come back at the address of the one just freed: the capacity says whether it is a new one)
        void *before = ctx->d_sched.ptr;
        const size_t cap_before = ctx->d_sched.cap;
        RSREG_HIP(ctx, ctx->d_sched.reserve((size_t)reduce_blocks(n) * 15 * 4 + 256));
        if (ctx->d_sched.ptr != before || ctx->d_sched.cap != cap_before) {
            RSREG_HIP(ctx, hipMemsetAsync(ctx->d_sched.ptr, 0, ctx->d_sched.cap, ctx->stream));
            ctx->sched_cap_tiles = (uint32_t)((ctx->d_sched.cap - 256) / (15 * 4));
            ctx->sched_keep_items = 0;
            ctx->sched_first_items = 0;
        }
    }
    RSREG_HIP(ctx, ctx->d_sums.reserve(64 * 8));
    RSREG_HIP(ctx, ctx->h_sums.reserve(64 * 8));
    RSREG_HIP(ctx, ctx->d_smisc.reserve((64 + 1024 * 8) * sizeof(uint32_t)));
    RSREG_HIP(ctx, ctx->h_smisc.reserve(64 * sizeof(uint32_t)));
    ctx->n_source = n;
    ctx->n_work = 0;
    ctx->src_cloud = nullptr;
    ctx->have_source = false;
    ctx->icp.active = 0;
    if (n) {
        // the raw cloud may have been produced (uploaded, filtered, transformed) on the main stream just now
        RSREG_HIP(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
        const bool no_worker = !tunables().worker;
        ctx->src_on_worker = false;
        ctx->src_plain = source_is_small(n);
        if (no_worker || source_is_small(n)) {   // (a small source is one launch: not worth a hand-over)
            int rc = load_source_queue(ctx, d_raw, n, stride, known);
            if (rc) return rc;
        } else {
            ctx->src_on_worker = true;   // (until it is joined)
            // the rest -- a bounding-box round trip and ~25 launches -- on the context's worker thread: the caller goes on
            // (to the target's index build, in the reference's order of calls) while the source's queue is being filled
            if (!ctx->src_worker) ctx->src_worker = new rsreg::SourceWorker();
            ctx->src_worker->post([ctx, d_raw, n, stride, known] { return load_source_queue(ctx, d_raw, n, stride, known); });
        }
        ctx->src_pending = true;
    }
    ctx->have_source = true;
    return RSREG_OK;
}

// DefaultConvergenceCriteria::hasConverged
bool criteria_has_converged(IcpState &s)
{
    const rsreg_icp_params &p = s.prm;
    if (s.state != RSREG_CONV_NOT_CONVERGED) { s.similar = 0; s.state = RSREG_CONV_NOT_CONVERGED; }
    bool is_similar = false;
    if (s.iterations >= p.max_iterations) { s.state = RSREG_CONV_ITERATIONS; return true; }
    if (p.criteria_mode == RSREG_CRITERIA_FIXED) return false;
    const double rot_thr = p.transformation_rotation_epsilon > 0 ? p.transformation_rotation_epsilon
                                                                  : 1.0 - p.transformation_epsilon;
    const double trans_thr = p.transformation_epsilon;
    const Mat4f &T = s.t_inc;
    const double cos_angle = 0.5 * ((double)T(0, 0) + (double)T(1, 1) + (double)T(2, 2) - 1.0);
    const double tsq = (double)T(0, 3) * T(0, 3) + (double)T(1, 3) * T(1, 3) + (double)T(2, 3) * T(2, 3);
    const int max_similar = 0;  // max_iterations_similar_transforms_
    if (cos_angle >= rot_thr && tsq <= trans_thr) {
        if (s.similar >= max_similar) { s.state = RSREG_CONV_TRANSFORM; return true; }
        is_similar = true;
    }
    if (std::fabs(s.cur_mse - s.prev_mse) < 1e-12) {
        if (s.similar >= max_similar) { s.state = RSREG_CONV_ABS_MSE; return true; }
        is_similar = true;
    }
    if (std::fabs(s.cur_mse - s.prev_mse) / s.prev_mse < p.euclidean_fitness_epsilon) {
        if (s.similar >= max_similar) { s.state = RSREG_CONV_REL_MSE; return true; }
        is_similar = true;
    }
    if (is_similar) ++s.similar; else s.similar = 0;
    s.prev_mse = s.cur_mse;
    return false;
}

int *seed_ptr(rsreg_ctx *ctx)
{
    return !tunables().seed ? nullptr : ctx->d_seed.as<int>();
}

// rsreg_icp_begin leaves "working copy = guess * source, no seeds" pending: the first search launch of the fused
// pipelines over the dense index does it on its way (launch_fused); whatever else touches the working copy first runs
// k_restart_source here.
int ensure_restarted(rsreg_ctx *ctx)
{
    IcpState &s = ctx->icp;
    if (!s.restart_pending) return RSREG_OK;
    s.restart_pending = false;
    const uint32_t n = (uint32_t)ctx->n_work;
    if (!n) return RSREG_OK;
    k_restart_source<<<div_up(n, kBlock), kBlock, 0, ctx->stream>>>(ctx->d_src.as<float4>(), ctx->d_first.as<uint32_t>(), n, to_mat34(s.final_t), s.final_t.is_identity() ? 0 : 1,
                                                                    ctx->d_cur.as<float4>(), ctx->d_seed.as<int>());   // (and: no seeds yet)
    RSREG_HIP(ctx, hipGetLastError());
    return RSREG_OK;
}

// diagnostic: per-wave start/end stamps of the last fused launch, dumped at rsreg_icp_end
unsigned long long *wave_times_ptr(rsreg_ctx *ctx, uint32_t n)
{
#ifndef RSREG_DIAG
    (void)ctx; (void)n;
    return nullptr;
#else
    if (!tunables().wave_times) return nullptr;
    if (ctx->d_brick.reserve(((size_t)n / 64 + 2) * 256 + (size_t)n * 4 + 64) != hipSuccess) return nullptr;   // (a scheduled launch has up to 2x the waves)
    return ctx->d_brick.as<unsigned long long>();
#endif
}

bool filters_on(const rsreg_icp_params &p)
{
    return p.use_reciprocal_correspondences != 0 || (p.trim_overlap_ratio > 0.0 && p.trim_overlap_ratio < 1.0);
}

// reciprocal check of one matched pair: is the source point the nearest source point of its target point?
template <bool kDense>
__global__ __launch_bounds__(kBlock) void k_recip_filter(const float4 *tgt, int *corr_pos, uint32_t *cw, uint32_t n, DenseDev gd, GridDev gh,
                                                         const float4 *child_pts, const uint32_t *perm, const uint32_t *first)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    const int pos = corr_pos[u];
    if (pos < 0) return;
    const float4 t = tgt[pos];
    const Best b = kDense ? nn_query_dense(gd, t.x, t.y, tgt_z(t), -1) : nn_query(gh, t.x, t.y, tgt_z(t), -1);
    // the index over the source keeps the lowest original index of equal points and prefers it among equidistant ones:
    // of the copies of a distinct source point only the first can be its target's nearest source point
    const bool ok = b.pos >= 0 && tgt_idx(child_pts[b.pos]) == perm[first[u]];
    cw[u] = ok ? 1u : 0u;
    if (!ok) corr_pos[u] = -1;
}

// the optional filters PCL's ICP applies between determineCorrespondences and the transformation estimate:
// reciprocal correspondences (CorrespondenceEstimation::determineReciprocalCorrespondences) and the trimmed rejector
int apply_filters(rsreg_ctx *ctx)
{
    IcpState &s = ctx->icp;
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)ctx->n_work, ns = (uint32_t)ctx->n_source;
    if (!n) return RSREG_OK;
    RSREG_HIP(ctx, ctx->d_corr_w.reserve((size_t)n * 4 + 16));
    uint32_t *cw = ctx->d_corr_w.as<uint32_t>();
    int *corr_pos = ctx->d_corr_pos.as<int>();
    const uint32_t nb = div_up(n, kBlock);
    k_corr_weights<<<nb, kBlock, 0, st>>>(corr_pos, ctx->d_cur.as<float4>(), n, cw);
    RSREG_HIP(ctx, hipGetLastError());
    if (s.prm.use_reciprocal_correspondences) {
        // index the CURRENT source (it moves every iteration; PCL rebuilds its reciprocal kd-tree too), in the caller's order
        if (!ctx->recip) {
            int rc = rsreg_ctx_create(ctx->device, ctx->stream, &ctx->recip);
            if (rc) return fail(ctx, rc, "context of the reciprocal index");
        }
        RSREG_HIP(ctx, ctx->d_recip_pts.reserve(((size_t)ns + 1) * sizeof(float4)));
        k_recip_points<<<div_up(ns, kBlock), kBlock, 0, st>>>(ctx->d_cur.as<float4>(), ctx->d_perm.as<uint32_t>(), ctx->d_uniq_of.as<uint32_t>(), ns,
                                                           ctx->d_recip_pts.as<float4>());
        RSREG_HIP(ctx, hipGetLastError());
        rsreg_ctx *c = ctx->recip;
        int rc = build_grid(c, ctx->d_recip_pts.as<char>(), ns, sizeof(float4), s.prm.max_correspondence_distance, 1.0);
        if (rc) {
            std::string why;
            {
                std::lock_guard<std::mutex> lk(c->error_mutex);
                why = c->last_error;
            }
            return fail(ctx, rc, why.c_str());
        }
        if (c->grid.n_points) {   // (a bound while the child's counting build has not been waited for: never 0 for a cloud with a finite point)
            const DenseDev gd = c->grid.dense ? dense_dev(c, s.prm.max_correspondence_distance) : DenseDev{};
            const GridDev gh = grid_dev(c, s.prm.max_correspondence_distance);
            auto kern = c->grid.dense ? k_recip_filter<true> : k_recip_filter<false>;
            kern<<<nb, kBlock, 0, st>>>(ctx->d_tgt_sorted.as<float4>(), corr_pos, cw, n, gd, gh, c->d_tgt_sorted.as<float4>(),
                                        ctx->d_perm.as<uint32_t>(), ctx->d_first.as<uint32_t>());
            RSREG_HIP(ctx, hipGetLastError());
        }
    }
    if (s.prm.trim_overlap_ratio > 0.0 && s.prm.trim_overlap_ratio < 1.0) {
        RSREG_HIP(ctx, ctx->d_keys.reserve((size_t)n * 8));
        RSREG_HIP(ctx, ctx->d_keys_alt.reserve((size_t)n * 8));
        RSREG_HIP(ctx, ctx->d_vals.reserve((size_t)n * 4));
        RSREG_HIP(ctx, ctx->d_vals_alt.reserve((size_t)n * 4));
        uint32_t *keys = ctx->d_keys.as<uint32_t>(), *vals = keys + n, *keys2 = ctx->d_keys_alt.as<uint32_t>(), *order = keys2 + n;
        uint32_t *ws = ctx->d_vals.as<uint32_t>(), *cum = ctx->d_vals_alt.as<uint32_t>();
        // (the distances' sort: the library's own, four digit passes that end in the pair they start from -- the keys are
        // written into the pair that makes that (keys2, order); its state is cleared by the keys kernel on its way)
        const Radix32Plan plan = radix32_plan(n, 0, 32);
        const size_t sort_bytes = (size_t)plan.words * 4, scan_bytes = oscan_scratch_bytes<uint32_t>(n);
        const size_t off_scan = (sort_bytes + 255) & ~(size_t)255;
        RSREG_HIP(ctx, ctx->d_tmp.reserve(off_scan + scan_bytes + 256));
        char *tmp = ctx->d_tmp.as<char>();
        uint32_t *k_a = plan.ends_in_first ? keys2 : keys, *v_a = plan.ends_in_first ? order : vals;
        uint32_t *k_b = k_a == keys ? keys2 : keys, *v_b = v_a == vals ? order : vals;
        k_trim_keys<<<nb, kBlock, 0, st>>>(cw, ctx->d_corr_d2.as<float>(), n, k_a, v_a, ctx->d_tmp.as<uint32_t>(), plan.words);
        RSREG_HIP(ctx, hipGetLastError());
        {
            bool in_first = false;
            RSREG_HIP(ctx, radix32_sort_pairs<uint32_t>(plan, ctx->d_tmp.as<uint32_t>(), k_a, k_b, v_a, v_b, n, 0, 32, st, &in_first));   // stable: ties by position
            if ((in_first ? v_a : v_b) != order) return fail(ctx, RSREG_ERR_STATE, "osort: the sorted pairs are not where they belong");
        }
        k_trim_gather<<<nb, kBlock, 0, st>>>(cw, order, n, ws);
        RSREG_HIP(ctx, hipGetLastError());
        RSREG_HIP(ctx, (oscan<uint32_t, true>(ws, cum, (size_t)n, 0u, tmp + off_scan, st)));
        k_trim_apply<<<nb, kBlock, 0, st>>>(order, ws, cum, n, (float)s.prm.trim_overlap_ratio, cw, corr_pos);
        RSREG_HIP(ctx, hipGetLastError());
    }
    return RSREG_OK;
}

int launch_search(rsreg_ctx *ctx)
{
    ctx->icp.idle_after_sums = false;   // (something is queued on the main stream from here on)
    IcpState &s = ctx->icp;
    {
        int rcr = ensure_restarted(ctx);
        if (rcr) return rcr;
    }
    const uint32_t n = (uint32_t)ctx->n_work;
    const double gate2 = s.prm.max_correspondence_distance * s.prm.max_correspondence_distance;
    if (n) {
        ScopedEvents ev(ctx, &ctx->ev_nn);
        const GridDev g = grid_dev(ctx, s.prm.max_correspondence_distance);
        if (ctx->grid.dense == 2) {
            // the largest float that passes PCL's `if (distance > max_dist_sqr) continue` (compared in double)
            float gate_f = gate2 >= (double)FLT_MAX ? FLT_MAX : (float)gate2;
            if ((double)gate_f > gate2) gate_f = std::nextafterf(gate_f, 0.0f);
            unsigned long long *keys = ctx->d_scan_keys.as<unsigned long long>();
            const uint32_t nt = ctx->grid.n_points;
            RSREG_HIP(ctx, hipMemsetAsync(keys, 0xff, (size_t)n * 8, ctx->stream));
            if (nt)
                k_scan_nn<<<std::min(div_up(nt, kBlock), 2048u), kBlock, 0, ctx->stream>>>(ctx->d_tgt_sorted.as<float4>(), nt, ctx->d_cur.as<float4>(),
                                                                                          n, gate_f, keys);
            k_scan_finish<<<1, kScanMaxSource, 0, ctx->stream>>>(keys, ctx->d_cur.as<float4>(), n, ctx->d_corr_pos.as<int>(),
                                                                ctx->d_corr_d2.as<float>());
        } else if (ctx->grid.dense)
            (ctx->grid.have_nbr ? k_nn_search_dense<0> : k_nn_search_dense<3>)<<<div_up(n, kBlock), kBlock, 0, ctx->stream>>>(ctx->d_cur.as<float4>(), n,
                                                                             dense_dev(ctx, s.prm.max_correspondence_distance), gate2,
                                                                             ctx->d_corr_pos.as<int>(), ctx->d_corr_d2.as<float>(),
                                                                             seed_ptr(ctx));
        else
            k_nn_search<<<div_up(n, kBlock), kBlock, 0, ctx->stream>>>(ctx->d_cur.as<float4>(), n, g, gate2,
                                                                       ctx->d_corr_pos.as<int>(), ctx->d_corr_d2.as<float>(), seed_ptr(ctx));
        RSREG_HIP(ctx, hipGetLastError());
        s.n_nn_launches++;
    }
    if (filters_on(s.prm)) {
        int rc = apply_filters(ctx);
        if (rc) return rc;
    }
    s.have_search = true;
    return RSREG_OK;
}

extern "C" int rsreg_comm_allreduce_device_(rsreg_ctx *ctx, double *d_buf, int count);  // comm.cpp

// where k_final_reduce leaves the 17 sums the host is about to read: without a communicator straight in the pinned
// host buffer (no copy to queue behind the kernel), otherwise in HBM (the all-reduce works there)
double *host_sums_target(rsreg_ctx *ctx) { return ctx->comm ? ctx->d_sums.as<double>() : ctx->h_sums.as<double>(); }

int fetch_sums(rsreg_ctx *ctx, double *sums, bool global)
{
    if (global && ctx->comm) {   // also on a one-rank communicator: same calls, same stream order
        ScopedEvents ev(ctx, &ctx->ev_allreduce);
        int rc = rsreg_comm_allreduce_device_(ctx, ctx->d_sums.as<double>(), RSREG_NUM_SUMS);
        if (rc) return rc;
    }
    double *h = ctx->h_sums.as<double>();
    if (ctx->comm) RSREG_HIP(ctx, hipMemcpyAsync(h, ctx->d_sums.ptr, RSREG_NUM_SUMS * 8, hipMemcpyDeviceToHost, ctx->stream));
    RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->icp.idle_after_sums = true;
    (void)target_counts(ctx, false);
    std::memcpy(sums, h, RSREG_NUM_SUMS * 8);
    return RSREG_OK;
}

int launch_sums(rsreg_ctx *ctx, double *sums, bool global)
{
    ctx->icp.idle_after_sums = false;   // (something is queued on the main stream from here on)
    {
        int rcr = ensure_restarted(ctx);
        if (rcr) return rcr;
    }
    const uint32_t n = (uint32_t)ctx->n_work;
    {
        ScopedEvents ev(ctx, &ctx->ev_reduce);
        if (filters_on(ctx->icp.prm))
            k_cov_reduce_w<<<reduce_blocks(n), kTile, 0, ctx->stream>>>(ctx->d_cur.as<float4>(), ctx->d_corr_pos.as<int>(),
                                                                      ctx->d_corr_d2.as<float>(), ctx->d_corr_w.as<uint32_t>(),
                                                                      ctx->d_tgt_sorted.as<float4>(), n, ctx->d_partials.as<double>());
        else
            k_cov_reduce<<<reduce_blocks(n), kTile, 0, ctx->stream>>>(ctx->d_cur.as<float4>(), ctx->d_corr_pos.as<int>(),
                                                                    ctx->d_corr_d2.as<float>(), ctx->d_tgt_sorted.as<float4>(), n,
                                                                    ctx->d_partials.as<double>());
        RSREG_HIP(ctx, hipGetLastError());
        k_final_reduce<<<RSREG_NUM_SUMS, kReduceBlock, 0, ctx->stream>>>(ctx->d_partials.as<double>(), reduce_blocks(n), host_sums_target(ctx));
        RSREG_HIP(ctx, hipGetLastError());
    }
    return fetch_sums(ctx, sums, global);
}

// ---- tile schedule of the fused dense kernel (icp_dense.hpp: TileSched) -------------------------------------
// A launch of ~14 k waves on 8 k wave slots ends with a long tail: a few waves run 3x longer than the mean,
// they all sit on the same (near, densely sampled) surfaces in every iteration, and nothing is left to fill
// the slots around them.  The second launch of an alignment times every wave; from then on the tiles
// are launched longest first, and the longest few per cent are searched by 2 or 4 lanes per query.
// What is summed, and in which order, does not change.
struct SchedCfg {
    bool on = true;
    double f4 = 0.0, f2 = 0.10;    // fractions of the tiles searched with 4 and with 2 lanes per query (swept on the bench pair)
    uint32_t min_tiles = 1024;     // below this a launch does not even fill the wave slots once
    int at_launch = 1;             // the launch that is timed (0 = the first, which runs without seeds)
};

SchedCfg sched_cfg()
{
    const Tunables &t = tunables();
    SchedCfg c;
    c.on = t.sched;
    c.f4 = t.sched_f4;
    c.f2 = t.sched_f2;
    c.min_tiles = t.sched_min_tiles;
    c.at_launch = t.sched_at;
    return c;
}

constexpr int kSchedKeepFor = 8;   // alignments a tile schedule serves before a launch is timed again

struct SchedBufs {
    uint32_t *items, *cost, *done, *keys, *keys_alt, *vals, *vals_alt, *items_first;
};

SchedBufs sched_bufs(const rsreg_ctx *ctx, uint32_t n_tiles)
{
    uint32_t *p = ctx->d_sched.as<uint32_t>();
    (void)n_tiles;
    const size_t t = ctx->sched_cap_tiles;   // (capacity, not this source's tiles: the arrays stay put from one alignment to the next)
    return SchedBufs{p, p + 4 * t, p + 6 * t, p + 7 * t, p + 8 * t, p + 9 * t, p + 10 * t, p + 11 * t};
}

// The whole schedule in one workgroup (7 k tiles at 10^6 points; three kernels and a device-wide radix sort of 10-bit
// keys cost 35 us between two search launches, this one a few): a counting sort of the tiles by how long their slower
// wave ran in the timed launch, in steps of 0.64 us, longest first, then rank r -> its workgroups: the first n4 tiles
// get four, the next n2 two, the others one.  Tiles of one step come in whatever order the atomics hand out: the order
// of the workgroups decides when a tile runs, never what it computes.
__global__ __launch_bounds__(1024) void k_sched_build(const uint32_t *cost, uint32_t n_tiles, uint32_t n4, uint32_t n2, uint32_t *items, uint32_t *done)
{
    __shared__ uint32_t bins[1024];
    __shared__ uint32_t part[16];
    bins[threadIdx.x] = 0;
    __syncthreads();
    auto key_of = [&](uint32_t t) {
        uint32_t c = 0;
        for (int w = 0; w < kTileWaves; ++w) c = max(c, cost[t * kTileWaves + w]);
        return 1023u - min(c >> 6, 1023u);
    };
    for (uint32_t t = threadIdx.x; t < n_tiles; t += 1024u) {
        atomicAdd(&bins[key_of(t)], 1u);
        done[t] = 0u;
    }
    __syncthreads();
    // exclusive scan of the 1024 bins: one per thread
    const uint32_t mine = bins[threadIdx.x], lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if ((int)lane >= off) incl += v;
    }
    if (lane == 63u) part[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += part[w];
    __syncthreads();
    bins[threadIdx.x] = before + incl - mine;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n_tiles; t += 1024u) {
        const uint32_t r = atomicAdd(&bins[key_of(t)], 1u);
        uint32_t lg, base;
        if (r < n4) { lg = 2; base = 4 * r; }
        else if (r < n4 + n2) { lg = 1; base = 4 * n4 + 2 * (r - n4); }
        else { lg = 0; base = 4 * n4 + 2 * n2 + (r - n4 - n2); }
        for (uint32_t p = 0; p < (1u << lg); ++p) items[base + p] = t | p << 24 | lg << 28;
    }
}

// The same schedule, XCD-aware (RSREG_SCHED_XCD, default on): workgroups are dealt to the eight XCDs in turn (b and
// b + 8 share one, MI355X_MICROARCH.md "Workgroup dispatch"), and every XCD has a 4 MiB L2 of its own.  The tiles are in
// the source's Morton order, so a run of consecutive tiles is a compact piece of space: pieces of `deal` (32) consecutive
// tiles are dealt to eight runs in turn, each run is sorted longest first by itself, and the runs are dealt out to the
// workgroups in turn: workgroup b works on run b % 8 while that run lasts, so one XCD's L2 sees an eighth of the target's
// cells instead of all of them.  When a run is used up the others close ranks (no empty workgroups; those last tiles land
// on whatever XCD is next).  Which workgroup takes which tile never changes what a tile computes.
// Measured (profiles/r03_experiments/xcd_schedule_*): 300 k points 53.6 -> 48.7 us per launch (the eighth of the index
// fits the L2), 10^6 points level (96.9 against 96.5-98.3); pieces of 64+ tiles or eight contiguous runs of equal cost
// (deal = 0) are slower at 10^6 (102-103 us: the heavy tiles of a crowded region then share one XCD's 256 workgroup
// slots); a few cost classes with the Morton order kept inside a class are slower the coarser the classes (97 -> 117 us).
__global__ __launch_bounds__(1024) void k_sched_build_xcd(const uint32_t *cost, uint32_t n_tiles, uint32_t n4, uint32_t n2, uint32_t total_items,
                                                          uint32_t deal, uint32_t *items, uint32_t *done)
{
    constexpr uint32_t kRuns = 8;
    __shared__ uint32_t bins[kRuns * 1024];
    __shared__ uint32_t part[16];
    __shared__ uint32_t run_first[kRuns + 1];   // rank, in run-major sorted order, of the first tile of a run
    __shared__ uint32_t total_cost;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t k = tid; k < kRuns * 1024u; k += 1024u) bins[k] = 0;
    auto bucket_of = [&](uint32_t t) {   // 0 .. 1023, steps of 0.64 us, longer = larger
        uint32_t c = 0;
        for (int w = 0; w < kTileWaves; ++w) c = max(c, cost[t * kTileWaves + w]);
        return min(c >> 6, 1023u);
    };
    auto block_exclusive = [&](uint32_t mine, uint32_t *total) -> uint32_t {   // over the 1024 threads, in thread order
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off);
            if ((int)lane >= off) incl += v;
        }
        __syncthreads();
        if (lane == 63u) part[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 16u; ++w) {
            if (w < wave) before += part[w];
            all += part[w];
        }
        if (total) *total = all;
        return before + incl - mine;
    };
    // every thread owns a contiguous share of the tiles: their cost before them, in natural order
    const uint32_t per = (n_tiles + 1023u) / 1024u, lo = min(n_tiles, tid * per), hi = min(n_tiles, lo + per);
    uint32_t mine = 0;
    for (uint32_t t = lo; t < hi; ++t) mine += bucket_of(t) + 1u;
    uint32_t all = 0;
    const uint32_t before = block_exclusive(mine, &all);
    if (tid == 0) total_cost = all;
    __syncthreads();
    const uint32_t C = max(total_cost, 1u);
    // deal == 0: eight contiguous runs of equal cost; deal > 0: pieces of `deal` consecutive tiles dealt to the runs in turn
    auto run_of = [&](uint32_t t, uint32_t cost_before) {
        return deal ? (t / deal) % kRuns : min(kRuns - 1u, (uint32_t)(((unsigned long long)cost_before * kRuns) / C));
    };
    {
        uint32_t at = before;
        for (uint32_t t = lo; t < hi; ++t) {
            const uint32_t b = bucket_of(t);
            atomicAdd(&bins[run_of(t, at) * 1024u + (1023u - b)], 1u);
            at += b + 1u;
            done[t] = 0u;
        }
    }
    __syncthreads();
    // exclusive scan of the 8 x 1024 bins, run-major: eight consecutive bins per thread
    {
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = bins[tid * 8u + k]; sum += v[k]; }
        uint32_t at = block_exclusive(sum, nullptr);
        if ((tid & 127u) == 0) run_first[tid >> 7] = at;   // (bin x * 1024 is thread 128 x's first)
        if (tid == 0) run_first[kRuns] = n_tiles;
#pragma unroll
        for (int k = 0; k < 8; ++k) { bins[tid * 8u + k] = at; at += v[k]; }
    }
    __syncthreads();
    // how long every run's list of workgroups is: its tiles, the first n4 / 8 of them four times, the next n2 / 8 twice
    uint32_t len[kRuns], r4[kRuns], r2[kRuns], used = 0;
#pragma unroll
    for (uint32_t x = 0; x < kRuns; ++x) {
        const uint32_t cnt = run_first[x + 1] - run_first[x];
        r4[x] = min(n4 / kRuns, cnt);
        r2[x] = min(n2 / kRuns, cnt - r4[x]);
        len[x] = cnt + 3u * r4[x] + r2[x];
        used += len[x];
    }
    auto place = [&](uint32_t x, uint32_t p) {   // list position p of run x -> workgroup: the runs dealt out in turn while they last
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t y = 0; y < kRuns; ++y) pos += min(len[y], p + (y < x ? 1u : 0u));
        return pos;
    };
    {
        uint32_t at = before;
        for (uint32_t t = lo; t < hi; ++t) {
            const uint32_t b = bucket_of(t), x = run_of(t, at);
            at += b + 1u;
            const uint32_t r = atomicAdd(&bins[x * 1024u + (1023u - b)], 1u) - run_first[x];   // rank inside its run, longest first
            uint32_t lg, p;
            if (r < r4[x]) { lg = 2; p = 4u * r; }
            else if (r < r4[x] + r2[x]) { lg = 1; p = 4u * r4[x] + 2u * (r - r4[x]); }
            else { lg = 0; p = 4u * r4[x] + 2u * r2[x] + (r - r4[x] - r2[x]); }
            for (uint32_t q = 0; q < (1u << lg); ++q) items[place(x, p + q)] = t | q << 24 | lg << 28;
        }
    }
    for (uint32_t k = used + tid; k < total_items; k += 1024u) items[k] = 0xffffffffu;   // (a run shorter than its share of the splits: nothing to do)
}

// first: the schedule of the alignments' FIRST launches (from a timed first launch; kept beside the steady one)
int build_schedule(rsreg_ctx *ctx, uint32_t n_tiles, bool first = false)
{
    const SchedCfg cfg = sched_cfg();
    if (n_tiles > ctx->sched_cap_tiles) return fail(ctx, RSREG_ERR_STATE, "tile schedule: more tiles than the schedule buffer was laid out for");
    const SchedBufs sb = sched_bufs(ctx, n_tiles);
    hipStream_t st = ctx->stream;
    uint32_t n4 = (uint32_t)(cfg.f4 * n_tiles), n2 = (uint32_t)(cfg.f2 * n_tiles);
    const bool xcd = tunables().sched_xcd;
    uint32_t *items = first ? sb.items_first : sb.items;
    if (xcd) {
        n4 -= n4 % 8u;   // (an eighth of the splits to every run)
        n2 -= n2 % 8u;
        const uint32_t deal = tunables().sched_xcd_deal;
        k_sched_build_xcd<<<1, 1024, 0, st>>>(sb.cost, n_tiles, n4, n2, n_tiles + 3 * n4 + n2, deal, items, sb.done);
    } else {
        k_sched_build<<<1, 1024, 0, st>>>(sb.cost, n_tiles, n4, n2, items, sb.done);
    }
    RSREG_HIP(ctx, hipGetLastError());
    const uint32_t n_items = n_tiles + 3 * n4 + n2;
    if (first) {
        ctx->sched_first_items = n_items;
        ctx->sched_first_tiles = n_tiles;
        ctx->sched_first_age = 0;
        return RSREG_OK;
    }
    ctx->icp.sched_items = n_items;
    ctx->sched_keep_items = ctx->icp.sched_items;   // (kept for the alignments to come: launch_fused)
    ctx->sched_keep_tiles = n_tiles;
    ctx->sched_keep_age = 0;
#ifdef RSREG_DIAG
    if (tunables().sched_verbose)
        std::fprintf(stderr, "[rsreg] tile schedule: %u tiles, %u searched by 4 lanes per query, %u by 2, %u workgroups\n", n_tiles, n4, n2,
                     ctx->icp.sched_items);
#endif
    ctx->icp.sched_ready = true;
    return RSREG_OK;
}

// fused pass: applies the pending increment (if any), searches, gates and reduces
int launch_fused(rsreg_ctx *ctx, double *sums, bool want_corr, bool device_loop = false)
{
    ctx->icp.idle_after_sums = false;   // (something is queued on the main stream from here on)
    IcpState &s = ctx->icp;
    const uint32_t n = (uint32_t)ctx->n_work;
    const IcpDevState *dev = device_loop ? ctx->d_icp_state.as<IcpDevState>() : nullptr;
    const double gate2 = s.prm.max_correspondence_distance * s.prm.max_correspondence_distance;
    // the first launch of an alignment over the dense index reads the source itself, applies the guess and starts without
    // seeds: k_restart_source's work, one launch and a pass over the working copy saved (never with a schedule of this
    // alignment's own in place: that is built from one of its launches; a carried one may serve it, below)
    const bool restart_apart = tunables().restart_apart;
    const bool restart_here = s.restart_pending && ctx->grid.dense && !s.sched_ready && !s.pending_transform && !restart_apart;
    if (restart_here) {
        s.restart_pending = false;
    } else {
        int rcr = ensure_restarted(ctx);
        if (rcr) return rcr;
    }
    {
        ScopedEvents ev(ctx, &ctx->ev_nn);
        const GridDev g = grid_dev(ctx, s.prm.max_correspondence_distance);
        if (ctx->grid.dense) {
            unsigned long long *wt = wave_times_ptr(ctx, n);
            // (the search beyond ring 1 comes in two forms; each instantiation carries only the one its grid uses)
            const bool blocks = ctx->grid.max_ring <= 4 && !tunables().far_rows;
#ifdef RSREG_DIAG
            const bool light = tunables().wave_times_light;
            const bool tab = !ctx->grid.have_nbr;   // (an index without occupancy words: read off the table)
            auto kern = wt ? (light ? (tab ? k_icp_fused_dense<2, 3> : blocks ? k_icp_fused_dense<2, 1> : k_icp_fused_dense<2, 2>)
                                    : (tab ? k_icp_fused_dense<1, 3> : blocks ? k_icp_fused_dense<1, 1> : k_icp_fused_dense<1, 2>))
                           : (tab ? k_icp_fused_dense<0, 3> : blocks ? k_icp_fused_dense<0, 1> : k_icp_fused_dense<0, 2>);
#else
            const bool light = false;
            auto kern = !ctx->grid.have_nbr ? k_icp_fused_dense<0, 3> : blocks ? k_icp_fused_dense<0, 1> : k_icp_fused_dense<0, 2>;
#endif
            const SchedCfg cfg = sched_cfg();
            const uint32_t n_tiles = reduce_blocks(n);
            const bool sched_ok = cfg.on && (!wt || light) && n_tiles >= cfg.min_tiles && n_tiles < (1u << 24) &&
                                  n_tiles <= ctx->sched_cap_tiles;   // (the arrays of sched_bufs are laid out for that many tiles)
            TileSched sc{};
            sc.n_tiles = n_tiles;
            uint32_t grid = n_tiles;
            if (sched_ok) {
                const SchedBufs sb = sched_bufs(ctx, n_tiles);
                sc.done = sb.done;
                sc.pos = ctx->d_corr_pos.as<int>();
                sc.d2 = ctx->d_corr_d2.as<float>();
                // The schedule of an earlier alignment of this context serves this one too, from the launch that would otherwise be
                // timed: the long tiles sit on the same (near, densely sampled) surfaces from one frame to the next, a schedule is
                // an order of work and never wrong, and the timed launch runs unscheduled (140 against 93 us at 10^6 points).
                // Carried for at most kSchedKeepFor alignments and only to a source of about as many tiles; RSREG_SCHED_KEEP=0: never.
                auto fits = [&](uint32_t tiles) { return n_tiles + n_tiles / 8 >= tiles && tiles + tiles / 8 >= n_tiles; };
                const bool keep = tunables().sched_keep;
                bool first_sched = false, time_first = false;
                if (restart_here && keep) {
                    // the FIRST launch (unseeded, from the source itself under the guess: the only launch the reference's
                    // parameters ever run) has a cost profile of its own: it is timed once and scheduled from its own
                    // kind's costs in the alignments that follow
                    if (ctx->sched_first_items && ctx->sched_first_age < kSchedKeepFor && fits(ctx->sched_first_tiles)) {
                        first_sched = true;
                        ++ctx->sched_first_age;
                    } else {
                        time_first = true;
                    }
                }
                if (!s.sched_ready && !restart_here && s.fused_launches >= cfg.at_launch && ctx->sched_keep_items && keep &&
                    ctx->sched_keep_age < kSchedKeepFor && fits(ctx->sched_keep_tiles)) {
                    s.sched_ready = true;
                    s.sched_carried = true;
                    s.sched_items = ctx->sched_keep_items + (n_tiles > ctx->sched_keep_tiles ? n_tiles - ctx->sched_keep_tiles : 0u);
                    ++ctx->sched_keep_age;
                }
                if (first_sched) {
                    sc.items = sb.items_first;
                    sc.n_items = ctx->sched_first_items;
                    sc.first_extra = ctx->sched_first_tiles;
                    grid = ctx->sched_first_items + (n_tiles > ctx->sched_first_tiles ? n_tiles - ctx->sched_first_tiles : 0u);
                } else if (s.sched_ready) {
                    sc.items = sb.items;
                    grid = s.sched_items;
                    sc.n_items = s.sched_carried ? ctx->sched_keep_items : grid;
                    sc.first_extra = ctx->sched_keep_tiles;
                } else if (time_first || (!restart_here && s.fused_launches == cfg.at_launch)) {
                    sc.cost = sb.cost;
                }
            }
            kern<<<grid, kTile, 0, ctx->stream>>>(
                ctx->d_cur.as<float4>(), restart_here ? ctx->d_src.as<float4>() : nullptr, ctx->d_first.as<uint32_t>(), n, to_mat34(restart_here ? s.final_t : s.t_inc),
                restart_here ? (s.final_t.is_identity() ? 0 : 1) : (s.pending_transform ? 1 : 0),
                dense_dev(ctx, s.prm.max_correspondence_distance), gate2, want_corr ? ctx->d_corr_pos.as<int>() : nullptr,
                ctx->d_corr_d2.as<float>(), ctx->d_partials.as<double>(), seed_ptr(ctx), wt, dev, sc);
            RSREG_HIP(ctx, hipGetLastError());
            if (sc.cost) {
                int rc = build_schedule(ctx, n_tiles, restart_here);
                if (rc) return rc;
                if (restart_here && cfg.at_launch == 0) {   // (RSREG_SCHED_AT=0: the steady schedule from the first launch's costs too)
                    rc = build_schedule(ctx, n_tiles, false);
                    if (rc) return rc;
                }
            }
            s.fused_launches++;
            if (sc.items) s.n_sched_launches++;
        }
        else
            k_icp_fused<<<reduce_blocks(n), kTile, 0, ctx->stream>>>(
                ctx->d_cur.as<float4>(), n, to_mat34(s.t_inc), s.pending_transform ? 1 : 0, g, gate2,
                want_corr ? ctx->d_corr_pos.as<int>() : nullptr, ctx->d_corr_d2.as<float>(), ctx->d_partials.as<double>(),
                seed_ptr(ctx), dev);
        RSREG_HIP(ctx, hipGetLastError());
        s.n_nn_launches++;
    }
    s.pending_transform = false;
    // (no events of their own for what follows: in the fused pipelines ms_reduce is the time
    // between consecutive search kernels, read off the search kernels' events)
    s.have_search = want_corr;
    if (device_loop && !ctx->comm) {   // final reduce + solve in one launch
        auto *st = ctx->d_icp_state.as<IcpDevState>();
        k_final_reduce_solve<<<RSREG_NUM_SUMS, kReduceBlock, 0, ctx->stream>>>(ctx->d_partials.as<double>(), reduce_blocks(n), ctx->d_sums.as<double>(),
                                                                          st, reinterpret_cast<unsigned int *>(st + 1));
        RSREG_HIP(ctx, hipGetLastError());
        return RSREG_OK;
    }
    k_final_reduce<<<RSREG_NUM_SUMS, kReduceBlock, 0, ctx->stream>>>(ctx->d_partials.as<double>(), reduce_blocks(n),
                                                               device_loop ? ctx->d_sums.as<double>() : host_sums_target(ctx));
    RSREG_HIP(ctx, hipGetLastError());
    if (device_loop) {   // the sums stay on the device: (all-reduce,) solve, next pass
        if (ctx->comm) {
            ScopedEvents ev(ctx, &ctx->ev_allreduce);
            int rc = rsreg_comm_allreduce_device_(ctx, ctx->d_sums.as<double>(), RSREG_NUM_SUMS);
            if (rc) return rc;
        }
        k_icp_solve<<<1, 64, 0, ctx->stream>>>(ctx->d_sums.as<double>(), ctx->d_icp_state.as<IcpDevState>());
        RSREG_HIP(ctx, hipGetLastError());
        return RSREG_OK;
    }
    return fetch_sums(ctx, sums, true);
}

// RSREG_PIPELINE_DEVICE_LOOP with fixed-count criteria: every iteration is queued up front, the
// 136-byte state comes back once at the end.  Same kernels, same arithmetic as the host loop.
int run_device_loop(rsreg_ctx *ctx)
{
    IcpState &s = ctx->icp;
    RSREG_HIP(ctx, ctx->d_icp_state.reserve(sizeof(IcpDevState) + 16));   // + the reduce kernel's ticket
    RSREG_HIP(ctx, ctx->h_sums.reserve(1024));
    IcpDevState *h = ctx->h_sums.as<IcpDevState>();
    std::memset(h, 0, sizeof(IcpDevState) + 16);
    h->t_inc = to_mat34(Mat4f::identity());
    h->final_t = s.final_t;
    for (int k = 0; k < 9; ++k) h->svd_v[k] = (k % 4 == 0) ? 1.0 : 0.0;
    const auto t_q0 = std::chrono::steady_clock::now();
    RSREG_HIP(ctx, hipMemcpyAsync(ctx->d_icp_state.ptr, h, sizeof(IcpDevState) + 16, hipMemcpyHostToDevice, ctx->stream));
    const int iters = std::max(1, s.prm.max_iterations);
    int it = 0;
    for (; it < iters; ++it) {
        int rc = launch_fused(ctx, nullptr, false, true);
        if (rc) return rc;
    }
    RSREG_HIP(ctx, hipMemcpyAsync(h, ctx->d_icp_state.ptr, sizeof(IcpDevState), hipMemcpyDeviceToHost, ctx->stream));
    // (what queueing the whole loop took this thread: with several alignments in flight the threads share the runtime's launch path)
    ctx->host_timing.loop_enqueue = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_q0).count();
    RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    s.idle_after_sums = true;
    (void)target_counts(ctx, false);
    std::memcpy(s.sums_last, h->sums_last, sizeof(s.sums_last));
    s.ncorr = h->ncorr;
    s.final_t = h->final_t;
    s.iterations = h->iterations;
    s.cur_mse = h->cur_mse;
    s.pending_transform = h->apply != 0;
    for (int c = 0; c < 4; ++c) { s.t_inc(0, c) = h->t_inc.r0[c]; s.t_inc(1, c) = h->t_inc.r1[c]; s.t_inc(2, c) = h->t_inc.r2[c]; }
    if (h->stopped) {
        s.state = RSREG_CONV_NO_CORRESPONDENCES;
        s.converged = 0;
    } else {
        s.state = RSREG_CONV_ITERATIONS;
        s.converged = 1;
    }
    return RSREG_OK;
}

int apply_pending_transform(rsreg_ctx *ctx)
{
    ctx->icp.idle_after_sums = false;   // (something is queued on the main stream from here on)
    IcpState &s = ctx->icp;
    {
        int rcr = ensure_restarted(ctx);
        if (rcr) return rcr;
    }
    if (!s.pending_transform) return RSREG_OK;
    const uint32_t n = (uint32_t)ctx->n_work;
    if (n) {
        ScopedEvents ev(ctx, &ctx->ev_transform);
        k_transform<<<div_up(n, kBlock), kBlock, 0, ctx->stream>>>(ctx->d_cur.as<float4>(), n, to_mat34(s.t_inc));
        RSREG_HIP(ctx, hipGetLastError());
    }
    s.pending_transform = false;
    return RSREG_OK;
}

// Umeyama + compose + criteria for one iteration's (global) sums; the transform of the
// source cloud itself is left pending so that the fused kernel can fold it into its pass.
int update_from_sums(rsreg_ctx *ctx, const double *sums, int *done)
{
    IcpState &s = ctx->icp;
    std::memcpy(s.sums_last, sums, sizeof(s.sums_last));
    s.ncorr = (uint64_t)(sums[0] + 0.5);
    if (s.ncorr < 3) {  // min_number_correspondences_
        s.state = RSREG_CONV_NO_CORRESPONDENCES;
        s.converged = 0;
        *done = 1;
        return RSREG_OK;
    }
    umeyama_from_sums(sums, s.t_inc, s.svd_v);
    s.pending_transform = true;
    s.final_t = mul(s.t_inc, s.final_t);
    s.iterations++;
    s.cur_mse = sums[16] / sums[0];
    s.converged = criteria_has_converged(s) ? 1 : 0;
    s.have_search = false;
    *done = s.converged;
    return RSREG_OK;
}

// A host cloud on its way into HBM, packed xyz (PCL's ICP reads nothing else of a point): the caller's records are
// packed into pinned memory piece by piece (the pool of host threads, workers.hpp) and every piece goes over the PCIe link
// on the context's upload stream while the next one is being packed; `stage` / `ev`: the staging buffer of this kind of
// cloud (source / target: one each, so that the target is packed while the source is still on the link) and the event behind
// its last piece.  The stream `waiter` is made to wait for that event; the caller's buffer has been read when this returns.
int upload_packed(rsreg_ctx *ctx, PinnedBuf &stage, hipEvent_t &ev, DevBuf &d_raw, const void *points, size_t n, size_t stride,
                  hipStream_t waiter, double *ms_pack, double *ms_wait)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    if (!ctx->stream_h2d) RSREG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_h2d, hipStreamNonBlocking));
    if (!ev) RSREG_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else RSREG_HIP(ctx, hipEventSynchronize(ev));   // (the staging buffer's last trip over the link is over)
    const auto t1 = clk::now();
    RSREG_HIP(ctx, stage.reserve(n * 12 + 16));
    RSREG_HIP(ctx, d_raw.reserve(n * 12 + 16));
    float *dst = stage.as<float>();
    const char *src = static_cast<const char *>(points);
    const size_t piece = (size_t)1 << 18;   // 3 MB of packed xyz: ~0.1 ms on the link
    for (size_t lo = 0; lo < n; lo += piece) {
        const size_t hi = std::min(n, lo + piece);
        host_parallel_for(hi - lo, [=](size_t a, size_t b) {
            if (stride == 12) {
                std::memcpy(dst + 3 * (lo + a), src + 12 * (lo + a), (b - a) * 12);
            } else {
                for (size_t i = lo + a; i < lo + b; ++i) std::memcpy(dst + 3 * i, src + i * stride, 12);
            }
        });
        RSREG_HIP(ctx, hipMemcpyAsync(d_raw.as<char>() + lo * 12, dst + 3 * lo, (hi - lo) * 12, hipMemcpyHostToDevice, ctx->stream_h2d));
    }
    RSREG_HIP(ctx, hipEventRecord(ev, ctx->stream_h2d));
    RSREG_HIP(ctx, hipStreamWaitEvent(waiter, ev, 0));
    const auto t2 = clk::now();
    if (ms_wait) *ms_wait = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (ms_pack) *ms_pack = std::chrono::duration<double, std::milli>(t2 - t1).count();
    return RSREG_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

int rsreg_version(void) { return RSREG_VERSION_MAJOR * 1000 + RSREG_VERSION_MINOR; }

const char *rsreg_status_string(int status)
{
    switch (status) {
        case RSREG_OK: return "ok";
        case RSREG_ERR_INVALID_ARG: return "invalid argument";
        case RSREG_ERR_EMPTY_CLOUD: return "empty cloud";
        case RSREG_ERR_HIP: return "HIP error";
        case RSREG_ERR_RCCL: return "RCCL error";
        case RSREG_ERR_NO_TARGET: return "no target set";
        case RSREG_ERR_NO_DEVICE: return "no usable HIP device";
        case RSREG_ERR_ALLOC: return "allocation failed";
        case RSREG_ERR_NO_SOURCE: return "no source set";
        case RSREG_ERR_STATE: return "call out of sequence";
        default: return "unknown status";
    }
}

// (the message is copied under the mutex the context's helper threads report under, into a buffer of the calling thread:
// the pointer stays valid until this thread's next call, whatever the helpers write meanwhile)
const char *rsreg_last_error(const rsreg_ctx *ctx)
{
    if (!ctx) return "null ctx";
    thread_local std::string copy;
    {
        std::lock_guard<std::mutex> lk(const_cast<rsreg_ctx *>(ctx)->error_mutex);
        copy = ctx->last_error;
    }
    return copy.c_str();
}

int rsreg_device_count(int *count)
{
    if (!count) return RSREG_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return RSREG_OK;
}

int rsreg_ctx_create(int device_id, void *stream, rsreg_ctx **out)
{
    rsreg::tunables_refresh();   // (the switches of csrc/tunables.hpp: as the environment has them now)
    if (!out) return RSREG_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RSREG_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return RSREG_ERR_INVALID_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return RSREG_ERR_NO_DEVICE;
    rsreg_ctx *ctx = new (std::nothrow) rsreg_ctx();
    if (!ctx) return RSREG_ERR_ALLOC;
    ctx->device = device_id;
    if (stream) {
        ctx->stream = static_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return RSREG_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return RSREG_OK;
}

int rsreg_comm_destroy(rsreg_ctx *ctx);

int rsreg_ctx_destroy(rsreg_ctx *ctx)
{
    if (!ctx) return RSREG_OK;
    (void)hipSetDevice(ctx->device);
    ctx->prep_join();   // (rsreg_ctx_prepare's thread: what it made is released with everything else below)
    // The helper threads go first, in the order of who waits for whom -- a queued side job waits for the upload worker
    // (side_wait_input), an upload or a download for nothing of the others -- and every stream they fed is drained before
    // a buffer, an event or a stream is released.
    if (ctx->src_worker) {
        ctx->src_worker->shutdown();
        delete ctx->src_worker;
        ctx->src_worker = nullptr;
    }
    for (rsreg::TicketWorker *w : ctx->side_workers)
        if (w) w->shutdown();
    if (ctx->up_worker) ctx->up_worker->shutdown();
    if (ctx->down_worker) ctx->down_worker->shutdown();
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream_src) (void)hipStreamSynchronize(ctx->stream_src);
    if (ctx->stream_copy) (void)hipStreamSynchronize(ctx->stream_copy);
    if (ctx->stream_down) (void)hipStreamSynchronize(ctx->stream_down);
    for (rsreg_ctx::SideSet &ss : ctx->side_sets)
        if (ss.stream) (void)hipStreamSynchronize(ss.stream);
    rsreg_comm_destroy(ctx);
    if (ctx->recip) {   // the child context of the reciprocal index runs on this context's stream: it goes first
        rsreg_ctx_destroy(ctx->recip);
        ctx->recip = nullptr;
    }
    DevBuf *bufs[] = {&ctx->d_tgt_raw, &ctx->d_tgt_sorted, &ctx->d_table, &ctx->d_keys, &ctx->d_keys_alt, &ctx->d_vals,
                      &ctx->d_vals_alt, &ctx->d_flags, &ctx->d_scan, &ctx->d_cellpos, &ctx->d_dense, &ctx->d_pos_of, &ctx->d_sched, &ctx->d_brick, &ctx->d_perm, &ctx->d_tmp, &ctx->d_plain_ticket,
                      &ctx->d_misc, &ctx->d_src_raw, &ctx->d_src_all, &ctx->d_uniq_of, &ctx->d_first, &ctx->d_src, &ctx->d_cur, &ctx->d_corr_pos, &ctx->d_corr_d2, &ctx->d_seed,
                      &ctx->d_partials, &ctx->d_sums, &ctx->d_icp_state, &ctx->d_corr_w, &ctx->d_recip_pts, &ctx->d_vox_in, &ctx->d_vox_out, &ctx->d_vox_cent, &ctx->d_ndt_vox, &ctx->d_ndt_src, &ctx->d_ndt_trans,
                      &ctx->d_ndt_partials, &ctx->d_ndt_out, &ctx->d_ndt_ctl, &ctx->d_ndt_seg, &ctx->d_comm, &ctx->d_skeys, &ctx->d_skeys_alt, &ctx->d_svals,
                      &ctx->d_sflags, &ctx->d_sscan, &ctx->d_stmp, &ctx->d_smisc, &ctx->d_shist, &ctx->d_scan_keys, &ctx->d_cnt, &ctx->d_arrived};
    if (ctx->stream_h2d) { (void)hipStreamSynchronize(ctx->stream_h2d); (void)hipStreamDestroy(ctx->stream_h2d); }
    ctx->h_stage_src.release();
    ctx->h_stage_tgt.release();
    if (ctx->ev_stage_src) (void)hipEventDestroy(ctx->ev_stage_src);
    if (ctx->ev_stage_tgt) (void)hipEventDestroy(ctx->ev_stage_tgt);
    for (hipEvent_t e : ctx->ev_home) (void)hipEventDestroy(e);
    for (DevBuf *b : bufs) b->release();
    ctx->h_sums.release();
    ctx->h_smisc.release();
    ctx->h_stage.release();
    ctx->h_ndt.release();
    ctx->h_ndt_build.release();
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_ndt)
        if (e) (void)hipEventDestroy(e);
    cloud_pool_clear(ctx);
    for (hipEvent_t e : ctx->ev_copy) (void)hipEventDestroy(e);
    if (ctx->up_worker) {
        ctx->up_worker->shutdown();
        delete ctx->up_worker;
        ctx->up_worker = nullptr;
    }
    if (ctx->stream_copy) {
        (void)hipStreamSynchronize(ctx->stream_copy);
        (void)hipStreamDestroy(ctx->stream_copy);
        (void)hipEventDestroy(ctx->ev_copy_gate);
        for (hipEvent_t e : ctx->ev_up) (void)hipEventDestroy(e);
    }
    for (rsreg::PinnedBuf &b : ctx->h_up) b.release();
    if (ctx->down_worker) {
        ctx->down_worker->shutdown();
        delete ctx->down_worker;
        ctx->down_worker = nullptr;
    }
    if (ctx->stream_down) {
        (void)hipStreamSynchronize(ctx->stream_down);
        (void)hipStreamDestroy(ctx->stream_down);
        (void)hipEventDestroy(ctx->ev_down_gate);
        for (hipEvent_t e : ctx->ev_down) (void)hipEventDestroy(e);
    }
    for (rsreg::PinnedBuf &b : ctx->h_down) b.release();
    for (rsreg::TicketWorker *&w : ctx->side_workers) {
        if (!w) continue;
        w->shutdown();
        delete w;
        w = nullptr;
    }
    for (rsreg_ctx::SideSet &ss : ctx->side_sets) {
        if (ss.stream) {
            (void)hipStreamSynchronize(ss.stream);
            (void)hipStreamDestroy(ss.stream);
        }
        for (DevBuf *b : {&ss.out, &ss.keys, &ss.keys_alt, &ss.vals, &ss.vals_alt, &ss.flags, &ss.scan, &ss.cent, &ss.misc, &ss.tmp}) b->release();
        ss.host.release();
    }
    if (ctx->ev_side_gate) (void)hipEventDestroy(ctx->ev_side_gate);
    if (ctx->stream_src) {
        (void)hipStreamDestroy(ctx->stream_src);
        (void)hipEventDestroy(ctx->ev_src_done);
        (void)hipEventDestroy(ctx->ev_main);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RSREG_OK;
}

int rsreg_ctx_synchronize(rsreg_ctx *ctx)
{
    if (!ctx) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)target_counts(ctx, false);
    return RSREG_OK;
}

int rsreg_ctx_set_profiling(rsreg_ctx *ctx, int enabled)
{
    if (!ctx) return RSREG_ERR_INVALID_ARG;
    ctx->profiling = enabled != 0;
    return RSREG_OK;
}

void rsreg_icp_params_default(rsreg_icp_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->max_iterations = 10;
    p->criteria_mode = RSREG_CRITERIA_PCL;
    p->pipeline_mode = RSREG_PIPELINE_FUSED;
    p->max_correspondence_distance = std::sqrt(DBL_MAX);
    p->transformation_epsilon = 0.0;
    p->transformation_rotation_epsilon = 0.0;
    p->euclidean_fitness_epsilon = -DBL_MAX;
}

void rsreg_icp_params_reference(rsreg_icp_params *p)
{
    if (!p) return;
    rsreg_icp_params_default(p);
    p->max_iterations = 100;                 // incremental_icp.hpp:46
    p->max_correspondence_distance = 0.01;   // :47
    p->transformation_epsilon = 1;           // :48
    p->euclidean_fitness_epsilon = 1000;     // :49
}

int rsreg_icp_set_target_device(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride, int is_dense,
                                double max_correspondence_distance)
{
    (void)is_dense;
    if (!ctx || (n && !d_points) || stride < 12 || (stride & 3)) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    return build_grid(ctx, static_cast<const char *>(d_points), n, stride, max_correspondence_distance);
}

// (internal, cloud.hip) a target for the handful of source points already loaded: no index, see scan_target
int rsreg_icp_set_target_scan_(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride, double max_correspondence_distance)
{
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    return scan_target(ctx, static_cast<const char *>(d_points), n, stride, max_correspondence_distance);
}

int rsreg_icp_set_target(rsreg_ctx *ctx, const void *points, size_t n, size_t stride, int is_dense,
                         double max_correspondence_distance)
{
    (void)is_dense;
    if (!ctx || (n && !points) || stride < 12) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    rsreg_host_timing &ht = ctx->host_timing;
    // a counting build that was queued and not waited for (build_dense) still reads d_tgt_raw on the main stream, and the
    // upload stream is ordered behind nothing: a second set_target with no alignment in between would overwrite the records
    // under k_cc_count / k_cc_scatter (a changed point re-derives another slot: a write past d_arrived).  Wait for it first.
    int rc = target_counts(ctx, true);
    if (rc) return rc;
    rc = upload_packed(ctx, ctx->h_stage_tgt, ctx->ev_stage_tgt, ctx->d_tgt_raw, points, n, stride, ctx->stream, &ht.target_pack, &ht.target_stage_wait);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    rc = build_grid(ctx, ctx->d_tgt_raw.as<char>(), n, 12, max_correspondence_distance);   // (behind the upload, by the stream's order)
    ht.target_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int rsreg_icp_set_source_device(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride, int is_dense)
{
    (void)is_dense;
    if (!ctx || (n && !d_points) || stride < 12 || (stride & 3)) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    return load_source(ctx, static_cast<const char *>(d_points), n, stride);
}

int rsreg_icp_set_source(rsreg_ctx *ctx, const void *points, size_t n, size_t stride, int is_dense)
{
    (void)is_dense;
    if (!ctx || (n && !points) || stride < 12) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = join_source(ctx);   // (a load still in flight reads d_src_raw)
    if (rc) return rc;
    // packed and sent piece by piece on the upload stream; the load itself (stream_src, the worker thread) starts behind the
    // event the main stream is made to wait for.  Nothing is waited for here: the caller goes on to rsreg_icp_set_target,
    // whose cloud is packed (into a staging buffer of its own) while this one is on the link and being sorted
    rsreg_host_timing &ht = ctx->host_timing;
    rc = upload_packed(ctx, ctx->h_stage_src, ctx->ev_stage_src, ctx->d_src_raw, points, n, stride, ctx->stream, &ht.source_pack, &ht.source_stage_wait);
    if (rc) return rc;
    return load_source(ctx, ctx->d_src_raw.as<char>(), n, 12);
}

int rsreg_icp_begin(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params)
{
    if (!ctx || !params) return RSREG_ERR_INVALID_ARG;
    if (!ctx->have_target) return fail(ctx, RSREG_ERR_NO_TARGET, "rsreg_icp_set_target not called");
    if (!ctx->have_source) return fail(ctx, RSREG_ERR_NO_SOURCE, "rsreg_icp_set_source not called");
    if (params->max_iterations < 0 || !(params->max_correspondence_distance >= 0))
        return fail(ctx, RSREG_ERR_INVALID_ARG, "bad ICP parameters");
    if (params->max_correspondence_distance > ctx->gate_built_for)
        return fail(ctx, RSREG_ERR_INVALID_ARG, "max_correspondence_distance exceeds the one the target index was built for");
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    {
        int rcj = join_source(ctx);
        if (rcj) return rcj;
    }
    if (ctx->grid.dense == 2 && (ctx->n_work > kScanMaxSource || filters_on(*params))) {
        // the target was set for a handful of queries (scan_target); this alignment needs the index after all
        if (!ctx->scan_raw) return fail(ctx, RSREG_ERR_NO_TARGET, "the target cloud was released before its index was built");
        int rcb = build_grid(ctx, ctx->scan_raw, ctx->n_target_raw, ctx->scan_stride, ctx->gate_built_for);
        if (rcb) return rcb;
    }
    IcpState &s = ctx->icp;
    s = IcpState();
    s.prm = *params;
    // the rings must cover THIS call's gate (it may be smaller than the one built for)
    s.final_t = Mat4f::identity();
    if (guess) std::memcpy(s.final_t.m, guess, sizeof(s.final_t.m));
    s.t_inc = Mat4f::identity();
    s.prev_mse = DBL_MAX;
    s.active = 1;
    ctx->ev_used = 0;
    ctx->ev_nn.clear();
    ctx->ev_reduce.clear();
    ctx->ev_transform.clear();
    ctx->ev_allreduce.clear();
    s.restart_pending = true;   // input_transformed = guess * input (App. A.2 prologue): ensure_restarted / launch_fused
    return RSREG_OK;
}

int rsreg_icp_search(rsreg_ctx *ctx, int32_t *index_out, float *sqr_dist_out)
{
    if (!ctx) return RSREG_ERR_INVALID_ARG;
    if (!ctx->icp.active) return fail(ctx, RSREG_ERR_STATE, "rsreg_icp_begin not called");
    int rc = apply_pending_transform(ctx);
    if (rc) return rc;
    rc = launch_search(ctx);
    if (rc) return rc;
    const size_t n = ctx->n_source;
    if ((index_out || sqr_dist_out) && n) {
        RSREG_HIP(ctx, ctx->d_tmp.reserve(n * 8 + 32));
        int *d_idx = ctx->d_tmp.as<int>();
        float *d_d2 = reinterpret_cast<float *>(d_idx + n);
        if (filters_on(ctx->icp.prm))
            k_export_corr_w<<<div_up((uint32_t)n, kBlock), kBlock, 0, ctx->stream>>>(
                ctx->d_corr_pos.as<int>(), ctx->d_corr_d2.as<float>(), ctx->d_corr_w.as<uint32_t>(), ctx->d_tgt_sorted.as<float4>(),
                ctx->d_perm.as<uint32_t>(), ctx->d_uniq_of.as<uint32_t>(), ctx->d_first.as<uint32_t>(), (uint32_t)n, d_idx, d_d2);
        else
            k_export_corr<<<div_up((uint32_t)n, kBlock), kBlock, 0, ctx->stream>>>(
                ctx->d_corr_pos.as<int>(), ctx->d_corr_d2.as<float>(), ctx->d_tgt_sorted.as<float4>(), ctx->d_perm.as<uint32_t>(),
                ctx->d_uniq_of.as<uint32_t>(), (uint32_t)n, d_idx, d_d2);
        RSREG_HIP(ctx, hipGetLastError());
        if (index_out) RSREG_HIP(ctx, hipMemcpyAsync(index_out, d_idx, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (sqr_dist_out) RSREG_HIP(ctx, hipMemcpyAsync(sqr_dist_out, d_d2, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RSREG_OK;
}

int rsreg_icp_sums(rsreg_ctx *ctx, double sums[RSREG_NUM_SUMS])
{
    if (!ctx || !sums) return RSREG_ERR_INVALID_ARG;
    if (!ctx->icp.active) return fail(ctx, RSREG_ERR_STATE, "rsreg_icp_begin not called");
    if (!ctx->icp.have_search) return fail(ctx, RSREG_ERR_STATE, "rsreg_icp_search not called for this iteration");
    return launch_sums(ctx, sums, false);
}

int rsreg_umeyama_from_sums(const double sums[RSREG_NUM_SUMS], float t_out[16])
{
    if (!sums || !t_out) return RSREG_ERR_INVALID_ARG;
    Mat4f T;
    if (!umeyama_from_sums(sums, T)) return RSREG_ERR_INVALID_ARG;
    std::memcpy(t_out, T.m, sizeof(T.m));
    return RSREG_OK;
}

int rsreg_icp_update(rsreg_ctx *ctx, const double sums[RSREG_NUM_SUMS], float *t_inc_out, int *done)
{
    if (!ctx || !sums || !done) return RSREG_ERR_INVALID_ARG;
    if (!ctx->icp.active) return fail(ctx, RSREG_ERR_STATE, "rsreg_icp_begin not called");
    int rc = update_from_sums(ctx, sums, done);
    if (rc) return rc;
    if (t_inc_out) std::memcpy(t_inc_out, ctx->icp.t_inc.m, 64);
    return RSREG_OK;
}

namespace {
int icp_end(rsreg_ctx *ctx, rsreg_icp_result *result, void *aligned_out, size_t out_stride, const void *source_records);
}

int rsreg_icp_end(rsreg_ctx *ctx, rsreg_icp_result *result, void *aligned_out, size_t out_stride)
{
    return icp_end(ctx, result, aligned_out, out_stride, nullptr);
}

namespace {
// source_records (nullable): the caller's source records, out_stride bytes each -- every record of aligned_out is then the
// WHOLE source record with xyz rewritten (what PCL's align(output) leaves: output = input, then xyz <- final * xyz), copied by
// the host threads that write the aligned positions anyway
int icp_end(rsreg_ctx *ctx, rsreg_icp_result *result, void *aligned_out, size_t out_stride, const void *source_records)
{
    if (!ctx) return RSREG_ERR_INVALID_ARG;
    IcpState &s = ctx->icp;
    if (!s.active) return fail(ctx, RSREG_ERR_STATE, "rsreg_icp_begin not called");
    const size_t n = ctx->n_source;
    if (aligned_out && n) {
        if (out_stride < 12) return RSREG_ERR_INVALID_ARG;
        RSREG_HIP(ctx, ctx->d_tmp.reserve(n * 12 + 16));
        RSREG_HIP(ctx, ctx->h_stage.reserve(n * 12 + 16));
        k_apply_final<<<div_up((uint32_t)n, kBlock), kBlock, 0, ctx->stream>>>(ctx->d_src_all.as<float4>(), (uint32_t)n, to_mat34(s.final_t),
                                                                                ctx->d_perm.as<uint32_t>(), ctx->d_tmp.as<float>());
        RSREG_HIP(ctx, hipGetLastError());
        // home in pieces: while piece k + 1 is on the link, piece k is written into the caller's records by the host's threads
        const auto t0 = std::chrono::steady_clock::now();
        const size_t piece = (size_t)1 << 18;
        const size_t pieces = (n + piece - 1) / piece;
        while (ctx->ev_home.size() < pieces) {
            hipEvent_t e;
            RSREG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->ev_home.push_back(e);
        }
        for (size_t k = 0; k < pieces; ++k) {
            const size_t lo = k * piece, hi = std::min(n, lo + piece);
            RSREG_HIP(ctx, hipMemcpyAsync(ctx->h_stage.as<char>() + lo * 12, ctx->d_tmp.as<char>() + lo * 12, (hi - lo) * 12, hipMemcpyDeviceToHost, ctx->stream));
            RSREG_HIP(ctx, hipEventRecord(ctx->ev_home[k], ctx->stream));
        }
        const float *src = ctx->h_stage.as<float>();
        char *dst = static_cast<char *>(aligned_out);
        const char *rec = static_cast<const char *>(source_records);
        // (the caller's records first, all of them, while the positions are still on the link: the positions then land in lines
        // the cores already own)
        // (icp_align has started that copy on a thread of its own, beside the iterations: it ends here)
        if (ctx->records_copy.joinable()) ctx->records_copy.join();
        else if (rec && rec != dst)
            host_parallel_for(n, [=](size_t a, size_t b) { std::memcpy(dst + a * out_stride, rec + a * out_stride, (b - a) * out_stride); });
        for (size_t k = 0; k < pieces; ++k) {
            const size_t lo = k * piece, hi = std::min(n, lo + piece);
            RSREG_HIP(ctx, hipEventSynchronize(ctx->ev_home[k]));
            host_parallel_for(hi - lo, [=](size_t a, size_t b) {
                const float one = 1.0f;
                for (size_t i = lo + a; i < lo + b; ++i) {
                    std::memcpy(dst + i * out_stride, src + 3 * i, 12);
                    if (out_stride >= 16) std::memcpy(dst + i * out_stride + 12, &one, 4);
                }
            });
        }
        ctx->host_timing.aligned_copy = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    // (the usual end of an alignment: the last thing this thread did was wait for the sums, and nothing has been queued since)
    if (!(s.idle_after_sums && !(aligned_out && n))) RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)target_counts(ctx, false);
#ifdef RSREG_DIAG
    if (const char *sd_path = tunables().dump_seed) {   // dev: the position every query matched last, and the queries
        const size_t nq = ctx->n_work;
        std::vector<int> h(nq);
        std::vector<float> hq(nq * 4);
        (void)hipMemcpy(h.data(), ctx->d_seed.ptr, nq * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hq.data(), ctx->d_cur.ptr, nq * 16, hipMemcpyDeviceToHost);
        if (FILE *f = std::fopen(sd_path, "wb")) {
            std::fwrite(h.data(), 4, nq, f);
            std::fwrite(hq.data(), 4, nq * 4, f);
            const size_t np_ = ctx->grid.n_points;
            std::vector<float> hp(np_ * 4);
            (void)hipMemcpy(hp.data(), ctx->d_tgt_sorted.ptr, np_ * 16, hipMemcpyDeviceToHost);
            std::fwrite(hp.data(), 4, np_ * 4, f);
            std::fclose(f);
        }
    }
    if (const char *wt_path = tunables().wave_times) {
        if (ctx->grid.dense == 1 && ctx->n_work) {
            const bool light = tunables().wave_times_light;
            const size_t nw = light ? (size_t)(ctx->icp.sched_ready ? ctx->icp.sched_items : reduce_blocks(ctx->n_work)) * kTileWaves
                                    : (ctx->n_work + 63) / 64;
            std::vector<unsigned long long> h(16 * nw + (light ? 0 : (ctx->n_work + 1) / 2));   // wave records, then a uint32 of step counts per lane
            (void)hipMemcpy(h.data(), ctx->d_brick.ptr, h.size() * 8, hipMemcpyDeviceToHost);
            if (FILE *f = std::fopen(wt_path, "wb")) {
                std::fwrite(h.data(), 8, h.size(), f);
                std::fclose(f);
            }
        }
    }
#endif
    if (result) {
        std::memset(result, 0, sizeof(*result));
        std::memcpy(result->transform, s.final_t.m, 64);
        result->converged = s.converged;
        result->state = s.state;
        result->iterations = s.iterations;
        result->n_correspondences = s.ncorr;
        result->mse = s.cur_mse;
        std::memcpy(result->sums_last, s.sums_last, sizeof(s.sums_last));
        result->n_nn_launches = s.n_nn_launches;
        result->n_scheduled_launches = s.n_sched_launches;
        if (ctx->profiling) {
            result->ms_nn = sum_events(ctx, ctx->ev_nn);
#ifdef RSREG_DIAG
            if (tunables().dump_nn_ms) {   // diagnostic builds: duration of every search launch of this call
                std::fprintf(stderr, "[rsreg] search launches (us):");
                for (auto &p : ctx->ev_nn) {
                    float t = 0;
                    (void)hipEventElapsedTime(&t, ctx->ev_pool[p.first], ctx->ev_pool[p.second]);
                    std::fprintf(stderr, " %.1f", t * 1e3);
                }
                std::fprintf(stderr, "\n");
            }
#endif
            result->ms_reduce = sum_events(ctx, ctx->ev_reduce);
            if (ctx->ev_reduce.empty())   // fused pipelines: everything between two consecutive search kernels
                for (size_t k = 0; k + 1 < ctx->ev_nn.size(); ++k) {
                    float t = 0;
                    if (hipEventElapsedTime(&t, ctx->ev_pool[ctx->ev_nn[k].second], ctx->ev_pool[ctx->ev_nn[k + 1].first]) == hipSuccess)
                        result->ms_reduce += t;
                }
            result->ms_transform = sum_events(ctx, ctx->ev_transform);
            result->ms_allreduce = sum_events(ctx, ctx->ev_allreduce);   // (already inside ms_reduce in the fused pipelines)
            result->ms_total = result->ms_nn + result->ms_reduce + result->ms_transform;
        }
    }
    s.active = 0;
    return RSREG_OK;
}
}  // namespace

int rsreg_ctx_host_timing(rsreg_ctx *ctx, rsreg_host_timing *out)
{
    if (!ctx || !out) return RSREG_ERR_INVALID_ARG;
    *out = ctx->host_timing;
    return RSREG_OK;
}

namespace {
int icp_align(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params, rsreg_icp_result *result, void *aligned_out, size_t out_stride,
              const void *source_records);
}

int rsreg_icp_align(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params, rsreg_icp_result *result,
                    void *aligned_out, size_t out_stride)
{
    return icp_align(ctx, guess, params, result, aligned_out, out_stride, nullptr);
}

int rsreg_icp_align_records(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params, rsreg_icp_result *result,
                            const void *source_records, void *aligned_out, size_t stride)
{
    if (!source_records || !aligned_out) return RSREG_ERR_INVALID_ARG;
    return icp_align(ctx, guess, params, result, aligned_out, stride, source_records);
}

namespace {
int icp_align(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params, rsreg_icp_result *result, void *aligned_out, size_t out_stride,
              const void *source_records)
{
    const auto t_align0 = std::chrono::steady_clock::now();
    struct AlignClock {   // (whatever way the call ends: begin .. the last iteration + the aligned cloud's way home)
        rsreg_ctx *c;
        std::chrono::steady_clock::time_point t0;
        ~AlignClock() { if (c) c->host_timing.align = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - c->host_timing.aligned_copy; }
    } align_clock{ctx, t_align0};
    if (ctx) ctx->host_timing.aligned_copy = 0;
    // `output = input`: the caller's records go into the output on a thread of their own (the pool's threads under it) while this
    // thread queues and waits for the iterations; icp_end joins it before the aligned positions are written over the records
    struct CopyGuard {   // (an error on the way: the copy still owns aligned_out)
        rsreg_ctx *c;
        ~CopyGuard() { if (c && c->records_copy.joinable()) c->records_copy.join(); }
    } copy_guard{ctx};
    if (ctx && source_records && aligned_out && source_records != aligned_out && ctx->n_source && out_stride >= 12) {
        const size_t n_rec = ctx->n_source;
        const char *rec = static_cast<const char *>(source_records);
        char *dst = static_cast<char *>(aligned_out);
        ctx->records_copy = std::thread([=] {
            host_parallel_for(n_rec, [=](size_t a, size_t b) { std::memcpy(dst + a * out_stride, rec + a * out_stride, (b - a) * out_stride); });
        });
    }
    int rc = rsreg_icp_begin(ctx, guess, params);
    if (rc) return rc;
    int done = 0;
    double sums[RSREG_NUM_SUMS];
    const bool filtered = filters_on(*params);   // the optional correspondence filters run between the staged kernels
    const bool scan = ctx->grid.dense == 2;   // no index: the staged kernels (search over the whole target, sums)
    const bool fused = !filtered && !scan && (params->pipeline_mode == RSREG_PIPELINE_FUSED || params->pipeline_mode == RSREG_PIPELINE_DEVICE_LOOP);
    if (!filtered && !scan && params->pipeline_mode == RSREG_PIPELINE_DEVICE_LOOP && params->criteria_mode == RSREG_CRITERIA_FIXED) {
        rc = run_device_loop(ctx);
        if (rc) return rc;
        done = 1;
    }
    while (!done) {
        if (fused) {
            rc = launch_fused(ctx, sums, false);
        } else {
            rc = apply_pending_transform(ctx);
            if (!rc) rc = launch_search(ctx);
            if (!rc) rc = launch_sums(ctx, sums, true);
        }
        if (rc) return rc;
        rc = update_from_sums(ctx, sums, &done);
        if (rc) return rc;
    }
    return icp_end(ctx, result, aligned_out, out_stride, source_records);
}
}  // namespace

int rsreg_icp_grid_info(rsreg_ctx *ctx, rsreg_grid_info *info)
{
    if (!ctx || !info) return RSREG_ERR_INVALID_ARG;
    if (!ctx->have_target) return RSREG_ERR_NO_TARGET;
    {
        int rcc = target_counts(ctx, true);   // (a counting build set_target did not wait for)
        if (rcc) return rcc;
    }
#ifdef RSREG_DIAG
    const bool want_max = tunables().grid_stats;   // (diagnostic builds: a pass over the occupied cells, only on request)
#else
    const bool want_max = false;
#endif
    if (want_max && ctx->grid.dense == 1 && ctx->grid_info.max_points_per_cell == 0 && ctx->grid.n_points > 0) {
        const size_t total = (size_t)(ctx->grid.dims[0] + 2) * (ctx->grid.dims[1] + 2) * (ctx->grid.dims[2] + 2);
        uint32_t *d = ctx->d_misc.as<uint32_t>() + 20;
        uint32_t h = 0;
        RSREG_HIP(ctx, hipMemsetAsync(d, 0, 4, ctx->stream));
        (void)total;
        k_dense_max_count<<<256, kBlock, 0, ctx->stream>>>(ctx->d_cellpos.as<uint32_t>(), ctx->d_misc.as<uint32_t>() + 8, d);
        RSREG_HIP(ctx, hipGetLastError());
        RSREG_HIP(ctx, hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, ctx->stream));
        RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->grid_info.max_points_per_cell = h;
    }
    {
        int rcj = join_source(ctx);
        if (rcj) return rcj;
    }
    *info = ctx->grid_info;
    info->n_source_distinct = ctx->have_source ? (uint32_t)ctx->n_work : 0u;
    return RSREG_OK;
}

// pcl::transformPointCloud: whole records copied, xyz rewritten
int rsreg_transform_cloud(rsreg_ctx *ctx, const void *in, void *out, size_t n, size_t stride, int is_dense,
                          const float transform[16])
{
    (void)is_dense;  // non-finite points are left unchanged either way (xform of NaN stays NaN-free here)
    if (!ctx || !transform || (n && (!in || !out)) || stride < 12) return RSREG_ERR_INVALID_ARG;
    if (n == 0) return RSREG_OK;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = pack_to_stage(ctx, in, n, stride);
    if (rc) return rc;
    RSREG_HIP(ctx, ctx->d_tmp.reserve(n * 12 + n * 16 + 64));
    char *d_raw = ctx->d_tmp.as<char>();
    float4 *d_pts = reinterpret_cast<float4 *>(d_raw + ((n * 12 + 15) & ~size_t(15)));
    RSREG_HIP(ctx, ctx->d_tmp.reserve(((n * 12 + 15) & ~size_t(15)) + n * 16 + 64));
    d_raw = ctx->d_tmp.as<char>();
    d_pts = reinterpret_cast<float4 *>(d_raw + ((n * 12 + 15) & ~size_t(15)));
    RSREG_HIP(ctx, hipMemcpyAsync(d_raw, ctx->h_stage.ptr, n * 12, hipMemcpyHostToDevice, ctx->stream));
    Mat4f T;
    std::memcpy(T.m, transform, 64);
    k_gather_source<uint32_t><<<div_up((uint32_t)n, kBlock), kBlock, 0, ctx->stream>>>(d_raw, 12, (uint32_t)n, nullptr, d_pts, nullptr);
    RSREG_HIP(ctx, hipGetLastError());
    k_apply_final<<<div_up((uint32_t)n, kBlock), kBlock, 0, ctx->stream>>>(d_pts, (uint32_t)n, to_mat34(T), nullptr,
                                                                            reinterpret_cast<float *>(d_raw));
    RSREG_HIP(ctx, hipGetLastError());
    RSREG_HIP(ctx, hipMemcpyAsync(ctx->h_stage.ptr, d_raw, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const char *src = static_cast<const char *>(in);
    char *dst = static_cast<char *>(out);
    const float *xyz = ctx->h_stage.as<float>();
    host_parallel_for(n, [=](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            if (dst != src) std::memmove(dst + i * stride, src + i * stride, stride);
            std::memcpy(dst + i * stride, xyz + 3 * i, 12);
        }
    });
    return RSREG_OK;
}

}  // extern "C"
