// rsreg_ctx.hpp — the per-(device, stream) context behind the C ABI (include/rsreg.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <thread>
#include <vector>

#include "../../include/rsreg.h"
#include "host_linalg.hpp"
#include "tunables.hpp"
#include "workers.hpp"

namespace rsreg {

// Growable device allocation; never shrinks, so steady-state calls allocate nothing.
struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&ptr, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

// Buffers of dropped device clouds, kept for the clouds to come (cloud.hip).  `limit`: bytes kept at most
// (RSREG_CLOUD_POOL_MB, default 4096; 0 = every drop is a hipFree).
struct CloudPool {
    struct Slot {
        void *ptr;
        size_t cap;
    };
    std::vector<Slot> slots;
    size_t held = 0;
    size_t limit = (size_t)tunables().cloud_pool_mb << 20;
};

struct PinnedBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

// Bounding box and number of the finite points of a cloud: what an index build or a source load starts from (one kernel
// pair and a round trip to the host).  A cloud handle keeps the box of its records as they are (cloud.hip: version), so a
// frame that was the source of one pair and is the target of the next is not measured twice.
// the way back from the order-preserving uint the kernels keep float minima / maxima in (icp_kernels.hpp: float_ordered)
inline float ordered_float(uint32_t u)
{
    const uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    std::memcpy(&f, &v, 4);
    return f;
}

struct CloudBox {
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    uint32_t nfin = 0;
    bool valid = false;
    // false: a box that CONTAINS the finite points without being measured on them (the transformed corners of a measured box,
    // a union with such a box).  Good enough for a target's index (the grid's origin and extent; the search is exact whatever
    // they are), not for a source's spatial order (its keys are quantised from the box: another box, another order of the sums)
    bool exact = true;
};

// a box around what `T` makes of everything inside `b` (eight corners, in double, widened by a margin far above the float
// rounding of the per-point transform): cloud.hip hands it to the output of a transform / an alignment
inline CloudBox transformed_box(const CloudBox &b, const float *T /* 4x4 column-major */)
{
    CloudBox o = b;
    if (!b.valid || b.nfin == 0) return o;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, big = 0;
    for (int c = 0; c < 8; ++c) {
        const double p[3] = {(c & 1) ? b.mx[0] : b.mn[0], (c & 2) ? b.mx[1] : b.mn[1], (c & 4) ? b.mx[2] : b.mn[2]};
        for (int r = 0; r < 3; ++r) {
            const double v = (double)T[r] * p[0] + (double)T[4 + r] * p[1] + (double)T[8 + r] * p[2] + (double)T[12 + r];
            lo[r] = v < lo[r] ? v : lo[r];
            hi[r] = v > hi[r] ? v : hi[r];
            big = std::fabs(v) > big ? std::fabs(v) : big;
        }
    }
    const double margin = 1e-5 * big + 1e-6;   // (float rounding of R x + t is ~1e-7 of the coordinates' size)
    for (int r = 0; r < 3; ++r) {
        o.mn[r] = (float)(lo[r] - margin);
        o.mx[r] = (float)(hi[r] + margin);
        if (!(o.mn[r] == o.mn[r]) || !(o.mx[r] == o.mx[r]) || std::fabs(o.mn[r]) > 1e30f || std::fabs(o.mx[r]) > 1e30f) o.valid = false;   // (a transform that is not finite)
    }
    o.exact = false;
    return o;
}

inline CloudBox union_box(const CloudBox &a, const CloudBox &b)
{
    if (!a.valid || !b.valid) return CloudBox{};
    if (a.nfin == 0) return b;
    if (b.nfin == 0) return a;
    CloudBox o;
    for (int r = 0; r < 3; ++r) {
        o.mn[r] = a.mn[r] < b.mn[r] ? a.mn[r] : b.mn[r];
        o.mx[r] = a.mx[r] > b.mx[r] ? a.mx[r] : b.mx[r];
    }
    const unsigned long long nf = (unsigned long long)a.nfin + b.nfin;
    o.nfin = (uint32_t)nf;
    o.valid = nf < 0xfffffff0ull;
    o.exact = a.exact && b.exact;
    return o;
}

// Uniform-grid index over the target cloud, all device-resident (layout: DESIGN.md §3).
struct GridParams {
    float origin[3];
    float inv_cell;
    float cell;
    int dims[3];
    uint32_t table_mask;   // brick hash table has table_mask + 1 slots
    uint32_t n_points;     // unique finite target points in `sorted`
    uint32_t n_cells;      // occupied cells
    uint32_t n_bricks;     // occupied 4x4x4-cell bricks
    int max_ring;          // rings (cells) needed to cover the correspondence gate
    int dense;             // 1: dense cell-start table (d_dense), 0: brick hash (d_table + d_cellpos), 2: no index (scan_target)
    int xbits;             // dense: bits of the x position inside a cell in the sort key (16, or what a 32-bit key leaves)
    int table_sparse;      // dense: only the table entries of occupied cells (and the slot behind each) are valid
    int have_nbr;          // dense: the neighbourhood occupancy words are built (0: a counting build for a small source left them out, icp.hip)
};

struct BrickEntry {          // 32 B, one hash-table slot
    unsigned long long key;  // packed brick coordinate, ~0 = empty
    unsigned long long mask; // occupancy of its 64 cells, bit = lz*16 + ly*4 + lx
    uint32_t base;           // id of its first occupied cell (cellpos index)
    uint32_t pad[3];
};

struct IcpState {
    rsreg_icp_params prm;
    Mat4f final_t, t_inc;
    int iterations = 0, state = 0, converged = 0, similar = 0, active = 0;
    double prev_mse = 0, cur_mse = 0;
    uint64_t ncorr = 0;
    double sums_last[RSREG_NUM_SUMS] = {0};
    double svd_v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // V of the previous Umeyama solve (warm start of the next)
    bool have_search = false;     // corr buffers hold the current iteration's search
    bool pending_transform = false;  // fused mode: t_inc not yet applied to d_cur
    bool restart_pending = false;    // d_cur = guess * d_src (and "no seeds") not done yet: ensure_restarted / launch_fused
    double ms_nn = 0, ms_reduce = 0, ms_transform = 0;
    int n_nn_launches = 0;
    int n_sched_launches = 0;      // ... of them launched from a tile schedule
    int fused_launches = 0;        // fused dense launches of this alignment so far
    bool sched_ready = false;      // d_sched holds a tile schedule for this alignment
    uint32_t sched_items = 0;      // workgroups of a scheduled launch
    bool sched_carried = false;    // ... whose schedule an earlier alignment of the context built
    bool idle_after_sums = false;  // the caller's thread has waited for the main stream (the sums) and queued nothing since: rsreg_icp_end need not wait again
};

}  // namespace rsreg

struct rsreg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    std::string last_error;
    std::mutex error_mutex;

    // ---- ICP target index
    bool have_target = false;
    rsreg::GridParams grid{};
    rsreg_grid_info grid_info{};
    double gate_built_for = 0;
    rsreg::DevBuf d_tgt_raw;      // packed xyz of the caller's target (n x float3)
    rsreg::DevBuf d_tgt_sorted;   // float4 {x,y,z,bits(orig index)}, cell-sorted, de-duplicated
    rsreg::DevBuf d_table;        // BrickEntry[table_mask+1]
    rsreg::DevBuf d_cellpos;      // uint32[n_cells+1]: first sorted point of each occupied cell
    rsreg::DevBuf d_dense;        // dense mode: uint32[(nx+2)(ny+2)(nz+2)+1] first sorted point of EVERY cell, followed by one uint32 per cell: occupancy of its 27-cell neighbourhood
    rsreg::DevBuf d_pos_of;       // dense mode: uint32 per target record, its position in d_tgt_sorted
    rsreg::DevBuf d_sched;        // tile schedule of the fused dense kernel: items (4 per tile) | wave costs | done counters | sort scratch
    uint32_t sched_cap_tiles = 0;        // tiles the buffer was laid out for (the arrays' offsets)
    uint32_t sched_keep_items = 0, sched_keep_tiles = 0;   // the last schedule built in it: workgroups, tiles of its source (0: none)
    int sched_keep_age = 0;              // alignments it has served since
    // the same for the FIRST launch of an alignment (unseeded, from the source itself: another cost profile), built from a timed
    // first launch and kept beside the steady one
    uint32_t sched_first_items = 0, sched_first_tiles = 0;
    int sched_first_age = 0;
    rsreg::DevBuf d_keys, d_keys_alt, d_vals, d_vals_alt, d_flags, d_scan, d_brick, d_tmp;
    rsreg::DevBuf d_misc;         // small: bbox, counters
    rsreg::DevBuf d_cnt;          // counting build (cellsort.hpp): points per table slot, all zero between builds
    rsreg::DevBuf d_arrived;      // ... and the records in arrival order, cell by cell
    void *cnt_zero_ptr = nullptr;
    size_t cnt_zero_cap = 0;
    bool cnt_flip = false;        // which of the two sets of per-span totals the next counting build fills
    bool cnt_dirty = false;       // a counting build is under way (or did not finish): the counts are not known to be zero
    // The two counts a counting build ends with (occupied cells, records in the sorted array) lie in pinned words; nothing the
    // searches do needs them on the host (they bound the record array by the number of finite points), so set_target does
    // not wait for the build: whoever wants the counts, or has just waited for the stream anyway, takes them over (icp.hip:
    // target_counts).
    bool counts_pending = false;
    size_t counts_total = 0, counts_n = 0;   // (table entries and input points of that build: for index_bytes)
    size_t n_target_raw = 0;

    // ---- ICP source
    bool have_source = false;
    const struct rsreg_cloud *src_cloud = nullptr;   // set by rsreg_icp_set_source_cloud: where the aligned cloud's records come from
    uint64_t src_cloud_id = 0, src_cloud_version = 0;   // ... as it was then: a handle destroyed or rewritten since is refused (RSREG_ERR_STATE)
    size_t n_source = 0;          // source points handed in
    size_t n_work = 0;            // distinct source points the iteration works on (exact copies merged)
    // the source is loaded on a stream of its own (so that it runs beside the target's index build when the
    // caller sets the source first, as the reference does) and joined where the alignment begins
    hipStream_t stream_src = nullptr;
    hipEvent_t ev_src_done = nullptr, ev_main = nullptr;
    bool src_pending = false;
    rsreg::DevBuf d_plain_ticket; // k_source_plain's ticket word (zero between launches)
    uint32_t plain_box_seq = 0, plain_box_counter = 0;   // the stamp the pending plain load leaves behind its box in h_smisc[48 .. 55] (0: none)
    bool src_plain = false;       // the pending load is k_source_plain on the MAIN stream (every record a query of its own): nothing to wait for at the join
    bool src_on_worker = false;   // ... and its launches are being queued by the context's worker thread right now
    rsreg::SourceWorker *src_worker = nullptr;   // (created with the first source load; RSREG_NO_WORKER=1: never, the load runs on the caller's thread)
    // everything of a pending source load has been queued on stream_src (so that ev_src_done is the event of THIS load)
    int source_enqueued() { return src_worker ? src_worker->wait() : 0; }
    rsreg::DevBuf d_skeys, d_skeys_alt, d_svals, d_sflags, d_sscan, d_stmp, d_smisc;   // its scratch (the target build has its own)
    rsreg::DevBuf d_shist;        // two sets of digit histograms of the source's sort, used in turn (k_source_keys_hist); the worker's
    bool shist_flip = false, shist_dirty = false;
    rsreg::PinnedBuf h_smisc;
    rsreg::DevBuf d_src_all;      // float4 {x,y,z,valid} of every source point, spatially sorted
    rsreg::DevBuf d_uniq_of;      // uint32: sorted position -> distinct point id
    rsreg::DevBuf d_first;        // uint32[n_work+1]: first sorted position of each distinct point
    rsreg::DevBuf d_src_raw;      // packed xyz as handed in
    rsreg::DevBuf d_perm;         // uint32: spatially sorted position -> caller's index
    rsreg::DevBuf d_src;          // float4 {x,y,z,weight} of the distinct points, spatially sorted
    rsreg::DevBuf d_cur;          // float4 current (transformed) source
    rsreg::DevBuf d_corr_pos;     // int32: position in d_tgt_sorted, -1 = none
    rsreg::DevBuf d_corr_d2;      // float
    rsreg::DevBuf d_seed;         // int32: nearest target found in the previous iteration (-1 none)
    rsreg::DevBuf d_partials;     // double[blocks][17]
    rsreg::DevBuf d_sums;         // double[17]
    rsreg::DevBuf d_icp_state;    // IcpDevState of the device-resident loop
    // optional correspondence filters (reciprocal correspondences, trimmed rejector)
    rsreg::DevBuf d_corr_w;       // uint32 per distinct source point: how many of its copies keep their match
    rsreg::DevBuf d_recip_pts;    // float4 per ORIGINAL source point: the current source in the caller's order
    rsreg_ctx *recip = nullptr;   // child context holding the index over the current source (reciprocal search)
    rsreg::PinnedBuf h_sums;      // pinned double[64]
    // next_*: the box of the cloud about to be set, when its handle knows it (consumed by build_grid / load_source_queue);
    // last_*: what the last index build / source load started from (valid: computed or taken over)
    rsreg::CloudBox next_tgt_box, last_tgt_box, next_src_box, last_src_box;
    rsreg::CloudBox next_ndt_box, last_ndt_box;   // the same for the NDT target's grid (rsreg_ndt_set_target_cloud)

    rsreg::PinnedBuf h_stage;     // pinned staging for H2D / D2H of clouds
    // host clouds of rsreg_icp_set_source / _set_target: a staging buffer each, the upload stream, the event behind each buffer's last copy
    rsreg::PinnedBuf h_stage_src, h_stage_tgt;
    hipStream_t stream_h2d = nullptr;
    hipEvent_t ev_stage_src = nullptr, ev_stage_tgt = nullptr;
    rsreg_host_timing host_timing{};
    std::vector<hipEvent_t> ev_home;   // one per piece of an aligned cloud on its way to the host (rsreg_icp_end)
    rsreg::IcpState icp;

    // ---- ApproximateVoxelGrid on the device (voxel.hip)
    rsreg::DevBuf d_vox_in, d_vox_out, d_vox_cent;
    // rsreg_cloud_filter_async: scratch sets and streams of their own, so that the filters of the next frames (one wave
    // per long run, latency-bound: 0.4 ms for PCL's default 1 m leaf) run under the alignment of this one AND beside each
    // other; a set is used again in turn, behind the filter that used it last (same stream).  ev_side_gate lets a filter
    // start after what the main stream holds.  Round 6: TWO side workers (threads) -- a frame's extraction and filter are ~20
    // launches and two waits for counts, as long as the frame's two alignments on the caller's thread, and every other
    // frame waited for them; jobs that do not depend on each other alternate between the workers, a job whose input is the
    // output of a job still queued follows it on the same worker.  Worker w owns the sets w and w + 2 and uses them in turn.
    static constexpr int kSideWorkers = 2;
    static constexpr int kSideSets = 2 * kSideWorkers;
    struct SideSet {
        hipStream_t stream = nullptr;
        rsreg::DevBuf out, keys, keys_alt, vals, vals_alt, flags, scan, cent, misc, tmp;
        rsreg::PinnedBuf host;
    } side_sets[kSideSets];
    int side_turn[kSideWorkers] = {0, 0};   // which of its two sets a worker's next job takes
    int side_rr = 0;                          // the worker of the next job that follows no other
    hipEvent_t ev_side_gate = nullptr;
    rsreg::TicketWorker *side_workers[kSideWorkers] = {nullptr, nullptr};   // (queue those jobs: rsreg_cloud_filter_async returns at once)

    // ---- NDT
    bool have_ndt_target = false;
    double ndt_resolution = 0;
    int ndt_centroid_mode = 0;              // 1: PCL's float running sum per voxel (rsreg_ndt_set_centroid_mode)
    int ndt_n_voxels = 0;
    rsreg::DevBuf d_scan_keys;           // no-index search (scan_target): one (distance, index) key per source point
    const char *scan_raw = nullptr;      // ... and the records the index is built from if an alignment needs it after all
    size_t scan_stride = 0;
    uint64_t tgt_cloud_id = 0, tgt_cloud_version = 0;   // the device cloud the ICP target index was built from (0: none)
    rsreg::CloudPool cloud_pool;
    std::vector<hipEvent_t> ev_copy;   // one per piece of a cloud download in flight (cloud.hip)
    // rsreg_cloud_upload_async: a copy stream, two pinned staging buffers used in turn, the event behind the last copy
    // out of each, and the event that lets the copy stream start only after what the main stream holds
    hipStream_t stream_copy = nullptr;
    hipEvent_t ev_copy_gate = nullptr, ev_up[2] = {nullptr, nullptr};
    rsreg::PinnedBuf h_up[2];
    bool up_busy[2] = {false, false};   // (these two: the upload worker's, once it exists)
    int up_next = 0;
    rsreg::TicketWorker *up_worker = nullptr;
    // rsreg_cloud_download_async: a download stream, three pinned staging buffers with an event each, the copy-out thread
    hipStream_t stream_down = nullptr;
    hipEvent_t ev_down_gate = nullptr, ev_down[rsreg::DownloadWorker::kSlots] = {};
    rsreg::PinnedBuf h_down[rsreg::DownloadWorker::kSlots];
    int down_next = 0;
    rsreg::DownloadWorker *down_worker = nullptr;
    uint64_t ndt_seq = 0;  // derivative passes launched; the final reduce stamps it into h_ndt
    rsreg::DevBuf d_ndt_vox;      // per voxel: 3 mean + 9 icov doubles + centroid float3 ...
    rsreg::DevBuf d_ndt_src, d_ndt_trans, d_ndt_partials, d_ndt_out;
    rsreg::DevBuf d_ndt_ctl;      // a line search in one launch (ndt_kernels.hpp: NdtLsCtl): its state, the next pass's parameters, its counters
    bool ndt_ls_failed = false;   // ... ran into one of its bounded waits once: the host advances the searches of this context from then on
    rsreg::DevBuf d_ndt_seg;      // first sorted point of every occupied leaf (NDT's own: d_cellpos belongs to the live ICP hash index)
    hipEvent_t ev_ndt[2] = {nullptr, nullptr};   // NDT's own event pair (the pool's indices belong to an ICP begin..end)
    std::vector<double> ndt_mean_cov_icov;   // 21 per voxel (host copy)
    std::vector<int> ndt_counts;
    std::vector<float> ndt_centroid;         // 3 per voxel
    rsreg::PinnedBuf h_ndt;
    rsreg::PinnedBuf h_ndt_build;   // the target build's transfers: the voxels' partial moments home, the finished table out (pageable copies cost a frame of the NDT-edge loop 0.1 ms)

    // ---- RCCL
    // rsreg_ctx_prepare: what a frame loop is about to need, made on a thread of its own while the caller goes on; whoever is
    // about to create one of these lazily joins that thread first (prep_join) and finds them there
    std::thread prep_thread;
    rsreg::DevBuf prep_model;     // a device buffer for the merged model, handed to the cloud pool at the join
    int prep_rc = 0;
    void prep_join();
    std::thread records_copy;     // rsreg_icp_align_records: the caller's source records on their way into aligned_out (joined by icp_end)
    void *comm = nullptr;         // ncclComm_t
    int rank = 0, nranks = 1;
    rsreg::DevBuf d_comm;

    // ---- profiling events
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<size_t, size_t>> ev_nn, ev_reduce, ev_transform, ev_allreduce;
};

namespace rsreg {

void cloud_pool_clear(rsreg_ctx *ctx);   // cloud.hip: frees the buffers kept in ctx->cloud_pool

inline int fail(rsreg_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->error_mutex);   // (the context's source worker thread reports through here too)
        ctx->last_error = what;
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return code;
}

// Host-side record loops (32-byte AoS records <-> packed xyz in pinned staging) are memory-bound
// copies of tens of MB: split over the process's pool of host threads (workers.hpp).  f(lo, hi) handles records [lo, hi).
template <typename F> inline void host_parallel_for(size_t n, F f)
{
    const size_t pieces = std::min<size_t>(n / 32768 + 1, 4 * (host_pool().th.size() + 1));
    const std::function<void(size_t, size_t)> fn = f;
    host_pool().run(n, pieces, fn);
}

// 32-byte (or any stride) AoS records -> packed xyz in the pinned staging buffer
inline int pack_to_stage(rsreg_ctx *ctx, const void *points, size_t n, size_t stride)
{
    hipError_t e = ctx->h_stage.reserve(n * 12 + 16);
    if (e != hipSuccess) return fail(ctx, RSREG_ERR_ALLOC, "pinned staging", e);
    float *dst = ctx->h_stage.as<float>();
    const char *src = static_cast<const char *>(points);
    host_parallel_for(n, [=](size_t lo, size_t hi) {
        if (stride == 12) {
            std::memcpy(dst + 3 * lo, src + 12 * lo, (hi - lo) * 12);
        } else {
            for (size_t i = lo; i < hi; ++i) std::memcpy(dst + 3 * i, src + i * stride, 12);
        }
    });
    return RSREG_OK;
}

#define RSREG_HIP(ctx, expr)                                              \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) return rsreg::fail((ctx), RSREG_ERR_HIP, #expr, _e); \
    } while (0)

}  // namespace rsreg
