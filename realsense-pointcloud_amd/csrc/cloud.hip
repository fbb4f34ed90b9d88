// cloud.hip — clouds that stay in HBM across the reference's frame loop (C ABI: include/rsreg.h,
// "device-resident clouds").
//
// The reference's schemes run, per frame, ApproximateVoxelGrid::filter -> align (-> align) ->
// transformPointCloud x2 -> operator+ (src/icp_edge_based_registration.hpp:75-76,95-120,
// src/ndt_edge_based_registration.hpp:68-108, src/incremental_icp.hpp:54-64) on host clouds.
// With cloud handles each of those steps takes and leaves its clouds in HBM: a frame is uploaded
// once, the merged cloud is downloaded once, nothing else crosses PCIe (and no 32 -> 12 byte packing
// on the host: the kernels read the 32-byte records as they are, through their stride).
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "records.hpp"

using namespace rsreg;

struct rsreg_cloud {
    rsreg_ctx *ctx = nullptr;
    DevBuf buf;
    size_t n = 0, stride = 32;
    uint32_t width = 0, height = 1;
    int is_dense = 0;
    // rsreg_cloud_upload_async: the copy that fills this cloud may still be on the link
    hipEvent_t ev_filled = nullptr;
    mutable bool filling = false;
    uint64_t up_ticket = 0;   // the upload worker's job that stages the records and queues their copy
    // rsreg_cloud_filter_async: the side worker's job that runs the filter INTO this cloud; until it has run, n / width are
    // not known (resolve)
    mutable bool filter_pending = false;
    uint64_t filter_ticket = 0;
    int filter_worker = 0;   // which side worker of the context runs (ran) that job
    int filter_rc = 0;
    // rsreg_cloud_download_async: a copy of these records to the host may still be reading them
    hipEvent_t ev_down = nullptr;
    mutable bool downloading = false;
    // which cloud this is and how often its records have been rewritten: an index built from (id, version) is still
    // good while both are unchanged (rsreg_icp_set_target_cloud)
    uint64_t id = 0, version = 0;
    // bounding box and finite count of the records of version `box_version`, once an index build or a source load has
    // measured them (rsreg_ctx.hpp: CloudBox): the frame that was a pair's source is the next pair's target
    mutable rsreg::CloudBox box;
    mutable uint64_t box_version = ~0ull;
};

extern "C" int rsreg_icp_set_target_scan_(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride, double max_correspondence_distance);   // icp.hip
constexpr size_t kScanSourceLimit = 64;       // (= kScanMaxSource of icp_kernels.hpp) source points at most, and ...
constexpr size_t kScanTargetFloor = 32768;    // ... target points at least, for the search without an index

namespace rsreg {
int edge_features_device(rsreg_ctx *ctx, const char *d_rec, size_t stride, uint32_t width, uint32_t height, uint32_t *n_out, int side_set);   // edges.hip
int voxel_filter_device(rsreg_ctx *ctx, const char *d_in, uint32_t N, size_t stride, const float leaf[3], uint32_t *n_out, int side_set);   // voxel.hip
}

namespace {

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// pcl::transformPointCloud / the aligned cloud of icp.align(): whole records copied, xyz <- T * xyz for
// finite points; set_w: data[3] = 1 like Registration::align does.  in == out allowed.
__global__ __launch_bounds__(kBlock) void k_records_transform(const char *in, char *out, uint32_t n, size_t stride, Mat34 T, int set_w)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(in + (size_t)i * stride);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)i * stride);
    const float x = __uint_as_float(src[0]), y = __uint_as_float(src[1]), z = __uint_as_float(src[2]);
    if (out != in)
        for (uint32_t k = 3; k < stride / 4; ++k) dst[k] = src[k];
    float3 t = make_float3(x, y, z);
    if (finite3(x, y, z)) t = xform(T, x, y, z);
    dst[0] = __float_as_uint(t.x);
    dst[1] = __float_as_uint(t.y);
    dst[2] = __float_as_uint(t.z);
    if (set_w && stride >= 16) dst[3] = __float_as_uint(1.0f);
}

// The same for PointXYZRGB's own layout (32-byte records, 16-byte aligned: SURVEY.md App. A.0): two lanes per record, each
// moving one 16-byte half with one load and one store -- a wave reads and writes 1 KiB runs instead of 64 words 32 bytes
// apart, eight times over.  The lane of the first half transforms; the lane of the second half only copies (and has
// nothing to do in place).
__global__ __launch_bounds__(kBlock) void k_records_transform32(const uint4 *in, uint4 *out, uint32_t n_halves, Mat34 T, int set_w)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_halves) return;
    if (h & 1u) {
        if (out != in) out[h] = in[h];
        return;
    }
    uint4 r = in[h];
    const float x = __uint_as_float(r.x), y = __uint_as_float(r.y), z = __uint_as_float(r.z);
    if (finite3(x, y, z)) {
        const float3 t = xform(T, x, y, z);
        r.x = __float_as_uint(t.x);
        r.y = __float_as_uint(t.y);
        r.z = __float_as_uint(t.z);
    }
    if (set_w) r.w = __float_as_uint(1.0f);
    out[h] = r;
}

hipError_t launch_records_transform(hipStream_t st, const char *in, char *out, uint32_t n, size_t stride, const Mat34 &T, int set_w)
{
    if (stride == 32 && ((uintptr_t)in & 15u) == 0 && ((uintptr_t)out & 15u) == 0 && n < 0x7fffffffu)
        k_records_transform32<<<div_up(2u * n, kBlock), kBlock, 0, st>>>(reinterpret_cast<const uint4 *>(in), reinterpret_cast<uint4 *>(out), 2u * n, T, set_w);
    else
        k_records_transform<<<div_up(n, kBlock), kBlock, 0, st>>>(in, out, n, stride, T, set_w);
    return hipGetLastError();
}

// Whoever reads or rewrites a cloud first waits (on the host) for an upload of it that is still in flight: by then a
// frame prefetched one step of the frame loop earlier has long arrived, and a host wait is right whatever stream the
// reader works on.
// A cloud a filter is still being queued INTO (rsreg_cloud_filter_async) has no size yet: whoever wants to know it, or
// to touch the records, waits for the side worker to have run that job.
hipError_t resolve(const rsreg_cloud *c)
{
    if (!c || !c->filter_pending) return hipSuccess;
    c->filter_pending = false;
    (void)c->ctx->side_workers[c->filter_worker]->wait(c->filter_ticket);
    return c->filter_rc ? hipErrorUnknown : hipSuccess;   // (the job has left its message in the context: fail())
}

hipError_t settle(const rsreg_cloud *c)
{
    if (!c) return hipSuccess;
    {
        const hipError_t e = resolve(c);
        if (e != hipSuccess) return e;
    }
    if (c->downloading) {   // whatever the main stream does to the cloud next comes after the copy that is reading it
        c->downloading = false;
        hipError_t e = hipStreamWaitEvent(c->ctx->stream, c->ev_down, 0);
        if (e != hipSuccess) return e;
    }
    if (!c->filling) return hipSuccess;
    c->filling = false;
    // first the worker has to have staged the records and queued their copy (the event is recorded behind it) ...
    const int e = c->ctx->up_worker ? c->ctx->up_worker->wait(c->up_ticket) : 0;
    if (e) return (hipError_t)e;
    return hipEventSynchronize(c->ev_filled);   // ... then the copy has to have arrived
}

// A dropped cloud's buffer goes to the context's pool and the next cloud takes it from there (rsreg_ctx.hpp, CloudPool):
// the frame loops create and drop half a dozen clouds per frame, and every hipFree is a device-wide synchronisation
// (0.16 ms).  Every kernel and copy that touches a cloud runs on ctx->stream, so a buffer handed on is written only
// after the work queued on its previous owner; the one other reader is a source load on ctx->stream_src, which the main
// stream is made to wait for before a buffer changes hands.
void cloud_drop(rsreg_ctx *ctx, DevBuf &b)
{
    if (!b.ptr) return;
    // a target set without an index (scan_target) still reads the cloud's own records, and rsreg_icp_begin may build the
    // index from them later: once the buffer changes hands that target is gone
    if (ctx->scan_raw && ctx->scan_raw >= static_cast<const char *>(b.ptr) && ctx->scan_raw < static_cast<const char *>(b.ptr) + b.cap) {
        ctx->scan_raw = nullptr;
        if (ctx->grid.dense == 2) ctx->have_target = false;
    }
    CloudPool &pool = ctx->cloud_pool;
    if (b.cap <= pool.limit && pool.held + b.cap <= pool.limit) {
        if (ctx->src_pending) { (void)ctx->source_enqueued(); (void)hipStreamWaitEvent(ctx->stream, ctx->ev_src_done, 0); }
        pool.slots.push_back({b.ptr, b.cap});
        pool.held += b.cap;
        b.ptr = nullptr;
        b.cap = 0;
        return;
    }
    // the buffer is freed: whatever the source worker still has to queue reads it (load_source_queue on stream_src), so
    // that work has to be queued AND finished first -- hipFree only waits for what is already on a stream
    if (ctx->src_pending) { (void)ctx->source_enqueued(); (void)hipEventSynchronize(ctx->ev_src_done); }
    (void)hipStreamSynchronize(ctx->stream);
    b.release();
}

// like DevBuf::reserve (the contents are not kept), through the pool: the smallest kept buffer that is large enough
// and not more than twice too large
// `growing`: the buffer of a cloud that keeps growing (`model += moved`): the LARGEST kept buffer that fits, however large --
// rsreg_ctx_prepare may have left one there for the whole model
hipError_t cloud_reserve(rsreg_ctx *ctx, DevBuf &b, size_t bytes, bool growing = false)
{
    if (bytes <= b.cap) return hipSuccess;
    ctx->prep_join();   // (rsreg_ctx_prepare may be holding a buffer for the pool)
    cloud_drop(ctx, b);
    CloudPool &pool = ctx->cloud_pool;
    size_t best = pool.slots.size();
    for (size_t i = 0; i < pool.slots.size(); ++i) {
        if (pool.slots[i].cap < bytes) continue;
        if (growing ? (best == pool.slots.size() || pool.slots[i].cap > pool.slots[best].cap)
                    : (pool.slots[i].cap <= 2 * bytes + (1u << 20) && (best == pool.slots.size() || pool.slots[i].cap < pool.slots[best].cap)))
            best = i;
    }
    if (best != pool.slots.size()) {
        b.ptr = pool.slots[best].ptr;
        b.cap = pool.slots[best].cap;
        pool.held -= b.cap;
        pool.slots[best] = pool.slots.back();
        pool.slots.pop_back();
        return hipSuccess;
    }
    hipError_t e = b.reserve(bytes);
    if (e != hipSuccess && !pool.slots.empty()) {   // out of memory with buffers kept aside: give them back and try once more
        (void)hipGetLastError();
        rsreg::cloud_pool_clear(ctx);
        e = b.reserve(bytes);
    }
    return e;
}

// A download comes over PCIe in pieces, and a piece is copied out of the pinned staging buffer while the next ones are
// still in flight (the merged cloud of 16 frames, 157 MB: 5.1 -> 4.0 ms).  f(p) handles piece p; pieces are dealt
// round-robin to a few threads.
constexpr size_t kPieceMin = (size_t)1 << 20, kPieceMax = (size_t)8 << 20;
inline size_t piece_bytes(size_t bytes) { return std::min(kPieceMax, std::max(kPieceMin, bytes / 32)); }

template <typename F> void parallel_pieces(size_t n_pieces, F f)
{
    static const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const unsigned nt = (unsigned)std::min<size_t>(hw, n_pieces);
    std::vector<std::thread> th;
    th.reserve(nt ? nt - 1 : 0);
    for (unsigned t = 1; t < nt; ++t)
        th.emplace_back([=] {
            for (size_t p = t; p < n_pieces; p += nt) f(p);
        });
    for (size_t p = 0; p < n_pieces; p += std::max(1u, nt)) f(p);
    for (auto &t : th) t.join();
}

int check_pair(const rsreg_ctx *ctx, const rsreg_cloud *a, const rsreg_cloud *b)
{
    if (!ctx || !a || !b || a->ctx != ctx || b->ctx != ctx) return RSREG_ERR_INVALID_ARG;
    return RSREG_OK;
}

}  // namespace

namespace rsreg {
void cloud_pool_clear(rsreg_ctx *ctx)
{
    CloudPool &pool = ctx->cloud_pool;
    if (pool.slots.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (const CloudPool::Slot &sl : pool.slots) (void)hipFree(sl.ptr);
    pool.slots.clear();
    pool.held = 0;
}
}  // namespace rsreg

void rsreg_ctx::prep_join()
{
    if (!prep_thread.joinable()) return;
    prep_thread.join();
    if (prep_model.ptr) {   // (the pool is the caller's thread's: the buffer joins it here)
        cloud_pool.slots.push_back({prep_model.ptr, prep_model.cap});
        cloud_pool.held += prep_model.cap;
        prep_model.ptr = nullptr;
        prep_model.cap = 0;
    }
}

extern "C" {

// What the frame loops of the schemes need besides the main stream, requested AHEAD of the need (incremental_icp.hpp:51-66,
// icp_edge_based_registration.hpp:71-123: the first registration() of a process -- main.cpp:85 -- otherwise creates them one by
// one on its critical path): the upload / download / source / side streams (the process's hardware queues: 12 ms each for
// the first four, profiles/r06_cold_run.txt), the pinned staging buffers of the upload and download workers for frames of
// `frame_bytes`, and ONE device buffer of `model_bytes` for a cloud that grows to that size (the merged model: no
// re-allocation while it grows).  Returns at once; a thread of the context makes them while the caller goes on (reading its
// frames, allocating its result), and every entry point that would create one of them waits for that thread first.
int rsreg_ctx_prepare(rsreg_ctx *ctx, size_t frame_bytes, size_t model_bytes, unsigned flags)
{
    if (!ctx) return RSREG_ERR_INVALID_ARG;
    ctx->prep_join();
    ctx->prep_rc = 0;
    const bool want_side = (flags & 1u) != 0;
    // (a second registration() of the process: the model's buffer of the first is in the pool already)
    for (const CloudPool::Slot &sl : ctx->cloud_pool.slots)
        if (sl.cap >= model_bytes + 16) model_bytes = 0;
    ctx->prep_thread = std::thread([ctx, frame_bytes, model_bytes, want_side] {
        auto ok = [&](hipError_t e) { if (e != hipSuccess && !ctx->prep_rc) ctx->prep_rc = (int)e; return e == hipSuccess; };
        if (!ok(hipSetDevice(ctx->device))) return;
        // in the order a frame loop needs them: the upload's stream and staging, the source's stream, the download's, the side sets'
        if (!ctx->stream_copy) {
            if (!ok(hipStreamCreateWithFlags(&ctx->stream_copy, hipStreamNonBlocking))) return;
            ok(hipEventCreateWithFlags(&ctx->ev_copy_gate, hipEventDisableTiming));
            for (hipEvent_t &e : ctx->ev_up) ok(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (frame_bytes) { ok(ctx->h_up[0].reserve(frame_bytes)); ok(ctx->h_up[1].reserve(frame_bytes)); }
        if (!ctx->stream_src) {
            if (!ok(hipStreamCreateWithFlags(&ctx->stream_src, hipStreamNonBlocking))) return;
            ok(hipEventCreateWithFlags(&ctx->ev_src_done, hipEventDisableTiming));
            ok(hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
        }
        if (!ctx->stream_down) {
            if (!ok(hipStreamCreateWithFlags(&ctx->stream_down, hipStreamNonBlocking))) return;
            ok(hipEventCreateWithFlags(&ctx->ev_down_gate, hipEventDisableTiming));
            for (hipEvent_t &e : ctx->ev_down) ok(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (frame_bytes) for (auto &h : ctx->h_down) ok(h.reserve(frame_bytes));
        if (want_side)
            for (auto &ss : ctx->side_sets)
                if (!ss.stream) ok(hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking));
        if (model_bytes) ok(ctx->prep_model.reserve(model_bytes + 16));
    });
    return RSREG_OK;
}

int rsreg_cloud_create(rsreg_ctx *ctx, rsreg_cloud **out)
{
    if (!ctx || !out) return RSREG_ERR_INVALID_ARG;
    rsreg_cloud *c = new (std::nothrow) rsreg_cloud();
    if (!c) return RSREG_ERR_ALLOC;
    static std::atomic<uint64_t> next_id{1};
    c->ctx = ctx;
    c->id = next_id.fetch_add(1);
    *out = c;
    return RSREG_OK;
}

int rsreg_cloud_destroy(rsreg_cloud *c)
{
    if (!c) return RSREG_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)settle(c);
    if (c->ev_filled) (void)hipEventDestroy(c->ev_filled);
    if (c->ev_down) (void)hipEventDestroy(c->ev_down);
    if (c->ctx->src_cloud == c) c->ctx->src_cloud = nullptr;   // (rsreg_icp_align_cloud then refuses to write an aligned cloud)
    cloud_drop(c->ctx, c->buf);
    delete c;
    return RSREG_OK;
}

int rsreg_cloud_upload(rsreg_cloud *c, const void *points, size_t n, size_t stride, uint32_t width, uint32_t height, int is_dense)
{
    if (!c || (n && !points) || stride < 12 || (stride & 3)) return RSREG_ERR_INVALID_ARG;
    rsreg_ctx *ctx = c->ctx;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, settle(c));
    RSREG_HIP(ctx, cloud_reserve(ctx, c->buf, n * stride + 16));
    if (n) {
        // through pinned staging, copied by a few threads (a pageable hipMemcpy of tens of MB is several times slower).
        // (Staging and PCIe copy in overlapping pieces gained nothing: 0.34 ms for a 9.8 MB frame is the link's rate.)
        RSREG_HIP(ctx, ctx->h_stage.reserve(n * stride));
        char *stage = ctx->h_stage.as<char>();
        const char *src = static_cast<const char *>(points);
        host_parallel_for(n, [=](size_t lo, size_t hi) { rsreg::stream_copy(stage + lo * stride, src + lo * stride, (hi - lo) * stride); });
        RSREG_HIP(ctx, hipMemcpyAsync(c->buf.ptr, stage, n * stride, hipMemcpyHostToDevice, ctx->stream));
        RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the staging buffer is reused by the next call
    }
    c->version++;
    c->n = n;
    c->stride = stride;
    c->width = width;
    c->height = height;
    c->is_dense = is_dense;
    return RSREG_OK;
}

// rsreg_cloud_upload that returns as soon as the records are in a pinned staging buffer of their own: the PCIe copy
// runs on the context's copy stream beside whatever the main stream is doing (the frame loops upload frame k + 1 while
// frame k is being aligned).  Every entry point that reads or rewrites the cloud waits for the copy first (settle).
namespace {

// rsreg_cloud_upload_async / _deferred: the caller's thread makes room, orders the copy stream behind what the buffer's
// previous owner has queued, and hands the rest to the context's upload worker: staging the records in pinned memory
// (0.2 ms for a 9.8 MB frame), queueing the PCIe copy and recording the events behind it.  `wait_staged`: return only
// when `points` has been read.
int upload_on_worker(rsreg_cloud *c, const void *points, size_t n, size_t stride, uint32_t width, uint32_t height, int is_dense, bool wait_staged)
{
    if (!c || (n && !points) || stride < 12 || (stride & 3)) return RSREG_ERR_INVALID_ARG;
    rsreg_ctx *ctx = c->ctx;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, settle(c));
    RSREG_HIP(ctx, cloud_reserve(ctx, c->buf, n * stride + 16));
    if (n) {
        ctx->prep_join();
        if (!ctx->stream_copy) {
            RSREG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_copy, hipStreamNonBlocking));
            RSREG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_copy_gate, hipEventDisableTiming));
            for (hipEvent_t &e : ctx->ev_up) RSREG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (!ctx->up_worker) ctx->up_worker = new rsreg::TicketWorker();
        if (!c->ev_filled) RSREG_HIP(ctx, hipEventCreateWithFlags(&c->ev_filled, hipEventDisableTiming));
        // the buffer may come from the pool: work queued on its previous owner (main stream) goes first
        RSREG_HIP(ctx, hipEventRecord(ctx->ev_copy_gate, ctx->stream));
        RSREG_HIP(ctx, hipStreamWaitEvent(ctx->stream_copy, ctx->ev_copy_gate, 0));
        if (ctx->src_pending) { (void)ctx->source_enqueued(); RSREG_HIP(ctx, hipStreamWaitEvent(ctx->stream_copy, ctx->ev_src_done, 0)); }
        char *dst = c->buf.as<char>();
        const char *src = static_cast<const char *>(points);
        hipEvent_t ev_filled = c->ev_filled;
        const size_t bytes = n * stride;
        c->up_ticket = ctx->up_worker->post([ctx, dst, src, n, stride, bytes, ev_filled]() -> int {
            hipError_t e = hipSetDevice(ctx->device);
            if (e != hipSuccess) return (int)e;
            const int slot = ctx->up_next;
            ctx->up_next ^= 1;
            if (ctx->up_busy[slot]) {   // the copy that last used this staging buffer
                if ((e = hipEventSynchronize(ctx->ev_up[slot])) != hipSuccess) return (int)e;
                ctx->up_busy[slot] = false;
            }
            if ((e = ctx->h_up[slot].reserve(bytes)) != hipSuccess) return (int)e;
            char *stage = ctx->h_up[slot].as<char>();
            host_parallel_for(n, [=](size_t lo, size_t hi) { rsreg::stream_copy(stage + lo * stride, src + lo * stride, (hi - lo) * stride); });
            if ((e = hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, ctx->stream_copy)) != hipSuccess) return (int)e;
            if ((e = hipEventRecord(ctx->ev_up[slot], ctx->stream_copy)) != hipSuccess) return (int)e;
            ctx->up_busy[slot] = true;
            return (int)hipEventRecord(ev_filled, ctx->stream_copy);
        });
        c->filling = true;
    }
    c->version++;
    c->n = n;
    c->stride = stride;
    c->width = width;
    c->height = height;
    c->is_dense = is_dense;
    if (n && wait_staged) {
        const int e = ctx->up_worker->wait(c->up_ticket);
        if (e) return fail(ctx, RSREG_ERR_HIP, "an asynchronous upload failed", (hipError_t)e);
    }
    return RSREG_OK;
}

}  // namespace

int rsreg_cloud_upload_async(rsreg_cloud *c, const void *points, size_t n, size_t stride, uint32_t width, uint32_t height, int is_dense)
{
    return upload_on_worker(c, points, n, stride, width, height, is_dense, true);
}

// rsreg_cloud_upload_async that returns before `points` has been read: the frame loops hand over frames that stay where
// they are for the whole registration (types.hpp:19: the caller's vector of clouds), two frames ahead of the one being
// aligned, and the 0.2 ms it takes to stage a frame no longer sit on the caller's thread with the GPU idle.
int rsreg_cloud_upload_deferred(rsreg_cloud *c, const void *points, size_t n, size_t stride, uint32_t width, uint32_t height, int is_dense)
{
    const bool on_caller = rsreg::tunables().upload_wait_staged;   // (dev: A/B)
    return upload_on_worker(c, points, n, stride, width, height, is_dense, on_caller);
}

int rsreg_cloud_download(const rsreg_cloud *c, void *out, size_t capacity)
{
    if (!c) return RSREG_ERR_INVALID_ARG;
    rsreg_ctx *ctx = c->ctx;
    RSREG_HIP(ctx, resolve(c));
    if ((c->n && !out) || capacity < c->n) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, settle(c));
    if (!c->n) return RSREG_OK;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = c->n * c->stride, piece = piece_bytes(bytes), n_pieces = (bytes + piece - 1) / piece;
    RSREG_HIP(ctx, ctx->h_stage.reserve(bytes));
    while (ctx->ev_copy.size() < n_pieces) {
        hipEvent_t e = nullptr;
        RSREG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->ev_copy.push_back(e);
    }
    char *stage = ctx->h_stage.as<char>();
    const char *src = c->buf.as<char>();
    for (size_t p = 0; p < n_pieces; ++p) {   // all pieces queued; an event behind each
        const size_t off = p * piece, len = std::min(piece, bytes - off);
        RSREG_HIP(ctx, hipMemcpyAsync(stage + off, src + off, len, hipMemcpyDeviceToHost, ctx->stream));
        RSREG_HIP(ctx, hipEventRecord(ctx->ev_copy[p], ctx->stream));
    }
    char *dst = static_cast<char *>(out);
    std::atomic<int> err{(int)hipSuccess};
    const int device = ctx->device;
    hipEvent_t *ev = ctx->ev_copy.data();
    parallel_pieces(n_pieces, [&, stage, dst, bytes, piece, device, ev](size_t p) {   // a piece is copied out while the next ones arrive
        const size_t off = p * piece, len = std::min(piece, bytes - off);
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipEventSynchronize(ev[p]);
        if (e != hipSuccess) { err.store((int)e); return; }
        std::memcpy(dst + off, stage + off, len);
    });
    RSREG_HIP(ctx, (hipError_t)err.load());
    return RSREG_OK;
}

int rsreg_cloud_info(const rsreg_cloud *c, size_t *n, size_t *stride, uint32_t *width, uint32_t *height, int *is_dense)
{
    if (!c) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(c->ctx, resolve(c));
    if (n) *n = c->n;
    if (stride) *stride = c->stride;
    if (width) *width = c->width;
    if (height) *height = c->height;
    if (is_dense) *is_dense = c->is_dense;
    return RSREG_OK;
}

const void *rsreg_cloud_device_ptr(const rsreg_cloud *c)
{
    if (!c || settle(c) != hipSuccess) return nullptr;   // (the pointer may be used on any stream: an upload in flight is waited for)
    return c->buf.ptr;
}

// (internal, edges.hip) the cloud takes a copy of the first n records of `buf`
int rsreg_cloud_adopt_(rsreg_cloud *c, DevBuf *buf, size_t n, size_t stride, uint32_t width, uint32_t height, int is_dense)
{
    rsreg_ctx *ctx = c->ctx;
    RSREG_HIP(ctx, settle(c));
    RSREG_HIP(ctx, cloud_reserve(ctx, c->buf, n * stride + 16));
    if (n) RSREG_HIP(ctx, hipMemcpyAsync(c->buf.ptr, buf->ptr, n * stride, hipMemcpyDeviceToDevice, ctx->stream));
    c->version++;
    c->n = n; c->stride = stride; c->width = width; c->height = height; c->is_dense = is_dense;
    return RSREG_OK;
}

int rsreg_cloud_copy(rsreg_ctx *ctx, const rsreg_cloud *in, rsreg_cloud *out)
{
    int rc = check_pair(ctx, in, out);
    if (rc) return rc;
    RSREG_HIP(ctx, settle(in));
    RSREG_HIP(ctx, settle(out));
    if (in == out) return RSREG_OK;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, cloud_reserve(ctx, out->buf, in->n * in->stride + 16));
    if (in->n) RSREG_HIP(ctx, hipMemcpyAsync(out->buf.ptr, in->buf.ptr, in->n * in->stride, hipMemcpyDeviceToDevice, ctx->stream));
    out->version++;
    out->n = in->n; out->stride = in->stride; out->width = in->width; out->height = in->height; out->is_dense = in->is_dense;
    return RSREG_OK;
}

// pcl::ApproximateVoxelGrid::filter (incremental_icp.hpp:54-55, icp_edge...hpp:59-60,75-76): in == out allowed
int rsreg_cloud_filter(rsreg_ctx *ctx, const rsreg_cloud *in, const float leaf[3], rsreg_cloud *out)
{
    int rc = check_pair(ctx, in, out);
    if (rc || !leaf) return rc ? rc : RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, resolve(in));
    if (!(leaf[0] > 0) || !(leaf[1] > 0) || !(leaf[2] > 0) || in->stride < 20) return RSREG_ERR_INVALID_ARG;
    if (in->n > 0x7ffffff0ull) return fail(ctx, RSREG_ERR_INVALID_ARG, "cloud too large");
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, settle(in));
    RSREG_HIP(ctx, settle(out));
    uint32_t nr = 0;
    const size_t stride = in->stride;
    rc = voxel_filter_device(ctx, in->buf.as<char>(), (uint32_t)in->n, stride, leaf, &nr, -1);
    if (rc) return rc;
    RSREG_HIP(ctx, cloud_reserve(ctx, out->buf, (size_t)nr * stride + 16));   // (in == out: the input has been consumed by now)
    if (nr) RSREG_HIP(ctx, hipMemcpyAsync(out->buf.ptr, ctx->d_vox_out.ptr, (size_t)nr * stride, hipMemcpyDeviceToDevice, ctx->stream));
    out->version++;
    out->n = nr; out->stride = stride; out->width = nr; out->height = 1; out->is_dense = 0;
    return RSREG_OK;
}

// rsreg_cloud_filter on the context's side stream, with scratch of its own: the call returns once the number of output
// records is known (a short host wait on the side stream), with the runs' sums -- for PCL's default 1 m leaf one wave
// adding 10^5 floats one after the other -- still running; whoever touches `out` next waits for them (settle).  The
// frame loops filter frame k + 1 this way before they align frame k.  `in` must stay alive and unchanged until `out`
// has been used; in != out.
namespace {

// The side worker of a job and its next scratch set, the set's stream behind what the main stream holds so far (`in` may
// have been produced there, and the buffer `out` is about to get may come from the pool with work of its previous owner
// queued).  follow >= 0: the worker that runs the job this one's input comes from (jobs of one worker run in the order
// they were posted); -1: the workers take turns.  A worker's two sets take turns: a job queues behind the one that used
// its set last (same stream) and runs beside the others.
int side_begin(rsreg_ctx *ctx, int follow, int *worker_out, int *set_out)
{
    int worker = follow;
    if (worker < 0) {
        worker = tunables().one_side_worker ? 0 : ctx->side_rr;
        ctx->side_rr = (ctx->side_rr + 1) % rsreg_ctx::kSideWorkers;
    }
    const int set = worker + rsreg_ctx::kSideWorkers * ctx->side_turn[worker];
    ctx->side_turn[worker] ^= 1;
    rsreg_ctx::SideSet &ss = ctx->side_sets[set];
    ctx->prep_join();
    if (!ss.stream) RSREG_HIP(ctx, hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking));
    if (!ctx->ev_side_gate) RSREG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_side_gate, hipEventDisableTiming));
    if (!ctx->side_workers[worker]) ctx->side_workers[worker] = new rsreg::TicketWorker();
    RSREG_HIP(ctx, hipEventRecord(ctx->ev_side_gate, ctx->stream));
    RSREG_HIP(ctx, hipStreamWaitEvent(ss.stream, ctx->ev_side_gate, 0));
    if (ctx->src_pending) { (void)ctx->source_enqueued(); RSREG_HIP(ctx, hipStreamWaitEvent(ss.stream, ctx->ev_src_done, 0)); }
    *worker_out = worker;
    *set_out = set;
    return RSREG_OK;
}

// what a side job needs to know about an input that may still be on its way: an upload in flight (the job waits for the
// upload worker to have queued the copy, then makes its stream wait for it) -- captured on the caller's thread
struct InFlight {
    bool filling;
    uint64_t ticket;
    hipEvent_t ev;
};
InFlight in_flight_of(const rsreg_cloud *c) { return InFlight{c->filling, c->up_ticket, c->ev_filled}; }
int side_wait_input(rsreg_ctx *ctx, const InFlight &f, hipStream_t st)
{
    if (!f.filling || !f.ev) return RSREG_OK;
    // (a third party's wait: a failed upload stays for the poster's own settle() to report, and this job does not run on
    // a buffer that was never filled)
    if (ctx->up_worker && ctx->up_worker->peek(f.ticket)) return fail(ctx, RSREG_ERR_HIP, "the upload of this job's input cloud failed");
    RSREG_HIP(ctx, hipStreamWaitEvent(st, f.ev, 0));
    return RSREG_OK;
}

}  // namespace

// rsreg_cloud_filter queued by the context's side worker on a scratch set and a stream of its own: the call returns at
// once; the number of output records is known when the job has run (every call that takes `out` waits for that first),
// the runs' sums -- for PCL's default 1 m leaf one wave adding 10^5 floats one after the other -- may then still be
// running (settle).  The frame loops filter the next frames this way while they align this one.  `in` must stay alive and
// unchanged until `out` has been used; in != out.  `in` may itself be the output of a side job that has not run yet
// (rsreg_cloud_edge_features_async, then this): the jobs run in the order they were posted.
int rsreg_cloud_filter_async(rsreg_ctx *ctx, const rsreg_cloud *in, const float leaf[3], rsreg_cloud *out)
{
    int rc = check_pair(ctx, in, out);
    if (rc || !leaf || in == out) return rc ? rc : RSREG_ERR_INVALID_ARG;
    const bool chained = in->filter_pending;   // its size is not known yet: the job reads it
    if (!(leaf[0] > 0) || !(leaf[1] > 0) || !(leaf[2] > 0) || in->stride < 20) return RSREG_ERR_INVALID_ARG;
    if (!chained && in->n > 0x7ffffff0ull) return fail(ctx, RSREG_ERR_INVALID_ARG, "cloud too large");
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    if (!chained) RSREG_HIP(ctx, settle(in));
    RSREG_HIP(ctx, settle(out));
    int set = 0, worker = 0;
    rc = side_begin(ctx, chained ? in->filter_worker : -1, &worker, &set);
    if (rc) return rc;
    const size_t stride = in->stride;
    // the filter's twenty launches and its round trip for the number of output records are a third of a frame's host time
    // in the frame loops: they run on the side worker's thread.  The output cannot be larger than the input: room for that
    // is made here (the buffer pool is the caller's thread's).
    RSREG_HIP(ctx, cloud_reserve(ctx, out->buf, (chained ? in->buf.cap : in->n * stride) + 16));
    if (!out->ev_filled) RSREG_HIP(ctx, hipEventCreateWithFlags(&out->ev_filled, hipEventDisableTiming));
    const float l0 = leaf[0], l1 = leaf[1], l2 = leaf[2];
    out->version++;
    out->n = 0; out->stride = stride; out->width = 0; out->height = 1; out->is_dense = 0;
    out->filter_rc = 0;
    out->filter_worker = worker;
    out->filter_ticket = ctx->side_workers[worker]->post([ctx, in, out, chained, stride, l0, l1, l2, set]() -> int {
        rsreg_ctx::SideSet &ws = ctx->side_sets[set];
        const float lf[3] = {l0, l1, l2};
        uint32_t nr = 0;
        int r = hipSetDevice(ctx->device) == hipSuccess ? RSREG_OK : RSREG_ERR_HIP;
        if (!r && chained) {   // the job that makes `in` has run (same thread, posted earlier); its records may still be on their way
            r = in->filter_rc;
            if (!r && in->n && hipStreamWaitEvent(ws.stream, in->ev_filled, 0) != hipSuccess) r = RSREG_ERR_HIP;
        }
        if (!r) r = voxel_filter_device(ctx, in->buf.as<char>(), (uint32_t)in->n, stride, lf, &nr, set);
        if (!r && nr) {
            if (hipMemcpyAsync(out->buf.ptr, ws.out.ptr, (size_t)nr * stride, hipMemcpyDeviceToDevice, ws.stream) != hipSuccess ||
                hipEventRecord(out->ev_filled, ws.stream) != hipSuccess)
                r = fail(ctx, RSREG_ERR_HIP, "queueing the filtered records");
        }
        out->filter_rc = r;
        out->n = r ? 0 : nr;
        out->width = r ? 0 : nr;
        return 0;
    });
    out->filter_pending = true;
    out->filling = true;   // (settle: the event behind the copy; never recorded when the output is empty)
    return RSREG_OK;
}

// rsreg_cloud_edge_features queued by the side worker the same way: the edge schemes extract and filter the features of
// frame k + 1 (eleven launches and a round trip, then the filter's) beside the two alignments of frame k -- the
// reference extracts the features of ALL frames before it registers any (types.hpp:30-43), so nothing of frame k + 1's
// depends on frame k.  `in` (an organized cloud) may still be uploading: the job waits for it, not the caller.
int rsreg_cloud_edge_features_async(rsreg_ctx *ctx, const rsreg_cloud *in, rsreg_cloud *out)
{
    int rc = check_pair(ctx, in, out);
    if (rc || in == out) return rc ? rc : RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, resolve(in));
    if ((size_t)in->width * in->height != in->n || in->stride < 20) return fail(ctx, RSREG_ERR_INVALID_ARG, "edge extraction needs an organized XYZRGB cloud");
    if (in->n > 0x3fffffffull) return fail(ctx, RSREG_ERR_INVALID_ARG, "image too large");
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, settle(out));
    if (in->downloading) RSREG_HIP(ctx, settle(in));   // (rare: only its own download pending needs the caller's stream)
    int set = 0, worker = 0;
    rc = side_begin(ctx, -1, &worker, &set);
    if (rc) return rc;
    const size_t stride = in->stride, n_in = in->n;
    const uint32_t w = in->width, h = in->height;
    RSREG_HIP(ctx, cloud_reserve(ctx, out->buf, n_in * stride + 16));
    if (!out->ev_filled) RSREG_HIP(ctx, hipEventCreateWithFlags(&out->ev_filled, hipEventDisableTiming));
    const InFlight inf = in_flight_of(in);
    const char *src = in->buf.as<char>();
    out->version++;
    out->n = 0; out->stride = stride; out->width = 0; out->height = 1; out->is_dense = in->is_dense;
    out->filter_rc = 0;
    out->filter_worker = worker;
    out->filter_ticket = ctx->side_workers[worker]->post([ctx, out, src, stride, w, h, inf, set]() -> int {
        rsreg_ctx::SideSet &ws = ctx->side_sets[set];
        uint32_t ne = 0;
        int r = hipSetDevice(ctx->device) == hipSuccess ? RSREG_OK : RSREG_ERR_HIP;
        if (!r) r = side_wait_input(ctx, inf, ws.stream);
        if (!r) r = rsreg::edge_features_device(ctx, src, stride, w, h, &ne, set);
        if (!r && ne) {
            if (hipMemcpyAsync(out->buf.ptr, ws.out.ptr, (size_t)ne * stride, hipMemcpyDeviceToDevice, ws.stream) != hipSuccess ||
                hipEventRecord(out->ev_filled, ws.stream) != hipSuccess)
                r = fail(ctx, RSREG_ERR_HIP, "queueing the edge points");
        }
        out->filter_rc = r;
        out->n = r ? 0 : ne;
        out->width = r ? 0 : ne;
        return 0;
    });
    out->filter_pending = true;
    out->filling = true;
    return RSREG_OK;
}

// pcl::transformPointCloud (incremental_icp.hpp:63, icp_edge...hpp:116-117): in == out allowed
int rsreg_cloud_transform(rsreg_ctx *ctx, const rsreg_cloud *in, const float transform[16], rsreg_cloud *out)
{
    int rc = check_pair(ctx, in, out);
    if (rc || !transform) return rc ? rc : RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, settle(in));
    RSREG_HIP(ctx, settle(out));
    if (in != out) RSREG_HIP(ctx, cloud_reserve(ctx, out->buf, in->n * in->stride + 16));
    Mat4f T;
    std::memcpy(T.m, transform, 64);
    if (in->n) {
        RSREG_HIP(ctx, launch_records_transform(ctx->stream, in->buf.as<char>(), out->buf.as<char>(), (uint32_t)in->n, in->stride, to_mat34(T), 0));
    }
    const rsreg::CloudBox moved_box = (in->box.valid && in->box_version == in->version) ? rsreg::transformed_box(in->box, transform) : rsreg::CloudBox{};
    out->version++;
    out->n = in->n; out->stride = in->stride; out->width = in->width; out->height = in->height; out->is_dense = in->is_dense;
    if (moved_box.valid && rsreg::tunables().box_cache) {   // (a box around the moved points, not measured on them: CloudBox::exact)
        out->box = moved_box;
        out->box_version = out->version;
    }
    return RSREG_OK;
}

// pcl::PointCloud::operator+ / += (incremental_icp.hpp:64, icp_edge...hpp:119-120): out = a followed by b;
// out may be a (the append of `target += transformed` then costs only the copy of b) or b
const rsreg_ctx *rsreg_cloud_ctx_(const rsreg_cloud *c) { return c ? c->ctx : nullptr; }   // (internal: edges.hip)

// rsreg_cloud_download that returns at once: the records as they are when the main stream gets here are copied to a pinned
// staging buffer on a download stream and from there to `out` by a thread of the context; the cloud may be rewritten
// or dropped right away (that work is queued behind the copy).  `out` belongs to the copy until
// rsreg_ctx_wait_downloads has returned.  The frame loops download every frame's moved points this way while the next
// frames are aligned: the merged cloud of sixteen 307 k-point frames is 157 MB, 4 ms on the link at the end otherwise.
int rsreg_cloud_download_async(const rsreg_cloud *c, void *out, size_t capacity)
{
    if (!c) return RSREG_ERR_INVALID_ARG;
    rsreg_ctx *ctx = c->ctx;
    RSREG_HIP(ctx, resolve(c));
    if ((c->n && !out) || capacity < c->n) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, settle(c));
    if (!c->n) return RSREG_OK;
    ctx->prep_join();
    if (!ctx->stream_down) {
        RSREG_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_down, hipStreamNonBlocking));
        RSREG_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_down_gate, hipEventDisableTiming));
        for (hipEvent_t &e : ctx->ev_down) RSREG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (!ctx->down_worker) {
        ctx->down_worker = new rsreg::DownloadWorker();
        ctx->down_worker->wait_ready = [](const rsreg::DownloadWorker::Job &j) -> int {
            hipError_t e = hipSetDevice(j.device);
            if (e == hipSuccess) e = hipEventSynchronize(static_cast<hipEvent_t>(j.ev));
            return (int)e;
        };
    }
    rsreg_cloud *mc = const_cast<rsreg_cloud *>(c);
    if (!mc->ev_down) RSREG_HIP(ctx, hipEventCreateWithFlags(&mc->ev_down, hipEventDisableTiming));
    const size_t bytes = c->n * c->stride;
    const int slot = ctx->down_next;
    ctx->down_next = (ctx->down_next + 1) % rsreg::DownloadWorker::kSlots;
    ctx->down_worker->wait_slot(slot);   // (the copy-out that last used this staging buffer)
    {
        // (a failure before the job is posted must give the slot back: nobody else would, and the third download after it
        // would wait for ever)
        hipError_t e = ctx->h_down[slot].reserve(bytes);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_down_gate, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream_down, ctx->ev_down_gate, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_down[slot].ptr, c->buf.ptr, bytes, hipMemcpyDeviceToHost, ctx->stream_down);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev_down[slot], ctx->stream_down);
        if (e == hipSuccess) e = hipEventRecord(mc->ev_down, ctx->stream_down);
        if (e != hipSuccess) {
            ctx->down_worker->release_slot(slot);
            return rsreg::fail(ctx, RSREG_ERR_HIP, "rsreg_cloud_download_async", e);
        }
    }
    c->downloading = true;
    ctx->down_worker->post(rsreg::DownloadWorker::Job{ctx->ev_down[slot], ctx->h_down[slot].as<char>(), static_cast<char *>(out), bytes, slot, ctx->device});
    return RSREG_OK;
}

int rsreg_ctx_wait_downloads(rsreg_ctx *ctx)
{
    if (!ctx) return RSREG_ERR_INVALID_ARG;
    if (!ctx->down_worker) return RSREG_OK;
    const int e = ctx->down_worker->wait_idle();
    if (e) return fail(ctx, RSREG_ERR_HIP, "an asynchronous download failed", (hipError_t)e);
    return RSREG_OK;
}

int rsreg_cloud_version(const rsreg_cloud *c, uint64_t *id, uint64_t *version)
{
    if (!c) return RSREG_ERR_INVALID_ARG;
    if (id) *id = c->id;
    if (version) *version = c->version;
    return RSREG_OK;
}

int rsreg_cloud_concat(rsreg_ctx *ctx, const rsreg_cloud *a, const rsreg_cloud *b, rsreg_cloud *out)
{
    int rc = check_pair(ctx, a, b);
    if (rc || !out || out->ctx != ctx) return rc ? rc : RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, resolve(a));
    RSREG_HIP(ctx, resolve(b));
    if (a->n && b->n && a->stride != b->stride) return fail(ctx, RSREG_ERR_INVALID_ARG, "record strides differ");
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, settle(a));
    RSREG_HIP(ctx, settle(b));
    RSREG_HIP(ctx, settle(out));
    const size_t stride = a->n ? a->stride : b->stride, na = a->n, nb = b->n, total = na + nb;
    const int dense = a->is_dense && b->is_dense;
    // the box of the parts' union, when both parts have one (`target = aligned + target`, icp_edge_based_registration.hpp:119-120:
    // the grown target's index then starts without a measuring launch and its round trip); taken before `out` changes
    auto box_of = [](const rsreg_cloud *c) {
        rsreg::CloudBox b = (c->box.valid && c->box_version == c->version) ? c->box : rsreg::CloudBox{};
        if (c->n == 0) { b = rsreg::CloudBox{}; b.valid = true; b.nfin = 0; }
        return b;
    };
    const rsreg::CloudBox merged_box = rsreg::union_box(box_of(a), box_of(b));
    if (out == a && out->buf.cap >= total * stride + 16) {
        if (nb) RSREG_HIP(ctx, hipMemcpyAsync(out->buf.as<char>() + na * stride, b->buf.ptr, nb * stride, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        DevBuf fresh;
        RSREG_HIP(ctx, cloud_reserve(ctx, fresh, total * stride + (out == a ? total * stride / 2 : 0) + 16, out == a));   // a growing model: room for the next frames
        hipError_t e = hipSuccess;
        if (na) e = hipMemcpyAsync(fresh.ptr, a->buf.ptr, na * stride, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess && nb) e = hipMemcpyAsync(static_cast<char *>(fresh.ptr) + na * stride, b->buf.ptr, nb * stride, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) {
            cloud_drop(ctx, fresh);   // (DevBuf has no destructor: hand the buffer back before reporting the error)
            RSREG_HIP(ctx, e);
        }
        cloud_drop(ctx, out->buf);   // (a or b may be `out`: its old buffer is reused only by work queued after these copies)
        out->buf = fresh;
    }
    out->version++;
    out->n = total; out->stride = stride; out->width = (uint32_t)total; out->height = 1; out->is_dense = dense;
    if (merged_box.valid && total && rsreg::tunables().box_cache) {
        out->box = merged_box;
        out->box_version = out->version;
    }
    return RSREG_OK;
}

// The source load that has been joined since left the box of the source cloud it read in the context: the handle keeps it
// (if it is still that cloud, unchanged).
static void harvest_source_box(rsreg_ctx *ctx)
{
    const rsreg_cloud *s = ctx->src_cloud;
    // a small source was loaded by one launch that measured its box on the way (k_source_plain): the words are complete once
    // the stamp behind them is this load's -- which it is as soon as anything queued behind the launch has been waited for
    if (s && !ctx->src_pending && ctx->plain_box_seq && ctx->h_smisc.ptr) {
        const uint32_t *hb = ctx->h_smisc.as<uint32_t>() + 48;
        if (__atomic_load_n(&hb[7], __ATOMIC_ACQUIRE) == ctx->plain_box_seq) {
            rsreg::CloudBox b;
            for (int k = 0; k < 3; ++k) { b.mn[k] = rsreg::ordered_float(hb[k]); b.mx[k] = rsreg::ordered_float(hb[3 + k]); }
            b.nfin = hb[6];
            b.valid = true;
            ctx->last_src_box = b;
            ctx->plain_box_seq = 0;
        }
    }
    if (!s || ctx->src_pending || !ctx->last_src_box.valid) return;
    if (s->id == ctx->src_cloud_id && s->version == ctx->src_cloud_version) {
        s->box = ctx->last_src_box;
        s->box_version = s->version;
    }
    ctx->last_src_box.valid = false;
}

// ---- ICP on cloud handles (the handles must stay alive and unchanged until the align has returned)
int rsreg_icp_set_target_cloud(rsreg_ctx *ctx, const rsreg_cloud *c, double max_correspondence_distance)
{
    if (!ctx || !c || c->ctx != ctx) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, settle(c));
    harvest_source_box(ctx);
    // IncrementalICP aligns the ten or twenty points a 1 m voxel filter leaves of a frame against the whole merged model
    // (incremental_icp.hpp:54-59): for so few queries the index is not worth building (icp.hip: scan_target).  The
    // source is set before the target in the reference; if it is not, or changes, rsreg_icp_begin builds the index.
    const bool few_queries = ctx->have_source && ctx->n_source > 0 && ctx->n_source <= kScanSourceLimit && c->n >= kScanTargetFloor &&
                             rsreg::tunables().scan_target;
    // (a cloud that has grown since its index was built -- icp_edge_based_registration.hpp:119-120: *target = *icp_aligned +
    // *target, then the next frame's setInputTarget -- is indexed afresh: rounds 3-4 merged the new records into the index
    // instead, bit for bit the same index, and it did not pay: profiles/r04_experiments/README.md)
    ctx->next_tgt_box.valid = false;
    if (!few_queries && c->box.valid && c->box_version == c->version) ctx->next_tgt_box = c->box;
    int rc = few_queries ? rsreg_icp_set_target_scan_(ctx, c->buf.ptr, c->n, c->stride, max_correspondence_distance)
                         : rsreg_icp_set_target_device(ctx, c->n ? c->buf.ptr : nullptr, c->n, c->stride, c->is_dense, max_correspondence_distance);
    ctx->next_tgt_box.valid = false;
    if (rc) return rc;
    if (!few_queries && ctx->last_tgt_box.valid) {
        c->box = ctx->last_tgt_box;
        c->box_version = c->version;
    }
    ctx->tgt_cloud_id = c->id;
    ctx->tgt_cloud_version = c->version;
    return RSREG_OK;
}

// 1 when the context's ICP target index was built from this cloud, as it is now, for this gate: the ICP edge scheme
// hands the same grown feature cloud to its coarse and to its refining ICP, one after the other
// (icp_edge_based_registration.hpp:94-95,108-109), and the second of them may keep the index instead of building it again
int rsreg_icp_target_is_cloud(const rsreg_ctx *ctx, const rsreg_cloud *c, double max_correspondence_distance)
{
    return ctx && c && c->ctx == ctx && ctx->have_target && ctx->tgt_cloud_id == c->id && ctx->tgt_cloud_version == c->version &&
           ctx->gate_built_for == max_correspondence_distance;
}

int rsreg_icp_set_source_cloud(rsreg_ctx *ctx, const rsreg_cloud *c)
{
    if (!ctx || !c || c->ctx != ctx) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, settle(c));
    harvest_source_box(ctx);   // (of the load before this one, if an alignment has joined it)
    ctx->next_src_box.valid = false;
    if (c->box.valid && c->box.exact && c->box_version == c->version) ctx->next_src_box = c->box;   // (a source's order is quantised from its box: a measured one only)
    int rc = rsreg_icp_set_source_device(ctx, c->n ? c->buf.ptr : nullptr, c->n, c->stride, c->is_dense);
    if (rc) {
        ctx->next_src_box.valid = false;
        return rc;
    }
    ctx->src_cloud = c;
    ctx->src_cloud_id = c->id;
    ctx->src_cloud_version = c->version;
    return RSREG_OK;
}

// icp.align(out[, guess]) with the aligned cloud left in HBM (nullable; may be the source cloud itself)
int rsreg_icp_align_cloud(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params, rsreg_icp_result *result,
                          rsreg_cloud *aligned_out)
{
    if (!ctx || !params || !result) return RSREG_ERR_INVALID_ARG;
    if (aligned_out && (aligned_out->ctx != ctx || !ctx->src_cloud)) return fail(ctx, RSREG_ERR_STATE, "rsreg_icp_set_source_cloud not called");
    // the aligned cloud is the SOURCE CLOUD's records under the final transform: the handle must still be the cloud that was loaded
    if (aligned_out && (ctx->src_cloud->id != ctx->src_cloud_id || ctx->src_cloud->version != ctx->src_cloud_version))
        return fail(ctx, RSREG_ERR_STATE, "the source cloud was rewritten after rsreg_icp_set_source_cloud");
    RSREG_HIP(ctx, settle(aligned_out));
    int rc = rsreg_icp_align(ctx, guess, params, result, nullptr, 0);
    if (!rc) harvest_source_box(ctx);   // (the alignment has been waited for: a small source's box is there; the aligned cloud below starts from it)
    if (rc || !aligned_out) return rc;
    const rsreg_cloud *src = ctx->src_cloud;
    if (aligned_out != src) RSREG_HIP(ctx, cloud_reserve(ctx, aligned_out->buf, src->n * src->stride + 16));
    Mat4f T;
    std::memcpy(T.m, result->transform, 64);
    if (src->n) {
        RSREG_HIP(ctx, launch_records_transform(ctx->stream, src->buf.as<char>(), aligned_out->buf.as<char>(), (uint32_t)src->n, src->stride, to_mat34(T), 1));
    }
    const rsreg::CloudBox moved_box = (src->box.valid && src->box_version == src->version) ? rsreg::transformed_box(src->box, result->transform) : rsreg::CloudBox{};
    aligned_out->version++;
    aligned_out->n = src->n; aligned_out->stride = src->stride; aligned_out->width = src->width; aligned_out->height = src->height;
    aligned_out->is_dense = src->is_dense;
    if (moved_box.valid && rsreg::tunables().box_cache) {
        aligned_out->box = moved_box;
        aligned_out->box_version = aligned_out->version;
    }
    return RSREG_OK;
}

// ---- NDT on cloud handles
int rsreg_ndt_set_target_device(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride, int is_dense, double resolution);
int rsreg_ndt_align_device(rsreg_ctx *ctx, const void *d_source, size_t n, size_t stride, int is_dense, const float *guess,
                           const rsreg_ndt_params *params, rsreg_ndt_result *result, void *d_aligned_out);

int rsreg_ndt_set_target_cloud(rsreg_ctx *ctx, const rsreg_cloud *c, double resolution)
{
    if (!ctx || !c || c->ctx != ctx) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, settle(c));
    ctx->next_ndt_box.valid = false;
    if (c->box.valid && c->box_version == c->version) ctx->next_ndt_box = c->box;
    const int rc = rsreg_ndt_set_target_device(ctx, c->n ? c->buf.ptr : nullptr, c->n, c->stride, c->is_dense, resolution);
    ctx->next_ndt_box.valid = false;
    if (rc == RSREG_OK && ctx->last_ndt_box.valid && !(c->box.valid && c->box_version == c->version)) {
        c->box = ctx->last_ndt_box;
        c->box_version = c->version;
    }
    return rc;
}

int rsreg_ndt_align_cloud(rsreg_ctx *ctx, const rsreg_cloud *source, const float *guess, const rsreg_ndt_params *params,
                          rsreg_ndt_result *result, rsreg_cloud *aligned_out)
{
    if (!ctx || !source || source->ctx != ctx || !params || (aligned_out && aligned_out->ctx != ctx)) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, settle(source));
    RSREG_HIP(ctx, settle(aligned_out));
    if (aligned_out && aligned_out != source) RSREG_HIP(ctx, cloud_reserve(ctx, aligned_out->buf, source->n * source->stride + 16));
    int rc = rsreg_ndt_align_device(ctx, source->n ? source->buf.ptr : nullptr, source->n, source->stride, source->is_dense, guess, params,
                                    result, aligned_out ? aligned_out->buf.ptr : nullptr);
    if (rc || !aligned_out) return rc;
    aligned_out->version++;
    aligned_out->n = source->n; aligned_out->stride = source->stride; aligned_out->width = source->width;
    aligned_out->height = source->height; aligned_out->is_dense = source->is_dense;
    return RSREG_OK;
}

}  // extern "C"
