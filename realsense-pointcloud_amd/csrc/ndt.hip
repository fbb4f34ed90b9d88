// ndt.hip — NDT half of the C ABI (include/rsreg.h): pcl::NormalDistributionsTransform as
// the reference drives it (src/ndt_edge_based_registration.hpp:38-43,71-72,83,92,104).
//
// Device: voxel binning, per-voxel moments, the per-point derivative pass (ndt_kernels.hpp).
// Host:   covariance regularisation + inversion per voxel (tens to hundreds of 3x3s), the
//         Newton step (6x6 SVD solve) and the More-Thuente line search.
#include <atomic>
#include <cstring>
#include <string.h>


#include <algorithm>
#include <cfloat>
#include <cmath>

#include "ndt_kernels.hpp"
#include "oscan.hpp"

using namespace rsreg;

extern "C" int rsreg_comm_allreduce_device_(rsreg_ctx *ctx, double *d_buf, int count);  // comm.cpp

namespace {

#ifndef RSREG_NDT_PASS_BLOCKS
#define RSREG_NDT_PASS_BLOCKS 512   // (dev: experiment builds with another count, RSREG_CXXFLAGS=-DRSREG_NDT_PASS_BLOCKS=...)
#endif
constexpr int kPassBlocks = RSREG_NDT_PASS_BLOCKS;   // fixed: the summation order does not depend on the GPU (2 workgroups per CU of an MI355X;
                                   // 36 k points against 23 voxels: 0.61-0.62 ms per alignment, 1024: 0.65-0.68, 256: 0.62-0.63)
constexpr int kMinPointsPerVoxel = 6;
constexpr double kMinCovarEigMult = 0.01;

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// order-preserving float <-> uint map (same as the ICP build's)
__device__ __forceinline__ uint32_t f2o(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
float o2f(uint32_t u)
{
    uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    std::memcpy(&f, &v, 4);
    return f;
}

__global__ __launch_bounds__(256) void k_ndt_bbox(const char *pts, size_t stride, uint32_t n, uint32_t *bbox)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    uint32_t cnt = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *p = reinterpret_cast<const float *>(pts + (size_t)i * stride);
        const float x = p[0], y = p[1], z = p[2];
        if (ndt_finite3(x, y, z)) {
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
    // one set of atomics per workgroup, not per wave: they all hit the same seven words
    __shared__ float smn[4][3], smx[4][3];
    __shared__ uint32_t scnt[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; ++k) { smn[wave][k] = mn[k]; smx[wave][k] = mx[k]; }
        scnt[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], smn[w][k]); mx[k] = fmaxf(mx[k], smx[w][k]); }
            cnt += scnt[w];
        }
        if (cnt) {
            for (int k = 0; k < 3; ++k) {
                atomicMin(&bbox[k], f2o(mn[k]));
                atomicMax(&bbox[3 + k], f2o(mx[k]));
            }
            atomicAdd(&bbox[6], cnt);
        }
    }
}

struct Pose {
    double p[6];
};

// (the pose matrix, the angle terms and the line search's arithmetic: ndt_math.hpp, one source for the host and the device)
Mat4f pose_matrix(const double *p) { return ndt_pose_matrix(p); }

// Eigen 3.3 Matrix3f::eulerAngles(0, 1, 2)
void euler_xyz(const Mat4f &M, float *res)
{
    const float kPi = 3.14159265358979323846f;
    res[0] = std::atan2(M(1, 2), M(2, 2));
    const float c2 = std::sqrt(M(0, 0) * M(0, 0) + M(0, 1) * M(0, 1));
    if (res[0] > 0.0f) {
        res[0] -= kPi;
        res[1] = std::atan2(-M(0, 2), -c2);
    } else {
        res[1] = std::atan2(-M(0, 2), c2);
    }
    const float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
    res[2] = std::atan2(s1 * M(2, 0) - c1 * M(1, 0), c1 * M(1, 1) - s1 * M(2, 1));
    res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2];
}

void gauss_constants(const rsreg_ndt_params &prm, double &d1, double &d2)
{
    const double c1 = 10.0 * (1 - prm.outlier_ratio);
    const double c2 = prm.outlier_ratio / std::pow(prm.resolution, 3);
    const double d3 = -std::log(c2);
    d1 = -std::log(c1 + c2) - d3;
    d2 = -2 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / d1);
}

void angle_terms(const double *p, NdtPassParams &pp) { ndt_angle_terms(p, pp.jang, pp.hang); }

constexpr int kNdtLsOffset = 64;   // h_ndt, in doubles: where a finished resident line search leaves its NdtLs (behind the sums and the flag)
constexpr size_t kNdtHostBytes = (kNdtLsOffset + 16) * 8 + sizeof(NdtLs) + sizeof(NdtLsCtl) + 64;   // sums, flag | NdtLs | the staged NdtLsCtl
constexpr int kNdtFlagSlot = 32;  // h_ndt: 28 sums, then the pass number the final reduce stamps

// partial sums of a pass + the counter the final reduce counts its workgroups on (zero between passes)
hipError_t reserve_partials(rsreg_ctx *ctx)
{
    const size_t tickets = 8;
    const size_t bytes = (size_t)kPassBlocks * kNdtAcc * 8 + tickets;
    if (ctx->d_ndt_partials.cap >= bytes) return hipSuccess;
    hipError_t e = ctx->d_ndt_partials.reserve(bytes);
    if (e != hipSuccess) return e;
    return hipMemsetAsync(static_cast<char *>(ctx->d_ndt_partials.ptr) + bytes - tickets, 0, tickets, ctx->stream);
}

struct NdtRun {
    rsreg_ctx *ctx;
    rsreg_ndt_params prm;
    uint32_t n;
    double d1, d2;
    Mat4f final_t;
    int passes = 0;
    double ms_derivatives = 0;
};

// One derivative pass with the parameters `pp` (pose matrix, angle terms, mode): the launch pair and the wait for its 28 sums
// (h_ndt: pinned).
int derivative_pass_pp(NdtRun &r, NdtPassParams &pp, bool store_trans)
{
    rsreg_ctx *ctx = r.ctx;
    pp.d1 = r.d1;
    pp.d2 = r.d2;
    pp.r2 = (float)(r.prm.resolution * r.prm.resolution);
    pp.n_vox = ctx->ndt_n_voxels;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->profiling) {
        for (hipEvent_t &e : ctx->ev_ndt)
            if (!e) (void)hipEventCreate(&e);
        e0 = ctx->ev_ndt[0];
        e1 = ctx->ev_ndt[1];
        (void)hipEventRecord(e0, ctx->stream);
    }
    double *h = ctx->h_ndt.as<double>();
    const bool watch = !ctx->comm && !ctx->profiling && tunables().ndt_watch;
    volatile uint64_t *flag = reinterpret_cast<volatile uint64_t *>(h + kNdtFlagSlot);
    const uint64_t seq = ++ctx->ndt_seq;
    *flag = 0;
    uint32_t *ticket = reinterpret_cast<uint32_t *>(ctx->d_ndt_partials.as<double>() + (size_t)kPassBlocks * kNdtAcc);
    // one GPU: the 28 sums go straight into the pinned host buffer (no copy to queue behind the kernel)
    double *sums_out = ctx->comm ? ctx->d_ndt_out.as<double>() : h;
    if (tunables().ndt_one_launch && kPassBlocks == 2 * kNdtBlock) {   // (k_ndt_pass_reduce adds slab t and slab t + 256)
        // the pass and its final reduce in one launch: the workgroup that finishes last adds the slabs (same tree, same bits; measured
        // 8 us per pass slower than the launch pair below -- round 6, as round 2's form was: opt-in)
        k_ndt_pass_reduce<<<kPassBlocks, kNdtBlock, 0, ctx->stream>>>(ctx->d_ndt_src.as<float4>(), r.n, ctx->d_ndt_vox.as<NdtVoxel>(), pp,
                                                                      store_trans ? ctx->d_ndt_trans.as<float>() : nullptr,
                                                                      ctx->d_ndt_partials.as<double>(), sums_out, ticket,
                                                                      watch ? const_cast<uint64_t *>(flag) : nullptr, seq);
        RSREG_HIP(ctx, hipGetLastError());
    } else {
        k_ndt_pass<<<kPassBlocks, kNdtBlock, 0, ctx->stream>>>(ctx->d_ndt_src.as<float4>(), r.n, ctx->d_ndt_vox.as<NdtVoxel>(), pp,
                                                               store_trans ? ctx->d_ndt_trans.as<float>() : nullptr,
                                                               ctx->d_ndt_partials.as<double>());
        RSREG_HIP(ctx, hipGetLastError());
        k_ndt_final_reduce<<<kNdtAcc, kNdtBlock, 0, ctx->stream>>>(ctx->d_ndt_partials.as<double>(), kPassBlocks, sums_out, ticket,
                                                                   watch ? const_cast<uint64_t *>(flag) : nullptr, seq);
        RSREG_HIP(ctx, hipGetLastError());
    }
    if (ctx->profiling) (void)hipEventRecord(e1, ctx->stream);
    if (ctx->comm) {   // also on a one-rank communicator: same calls, same stream order
        int rc = rsreg_comm_allreduce_device_(ctx, ctx->d_ndt_out.as<double>(), kNdtAcc);
        if (rc) return rc;
    }
    if (ctx->comm) RSREG_HIP(ctx, hipMemcpyAsync(h, ctx->d_ndt_out.ptr, kNdtAcc * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (watch) {
        // the pass number lands in pinned memory right behind the sums; a stream query now and then notices a fault
        for (uint32_t spins = 1; *flag != seq; ++spins)
            if ((spins & 0xFFFFu) == 0 && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*flag != seq) RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (ctx->profiling) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) r.ms_derivatives += ms;
    }
    r.passes++;
    return RSREG_OK;
}

// One derivative pass at pose p with transform M; mode as in NdtPassParams.
// On return grad (6) / hess (36, symmetric) hold the requested parts; returns the score.
int derivative_pass(NdtRun &r, const double *p, const Mat4f &M, int mode, bool store_trans, double *score,
                    double *grad, double *hess)
{
    NdtPassParams pp;
    for (int row = 0; row < 3; ++row)
        for (int c = 0; c < 4; ++c) pp.M[row * 4 + c] = M(row, c);
    angle_terms(p, pp);
    pp.mode = mode;
    int rc = derivative_pass_pp(r, pp, store_trans);
    if (rc) return rc;
    const double *h = r.ctx->h_ndt.as<double>();
    if (mode != 2) {
        *score = h[0];
        for (int i = 0; i < 6; ++i) grad[i] = h[1 + i];
    }
    for (int k = 0; k < 36; ++k) hess[k] = 0.0;  // PCL zeroes the Hessian at the start of every pass
    if (mode != 1) {
        int k = 7;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) {
                hess[a * 6 + b] = h[k];
                hess[b * 6 + a] = h[k];
                ++k;
            }
    }
    return RSREG_OK;
}

// Can the 512 workgroups of k_ndt_line_search be resident together on this device?  (They wait for each other inside the launch.)
bool line_search_fits(rsreg_ctx *ctx)
{
    static int fits[64] = {0};   // per device: 0 unknown, 1 yes, -1 no
    int &f = fits[ctx->device & 63];
    if (f == 0) {
        int per_cu = 0, cus = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_ndt_line_search, kNdtBlock, 0) == hipSuccess &&
            hipGetDeviceProperties(&prop, ctx->device) == hipSuccess)
            cus = prop.multiProcessorCount;
        f = (long long)per_cu * cus >= kPassBlocks ? 1 : -1;
    }
    return f > 0;
}

// computeStepLengthMT (More-Thuente): the state machine of ndt_math.hpp, advanced by the host, a launch pair and a wait per
// pass.  RSREG_NDT_RESIDENT_LS=1 (one GPU, no profiling): all the passes of the search in ONE launch (k_ndt_line_search,
// which advances the machine itself) -- the same source on the same sums in the same order, the same bits
// (tests/test_ndt_gpu.py), and measured no faster: a hand-over between workgroups of different XCDs costs what a kernel
// boundary costs, so it is not the default.
int step_length(NdtRun &r, const double *x, double *dir, double step_init, double step_max, double step_min,
                double &score, double *grad, double *hess, double &a_out)
{
    rsreg_ctx *ctx = r.ctx;
    NdtLs ls;
    ndt_ls_begin(ls, x, dir, step_init, step_max, step_min, score, grad, hess);
    const bool host_only = !tunables().ndt_resident_ls;
    bool resident = ls.phase != kNdtLsDone && !host_only && !ctx->comm && !ctx->profiling && !ctx->ndt_ls_failed && line_search_fits(ctx);
    if (resident) {
        RSREG_HIP(ctx, ctx->d_ndt_ctl.reserve(sizeof(NdtLsCtl) + 64));
        RSREG_HIP(ctx, ctx->h_ndt.reserve(kNdtHostBytes));
        double *h = ctx->h_ndt.as<double>();
        volatile uint64_t *flag = reinterpret_cast<volatile uint64_t *>(h + kNdtFlagSlot);
        NdtLs *host_out = reinterpret_cast<NdtLs *>(h + kNdtLsOffset);
        NdtLsCtl *stage = reinterpret_cast<NdtLsCtl *>(h + kNdtLsOffset + (sizeof(NdtLs) + 7) / 8 + 8);
        std::memset(stage, 0, sizeof(NdtLsCtl));
        stage->ls = ls;
        stage->pp.d1 = r.d1;
        stage->pp.d2 = r.d2;
        stage->pp.r2 = (float)(r.prm.resolution * r.prm.resolution);
        stage->pp.n_vox = ctx->ndt_n_voxels;
        ndt_fill_pass(stage->pp, ls);
        stage->release = 1;   // the first pass may start
        const uint64_t seq = ++ctx->ndt_seq;
        *flag = 0;
        RSREG_HIP(ctx, hipMemcpyAsync(ctx->d_ndt_ctl.ptr, stage, sizeof(NdtLsCtl), hipMemcpyHostToDevice, ctx->stream));
        k_ndt_line_search<<<kPassBlocks, kNdtBlock, 0, ctx->stream>>>(ctx->d_ndt_src.as<float4>(), r.n, ctx->d_ndt_vox.as<NdtVoxel>(), ctx->d_ndt_ctl.as<NdtLsCtl>(),
                                                                    ctx->d_ndt_trans.as<float>(), ctx->d_ndt_partials.as<double>(), host_out,
                                                                    const_cast<uint64_t *>(flag), seq);
        RSREG_HIP(ctx, hipGetLastError());
        // the outcome lands in pinned memory, the launch's number behind it; a stream query now and then notices a fault or a time-out
        for (uint32_t spins = 1; *flag != seq; ++spins)
            if ((spins & 0xFFFFu) == 0 && hipStreamQuery(ctx->stream) != hipErrorNotReady) break;
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*flag != seq) {
            RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (*flag == seq) {
            std::memcpy(&ls, host_out, sizeof(NdtLs));
            r.passes += ls.passes;
        } else {
            // (a bounded wait inside the launch ran out: not again on this context; this search pass by pass, from its start)
            ctx->ndt_ls_failed = true;
            resident = false;
            ndt_ls_begin(ls, x, dir, step_init, step_max, step_min, score, grad, hess);
        }
    }
    if (!resident) {
        while (ls.phase != kNdtLsDone) {
            NdtPassParams pp;
            ndt_fill_pass(pp, ls);
            int rc = derivative_pass_pp(r, pp, ls.next_mode != 2);
            if (rc) return rc;
            ndt_ls_consume(ls, ctx->h_ndt.as<double>());
        }
    }
    for (int i = 0; i < 6; ++i) dir[i] = ls.dir[i];
    if (ls.passes) r.final_t = pose_matrix(ls.x_t);
    score = ls.score;
    for (int i = 0; i < 6; ++i) grad[i] = ls.grad[i];
    for (int k = 0; k < 36; ++k) hess[k] = ls.hess[k];
    a_out = ls.a_t;
    return RSREG_OK;
}

int load_ndt_source_device(rsreg_ctx *ctx, const void *d_source, size_t n, size_t stride)
{
    RSREG_HIP(ctx, ctx->d_ndt_trans.reserve(n * 12 + 16));
    RSREG_HIP(ctx, ctx->d_ndt_src.reserve((n + 1) * sizeof(float4)));
    RSREG_HIP(ctx, reserve_partials(ctx));
    RSREG_HIP(ctx, ctx->d_ndt_out.reserve(64 * 8));
    RSREG_HIP(ctx, ctx->h_ndt.reserve(kNdtHostBytes));
    if (n) {
        k_ndt_load_source<<<div_up((uint32_t)n, kNdtBlock), kNdtBlock, 0, ctx->stream>>>(static_cast<const char *>(d_source), stride,
                                                                                        (uint32_t)n, ctx->d_ndt_src.as<float4>());
        RSREG_HIP(ctx, hipGetLastError());
    }
    return RSREG_OK;
}

int load_ndt_source(rsreg_ctx *ctx, const void *source, size_t n, size_t stride)
{
    int rc_pack = pack_to_stage(ctx, source, n, stride);
    if (rc_pack) return rc_pack;
    RSREG_HIP(ctx, ctx->d_ndt_trans.reserve(n * 12 + 16));
    RSREG_HIP(ctx, ctx->d_ndt_src.reserve((n + 1) * sizeof(float4)));
    RSREG_HIP(ctx, ctx->d_tmp.reserve(n * 12 + 16));
    RSREG_HIP(ctx, reserve_partials(ctx));
    RSREG_HIP(ctx, ctx->d_ndt_out.reserve(64 * 8));
    RSREG_HIP(ctx, ctx->h_ndt.reserve(kNdtHostBytes));
    if (n) {
        RSREG_HIP(ctx, hipMemcpyAsync(ctx->d_tmp.ptr, ctx->h_stage.ptr, n * 12, hipMemcpyHostToDevice, ctx->stream));
        k_ndt_load_source<<<div_up((uint32_t)n, kNdtBlock), kNdtBlock, 0, ctx->stream>>>(ctx->d_tmp.as<char>(), 12, (uint32_t)n,
                                                                                        ctx->d_ndt_src.as<float4>());
        RSREG_HIP(ctx, hipGetLastError());
        RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RSREG_OK;
}

}  // namespace

extern "C" {

void rsreg_ndt_params_default(rsreg_ndt_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->max_iterations = 35;
    p->transformation_epsilon = 0.1;
    p->step_size = 0.1;
    p->resolution = 1.0;
    p->outlier_ratio = 0.55;
}

void rsreg_ndt_params_reference(rsreg_ndt_params *p)
{
    if (!p) return;
    rsreg_ndt_params_default(p);
    p->transformation_epsilon = 0.01;  // ndt_edge_based_registration.hpp:39
    p->step_size = 0.1;                // :40
    p->resolution = 1.0;               // :41
    p->max_iterations = 50;            // :43
}

// VoxelGridCovariance over records already in HBM (d_points / stride); keeps no pointer to them
int rsreg_ndt_set_target_device(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride, int is_dense, double resolution)
{
    (void)is_dense;
    if (!ctx || (n && !d_points) || stride < 12 || (stride & 3) || !(resolution > 0)) return RSREG_ERR_INVALID_ARG;
    if (n > 0xfffffff0ull) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ctx->have_ndt_target = false;
    ctx->ndt_n_voxels = 0;
    ctx->ndt_resolution = resolution;
    ctx->ndt_mean_cov_icov.clear();
    ctx->ndt_counts.clear();
    ctx->ndt_centroid.clear();

    // ---- bounding box of the finite points (pcl::getMinMax3D)
    RSREG_HIP(ctx, ctx->d_misc.reserve(64 * 4));
    RSREG_HIP(ctx, ctx->h_ndt.reserve(kNdtHostBytes));
    const char *d_pts = static_cast<const char *>(d_points);
    const size_t pstride = stride;
    uint32_t *d_misc = ctx->d_misc.as<uint32_t>();
    uint32_t *h_misc = ctx->h_ndt.as<uint32_t>();
    h_misc[6] = 0;
    // A cloud handle that knows a box around its finite points and their number (measured by an earlier build, or the union / the
    // transformed corners of measured ones: rsreg_ctx.hpp CloudBox) saves the kernel and the round trip.  The box only has to
    // CONTAIN the points: a leaf is floor(x / leaf) whatever the grid's origin is, and the leaves are visited in the order of
    // (z, y, x) leaf coordinates whatever its extent is -- the same voxels in the same order as from the measured box.
    const rsreg::CloudBox known = ctx->next_ndt_box;
    ctx->next_ndt_box.valid = false;
    ctx->last_ndt_box.valid = false;
    if (n && known.valid && known.nfin <= n && tunables().box_cache) {
        auto host_f2o = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        for (int k = 0; k < 3; ++k) { h_misc[k] = host_f2o(known.mn[k]); h_misc[3 + k] = host_f2o(known.mx[k]); }
        h_misc[6] = known.nfin;
        ctx->last_ndt_box = known;
    } else if (n) {
        // minima start at all ones, maxima and the count at zero (ordered-float encoding): two memsets, no upload + sync
        RSREG_HIP(ctx, hipMemsetAsync(d_misc, 0xff, 12, st));
        RSREG_HIP(ctx, hipMemsetAsync(d_misc + 3, 0, 52, st));
        k_ndt_bbox<<<std::min<uint32_t>(div_up((uint32_t)n, 256), 256), 256, 0, st>>>(d_pts, pstride, (uint32_t)n, d_misc);
        RSREG_HIP(ctx, hipGetLastError());
        RSREG_HIP(ctx, hipMemcpyAsync(h_misc, d_misc, 64, hipMemcpyDeviceToHost, st));
        RSREG_HIP(ctx, hipStreamSynchronize(st));
        if (h_misc[6]) {   // (measured: the handle keeps it)
            for (int k = 0; k < 3; ++k) { ctx->last_ndt_box.mn[k] = o2f(h_misc[k]); ctx->last_ndt_box.mx[k] = o2f(h_misc[3 + k]); }
            ctx->last_ndt_box.nfin = h_misc[6];
            ctx->last_ndt_box.valid = true;
            ctx->last_ndt_box.exact = true;
        }
    }
    const uint32_t nfin = h_misc[6];
    if (nfin == 0) {
        ctx->have_ndt_target = true;
        RSREG_HIP(ctx, ctx->d_ndt_vox.reserve(sizeof(NdtVoxel)));
        return RSREG_OK;
    }
    NdtBinParams bp;
    const float leaf = (float)resolution;
    bp.inv_leaf = 1.0f / leaf;
    int div_b[3];
    for (int k = 0; k < 3; ++k) {
        volatile float lo = o2f(h_misc[k]) * bp.inv_leaf, hi = o2f(h_misc[3 + k]) * bp.inv_leaf;
        bp.min_b[k] = (int)std::floor(lo);
        div_b[k] = (int)std::floor(hi) - bp.min_b[k] + 1;
    }
    bp.mul[0] = 1;
    bp.mul[1] = div_b[0];
    bp.mul[2] = (long long)div_b[0] * div_b[1];

    // ---- bin: key per point, sort, segment
    RSREG_HIP(ctx, ctx->d_keys.reserve(n * 8));
    RSREG_HIP(ctx, ctx->d_keys_alt.reserve(n * 8));
    RSREG_HIP(ctx, ctx->d_vals.reserve(n * 4));
    RSREG_HIP(ctx, ctx->d_vals_alt.reserve(n * 4));
    RSREG_HIP(ctx, ctx->d_flags.reserve(n * 4));
    RSREG_HIP(ctx, ctx->d_scan.reserve(n * 4));
    RSREG_HIP(ctx, ctx->d_ndt_seg.reserve(((size_t)nfin + 2) * 4));
    auto *vals = ctx->d_vals.as<uint32_t>();
    auto *vals2 = ctx->d_vals_alt.as<uint32_t>();
    uint32_t *start = ctx->d_flags.as<uint32_t>(), *sid = ctx->d_scan.as<uint32_t>(), *seg_begin = ctx->d_ndt_seg.as<uint32_t>();
    size_t sort_bytes = 0;
    const size_t scan_bytes = oscan_scratch_bytes<uint32_t>(nfin);
    // the leaf keys are < div0 * div1 * div2 (a few dozen leaves at the reference's 1 m resolution) and the key of a
    // non-finite point is that product itself: the sort looks at those bits only -- one or two radix passes where all
    // 64 bits of the key type were ten merge passes (23 launches) for a grown edge target of 5 x 10^5 points
    const unsigned long long n_leaves = (unsigned long long)div_b[0] * (unsigned long long)div_b[1] * (unsigned long long)div_b[2];
    unsigned key_bits = 1;
    while (key_bits < 64 && (n_leaves >> key_bits)) ++key_bits;
    if (n_leaves < 0x7fffffffull) {
        auto *keys = ctx->d_keys.as<uint32_t>();
        auto *keys2 = ctx->d_keys_alt.as<uint32_t>();
        // (the library's own radix sort, its state cleared by the keys kernel: osort.hpp)
        const Radix32Plan plan = radix32_plan(n, 0, key_bits);
        sort_bytes = (size_t)plan.words * 4;
        RSREG_HIP(ctx, ctx->d_tmp.reserve(std::max(sort_bytes, scan_bytes) + 256));
        const bool start_in_out = plan.ends_in_first;   // (an even number of passes ends in the pair it started from)
        k_ndt_keys<uint32_t><<<div_up((uint32_t)n, kNdtBlock), kNdtBlock, 0, st>>>(d_pts, pstride, (uint32_t)n, bp, (uint32_t)n_leaves, start_in_out ? keys2 : keys,
                                                                                 start_in_out ? vals2 : vals, ctx->d_tmp.as<uint32_t>(), plan.words);
        RSREG_HIP(ctx, hipGetLastError());
        {
            bool in_first = false;
            RSREG_HIP(ctx, radix32_sort_pairs<uint32_t>(plan, ctx->d_tmp.as<uint32_t>(), start_in_out ? keys2 : keys, start_in_out ? keys : keys2,
                                                        start_in_out ? vals2 : vals, start_in_out ? vals : vals2, n, 0, key_bits, st, &in_first));
            if (in_first != start_in_out) return fail(ctx, RSREG_ERR_STATE, "osort: the sorted pairs are not where they belong");
        }
        k_ndt_flag_starts<uint32_t><<<div_up(nfin, kNdtBlock), kNdtBlock, 0, st>>>(keys2, nfin, start);
        RSREG_HIP(ctx, hipGetLastError());
    } else {
        auto *keys = ctx->d_keys.as<unsigned long long>();
        auto *keys2 = ctx->d_keys_alt.as<unsigned long long>();
        const Radix32Plan plan = radix32_plan<unsigned long long>(n, 0, key_bits);
        sort_bytes = (size_t)plan.words * 4;
        RSREG_HIP(ctx, ctx->d_tmp.reserve(std::max(sort_bytes, scan_bytes) + 256));
        const bool start_in_out = plan.ends_in_first;
        k_ndt_keys<unsigned long long><<<div_up((uint32_t)n, kNdtBlock), kNdtBlock, 0, st>>>(d_pts, pstride, (uint32_t)n, bp, n_leaves, start_in_out ? keys2 : keys,
                                                                                           start_in_out ? vals2 : vals, ctx->d_tmp.as<uint32_t>(), plan.words);
        RSREG_HIP(ctx, hipGetLastError());
        {
            bool in_first = false;
            RSREG_HIP(ctx, radix32_sort_pairs<unsigned long long>(plan, ctx->d_tmp.as<uint32_t>(), start_in_out ? keys2 : keys, start_in_out ? keys : keys2,
                                                                  start_in_out ? vals2 : vals, start_in_out ? vals : vals2, n, 0, key_bits, st, &in_first));
            if (in_first != start_in_out) return fail(ctx, RSREG_ERR_STATE, "osort: the sorted pairs are not where they belong");
        }
        k_ndt_flag_starts<unsigned long long><<<div_up(nfin, kNdtBlock), kNdtBlock, 0, st>>>(keys2, nfin, start);
        RSREG_HIP(ctx, hipGetLastError());
    }
    RSREG_HIP(ctx, (oscan<uint32_t>(start, sid, (size_t)nfin, 0u, ctx->d_tmp.ptr, st)));
    k_ndt_seg_offsets<<<div_up(nfin, kNdtBlock), kNdtBlock, 0, st>>>(start, sid, nfin, seg_begin, d_misc + 8);
    RSREG_HIP(ctx, hipGetLastError());
    RSREG_HIP(ctx, hipMemcpyAsync(h_misc, d_misc, 64, hipMemcpyDeviceToHost, st));
    // A grid of a few dozen leaves (the reference's 1 m resolution): the moments are launched for EVERY leaf of the box, the
    // workgroups beyond the occupied ones leave at once, and the number of occupied leaves comes home with the moments --
    // one round trip instead of two.
    const bool few_leaves = n_leaves <= 256 && ctx->ndt_centroid_mode != 1;
    if (!few_leaves) RSREG_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t nseg_launch = few_leaves ? (uint32_t)std::min<unsigned long long>(n_leaves, nfin) : h_misc[8];

    // ---- per-voxel moments on the device, one block per occupied leaf
    const uint32_t n_parts = nseg_launch <= 2048 ? 16u : 1u;   // (the reference's 1 m voxels: two dozen of 10^4 points each)
    RSREG_HIP(ctx, ctx->d_ndt_out.reserve(std::max<size_t>((size_t)nseg_launch * n_parts * 10 * 8, 64 * 8)));
    k_ndt_voxel_stats<<<nseg_launch * n_parts, kNdtBlock, 0, st>>>(vals2, seg_begin, d_pts, pstride, n_parts, ctx->d_ndt_out.as<double>(),
                                                                   few_leaves ? d_misc + 8 : nullptr);
    RSREG_HIP(ctx, hipGetLastError());
    // (through pinned memory: [partial moments | PCL-mode centroid sums | later the finished table])
    const size_t n_parts_d = (size_t)nseg_launch * n_parts * 10, parts_bytes = n_parts_d * 8;
    const size_t csum_off = (parts_bytes + 255) & ~(size_t)255, csum_bytes = (size_t)nseg_launch * 12;
    const size_t table_off = (csum_off + csum_bytes + 255) & ~(size_t)255;
    RSREG_HIP(ctx, ctx->h_ndt_build.reserve(table_off + (size_t)nseg_launch * sizeof(NdtVoxel) + 256));
    const double *parts = ctx->h_ndt_build.as<double>();
    RSREG_HIP(ctx, hipMemcpyAsync(ctx->h_ndt_build.ptr, ctx->d_ndt_out.ptr, parts_bytes, hipMemcpyDeviceToHost, st));
    const float *csum = nullptr;   // PCL-mode centroids (rsreg_ndt_set_centroid_mode)
    if (ctx->ndt_centroid_mode == 1) {
        RSREG_HIP(ctx, ctx->d_scan.reserve(std::max<size_t>(n * 4, (size_t)nseg_launch * 12)));   // (sid is no longer needed)
        k_ndt_voxel_csum<<<nseg_launch, 64, 0, st>>>(vals2, seg_begin, d_pts, pstride, ctx->d_scan.as<float>());
        RSREG_HIP(ctx, hipGetLastError());
        RSREG_HIP(ctx, hipMemcpyAsync(ctx->h_ndt_build.as<char>() + csum_off, ctx->d_scan.ptr, csum_bytes, hipMemcpyDeviceToHost, st));
        csum = reinterpret_cast<const float *>(ctx->h_ndt_build.as<char>() + csum_off);
    }
    RSREG_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t nseg = h_misc[8];   // (few leaves: it came home with the moments)
    if (nseg > nseg_launch) return fail(ctx, RSREG_ERR_STATE, "ndt: more occupied leaves than the grid has");

    // ---- host: mean, single-pass covariance, eigenvalue floor, inverse (App. A.6)
    std::vector<double> stats((size_t)nseg * 10, 0.0);
    for (uint32_t v = 0; v < nseg; ++v)
        for (uint32_t part = 0; part < n_parts; ++part)   // (the parts of a voxel, in order)
            for (int k = 0; k < 10; ++k) stats[(size_t)v * 10 + k] += parts[((size_t)v * n_parts + part) * 10 + k];
    std::vector<NdtVoxel> table;
    for (uint32_t v = 0; v < nseg; ++v) {
        const double *s = &stats[(size_t)v * 10];
        const int cnt = (int)(s[0] + 0.5);
        if (cnt < kMinPointsPerVoxel) continue;
        const double nn = cnt;
        const double sum[3] = {s[1], s[2], s[3]};
        const double sxx[9] = {s[4], s[5], s[6], s[5], s[7], s[8], s[6], s[8], s[9]};
        double mean[3], cov[9], icov[9];
        for (int k = 0; k < 3; ++k) mean[k] = sum[k] / nn;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                cov[r * 3 + c] = (sxx[r * 3 + c] - 2.0 * (sum[r] * mean[c])) / nn + mean[r] * mean[c];
        for (int k = 0; k < 9; ++k) cov[k] *= (nn - 1.0) / nn;
        double sym[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) sym[r * 3 + c] = r >= c ? cov[r * 3 + c] : cov[c * 3 + r];
        double ev[3], evec[9];
        eig_sym3(sym, ev, evec);
        std::memset(icov, 0, sizeof(icov));
        if (!(ev[0] < 0 || ev[1] < 0 || ev[2] <= 0)) {
            const double floor_ev = kMinCovarEigMult * ev[2];
            if (ev[0] < floor_ev) {
                ev[0] = floor_ev;
                if (ev[1] < floor_ev) ev[1] = floor_ev;
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) {
                        double acc = 0;
                        for (int k = 0; k < 3; ++k) acc += evec[r * 3 + k] * ev[k] * evec[c * 3 + k];
                        cov[r * 3 + c] = acc;
                    }
            }
            if (!inv3(cov, icov)) std::memset(icov, 0, sizeof(icov));
        }
        NdtVoxel nv;
        std::memset(&nv, 0, sizeof(nv));
        for (int k = 0; k < 3; ++k) {
            nv.mean[k] = mean[k];
            // PCL: f32 running sum / n (mode 1); default: the f64 mean rounded
            nv.centroid[k] = !csum ? (float)mean[k] : csum[(size_t)v * 3 + k];
        }
        std::memcpy(nv.icov, icov, sizeof(icov));
        table.push_back(nv);
        ctx->ndt_counts.push_back(cnt);
        ctx->ndt_mean_cov_icov.insert(ctx->ndt_mean_cov_icov.end(), mean, mean + 3);
        ctx->ndt_mean_cov_icov.insert(ctx->ndt_mean_cov_icov.end(), cov, cov + 9);
        ctx->ndt_mean_cov_icov.insert(ctx->ndt_mean_cov_icov.end(), icov, icov + 9);
        for (int k = 0; k < 3; ++k) ctx->ndt_centroid.push_back(nv.centroid[k]);
    }
    ctx->ndt_n_voxels = (int)table.size();
    RSREG_HIP(ctx, ctx->d_ndt_vox.reserve(std::max<size_t>(table.size(), 1) * sizeof(NdtVoxel)));
    if (!table.empty()) {
        // (from pinned memory, and nobody waits for it: the first pass is queued behind it on the same stream, and the next
        //  build's moments land in front of this region, which is rewritten only after that build's own waits)
        std::memcpy(ctx->h_ndt_build.as<char>() + table_off, table.data(), table.size() * sizeof(NdtVoxel));
        RSREG_HIP(ctx, hipMemcpyAsync(ctx->d_ndt_vox.ptr, ctx->h_ndt_build.as<char>() + table_off, table.size() * sizeof(NdtVoxel), hipMemcpyHostToDevice, st));
    }
    ctx->have_ndt_target = true;
    return RSREG_OK;
}

int rsreg_ndt_set_target(rsreg_ctx *ctx, const void *points, size_t n, size_t stride, int is_dense, double resolution)
{
    if (!ctx || (n && !points) || stride < 12 || !(resolution > 0)) return RSREG_ERR_INVALID_ARG;
    if (n > 0xfffffff0ull) return RSREG_ERR_INVALID_ARG;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = pack_to_stage(ctx, points, n, stride);
    if (rc) return rc;
    RSREG_HIP(ctx, ctx->d_tgt_raw.reserve(n * 12 + 16));
    if (n) RSREG_HIP(ctx, hipMemcpyAsync(ctx->d_tgt_raw.ptr, ctx->h_stage.ptr, n * 12, hipMemcpyHostToDevice, ctx->stream));
    return rsreg_ndt_set_target_device(ctx, n ? ctx->d_tgt_raw.ptr : nullptr, n, 12, is_dense, resolution);
}

int rsreg_ndt_set_centroid_mode(rsreg_ctx *ctx, int mode)
{
    if (!ctx || (mode != 0 && mode != 1)) return RSREG_ERR_INVALID_ARG;
    ctx->ndt_centroid_mode = mode;
    return RSREG_OK;
}

int rsreg_ndt_get_centroids(rsreg_ctx *ctx, float *centroids, int32_t capacity)
{
    if (!ctx || (capacity > 0 && !centroids)) return RSREG_ERR_INVALID_ARG;
    if (!ctx->have_ndt_target) return fail(ctx, RSREG_ERR_NO_TARGET, "rsreg_ndt_set_target not called");
    const int m = std::min<int>(capacity, ctx->ndt_n_voxels);
    if (m > 0) std::memcpy(centroids, ctx->ndt_centroid.data(), (size_t)m * 3 * sizeof(float));
    return RSREG_OK;
}

int rsreg_ndt_get_voxels(rsreg_ctx *ctx, int32_t *n_voxels, double *mean_cov_icov, int32_t *counts, int32_t capacity)
{
    if (!ctx || !n_voxels) return RSREG_ERR_INVALID_ARG;
    if (!ctx->have_ndt_target) return fail(ctx, RSREG_ERR_NO_TARGET, "rsreg_ndt_set_target not called");
    *n_voxels = ctx->ndt_n_voxels;
    const int m = std::min<int>(capacity, ctx->ndt_n_voxels);
    if (mean_cov_icov && m > 0) std::memcpy(mean_cov_icov, ctx->ndt_mean_cov_icov.data(), (size_t)m * 21 * 8);
    if (counts && m > 0) std::memcpy(counts, ctx->ndt_counts.data(), (size_t)m * 4);
    return RSREG_OK;
}

int rsreg_ndt_derivatives(rsreg_ctx *ctx, const void *source, size_t n, size_t stride, int is_dense, const double pose[6],
                          double *score, double gradient[6], double hessian[36])
{
    (void)is_dense;
    if (!ctx || (n && !source) || stride < 12 || !pose || !score || !gradient || !hessian) return RSREG_ERR_INVALID_ARG;
    if (!ctx->have_ndt_target) return fail(ctx, RSREG_ERR_NO_TARGET, "rsreg_ndt_set_target not called");
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    int rc = load_ndt_source(ctx, source, n, stride);
    if (rc) return rc;
    NdtRun r;
    r.ctx = ctx;
    rsreg_ndt_params_default(&r.prm);
    r.prm.resolution = ctx->ndt_resolution;
    r.n = (uint32_t)n;
    gauss_constants(r.prm, r.d1, r.d2);
    return derivative_pass(r, pose, pose_matrix(pose), 0, false, score, gradient, hessian);
}

}  // extern "C"

namespace {

// Newton + More-Thuente on the source already loaded into d_ndt_src (n points); leaves the cloud at
// the last evaluated pose in d_ndt_trans (packed xyz)
int ndt_align_loaded(rsreg_ctx *ctx, size_t n, const float *guess, const rsreg_ndt_params *params, rsreg_ndt_result *result)
{
    int rc = RSREG_OK;
    NdtRun r;
    r.ctx = ctx;
    r.prm = *params;
    r.n = (uint32_t)n;
    gauss_constants(r.prm, r.d1, r.d2);
    r.final_t = Mat4f::identity();
    if (guess) {
        Mat4f G;
        std::memcpy(G.m, guess, 64);
        if (!G.is_identity()) r.final_t = G;
    }
    float er[3];
    euler_xyz(r.final_t, er);
    double p[6] = {r.final_t(0, 3), r.final_t(1, 3), r.final_t(2, 3), er[0], er[1], er[2]};
    double delta_p[6], grad[6], hess[36], score = 0;
    int nr_iterations = 0, converged = 0;
    // first pass on the guess-transformed cloud: the cloud is moved by the guess MATRIX, the
    // angle terms come from its Euler angles (ndt.hpp computeTransformation)
    rc = derivative_pass(r, p, r.final_t, 0, true, &score, grad, hess);
    if (rc) return rc;
    while (!converged) {
        double neg_g[6];
        for (int i = 0; i < 6; ++i) neg_g[i] = -grad[i];
        svd_solve<6>(hess, neg_g, delta_p);
        double nrm = 0;
        for (int i = 0; i < 6; ++i) nrm += delta_p[i] * delta_p[i];
        nrm = std::sqrt(nrm);
        if (nrm == 0 || nrm != nrm) {
            converged = (nrm == nrm) ? 1 : 0;
            break;
        }
        for (int i = 0; i < 6; ++i) delta_p[i] /= nrm;
        double a = 0;
        rc = step_length(r, p, delta_p, nrm, params->step_size, params->transformation_epsilon / 2, score, grad, hess, a);
        if (rc) return rc;
        nrm = a;
        for (int i = 0; i < 6; ++i) {
            delta_p[i] *= nrm;
            p[i] += delta_p[i];
        }
        if (nr_iterations > params->max_iterations || (nr_iterations && (std::fabs(nrm) < params->transformation_epsilon)))
            converged = 1;
        ++nr_iterations;
    }
    if (result) {
        std::memset(result, 0, sizeof(*result));
        std::memcpy(result->transform, r.final_t.m, 64);
        result->converged = converged;
        result->iterations = nr_iterations;
        result->score = score;
        result->trans_probability = score / (double)(n ? n : 1);
        result->n_voxels = ctx->ndt_n_voxels;
        result->n_derivative_passes = r.passes;
        result->ms_derivatives = r.ms_derivatives;
        result->ms_total = r.ms_derivatives;
    }
    return RSREG_OK;
}

int ndt_check(rsreg_ctx *ctx, const rsreg_ndt_params *params)
{
    if (!ctx->have_ndt_target) return fail(ctx, RSREG_ERR_NO_TARGET, "rsreg_ndt_set_target not called");
    if (std::fabs(params->resolution - ctx->ndt_resolution) > 0)
        return fail(ctx, RSREG_ERR_INVALID_ARG, "resolution differs from the one the NDT target was built with");
    return RSREG_OK;
}

}  // namespace

extern "C" {

int rsreg_ndt_align(rsreg_ctx *ctx, const void *source, size_t n, size_t stride, int is_dense, const float *guess,
                    const rsreg_ndt_params *params, rsreg_ndt_result *result, void *aligned_out, size_t out_stride)
{
    (void)is_dense;
    if (!ctx || !params || (n && !source) || stride < 12) return RSREG_ERR_INVALID_ARG;
    if (aligned_out && out_stride < 12) return RSREG_ERR_INVALID_ARG;
    int rc = ndt_check(ctx, params);
    if (rc) return rc;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    rc = load_ndt_source(ctx, source, n, stride);
    if (rc) return rc;
    rc = ndt_align_loaded(ctx, n, guess, params, result);
    if (rc) return rc;
    if (aligned_out && n) {  // output cloud = the cloud at the last evaluated pose
        RSREG_HIP(ctx, ctx->h_stage.reserve(n * 12 + 16));
        RSREG_HIP(ctx, hipMemcpyAsync(ctx->h_stage.ptr, ctx->d_ndt_trans.ptr, n * 12, hipMemcpyDeviceToHost, ctx->stream));
        RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const float *xyz = ctx->h_stage.as<float>();
        const char *in = static_cast<const char *>(source);
        char *dst = static_cast<char *>(aligned_out);
        host_parallel_for(n, [=](size_t lo, size_t hi) {
            const float one = 1.0f;
            for (size_t i = lo; i < hi; ++i) {
                float q[3];
                std::memcpy(q, in + i * stride, 12);
                const bool ok = std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]);
                std::memcpy(dst + i * out_stride, ok ? xyz + 3 * i : q, 12);
                if (out_stride >= 16) std::memcpy(dst + i * out_stride + 12, &one, 4);
            }
        });
    }
    return RSREG_OK;
}

// source records already in HBM; d_aligned_out (nullable, may be d_source): the records with xyz at the final pose
int rsreg_ndt_align_device(rsreg_ctx *ctx, const void *d_source, size_t n, size_t stride, int is_dense, const float *guess,
                           const rsreg_ndt_params *params, rsreg_ndt_result *result, void *d_aligned_out)
{
    (void)is_dense;
    if (!ctx || !params || (n && !d_source) || stride < 12 || (stride & 3)) return RSREG_ERR_INVALID_ARG;
    int rc = ndt_check(ctx, params);
    if (rc) return rc;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    rc = load_ndt_source_device(ctx, d_source, n, stride);
    if (rc) return rc;
    rc = ndt_align_loaded(ctx, n, guess, params, result);
    if (rc) return rc;
    if (d_aligned_out && n) {
        k_ndt_write_aligned<<<div_up((uint32_t)n, kNdtBlock), kNdtBlock, 0, ctx->stream>>>(static_cast<const char *>(d_source), stride, (uint32_t)n,
                                                                                          ctx->d_ndt_trans.as<float>(),
                                                                                          static_cast<char *>(d_aligned_out));
        RSREG_HIP(ctx, hipGetLastError());
    }
    return RSREG_OK;
}

}  // extern "C"
