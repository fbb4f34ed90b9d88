// pcl_compat.hpp — header-only C++ host layer over the C ABI (include/rsreg.h).
//
// The reference (hyunminch/realsense-pointcloud) is compiled C++ that calls PCL classes; this
// header offers the same class surface — names, argument meaning, error behaviour — for
// exactly the calls its registration schemes make, so that a scheme written
// against PCL compiles against `rsreg::` with a namespace switch:
//
//   pcl::PointXYZRGB / pcl::PointCloud<T> / ::Ptr          src/types.hpp:8-10
//   pcl::IterativeClosestPoint<S,T>                         src/incremental_icp.hpp:37,46-63
//   pcl::NormalDistributionsTransform<S,T>                  src/ndt_edge_based_registration.hpp:37-43,71-104
//   pcl::ApproximateVoxelGrid<T>                            src/icp_edge_based_registration.hpp:38,47,59-60
//   pcl::transformPointCloud(in, out, Matrix4f)             src/incremental_icp.hpp:63
//   pcl::io::loadPCDFile / savePCDFileBinary                src/main.cpp:81,87
//   Eigen::Matrix4f, AngleAxisf * Translation3f products    src/icp_edge...hpp:81-92 (as rsreg::Matrix4f helpers)
//
// No PCL, Eigen or Boost is needed.  All numerics run on the MI355X through librsreg.so; a
// missing GPU makes the calls throw rsreg::Error (there is no CPU fallback).
#pragma once

#include <sys/mman.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <memory>
#include <new>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../rsreg.h"
#include "lzf.hpp"

namespace rsreg {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

inline void check(int status, rsreg_ctx *ctx = nullptr)
{
    if (status == RSREG_OK) return;
    std::string msg = std::string("rsreg: ") + rsreg_status_string(status);
    if (ctx) msg += std::string(" (") + rsreg_last_error(ctx) + ")";
    throw Error(status, msg);
}

// (id, version) of a device cloud: the pair changes whenever its records in HBM are rewritten (rsreg_cloud_version)
inline std::pair<uint64_t, uint64_t> cloud_stamp(const rsreg_cloud *c)
{
    uint64_t id = 0, version = 0;
    check(rsreg_cloud_version(c, &id, &version));
    return {id, version};
}

// ---- pcl::PointXYZRGB: 32 bytes, 16-byte aligned, rgb at byte 16
struct alignas(16) PointXYZRGB {
    float x = 0.f, y = 0.f, z = 0.f, data3 = 1.f;
    union {
        struct { uint8_t b, g, r, a; };
        float rgb;
        uint32_t rgba;
    };
    uint32_t pad_[3] = {0, 0, 0};
    PointXYZRGB() : rgba(0xff000000u) {}
    PointXYZRGB(float x_, float y_, float z_) : x(x_), y(y_), z(z_), rgba(0xff000000u) {}
};
static_assert(sizeof(PointXYZRGB) == 32, "PointXYZRGB must stay byte-compatible with pcl::PointXYZRGB");

// ---- storage of a cloud's points.  pcl::PointCloud keeps a std::vector with Eigen's aligned allocator; this one is
// aligned too and can hand out records WITHOUT constructing them one by one, for the callers that overwrite every
// record right away (a download of the merged cloud of 16 frames spent 25 ms of a 60 ms scheme constructing 4.9 M
// points on one thread before the copy touched them).
namespace detail {
inline bool &skip_point_init()
{
    static thread_local bool skip = false;
    return skip;
}
template <class T> struct point_allocator {
    using value_type = T;
    point_allocator() = default;
    template <class U> point_allocator(const point_allocator<U> &) {}
    // Large clouds sit on 2 MB pages where the kernel hands them out on request (what numpy does for its arrays): the
    // first touch of a 157 MB merged cloud is 75 page faults instead of 38 000 (20 ms of a 54 ms scheme).
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T), huge = (size_t)2 << 20;
        const bool large = bytes >= 2 * huge;
        void *p = nullptr;
        if (posix_memalign(&p, large ? huge : (alignof(T) > sizeof(void *) ? alignof(T) : sizeof(void *)), bytes ? bytes : 1) != 0) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
        if (large) (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) { std::free(p); }
    template <class U> void construct(U *p)
    {
        if (!skip_point_init()) ::new (static_cast<void *>(p)) U();
    }
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a)
    {
        ::new (static_cast<void *>(p)) U(std::forward<A0>(a0), std::forward<A>(a)...);
    }
    template <class U> bool operator==(const point_allocator<U> &) const { return true; }
    template <class U> bool operator!=(const point_allocator<U> &) const { return false; }
};
}  // namespace detail

template <typename PointT> using PointVector = std::vector<PointT, detail::point_allocator<PointT>>;

// n records the caller is about to overwrite, all of them (their contents are unspecified until then)
template <typename PointT> inline PointVector<PointT> uninitialized_points(size_t n)
{
    static_assert(std::is_trivially_copyable<PointT>::value && std::is_trivially_destructible<PointT>::value,
                  "records that may stay unconstructed must be plain data");
#if defined(RSREG_PCL_COMPAT_FAST_UNINIT) && defined(__GLIBCXX__) && !defined(_GLIBCXX_DEBUG) && !defined(_GLIBCXX_SANITIZE_VECTOR)
    // OPT-IN (-DRSREG_PCL_COMPAT_FAST_UNINIT; this repository's own runners set it): libstdc++ only, not its debug mode nor
    // annotated containers.  The storage is reserved and the end pointer moved, no per-record call at all (the portable
    // default below goes through the allocator's construct() once per record: 1 ms of doing nothing for a merged cloud of
    // 4.9 M points).  It reaches into libstdc++'s private _M_impl through a cast to a type the vector is not -- formally
    // undefined behaviour, which is why an integrator's toolchain gets the portable path unless it asks.
    struct Open : PointVector<PointT> {
        void grow_unconstructed(size_t m)
        {
            this->reserve(m);
            this->_M_impl._M_finish = this->_M_impl._M_start + m;
        }
    };
    PointVector<PointT> v;
    static_cast<Open &>(v).grow_unconstructed(n);
    return v;
#else
    struct Guard {
        bool before = detail::skip_point_init();
        Guard() { detail::skip_point_init() = true; }
        ~Guard() { detail::skip_point_init() = before; }
    } guard;
    return PointVector<PointT>(n);
#endif
}

// ---- pcl::PointCloud<PointT>
template <typename PointT> struct PointCloud {
    using Ptr = std::shared_ptr<PointCloud<PointT>>;
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    PointVector<PointT> points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;

    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    bool isOrganized() const { return height > 1; }
    void clear() { points.clear(); width = height = 0; }
    void push_back(const PointT &p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    PointT &operator[](size_t i) { return points[i]; }
    const PointT &operator[](size_t i) const { return points[i]; }

    // concatenation: width = size, height = 1, dense only if both are
    PointCloud &operator+=(const PointCloud &rhs)
    {
        points.insert(points.end(), rhs.points.begin(), rhs.points.end());
        width = (uint32_t)points.size();
        height = 1;
        is_dense = is_dense && rhs.is_dense;
        return *this;
    }
    PointCloud operator+(const PointCloud &rhs) const
    {
        PointCloud out = *this;
        out += rhs;
        return out;
    }
};

// ---- Eigen::Matrix4f stand-in: 16 floats, column-major
struct Matrix4f {
    float m[16];
    Matrix4f() { std::memset(m, 0, sizeof(m)); }
    static Matrix4f Identity()
    {
        Matrix4f r;
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f;
        return r;
    }
    float &operator()(int r, int c) { return m[c * 4 + r]; }
    float operator()(int r, int c) const { return m[c * 4 + r]; }
    const float *data() const { return m; }
    float *data() { return m; }
    Matrix4f operator*(const Matrix4f &o) const
    {
        Matrix4f r;
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
                for (int k = 0; k < 4; ++k) s += m[k * 4 + i] * o.m[j * 4 + k];
                r.m[j * 4 + i] = s;
            }
        return r;
    }
    bool operator==(const Matrix4f &o) const { return std::memcmp(m, o.m, sizeof(m)) == 0; }
    bool operator!=(const Matrix4f &o) const { return !(*this == o); }
    // (Eigen::Translation3f(t) * Eigen::AngleAxisf(angle, axis)).matrix() pieces
    static Matrix4f RotationX(float a) { Matrix4f r = Identity(); const float c = std::cos(a), s = std::sin(a); r(1, 1) = c; r(1, 2) = -s; r(2, 1) = s; r(2, 2) = c; return r; }
    static Matrix4f RotationY(float a) { Matrix4f r = Identity(); const float c = std::cos(a), s = std::sin(a); r(0, 0) = c; r(0, 2) = s; r(2, 0) = -s; r(2, 2) = c; return r; }
    static Matrix4f RotationZ(float a) { Matrix4f r = Identity(); const float c = std::cos(a), s = std::sin(a); r(0, 0) = c; r(0, 1) = -s; r(1, 0) = s; r(1, 1) = c; return r; }
    static Matrix4f Translation(float x, float y, float z) { Matrix4f r = Identity(); r(0, 3) = x; r(1, 3) = y; r(2, 3) = z; return r; }
};

// ---- one execution context per (device, stream); shared default on device 0
class Context {
  public:
    explicit Context(int device = 0, void *stream = nullptr) { check(rsreg_ctx_create(device, stream, &ctx_)); }
    ~Context() { if (ctx_) rsreg_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    rsreg_ctx *get() const { return ctx_; }
    void wait_downloads() { check(rsreg_ctx_wait_downloads(ctx_), ctx_); }   // every DeviceCloud::download_async of this context has landed
    // what a frame loop is about to need, made on a thread of the context while the caller goes on (rsreg_ctx_prepare)
    void prepare(size_t frame_bytes, size_t model_bytes, bool side_streams) { check(rsreg_ctx_prepare(ctx_, frame_bytes, model_bytes, side_streams ? RSREG_PREPARE_SIDE_STREAMS : 0u), ctx_); }
    // A context holds ONE ICP target index, ONE ICP source and ONE NDT voxel grid.  The object that
    // uploaded each of them last is remembered here, so that a second registration object sharing
    // the context (e.g. the default one) re-uploads its own clouds instead of silently using another's.
    const void *icp_target_owner = nullptr, *icp_source_owner = nullptr, *ndt_target_owner = nullptr;
    static std::shared_ptr<Context> Default()
    {
        static std::shared_ptr<Context> ctx = std::make_shared<Context>(0);
        return ctx;
    }
  private:
    rsreg_ctx *ctx_ = nullptr;
};

// ---- a cloud resident in HBM (rsreg_cloud): what the frame loop hands from step to step without
// leaving the GPU (filter -> align -> transformPointCloud -> operator+, icp_edge_based_registration.hpp:75-120)
template <typename PointT> class DeviceCloud {
  public:
    using Ptr = std::shared_ptr<DeviceCloud<PointT>>;
    explicit DeviceCloud(std::shared_ptr<Context> ctx = Context::Default()) : ctx_(std::move(ctx))
    {
        check(rsreg_cloud_create(ctx_->get(), &h_), ctx_->get());
    }
    explicit DeviceCloud(const PointCloud<PointT> &host, std::shared_ptr<Context> ctx = Context::Default()) : DeviceCloud(std::move(ctx))
    {
        upload(host);
    }
    ~DeviceCloud() { if (h_) rsreg_cloud_destroy(h_); }
    DeviceCloud(const DeviceCloud &) = delete;
    DeviceCloud &operator=(const DeviceCloud &) = delete;
    void upload(const PointCloud<PointT> &host)
    {
        check(rsreg_cloud_upload(h_, host.points.data(), host.size(), sizeof(PointT), host.width, host.height, host.is_dense), ctx_->get());
    }
    // upload() that returns once the records are staged: the PCIe copy runs beside the main stream's work, and whoever
    // touches this cloud next waits for it (rsreg_cloud_upload_async).  `host` may change as soon as this returns.
    void upload_async(const PointCloud<PointT> &host)
    {
        check(rsreg_cloud_upload_async(h_, host.points.data(), host.size(), sizeof(PointT), host.width, host.height, host.is_dense), ctx_->get());
    }
    // upload_async() that returns before `host` has been read (rsreg_cloud_upload_deferred): a thread of the context
    // stages the records and queues their copy.  `host` must stay as it is until a call that reads or rewrites this
    // cloud has returned -- the frame loops hand over the caller's frames, which stay put for the whole registration.
    void upload_deferred(const PointCloud<PointT> &host)
    {
        check(rsreg_cloud_upload_deferred(h_, host.points.data(), host.size(), sizeof(PointT), host.width, host.height, host.is_dense), ctx_->get());
    }
    void download(PointCloud<PointT> &host) const
    {
        size_t n = 0, stride = 0;
        uint32_t w = 0, h = 0;
        int dense = 0;
        check(rsreg_cloud_info(h_, &n, &stride, &w, &h, &dense), ctx_->get());
        if (n && stride != sizeof(PointT)) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: record size of the device cloud differs");
        PointVector<PointT> pts = uninitialized_points<PointT>(n);
        check(rsreg_cloud_download(h_, pts.data(), n), ctx_->get());
        host.points = std::move(pts);
        host.width = w;
        host.height = h;
        host.is_dense = dense != 0;
    }
    // engine extra: download() that returns at once -- the records as they are now go to dst[0 .. size()) beside whatever
    // the GPU does next; dst belongs to the copy until Context::wait_downloads() has returned (rsreg_cloud_download_async)
    void download_async(PointT *dst, size_t capacity) const
    {
        check(rsreg_cloud_download_async(h_, dst, capacity), ctx_->get());
    }
    void info(size_t &n, uint32_t &width, uint32_t &height, bool &is_dense) const
    {
        size_t stride = 0;
        int dense = 0;
        check(rsreg_cloud_info(h_, &n, &stride, &width, &height, &dense), ctx_->get());
        is_dense = dense != 0;
    }
    size_t size() const
    {
        size_t n = 0;
        check(rsreg_cloud_info(h_, &n, nullptr, nullptr, nullptr, nullptr), ctx_->get());
        return n;
    }
    // *this += other (PointCloud::operator+=): grows in place
    DeviceCloud &operator+=(const DeviceCloud &other)
    {
        check(rsreg_cloud_concat(ctx_->get(), h_, other.h_, h_), ctx_->get());
        return *this;
    }
    // out = a + b (a's records first); out may be a or b
    static void concatenate(const DeviceCloud &a, const DeviceCloud &b, DeviceCloud &out)
    {
        check(rsreg_cloud_concat(a.ctx_->get(), a.h_, b.h_, out.h_), a.ctx_->get());
    }
    rsreg_cloud *handle() const { return h_; }
    const std::shared_ptr<Context> &context() const { return ctx_; }

  private:
    std::shared_ptr<Context> ctx_;
    rsreg_cloud *h_ = nullptr;
};

namespace detail {
template <typename PointT> void copy_aligned(const PointCloud<PointT> &src, PointCloud<PointT> &out)
{
    out.points = src.points;  // fields other than xyz are the source's (PCL copies the input first)
    out.width = src.width;
    out.height = src.height;
    out.is_dense = src.is_dense;
}
}  // namespace detail

// ---- pcl::IterativeClosestPoint
template <typename PointSource, typename PointTarget> class IterativeClosestPoint {
  public:
    using SourcePtr = typename PointCloud<PointSource>::Ptr;
    using TargetPtr = typename PointCloud<PointTarget>::Ptr;

    explicit IterativeClosestPoint(std::shared_ptr<Context> ctx = Context::Default()) : ctx_(std::move(ctx))
    {
        rsreg_icp_params_default(&prm_);
        final_ = Matrix4f::Identity();
    }
    void setMaximumIterations(int n) { prm_.max_iterations = n; }
    void setMaxCorrespondenceDistance(double d)
    {
        if (d != prm_.max_correspondence_distance) target_dirty_ = true;  // the index cell size derives from it
        prm_.max_correspondence_distance = d;
    }
    void setTransformationEpsilon(double e) { prm_.transformation_epsilon = e; }
    void setTransformationRotationEpsilon(double e) { prm_.transformation_rotation_epsilon = e; }
    void setEuclideanFitnessEpsilon(double e) { prm_.euclidean_fitness_epsilon = e; }
    void setInputSource(const SourcePtr &cloud) { source_ = cloud; dsource_ = nullptr; source_dirty_ = true; }
    void setInputTarget(const TargetPtr &cloud) { target_ = cloud; dtarget_ = nullptr; target_dirty_ = true; }  // PCL rebuilds its kd-tree here too
    // optional correspondence filters (off by default; the reference constructs a trimmed rejector and never attaches it)
    void setUseReciprocalCorrespondences(bool on) { prm_.use_reciprocal_correspondences = on ? 1 : 0; }
    // addCorrespondenceRejector (CorrespondenceRejectorTrimmed with setOverlapRatio (ratio)); <= 0 or >= 1: none
    void setTrimmedRejectorOverlapRatio(double ratio) { prm_.trim_overlap_ratio = ratio; }
    // engine knobs without a PCL counterpart
    void setFixedIterationCount(bool on) { prm_.criteria_mode = on ? RSREG_CRITERIA_FIXED : RSREG_CRITERIA_PCL; }
    void setPipelineMode(int mode) { prm_.pipeline_mode = mode; }

    void align(PointCloud<PointSource> &output) { align(output, Matrix4f::Identity()); }
    void align(PointCloud<PointSource> &output, const Matrix4f &guess)
    {
        if (!source_ || !target_) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: setInputSource / setInputTarget not called");
        rsreg_ctx *c = ctx_->get();
        // the source first (the reference's order, incremental_icp.hpp:57-58): it is loaded on a stream of its own,
        // beside the target's index build
        if (source_dirty_ || ctx_->icp_source_owner != this) {
            check(rsreg_icp_set_source(c, source_->points.data(), source_->size(), sizeof(PointSource), source_->is_dense), c);
            source_dirty_ = false;
            ctx_->icp_source_owner = this;
        }
        if (target_dirty_ || ctx_->icp_target_owner != this) {
            check(rsreg_icp_set_target(c, target_->points.data(), target_->size(), sizeof(PointTarget), target_->is_dense,
                                       prm_.max_correspondence_distance), c);
            target_dirty_ = false;
            ctx_->icp_target_owner = this;
        }
        // `output = input` (PCL copies the input cloud first, then rewrites xyz) is made inside the call, by the host threads that
        // write the aligned positions anyway (rsreg_icp_align_records): the records are never constructed or copied here
        PointCloud<PointSource> tmp;
        tmp.points = uninitialized_points<PointSource>(source_->size());
        tmp.width = source_->width;
        tmp.height = source_->height;
        tmp.is_dense = source_->is_dense;
        if (source_->size())
            check(rsreg_icp_align_records(c, guess.data(), &prm_, &res_, source_->points.data(), tmp.points.data(), sizeof(PointSource)), c);
        else
            check(rsreg_icp_align(c, guess.data(), &prm_, &res_, nullptr, 0), c);
        std::memcpy(final_.m, res_.transform, sizeof(final_.m));
        output = std::move(tmp);
    }
    // the same calls on clouds resident in HBM: nothing is uploaded or downloaded (the handles must
    // stay alive and unchanged until align has returned); output may be the source cloud itself
    void setInputSource(const DeviceCloud<PointSource> &cloud) { dsource_ = &cloud; source_.reset(); source_dirty_ = true; }
    void setInputTarget(const DeviceCloud<PointTarget> &cloud) { dtarget_ = &cloud; target_.reset(); target_dirty_ = true; }
    // engine extra: with a device cloud as the target, keep the context's index when it was built from that very cloud
    // (unchanged since) by another ICP object of the same context, instead of building it again.  Off by default: PCL
    // rebuilds its kd-tree at every setInputTarget.
    void setReuseTargetIndex(bool on) { reuse_target_index_ = on; }
    void align(DeviceCloud<PointSource> &output) { align(output, Matrix4f::Identity()); }
    void align(DeviceCloud<PointSource> &output, const Matrix4f &guess)
    {
        if (!dsource_ || !dtarget_) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: setInputSource / setInputTarget (device clouds) not called");
        rsreg_ctx *c = ctx_->get();
        const auto t0 = std::chrono::steady_clock::now();
        // a device cloud rewritten in place since it was loaded (filter(x, x), +=, a transform or an alignment into it) is
        // loaded again: PCL would see the new points through its pointer
        if (!source_dirty_ && cloud_stamp(dsource_->handle()) != source_stamp_) source_dirty_ = true;
        if (!target_dirty_ && cloud_stamp(dtarget_->handle()) != target_stamp_) target_dirty_ = true;
        if (source_dirty_ || ctx_->icp_source_owner != this) {
            check(rsreg_icp_set_source_cloud(c, dsource_->handle()), c);
            source_stamp_ = cloud_stamp(dsource_->handle());
            source_dirty_ = false;
            ctx_->icp_source_owner = this;
        }
        t1_ = std::chrono::steady_clock::now();
        if (target_dirty_ || ctx_->icp_target_owner != this) {
            if (!(reuse_target_index_ && rsreg_icp_target_is_cloud(c, dtarget_->handle(), prm_.max_correspondence_distance)))
                check(rsreg_icp_set_target_cloud(c, dtarget_->handle(), prm_.max_correspondence_distance), c);
            target_stamp_ = cloud_stamp(dtarget_->handle());
            target_dirty_ = false;
            ctx_->icp_target_owner = this;
        }
        const auto t2 = std::chrono::steady_clock::now();
        check(rsreg_icp_align_cloud(c, guess.data(), &prm_, &res_, output.handle()), c);
        std::memcpy(final_.m, res_.transform, sizeof(final_.m));
        const auto t3 = std::chrono::steady_clock::now();
        call_ms[0] += std::chrono::duration<double, std::milli>(t1_ - t0).count();
        call_ms[1] += std::chrono::duration<double, std::milli>(t2 - t1_).count();
        call_ms[2] += std::chrono::duration<double, std::milli>(t3 - t2).count();
    }
    // engine extra: host time of the device-cloud align() calls so far, in ms -- [0] loading the source (and waiting for whatever
    // still makes it), [1] the target's index, [2] the alignment itself
    double call_ms[3] = {0, 0, 0};
    bool hasConverged() const { return res_.converged != 0; }
    Matrix4f getFinalTransformation() const { return final_; }
    int getConvergenceState() const { return res_.state; }
    const rsreg_icp_result &result() const { return res_; }

  private:
    std::shared_ptr<Context> ctx_;
    rsreg_icp_params prm_;
    rsreg_icp_result res_{};
    Matrix4f final_;
    SourcePtr source_;
    TargetPtr target_;
    const DeviceCloud<PointSource> *dsource_ = nullptr;
    const DeviceCloud<PointTarget> *dtarget_ = nullptr;
    bool source_dirty_ = true, target_dirty_ = true, reuse_target_index_ = false;
    std::pair<uint64_t, uint64_t> source_stamp_{0, 0}, target_stamp_{0, 0};   // (id, version) of the device clouds as loaded
    std::chrono::steady_clock::time_point t1_;
};

// ---- pcl::NormalDistributionsTransform
template <typename PointSource, typename PointTarget> class NormalDistributionsTransform {
  public:
    using SourcePtr = typename PointCloud<PointSource>::Ptr;
    using TargetPtr = typename PointCloud<PointTarget>::Ptr;

    explicit NormalDistributionsTransform(std::shared_ptr<Context> ctx = Context::Default()) : ctx_(std::move(ctx))
    {
        rsreg_ndt_params_default(&prm_);
        final_ = Matrix4f::Identity();
    }
    void setTransformationEpsilon(double e) { prm_.transformation_epsilon = e; }
    void setStepSize(double s) { prm_.step_size = s; }
    void setResolution(float r)
    {
        if ((double)r != prm_.resolution) target_dirty_ = true;
        prm_.resolution = r;
    }
    void setMaximumIterations(int n) { prm_.max_iterations = n; }
    // engine extra: search the voxels by PCL's own centroid arithmetic (float running sum per voxel in input order,
    // rsreg_ndt_set_centroid_mode) instead of the rounded f64 mean
    void setPclCentroids(bool on)
    {
        if (on != pcl_centroids_) target_dirty_ = true;
        pcl_centroids_ = on;
    }
    void setInputSource(const SourcePtr &cloud) { source_ = cloud; dsource_ = nullptr; }
    void setInputTarget(const TargetPtr &cloud) { target_ = cloud; dtarget_ = nullptr; target_dirty_ = true; }

    void align(PointCloud<PointSource> &output) { align(output, Matrix4f::Identity()); }
    void align(PointCloud<PointSource> &output, const Matrix4f &guess)
    {
        if (!source_ || !target_) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: setInputSource / setInputTarget not called");
        rsreg_ctx *c = ctx_->get();
        if (target_dirty_ || ctx_->ndt_target_owner != this) {
            check(rsreg_ndt_set_centroid_mode(c, pcl_centroids_ ? 1 : 0), c);
            check(rsreg_ndt_set_target(c, target_->points.data(), target_->size(), sizeof(PointTarget), target_->is_dense,
                                       prm_.resolution), c);
            target_dirty_ = false;
            ctx_->ndt_target_owner = this;
        }
        PointCloud<PointSource> tmp;
        detail::copy_aligned(*source_, tmp);
        check(rsreg_ndt_align(c, source_->points.data(), source_->size(), sizeof(PointSource), source_->is_dense, guess.data(),
                              &prm_, &res_, tmp.points.data(), sizeof(PointSource)), c);
        std::memcpy(final_.m, res_.transform, sizeof(final_.m));
        output = std::move(tmp);
    }
    void setInputSource(const DeviceCloud<PointSource> &cloud) { dsource_ = &cloud; source_.reset(); }
    void setInputTarget(const DeviceCloud<PointTarget> &cloud) { dtarget_ = &cloud; target_.reset(); target_dirty_ = true; }
    void align(DeviceCloud<PointSource> &output) { align(output, Matrix4f::Identity()); }
    void align(DeviceCloud<PointSource> &output, const Matrix4f &guess)
    {
        if (!dsource_ || !dtarget_) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: setInputSource / setInputTarget (device clouds) not called");
        rsreg_ctx *c = ctx_->get();
        if (!target_dirty_ && cloud_stamp(dtarget_->handle()) != target_stamp_) target_dirty_ = true;   // rewritten in place since
        if (target_dirty_ || ctx_->ndt_target_owner != this) {
            check(rsreg_ndt_set_centroid_mode(c, pcl_centroids_ ? 1 : 0), c);
            check(rsreg_ndt_set_target_cloud(c, dtarget_->handle(), prm_.resolution), c);
            target_stamp_ = cloud_stamp(dtarget_->handle());
            target_dirty_ = false;
            ctx_->ndt_target_owner = this;
        }
        check(rsreg_ndt_align_cloud(c, dsource_->handle(), guess.data(), &prm_, &res_, output.handle()), c);
        std::memcpy(final_.m, res_.transform, sizeof(final_.m));
    }
    bool hasConverged() const { return res_.converged != 0; }
    Matrix4f getFinalTransformation() const { return final_; }
    double getTransformationProbability() const { return res_.trans_probability; }
    int getFinalNumIteration() const { return res_.iterations; }
    const rsreg_ndt_result &result() const { return res_; }

  private:
    std::shared_ptr<Context> ctx_;
    rsreg_ndt_params prm_;
    rsreg_ndt_result res_{};
    Matrix4f final_;
    SourcePtr source_;
    TargetPtr target_;
    const DeviceCloud<PointSource> *dsource_ = nullptr;
    const DeviceCloud<PointTarget> *dtarget_ = nullptr;
    bool target_dirty_ = true, pcl_centroids_ = false;
    std::pair<uint64_t, uint64_t> target_stamp_{0, 0};   // (id, version) of the device target as loaded
};

// ---- pcl::ApproximateVoxelGrid (host, sequential: order-dependent by definition)
// Default-constructed: the sequential host filter.  With a Context: the same filter on the GPU
// (csrc/voxel.hip), same records in the same order.
template <typename PointT> class ApproximateVoxelGrid {
  public:
    ApproximateVoxelGrid() = default;
    explicit ApproximateVoxelGrid(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) {}
    void setLeafSize(float lx, float ly, float lz) { leaf_[0] = lx; leaf_[1] = ly; leaf_[2] = lz; }
    void setInputCloud(const typename PointCloud<PointT>::Ptr &cloud) { input_ = cloud; }
    void filter(PointCloud<PointT> &output)  // output may be *input (the reference filters in place)
    {
        if (!input_) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: setInputCloud not called");
        PointVector<PointT> out = uninitialized_points<PointT>(input_->size());
        size_t n_out = 0;
        if (ctx_)
            check(rsreg_approx_voxel_grid_gpu(ctx_->get(), input_->points.data(), input_->size(), sizeof(PointT), leaf_, out.data(),
                                              &n_out), ctx_->get());
        else
            check(rsreg_approx_voxel_grid(input_->points.data(), input_->size(), sizeof(PointT), leaf_, out.data(), &n_out));
        out.resize(n_out);
        output.points = std::move(out);
        output.width = (uint32_t)n_out;
        output.height = 1;
        output.is_dense = false;
    }
    // the filter on a cloud resident in HBM (always the GPU filter); output may be the input
    void filter(const DeviceCloud<PointT> &input, DeviceCloud<PointT> &output)
    {
        check(rsreg_cloud_filter(input.context()->get(), input.handle(), leaf_, output.handle()), input.context()->get());
    }
    // the same on the context's side stream: returns once the size of the result is known, the voxel sums still
    // running; whatever touches `output` next waits for them.  `input` must stay alive and unchanged until then;
    // output must not be the input (rsreg_cloud_filter_async)
    void filter_async(const DeviceCloud<PointT> &input, DeviceCloud<PointT> &output)
    {
        check(rsreg_cloud_filter_async(input.context()->get(), input.handle(), leaf_, output.handle()), input.context()->get());
    }
  private:
    float leaf_[3] = {1.f, 1.f, 1.f};  // PCL default: IncrementalICP never sets it (incremental_icp.hpp:36)
    typename PointCloud<PointT>::Ptr input_;
    std::shared_ptr<Context> ctx_;
};

// ---- pcl::transformPointCloud(in, out, Matrix4f); in and out may be the same object
template <typename PointT>
void transformPointCloud(const PointCloud<PointT> &in, PointCloud<PointT> &out, const Matrix4f &T,
                         const std::shared_ptr<Context> &ctx = Context::Default())
{
    PointVector<PointT> pts = uninitialized_points<PointT>(in.size());
    check(rsreg_transform_cloud(ctx->get(), in.points.data(), pts.data(), in.size(), sizeof(PointT), in.is_dense, T.data()),
          ctx->get());
    const uint32_t w = in.width, h = in.height;
    const bool dense = in.is_dense;
    out.points = std::move(pts);
    out.width = w;
    out.height = h;
    out.is_dense = dense;
}

template <typename PointT> void transformPointCloud(const DeviceCloud<PointT> &in, DeviceCloud<PointT> &out, const Matrix4f &T)
{
    check(rsreg_cloud_transform(in.context()->get(), in.handle(), T.data(), out.handle()), in.context()->get());
}

// ---- extract_edge_features (src/edge_extractor.hpp:7-39): the RGB-Canny edge points of an organized cloud
inline std::shared_ptr<PointCloud<PointXYZRGB>> extract_edge_features(const std::shared_ptr<PointCloud<PointXYZRGB>> &cloud,
                                                                      const std::shared_ptr<Context> &ctx = Context::Default())
{
    auto out = std::make_shared<PointCloud<PointXYZRGB>>();
    if ((size_t)cloud->width * cloud->height != cloud->size()) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: edge extraction needs an organized cloud");
    PointVector<PointXYZRGB> pts = uninitialized_points<PointXYZRGB>(cloud->size());
    size_t n = 0;
    check(rsreg_extract_edge_features(ctx->get(), cloud->points.data(), cloud->width, cloud->height, sizeof(PointXYZRGB), pts.data(), nullptr, &n),
          ctx->get());
    pts.resize(n);
    out->points = std::move(pts);
    out->width = (uint32_t)n;
    out->height = 1;
    out->is_dense = cloud->is_dense;
    return out;
}

// the same on a frame that already lies in HBM (out: width = number of edge points, height = 1)
inline void extract_edge_features(const DeviceCloud<PointXYZRGB> &cloud, DeviceCloud<PointXYZRGB> &out)
{
    check(rsreg_cloud_edge_features(cloud.context()->get(), cloud.handle(), out.handle()), cloud.context()->get());
}
// ... queued by a thread of the context on a stream of its own (rsreg_cloud_edge_features_async): returns at once, `out` is
// complete when a call that takes it has waited for it.  `cloud` must stay as it is until then.
inline void extract_edge_features_async(const DeviceCloud<PointXYZRGB> &cloud, DeviceCloud<PointXYZRGB> &out)
{
    check(rsreg_cloud_edge_features_async(cloud.context()->get(), cloud.handle(), out.handle()), cloud.context()->get());
}

// ---- pcl::io: PCD files with FIELDS x y z rgb (ascii, binary, binary_compressed), as the reference reads/writes
// (src/main.cpp:53,81,87)
namespace io {

inline int loadPCDFile(const std::string &path, PointCloud<PointXYZRGB> &cloud)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    std::string line, data_mode;
    std::vector<std::string> fields, types;
    std::vector<int> sizes;
    size_t n = 0;
    uint32_t width = 0, height = 1;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "FIELDS") { std::string s; while (ss >> s) fields.push_back(s); }
        else if (key == "SIZE") { int s; while (ss >> s) sizes.push_back(s); }
        else if (key == "TYPE") { std::string s; while (ss >> s) types.push_back(s); }
        else if (key == "WIDTH") ss >> width;
        else if (key == "HEIGHT") ss >> height;
        else if (key == "POINTS") ss >> n;
        else if (key == "DATA") { ss >> data_mode; break; }
    }
    if (n == 0) n = (size_t)width * height;
    int ix = -1, iy = -1, iz = -1, ic = -1;
    for (size_t k = 0; k < fields.size(); ++k) {
        if (fields[k] == "x") ix = (int)k;
        else if (fields[k] == "y") iy = (int)k;
        else if (fields[k] == "z") iz = (int)k;
        else if (fields[k] == "rgb" || fields[k] == "rgba") ic = (int)k;
    }
    if (ix < 0 || iy < 0 || iz < 0) return -2;
    // the header is untrusted input: every field needs a SIZE and a TYPE, sizes are positive, and the fields this reader
    // copies four bytes of (x, y, z, the packed colour) are four bytes wide; the body must fit what is left of the file
    if (sizes.size() != fields.size() || types.size() != fields.size()) return -2;
    for (int sz : sizes)
        if (sz <= 0 || sz > 8) return -2;
    if (sizes[ix] != 4 || sizes[iy] != 4 || sizes[iz] != 4 || (ic >= 0 && sizes[ic] != 4)) return -2;
    size_t remaining = 0;
    {
        const std::streampos here = f.tellg();
        f.seekg(0, std::ios::end);
        const std::streampos end = f.tellg();
        f.seekg(here);
        if (here < 0 || end < here) return -4;
        remaining = (size_t)(end - here);
    }
    {
        size_t rec_bytes = 0;
        for (int sz : sizes) rec_bytes += (size_t)sz;
        // ascii needs at least "0 " per field; the binary forms are checked exactly below
        const size_t least = data_mode == "binary" ? rec_bytes : (data_mode == "ascii" ? 2 * fields.size() - 1 : 0);
        if (n > 0 && least > 0 && n > remaining / least + 1) return -4;
        if (n > (size_t)1 << 32) return -4;
    }
    cloud.points.assign(n, PointXYZRGB());
    cloud.width = width;
    cloud.height = height;
    bool dense = true;
    if (data_mode == "ascii") {
        for (size_t i = 0; i < n; ++i) {
            PointXYZRGB &p = cloud.points[i];
            for (size_t k = 0; k < fields.size(); ++k) {
                std::string tok;
                if (!(f >> tok)) return -4;
                if ((int)k == ic) {
                    if (types[k] == "F") { float v = std::stof(tok); std::memcpy(&p.rgba, &v, 4); }
                    else p.rgba = (uint32_t)std::stoul(tok);
                } else {
                    const float v = std::stof(tok);
                    if ((int)k == ix) p.x = v; else if ((int)k == iy) p.y = v; else if ((int)k == iz) p.z = v;
                }
            }
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) dense = false;
        }
    } else if (data_mode == "binary") {
        size_t rec = 0;
        std::vector<size_t> off(fields.size());
        for (size_t k = 0; k < fields.size(); ++k) { off[k] = rec; rec += (size_t)sizes[k]; }
        if (rec * n > remaining) return -4;
        std::vector<char> buf(rec * n);
        f.read(buf.data(), (std::streamsize)buf.size());
        if ((size_t)f.gcount() != buf.size()) return -4;
        for (size_t i = 0; i < n; ++i) {
            PointXYZRGB &p = cloud.points[i];
            const char *r = buf.data() + i * rec;
            std::memcpy(&p.x, r + off[ix], 4);
            std::memcpy(&p.y, r + off[iy], 4);
            std::memcpy(&p.z, r + off[iz], 4);
            if (ic >= 0) std::memcpy(&p.rgba, r + off[ic], 4);
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) dense = false;
        }
    } else if (data_mode == "binary_compressed") {
        // u32 compressed size, u32 uncompressed size, LZF stream of the fields one after the other
        uint32_t csize = 0, usize = 0;
        f.read(reinterpret_cast<char *>(&csize), 4);
        f.read(reinterpret_cast<char *>(&usize), 4);
        size_t rec = 0;
        std::vector<size_t> foff(fields.size());
        for (size_t k = 0; k < fields.size(); ++k) { foff[k] = rec * n; rec += (size_t)sizes[k]; }
        if (!f || (size_t)usize != rec * n || remaining < 8 || (size_t)csize > remaining - 8) return -4;
        std::vector<uint8_t> comp(csize), soa(usize);
        f.read(reinterpret_cast<char *>(comp.data()), (std::streamsize)csize);
        if (!f || (usize && lzf::decode(comp.data(), csize, soa.data(), usize) != usize)) return -4;
        for (size_t i = 0; i < n; ++i) {
            PointXYZRGB &p = cloud.points[i];
            std::memcpy(&p.x, soa.data() + foff[ix] + 4 * i, 4);
            std::memcpy(&p.y, soa.data() + foff[iy] + 4 * i, 4);
            std::memcpy(&p.z, soa.data() + foff[iz] + 4 * i, 4);
            if (ic >= 0) std::memcpy(&p.rgba, soa.data() + foff[ic] + 4 * i, 4);
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) dense = false;
        }
    } else {
        return -3;
    }
    cloud.is_dense = dense;
    return 0;
}

inline int savePCDFileBinary(const std::string &path, const PointCloud<PointXYZRGB> &cloud)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) return -1;
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
      << "WIDTH " << cloud.width << "\nHEIGHT " << cloud.height << "\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << cloud.size()
      << "\nDATA binary\n";
    for (const PointXYZRGB &p : cloud.points) {
        f.write(reinterpret_cast<const char *>(&p.x), 12);
        f.write(reinterpret_cast<const char *>(&p.rgba), 4);
    }
    return f ? 0 : -1;
}

// pcl::io::savePCDFileBinaryCompressed: same header, DATA binary_compressed, LZF over the SoA body
inline int savePCDFileBinaryCompressed(const std::string &path, const PointCloud<PointXYZRGB> &cloud)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) return -1;
    const size_t n = cloud.size();
    std::vector<uint8_t> soa(16 * n);
    for (size_t i = 0; i < n; ++i) {
        const PointXYZRGB &p = cloud.points[i];
        std::memcpy(soa.data() + 4 * i, &p.x, 4);
        std::memcpy(soa.data() + 4 * (n + i), &p.y, 4);
        std::memcpy(soa.data() + 4 * (2 * n + i), &p.z, 4);
        std::memcpy(soa.data() + 4 * (3 * n + i), &p.rgba, 4);
    }
    std::vector<uint8_t> comp(lzf::max_encoded_size(soa.size()));
    const uint32_t usize = (uint32_t)soa.size();
    const uint32_t csize = (uint32_t)lzf::encode(soa.data(), soa.size(), comp.data(), comp.size());
    if (usize && !csize) return -1;
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
      << "WIDTH " << cloud.width << "\nHEIGHT " << cloud.height << "\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n
      << "\nDATA binary_compressed\n";
    f.write(reinterpret_cast<const char *>(&csize), 4);
    f.write(reinterpret_cast<const char *>(&usize), 4);
    f.write(reinterpret_cast<const char *>(comp.data()), csize);
    return f ? 0 : -1;
}

}  // namespace io
}  // namespace rsreg
