// schemes.hpp — the reference's registration schemes on top of rsreg/pcl_compat.hpp.
//
// Same public surface as the reference:
//   RegistrationScheme::registration(std::vector<cloud_ptr>&) -> cloud_ptr        src/types.hpp:14-20
//   TwoPhaseRegistrationScheme: extract_features / global_registration             src/types.hpp:22-44
//   IncrementalICP                                                                 src/incremental_icp.hpp:33-70
//   ICPEdgeBasedRegistration  (static yaw or IMU Euler-angle guesses)              src/icp_edge_based_registration.hpp:10-136
//   NDTEdgeBasedRegistration  (NDT guess, ICP refine)                              src/ndt_edge_based_registration.hpp:7-123
// and the same observable behaviour: which clouds are filtered, which are aliased and
// mutated (frame 0 IS the accumulating target), the hard-coded parameters, the order of
// concatenation, frames whose ICP does not converge being skipped silently.
//
// `extract_features` is the reference's extract_edge_features (src/edge_extractor.hpp:7-39: the RGB-Canny
// edge points of the organized frame, rsreg_extract_edge_features) unless a `feature_fn` is plugged in.
#pragma once

#include <algorithm>
#include <atomic>
#include <cassert>
#include <condition_variable>
#include <mutex>
#include <functional>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "pcl_compat.hpp"

namespace rsreg {

using rgb_point = PointXYZRGB;
using rgb_point_cloud = PointCloud<rgb_point>;
using rgb_point_cloud_pointer = rgb_point_cloud::Ptr;

// src/utils.hpp:30-62 — only what the schemes use
struct float3 {
    float x, y, z;
    float3 operator*(float t) const { return {x * t, y * t, z * t}; }
    void add(float t1, float t2, float t3) { x += t1; y += t2; z += t3; }
};

using rgb_device_cloud = DeviceCloud<rgb_point>;

class RegistrationScheme {
  public:
    virtual ~RegistrationScheme()
    {
        if (release_.joinable()) release_.join();
    }
    virtual rgb_point_cloud_pointer registration(std::vector<rgb_point_cloud_pointer> &clouds) = 0;
    // true: the reference's progress lines on stdout, text for text (types.hpp:35-41, icp_edge_based_registration.hpp:27-32,
    // 94-96,103-104,110,113,122,127, ndt_edge_based_registration.hpp:24-29,82-84,91-93,98,101,110,114); IncrementalICP
    // prints nothing in the reference either.  Off by default: a library does not write to its caller's stdout.
    bool verbose = false;
    // true (default): the frame loop runs on clouds resident in HBM -- a frame is uploaded once, every
    // step takes and leaves its clouds on the GPU, only what the caller gets back is downloaded.
    // false: every step on host clouds (one upload + download per step).  Same records either way.
    bool device_resident = true;
    // device-resident loop only: every frame's moved points go to the host while the next frames are aligned
    // (DeviceCloud::download_async), so the merged cloud is complete when the loop ends instead of one 157 MB download
    // after it (16 frames of 307 k points: 4 ms on the link).  false: one download at the end.  Same records either way.
    bool stream_result = true;
    // engine extra (device-resident loops): milliseconds since registration() began at which the set-up before the frame
    // loop was done ([0]), every frame's pass through the loop ended ([1] .. [n-1]), and the merged cloud was complete on
    // the host (the last entry) -- the per-frame table of tools/cpp_scheme_times.py (RSREG_SCHEME_FRAMES=1)
    std::vector<double> frame_clock_ms;
    // engine extra (stream_result): what the end of the loop waited for, in ms -- the host copy of frame 0, the downloads still
    // on their way -- and the host time of all download_async calls together
    double stream_finish_ms[4] = {0, 0, 0, 0};   // ([3]: handing the records over to the caller's cloud: its old storage is let go of)
    // engine extra (device-resident loops): where the caller's thread spent the frame loop, in ms over all frames, call by call --
    // IncrementalICP: [0] queueing the uploads / filters of the frames ahead, [1] setInputSource, [2] setInputTarget, [3] align,
    // [4] transformPointCloud, [5] +=, [6] handing the moved points to the result's download; the edge schemes: [0] queueing the
    // uploads / extractions / filters of the frames ahead, [1] the coarse alignment (ICP on the filtered edges, or NDT), [2] the
    // refining ICP, [3] the two transformPointCloud and the grown target, [4] the result's download (or merged +=)
    double stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  protected:
    void clock_start()
    {
        clock0_ = lap_ = std::chrono::steady_clock::now();
        frame_clock_ms.clear();
        for (double &v : stage_ms) v = 0;
    }
    void clock_lap(int stage)   // the time since the last lap (or mark) goes to stage_ms[stage]
    {
        const auto now = std::chrono::steady_clock::now();
        if (stage >= 0) stage_ms[stage] += std::chrono::duration<double, std::milli>(now - lap_).count();
        lap_ = now;
    }
    template <class R> void note_finish(R &r)
    {
        // (the storage the caller's frame 0 had before it became the merged cloud is let go of by a thread: unmapping 9.8 MB
        //  took 0.7-1.0 ms of the call; it is over when this object is destroyed or registers again)
        if (release_.joinable()) release_.join();
        release_ = r.take_release();
        stream_finish_ms[0] = r.join_ms;
        stream_finish_ms[1] = r.wait_ms;
        stream_finish_ms[2] = r.append_ms;
        stream_finish_ms[3] = r.hand_over_ms;
    }
    void clock_mark() { frame_clock_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - clock0_).count()); }

  private:
    std::chrono::steady_clock::time_point clock0_, lap_;
    std::thread release_;
};

class TwoPhaseRegistrationScheme : public RegistrationScheme {
  public:
    using FeatureFn = std::function<rgb_point_cloud_pointer(rgb_point_cloud_pointer)>;
    using PairList = std::vector<std::pair<rgb_point_cloud_pointer, rgb_point_cloud_pointer>>;

    virtual rgb_point_cloud_pointer extract_features(rgb_point_cloud_pointer cloud)
    {
        if (feature_fn) return feature_fn(cloud);
        return extract_edge_features(cloud);   // icp_edge_based_registration.hpp:21-23, ndt_edge...hpp:18-20
    }
    // given (feature cloud, original cloud) pairs, compute the merged global cloud
    virtual rgb_point_cloud_pointer global_registration(PairList &clouds) = 0;

    rgb_point_cloud_pointer registration(std::vector<rgb_point_cloud_pointer> &clouds) override
    {
        PairList pairs;
        for (auto &c : clouds) {                                              // phase 1
            if (verbose) std::cout << "[PCL] Extracting features..." << std::flush;
            pairs.emplace_back(extract_features(c), c);
            if (verbose) std::cout << "OK" << std::endl;
        }
        if (verbose) std::cout << "[PCL] Performing global registration..." << std::endl;
        return global_registration(pairs);                                    // phase 2
    }
    FeatureFn feature_fn;
};

namespace detail {
// The merged cloud a scheme returns, filled while its frame loop runs: frame 0 is copied on the host (a thread of its
// own: it is 10 MB), every later frame's moved points arrive by download_async behind the work that made them.
class StreamedResult {
  public:
    StreamedResult(std::shared_ptr<Context> ctx, const rgb_point_cloud &frame0, size_t capacity)
        : ctx_(std::move(ctx)), pts_(uninitialized_points<rgb_point>(capacity)), n_(frame0.size()), dense_(frame0.is_dense)
    {
        rgb_point *dst = pts_.data();
        const rgb_point *src = frame0.points.data();
        const size_t n = n_;
        copy0_ = std::thread([dst, src, n] { std::memcpy(static_cast<void *>(dst), src, n * sizeof(rgb_point)); });
    }
    std::thread take_release() { return std::move(release_); }
    ~StreamedResult()
    {
        if (release_.joinable()) release_.join();
        if (copy0_.joinable()) copy0_.join();
        if (pending_) (void)rsreg_ctx_wait_downloads(ctx_->get());   // (an exception on the way: the copies still own pts_)
    }
    void append(const rgb_device_cloud &moved)
    {
        size_t n = 0;
        uint32_t w = 0, h = 0;
        bool dense = false;
        moved.info(n, w, h, dense);
        if (n_ + n > pts_.size()) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: streamed result overflows its capacity");
        const auto t0 = std::chrono::steady_clock::now();
        moved.download_async(pts_.data() + n_, n);
        append_us_ += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        pending_ = true;
        n_ += n;
        dense_ = dense_ && dense;
    }
    void finish(rgb_point_cloud &out)
    {
        const auto t0 = std::chrono::steady_clock::now();
        if (copy0_.joinable()) copy0_.join();
        const auto t1 = std::chrono::steady_clock::now();
        ctx_->wait_downloads();
        join_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        wait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        append_ms = append_us_ / 1e3;
        pending_ = false;
        const auto t2 = std::chrono::steady_clock::now();
        pts_.resize(n_);
        out.points.swap(pts_);   // (pts_: what `out` held before -- the caller's frame 0 in IncrementalICP, nothing in the edge schemes)
        if (!pts_.empty()) release_ = std::thread([old = std::move(pts_)]() mutable { PointVector<rgb_point>().swap(old); });
        out.width = (uint32_t)n_;
        out.height = 1;
        out.is_dense = dense_;
        hand_over_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
    }

    // what finish() waited for: the copy of frame 0, the downloads still on their way; and the host time of all download_async calls
    double join_ms = 0, wait_ms = 0, append_ms = 0, hand_over_ms = 0;

  private:
    std::shared_ptr<Context> ctx_;
    PointVector<rgb_point> pts_;
    size_t n_;
    bool dense_, pending_ = false;
    double append_us_ = 0;   // (host time inside download_async)
    std::thread copy0_, release_;
};

inline void reference_icp_parameters(IterativeClosestPoint<rgb_point, rgb_point> &icp)
{
    icp.setMaximumIterations(100);
    icp.setMaxCorrespondenceDistance(0.01);
    icp.setTransformationEpsilon(1);
    icp.setEuclideanFitnessEpsilon(1000);
}
}  // namespace detail

// Frame-to-model chain: every frame is registered against everything merged so far.
class IncrementalICP : public RegistrationScheme {
  public:
    rgb_point_cloud_pointer registration(std::vector<rgb_point_cloud_pointer> &clouds) override
    {
        if (device_resident) return registration_device(clouds);
        return registration_host(clouds);
    }
    std::vector<Matrix4f> transforms;  // per merged frame (engine extra, for tests)

  private:
    rgb_point_cloud_pointer registration_device(std::vector<rgb_point_cloud_pointer> &clouds)
    {
        clock_start();
        {   // the streams, the pinned staging and the model's device buffer are made while this thread sets the loop up
            size_t largest = 0, total = 0;
            for (auto &c : clouds) { largest = std::max(largest, c->size()); total += c->size(); }
            Context::Default()->prepare(largest * sizeof(rgb_point), total * sizeof(rgb_point), true);
        }
        ApproximateVoxelGrid<rgb_point> voxel;   // leaf never set: PCL's 1 m default applies
        IterativeClosestPoint<rgb_point, rgb_point> icp;
        detail::reference_icp_parameters(icp);
        // four frames ahead on the PCIe link, three frames ahead in the voxel filter: the 1 m filter is one wave adding
        // floats one after the other (0.4 ms a frame) on a stream of its own; three of them run side by side under the
        // alignments of the frames before
        const size_t n = clouds.size();
        constexpr size_t kFilters = 3, kUploads = 4, kRing = kUploads + 1;
        rgb_device_cloud model, frames[kRing], reduced_of[kFilters + 1], aligned, moved;   // (inputs of queued jobs outlive their outputs)
        for (size_t k = 1; k < std::min<size_t>(kUploads + 1, n); ++k) frames[k % kRing].upload_deferred(*clouds[k]);   // (the worker starts on these ...)
        model.upload(*clouds[0]);                                                                                        // (... while frame 0 goes up from here)
        std::unique_ptr<detail::StreamedResult> result;
        if (stream_result && n > 1) {
            size_t capacity = 0;
            for (auto &c : clouds) capacity += c->size();
            result.reset(new detail::StreamedResult(model.context(), *clouds[0], capacity));
        }
        for (size_t k = 1; k < std::min<size_t>(kFilters + 1, n); ++k) voxel.filter_async(frames[k % kRing], reduced_of[k % (kFilters + 1)]);
        size_t merged_frames = 0;
        clock_mark();
        for (size_t k = 1; k < n; ++k, clock_mark()) {
            rgb_device_cloud &frame = frames[k % kRing], &reduced = reduced_of[k % (kFilters + 1)];
            // (frame k + kUploads takes the buffer of frame k - 1, the filtered frame k + kFilters that of frame k - 1)
            clock_lap(-1);
            if (k + kUploads < n) frames[(k + kUploads) % kRing].upload_deferred(*clouds[k + kUploads]);
            if (k + kFilters < n) voxel.filter_async(frames[(k + kFilters) % kRing], reduced_of[(k + kFilters) % (kFilters + 1)]);
            clock_lap(0);
            icp.setInputSource(reduced);
            clock_lap(1);
            icp.setInputTarget(model);
            clock_lap(2);
            icp.align(aligned);
            clock_lap(3);
            if (!icp.hasConverged()) continue;
            transformPointCloud(frame, moved, icp.getFinalTransformation());
            clock_lap(4);
            model += moved;
            clock_lap(5);
            if (result) result->append(moved);   // on its way to the host while the next frame is aligned
            clock_lap(6);
            ++merged_frames;
            transforms.push_back(icp.getFinalTransformation());
        }
        // the caller's frame 0 has become the merged cloud (incremental_icp.hpp:40,64)
        if (!result) model.download(*clouds[0]);
        else if (merged_frames) { result->finish(*clouds[0]); note_finish(*result); }
        clock_mark();
        return clouds[0];
    }
    rgb_point_cloud_pointer registration_host(std::vector<rgb_point_cloud_pointer> &clouds)
    {
        ApproximateVoxelGrid<rgb_point> voxel(Context::Default());  // leaf never set: PCL's 1 m default applies
        IterativeClosestPoint<rgb_point, rgb_point> icp;
        detail::reference_icp_parameters(icp);
        rgb_point_cloud_pointer model = clouds[0];  // aliases (and grows) the caller's frame 0
        auto reduced = std::make_shared<rgb_point_cloud>();
        for (size_t k = 1; k < clouds.size(); ++k) {
            rgb_point_cloud aligned;
            voxel.setInputCloud(clouds[k]);
            voxel.filter(*reduced);
            icp.setInputSource(reduced);
            icp.setInputTarget(model);
            icp.align(aligned);
            if (!icp.hasConverged()) continue;
            rgb_point_cloud moved;
            transformPointCloud(*clouds[k], moved, icp.getFinalTransformation());
            *model += moved;
            transforms.push_back(icp.getFinalTransformation());
        }
        return model;
    }
};

// A chain of frames registered as the INDEPENDENT consecutive pairs (k - 1, k), `in_flight` of them side by side on one GPU
// (BASELINE configs[4]).  The reference's IncrementalICP (src/incremental_icp.hpp:51-66) is sequential by
// definition -- frame k is aligned with everything merged before it; this class is the documented throughput restatement of that
// loop: every pair is the same ICP (same parameters, incremental_icp.hpp:46-49, unless `params` is changed) between two raw
// frames, and the pair transforms are composed on the host afterwards, T_0k = T_01 * ... * T_(k-1)k.
//
// Why side by side: one alignment leaves a third of the chip-time of every search launch to an emptying tail and ~6-11 us of
// nothing between two dependent launches; kernels of OTHER alignments fill both.  Every pair runs on a context
// of its own (stream, scratch, index, a queueing thread: a context is not thread-safe, different contexts are independent --
// include/rsreg.h), reads the frames where the home context has put them (rsreg_cloud_device_ptr: any stream may read a
// settled cloud), and gets exactly the bits it gets alone: nothing of an alignment depends on what else the GPU is doing.
// More than three or four contexts stop paying: the streams of a process share a few hardware queues.
class ChainRegistrar {
  public:
    explicit ChainRegistrar(size_t in_flight = 3, int device = 0) : device_(device)
    {
        rsreg_icp_params_reference(&params);
        home_ = std::make_shared<Context>(device);
        set_in_flight(in_flight);
    }
    // contexts are created once and keep their buffers from one registration() to the next
    void set_in_flight(size_t k)
    {
        k = std::max<size_t>(1, k);
        while (workers_.size() < k) workers_.push_back(std::make_shared<Context>(device_));
        in_flight_ = k;
    }
    size_t in_flight() const { return in_flight_; }

    rsreg_icp_params params;                    // of every pair (default: the reference's)
    std::vector<Matrix4f> pair_transforms;      // [k]: frame k into frame k - 1 ([0]: identity)
    std::vector<rsreg_icp_result> pair_results; // [k]: converged / state / iterations / n_correspondences of pair (k - 1, k)
    std::vector<int> pair_context;              // [k]: which context registered pair k (engine extra, for traces)

    // frames on the host in, T_0k (frame k into frame 0) out; guesses[k] (optional, one per frame, [0] unused): the initial
    // guess of pair (k - 1, k), else the identity (incremental_icp.hpp:59).  A pair that did not converge contributes the
    // transform ICP stopped at (the reference would skip the frame; `pair_results[k].converged` says so).
    std::vector<Matrix4f> registration(const std::vector<rgb_point_cloud_pointer> &frames, const std::vector<Matrix4f> &guesses = {})
    {
        const size_t n = frames.size();
        if (!guesses.empty() && guesses.size() != n) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: one guess per frame (the first is unused) or none");
        while (dev_.size() < n) dev_.emplace_back(new rgb_device_cloud(home_));
        for (size_t k = 0; k < n; ++k) dev_[k]->upload_deferred(*frames[k]);   // (the context's upload worker stages and sends them in this order)
        std::vector<Shared> shared(n);
        register_shared(shared, guesses, [&](size_t k) {
            // settled = complete in HBM, waited for on the host: from here on any stream may read the records
            shared[k].ptr = rsreg_cloud_device_ptr(dev_[k]->handle());
            shared[k].n = frames[k]->size();
            shared[k].dense = frames[k]->is_dense;
            if (shared[k].n && !shared[k].ptr) throw Error(RSREG_ERR_HIP, "rsreg: a frame did not reach the device");
        });
        return compose();
    }
    // the same on frames already resident in HBM (clouds of ANY context of this device; they must stay unchanged until the call returns)
    std::vector<Matrix4f> registration(const std::vector<const rgb_device_cloud *> &frames, const std::vector<Matrix4f> &guesses = {})
    {
        const size_t n = frames.size();
        if (!guesses.empty() && guesses.size() != n) throw Error(RSREG_ERR_INVALID_ARG, "rsreg: one guess per frame (the first is unused) or none");
        std::vector<Shared> shared(n);
        register_shared(shared, guesses, [&](size_t k) {
            size_t m = 0;
            uint32_t w = 0, h = 0;
            bool dense = false;
            frames[k]->info(m, w, h, dense);
            shared[k].ptr = rsreg_cloud_device_ptr(frames[k]->handle());
            shared[k].n = m;
            shared[k].dense = dense;
        });
        return compose();
    }

  private:
    struct Shared {
        const void *ptr = nullptr;
        size_t n = 0;
        bool dense = true;
    };
    template <class Settle> void register_shared(std::vector<Shared> &shared, const std::vector<Matrix4f> &guesses, Settle settle)
    {
        const size_t n = shared.size();
        pair_transforms.assign(n, Matrix4f::Identity());
        pair_results.assign(n, rsreg_icp_result{});
        pair_context.assign(n, -1);
        if (n < 2) {
            for (size_t k = 0; k < n; ++k) settle(k);
            return;
        }
        std::mutex mu;
        std::condition_variable cv;
        size_t ready = 0;            // frames [0, ready) are settled
        bool failed = false;
        std::atomic<size_t> next{1};
        std::vector<std::string> errors(in_flight_);
        auto work = [&](size_t w) {
            rsreg_ctx *c = workers_[w]->get();
            try {
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= n) return;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return ready > k || failed; });
                        if (failed) return;
                    }
                    // the source first, the reference's order (incremental_icp.hpp:57-58): it is put into the engine's order beside the target's index build
                    check(rsreg_icp_set_source_device(c, shared[k].ptr, shared[k].n, sizeof(rgb_point), shared[k].dense), c);
                    check(rsreg_icp_set_target_device(c, shared[k - 1].ptr, shared[k - 1].n, sizeof(rgb_point), shared[k - 1].dense,
                                                      params.max_correspondence_distance), c);
                    check(rsreg_icp_align(c, guesses.empty() ? nullptr : guesses[k].data(), &params, &pair_results[k], nullptr, 0), c);
                    std::memcpy(pair_transforms[k].m, pair_results[k].transform, sizeof(pair_transforms[k].m));
                    pair_context[k] = (int)w;
                }
            } catch (const std::exception &e) {
                errors[w] = e.what();
                std::lock_guard<std::mutex> lk(mu);
                failed = true;
                cv.notify_all();
            }
        };
        // every context gets a thread of its own; the caller's thread brings the frames in (they arrive in order: a worker starts on
        // pair k as soon as frames k - 1 and k are there, while the later frames are still on the link)
        std::vector<std::thread> th;
        for (size_t w = 0; w < in_flight_; ++w) th.emplace_back(work, w);
        std::string home_error;
        try {
            for (size_t k = 0; k < n; ++k) {
                settle(k);
                std::lock_guard<std::mutex> lk(mu);
                ready = k + 1;
                cv.notify_all();
            }
        } catch (const std::exception &e) {
            home_error = e.what();
            std::lock_guard<std::mutex> lk(mu);
            failed = true;
            cv.notify_all();
        }
        for (auto &t : th) t.join();
        if (!home_error.empty()) throw Error(RSREG_ERR_HIP, home_error);
        for (const std::string &e : errors)
            if (!e.empty()) throw Error(RSREG_ERR_STATE, e);
    }
    // T_0k = T_0(k-1) * T_(k-1)k, accumulated in double (the pair transforms are float), rounded once per pose
    std::vector<Matrix4f> compose() const
    {
        std::vector<Matrix4f> poses(pair_transforms.size(), Matrix4f::Identity());
        double acc[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (size_t k = 1; k < pair_transforms.size(); ++k) {
            double nxt[16];
            for (int j = 0; j < 4; ++j)
                for (int i = 0; i < 4; ++i) {
                    double v = 0;
                    for (int q = 0; q < 4; ++q) v += acc[q * 4 + i] * (double)pair_transforms[k].m[j * 4 + q];
                    nxt[j * 4 + i] = v;
                }
            for (int i = 0; i < 16; ++i) {
                acc[i] = nxt[i];
                poses[k].m[i] = (float)nxt[i];
            }
        }
        return poses;
    }

    int device_;
    size_t in_flight_ = 1;
    std::shared_ptr<Context> home_;
    std::vector<std::shared_ptr<Context>> workers_;
    std::vector<std::unique_ptr<rgb_device_cloud>> dev_;
};

// Shared skeleton of the two edge-based schemes: a coarse aligner that takes an initial
// guess, then an ICP refinement from the coarse result; the full clouds only get moved.
class EdgeBasedRegistrationBase : public TwoPhaseRegistrationScheme {
  public:
    EdgeBasedRegistrationBase() = default;
    explicit EdgeBasedRegistrationBase(std::vector<float3> &input_thetas) : thetas(input_thetas), use_imu(true) {}
    explicit EdgeBasedRegistrationBase(float usr_def_rads) : rads(usr_def_rads) {}

    rgb_point_cloud_pointer global_registration(PairList &clouds) override
    {
        if (device_resident) return global_registration_device(&clouds, nullptr);
        return global_registration_host(clouds);
    }
    // both phases (types.hpp:30-43).  With the frame loop resident in HBM and the built-in feature extractor a frame is
    // uploaded ONCE: its edge points are extracted where it lies and both stay there (the feature clouds never exist
    // on the host); a plugged-in feature_fn is a host function and goes the reference's way through host pairs.
    rgb_point_cloud_pointer registration(std::vector<rgb_point_cloud_pointer> &clouds) override
    {
        if (!device_resident || feature_fn) return TwoPhaseRegistrationScheme::registration(clouds);
        if (verbose) {   // the same lines in the same order; the features themselves are extracted as each frame reaches the GPU
            for (size_t k = 0; k < clouds.size(); ++k) std::cout << "[PCL] Extracting features..." << "OK" << std::endl;
            std::cout << "[PCL] Performing global registration..." << std::endl;
        }
        return global_registration_device(nullptr, &clouds);
    }
    std::vector<std::pair<Matrix4f, Matrix4f>> frame_transforms;  // (coarse, refine) per merged frame
    // ICPEdgeBasedRegistration writes files while it runs (icp_edge_based_registration.hpp:66-69,126): every frame's edge
    // cloud as <dir>/edge-<k>.pcd (frame 0's already voxel-filtered, the others as extracted) and the grown edge target
    // as <dir>/edge_cloud.pcd at the end, all with savePCDFileBinary.  Opt-in here (a library does not write into its
    // caller's working directory unasked); the directory must exist, as in the reference ("dataset").  The NDT scheme
    // writes nothing in the reference and nothing here.
    bool write_byproducts = false;
    std::string byproduct_dir = "dataset";

  protected:
    virtual const char *coarse_name() const = 0;       // "ICP" / "NDT": the reference's "Performing <name> iteration [k]..."
    virtual bool has_byproducts() const { return false; }
    bool byproducts_on() const { return write_byproducts && has_byproducts(); }
    void save_edge(size_t k, const rgb_point_cloud &edge) const
    {
        if (io::savePCDFileBinary(byproduct_dir + "/edge-" + std::to_string(k) + ".pcd", edge) != 0)
            throw Error(RSREG_ERR_INVALID_ARG, "rsreg: cannot write " + byproduct_dir + "/edge-" + std::to_string(k) + ".pcd");
    }
    void save_edge(size_t k, const rgb_device_cloud &edge) const
    {
        rgb_point_cloud host;
        edge.download(host);
        save_edge(k, host);
    }
    void save_edge_cloud(const rgb_point_cloud &target) const
    {
        if (io::savePCDFileBinary(byproduct_dir + "/edge_cloud.pcd", target) != 0)
            throw Error(RSREG_ERR_INVALID_ARG, "rsreg: cannot write " + byproduct_dir + "/edge_cloud.pcd");
    }
    void say_header() const
    {
        if (!verbose) return;
        std::cout << "[PCL] Performing edge-based registration";
        if (use_imu) std::cout << " with dynamic initial rotation guesses..." << std::endl;
        else std::cout << " with static initial rotation guesses..." << std::endl;
    }
    void say_iteration(const char *what, size_t k) const
    {
        if (verbose) std::cout << "[PCL]   Performing " << what << " iteration [" << k << "]..." << std::flush;
    }
    void say(const char *text) const
    {
        if (verbose) std::cout << text << std::endl;
    }

    Matrix4f next_guess(size_t k, float &acc_rads)
    {
        if (use_imu) {
            const float3 rel = thetas[0] * -1.0f;
            thetas[k].add(rel.x, rel.y, rel.z);  // the reference mutates thetas in place
            return imu_guess(thetas[k]);
        }
        acc_rads += rads;
        return Matrix4f::RotationY(acc_rads);
    }

    // `pairs` (features and frames on the host) or `frames` (frames only: features extracted on the GPU), one of the two
    rgb_point_cloud_pointer global_registration_device(PairList *pairs, std::vector<rgb_point_cloud_pointer> *frames)
    {
        const size_t n_frames = pairs ? pairs->size() : frames->size();
        clock_start();
        {   // the streams and the pinned staging are made while this thread sets the loop up (the merged cloud is streamed home: no model buffer)
            size_t largest = 0;
            for (size_t k = 0; k < n_frames; ++k) largest = std::max(largest, pairs ? (*pairs)[k].second->size() : (*frames)[k]->size());
            Context::Default()->prepare(largest * sizeof(rgb_point), 0, true);
        }
        say_header();
        if (use_imu) assert(n_frames == thetas.size());
        IterativeClosestPoint<rgb_point, rgb_point> icp;
        detail::reference_icp_parameters(icp);
        icp.setReuseTargetIndex(true);   // the coarse ICP of the ICP scheme has just built the index of the same target
        ApproximateVoxelGrid<rgb_point> voxel;
        voxel.setLeafSize(0.01f, 0.01f, 0.01f);
        configure_coarse();
        const rgb_point_cloud &frame0 = pairs ? *(*pairs)[0].second : *(*frames)[0];
        // frames only: the features of the next kAhead frames are extracted and voxel-filtered (a thread, streams and scratch of the
        // context's own) while frame k goes through its two alignments here, and frame k + kAhead + 1 is on the PCIe link: the
        // reference extracts all features before it registers anything (types.hpp:30-43), none depends on a registration
        // (kAhead = 3, end of round 6: the coarse ICP waits 0.02 ms a frame for its source instead of 0.07, and the loops are no
        //  faster -- ICP-edge level, NDT-edge 1-2 ms slower with more side work beside its passes)
        constexpr size_t kAhead = 2, kFulls = kAhead + 2, kFeat = kAhead + 1;
        rgb_device_cloud target, merged, fulls[kFulls], features_of[kFeat], reduced_of[kFeat], coarse_out, refined, moved;   // (inputs of queued jobs outlive their outputs)
        auto prepare = [&](size_t k) {
            extract_edge_features_async(fulls[k % kFulls], features_of[k % kFeat]);
            voxel.filter_async(features_of[k % kFeat], reduced_of[k % kFeat]);
        };
        if (!pairs)
            for (size_t k = 1; k < std::min<size_t>(kAhead + 2, n_frames); ++k) fulls[k % kFulls].upload_deferred(*(*frames)[k]);   // (the worker starts on these ...)
        merged.upload(frame0);                                                                                                  // (... while frame 0 goes up from here)
        // (`merged` on the GPU: frame 0 for its features, and the whole merged cloud only when it is downloaded at the end)
        std::unique_ptr<detail::StreamedResult> result;
        if (stream_result) {
            size_t capacity = 0;
            for (size_t k = 0; k < n_frames; ++k) capacity += pairs ? (*pairs)[k].second->size() : (*frames)[k]->size();
            result.reset(new detail::StreamedResult(merged.context(), frame0, capacity));
        }
        if (!pairs)
            for (size_t k = 1; k < std::min<size_t>(kAhead + 1, n_frames); ++k) prepare(k);
        if (pairs) target.upload(*(*pairs)[0].first);
        else extract_edge_features(merged, target);
        voxel.filter(target, target);   // frame-0 features: filtered in place, then grown
        if (byproducts_on()) save_edge(0, target);   // (the reference writes all edge-k.pcd before the loop; the files are the same)
        float acc_rads = 0.f;
        frame_transforms.clear();
        double coarse0[3];
        coarse_call_ms(coarse0);
        clock_mark();
        for (size_t k = 1; k < n_frames; ++k, clock_mark()) {
            rgb_device_cloud &full = fulls[k % kFulls], &features = features_of[k % kFeat], &reduced = reduced_of[k % kFeat];
            clock_lap(-1);
            if (pairs) {
                features.upload(*(*pairs)[k].first);
                voxel.filter(features, reduced);
            } else {
                // (frame k + kAhead + 1 takes the buffer of frame k - 1, the features of frame k + kAhead those of frame k - 1)
                if (k + kAhead + 1 < n_frames) fulls[(k + kAhead + 1) % kFulls].upload_deferred(*(*frames)[k + kAhead + 1]);
                if (k + kAhead < n_frames) prepare(k + kAhead);
            }
            if (byproducts_on()) save_edge(k, features);
            const Matrix4f guess = next_guess(k, acc_rads);
            say_iteration(coarse_name(), k);
            clock_lap(0);
            const Matrix4f t_coarse = coarse_align_device(reduced, target, coarse_out, guess);
            clock_lap(1);
            say("OK");
            icp.setInputSource(coarse_out);
            icp.setInputTarget(target);
            say_iteration("ICP", k);
            icp.align(refined);
            clock_lap(2);
            if (!icp.hasConverged()) {   // frame dropped, like the reference
                say("");
                continue;
            }
            say("OK");
            if (pairs) full.upload(*(*pairs)[k].second);
            transformPointCloud(full, moved, t_coarse);
            transformPointCloud(moved, moved, icp.getFinalTransformation());
            rgb_device_cloud::concatenate(refined, target, target);   // new points first
            clock_lap(3);
            if (result) result->append(moved);   // `*global = *global + *transformed`: on its way to the host already
            else merged += moved;
            clock_lap(4);
            frame_transforms.emplace_back(t_coarse, icp.getFinalTransformation());
        }
        if (pairs) target.download(*(*pairs)[0].first);   // the caller's frame-0 feature cloud has become the grown target
        if (byproducts_on()) {
            rgb_point_cloud grown;
            if (pairs) save_edge_cloud(*(*pairs)[0].first);
            else { target.download(grown); save_edge_cloud(grown); }
        }
        say("[PCL] Done");
        {   // (stage_ms[5..7]: the coarse ICP's share of stage_ms[1], call by call -- its source, the target's index, the alignment)
            double c1[3];
            coarse_call_ms(c1);
            for (int q = 0; q < 3; ++q) stage_ms[5 + q] = c1[q] - coarse0[q];
        }
        auto out = std::make_shared<rgb_point_cloud>();
        if (result) { result->finish(*out); note_finish(*result); }
        else merged.download(*out);
        clock_mark();
        out->width = (uint32_t)out->size();   // `*merged = *merged + ...`: an unorganized cloud whatever came in
        out->height = 1;
        return out;
    }

    rgb_point_cloud_pointer global_registration_host(PairList &clouds)
    {
        say_header();
        if (use_imu) assert(clouds.size() == thetas.size());
        IterativeClosestPoint<rgb_point, rgb_point> icp;
        detail::reference_icp_parameters(icp);
        ApproximateVoxelGrid<rgb_point> voxel(Context::Default());
        voxel.setLeafSize(0.01f, 0.01f, 0.01f);
        configure_coarse();

        rgb_point_cloud_pointer target = clouds[0].first;   // frame-0 features: filtered in place, then grown
        auto merged = std::make_shared<rgb_point_cloud>();
        *merged = *merged + *clouds[0].second;
        voxel.setInputCloud(target);
        voxel.filter(*target);
        auto reduced = std::make_shared<rgb_point_cloud>();
        if (byproducts_on())
            for (size_t k = 0; k < clouds.size(); ++k) save_edge(k, *clouds[k].first);   // icp_edge_based_registration.hpp:66-69
        float acc_rads = 0.f;
        frame_transforms.clear();
        for (size_t k = 1; k < clouds.size(); ++k) {
            auto coarse_out = std::make_shared<rgb_point_cloud>();
            rgb_point_cloud refined;
            voxel.setInputCloud(clouds[k].first);
            voxel.filter(*reduced);
            const Matrix4f guess = next_guess(k, acc_rads);
            say_iteration(coarse_name(), k);
            const Matrix4f t_coarse = coarse_align(reduced, target, *coarse_out, guess);
            say("OK");
            icp.setInputSource(coarse_out);
            icp.setInputTarget(target);
            say_iteration("ICP", k);
            icp.align(refined);
            if (!icp.hasConverged()) {   // frame dropped, like the reference
                say("");
                continue;
            }
            say("OK");
            rgb_point_cloud moved;
            transformPointCloud(*clouds[k].second, moved, t_coarse);
            transformPointCloud(moved, moved, icp.getFinalTransformation());
            *target = refined + *target;   // new points first
            *merged = *merged + moved;
            frame_transforms.emplace_back(t_coarse, icp.getFinalTransformation());
        }
        if (byproducts_on()) save_edge_cloud(*target);   // icp_edge_based_registration.hpp:126
        say("[PCL] Done");
        return merged;
    }

    virtual void configure_coarse() = 0;
    virtual Matrix4f coarse_align(const rgb_point_cloud_pointer &src, const rgb_point_cloud_pointer &tgt, rgb_point_cloud &out,
                                  const Matrix4f &guess) = 0;
    virtual void coarse_call_ms(double out[3]) const { out[0] = out[1] = out[2] = 0; }   // (engine extra: IterativeClosestPoint::call_ms of the coarse ICP)
    virtual Matrix4f coarse_align_device(const rgb_device_cloud &src, const rgb_device_cloud &tgt, rgb_device_cloud &out,
                                         const Matrix4f &guess) = 0;
    virtual Matrix4f imu_guess(const float3 &theta) const = 0;

    std::vector<float3> thetas;
    bool use_imu = false;
    float rads = -0.523599f;
};

class ICPEdgeBasedRegistration : public EdgeBasedRegistrationBase {
  public:
    using EdgeBasedRegistrationBase::EdgeBasedRegistrationBase;
  protected:
    const char *coarse_name() const override { return "ICP"; }
    bool has_byproducts() const override { return true; }
    void configure_coarse() override { detail::reference_icp_parameters(coarse_); }
    Matrix4f coarse_align(const rgb_point_cloud_pointer &src, const rgb_point_cloud_pointer &tgt, rgb_point_cloud &out,
                          const Matrix4f &guess) override
    {
        coarse_.setInputSource(src);
        coarse_.setInputTarget(tgt);
        coarse_.align(out, guess);
        return coarse_.getFinalTransformation();
    }
    Matrix4f coarse_align_device(const rgb_device_cloud &src, const rgb_device_cloud &tgt, rgb_device_cloud &out, const Matrix4f &guess) override
    {
        coarse_.setInputSource(src);
        coarse_.setInputTarget(tgt);
        coarse_.align(out, guess);
        return coarse_.getFinalTransformation();
    }
    void coarse_call_ms(double out[3]) const override { for (int k = 0; k < 3; ++k) out[k] = coarse_.call_ms[k]; }
    // AngleAxis(theta.x, Z) * AngleAxis(-theta.y, Y) * AngleAxis(theta.z, X), no translation
    Matrix4f imu_guess(const float3 &t) const override
    {
        return Matrix4f::RotationZ(t.x) * Matrix4f::RotationY(-t.y) * Matrix4f::RotationX(t.z);
    }
  private:
    IterativeClosestPoint<rgb_point, rgb_point> coarse_;
};

class NDTEdgeBasedRegistration : public EdgeBasedRegistrationBase {
  public:
    using EdgeBasedRegistrationBase::EdgeBasedRegistrationBase;
  protected:
    const char *coarse_name() const override { return "NDT"; }
    void configure_coarse() override
    {
        ndt_.setTransformationEpsilon(0.01);
        ndt_.setStepSize(0.1);
        ndt_.setResolution(1.0f);
        ndt_.setMaximumIterations(50);
    }
    Matrix4f coarse_align(const rgb_point_cloud_pointer &src, const rgb_point_cloud_pointer &tgt, rgb_point_cloud &out,
                          const Matrix4f &guess) override
    {
        ndt_.setInputSource(src);
        ndt_.setInputTarget(tgt);
        ndt_.align(out, guess);
        return ndt_.getFinalTransformation();
    }
    Matrix4f coarse_align_device(const rgb_device_cloud &src, const rgb_device_cloud &tgt, rgb_device_cloud &out, const Matrix4f &guess) override
    {
        ndt_.setInputSource(src);
        ndt_.setInputTarget(tgt);
        ndt_.align(out, guess);
        return ndt_.getFinalTransformation();
    }
    Matrix4f imu_guess(const float3 &t) const override { return Matrix4f::RotationY(-t.y); }
  private:
    NormalDistributionsTransform<rgb_point, rgb_point> ndt_;
};

}  // namespace rsreg
