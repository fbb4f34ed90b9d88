// dev: what the library's upload / download workers give a frame loop, without any registration in between:
//   (a) 16 frames of 307 200 points up (upload_deferred, then wait for the last), (b) the same with every frame going home
//   again into one fresh 157 MB result (download_async) while the next ones come up, (c) downloads alone.
// g++ -std=c++17 -O2 -pthread -DRSREG_PCL_COMPAT_FAST_UNINIT -Iinclude tools/cpp/link_pipeline.cpp -o /tmp/link_pipeline -Lrealsense-pointcloud_amd -lrsreg -Wl,-rpath,$PWD/realsense-pointcloud_amd
#include <chrono>
#include <cstdio>
#include <memory>
#include <vector>
#include <cstdlib>
#include <sys/mman.h>
#include "rsreg/pcl_compat.hpp"
using namespace rsreg;
using rgb_point = PointXYZRGB;
using cloud_t = PointCloud<rgb_point>;
using dcloud_t = DeviceCloud<rgb_point>;
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 307200, frames = 16;
    std::vector<std::shared_ptr<cloud_t>> host;
    for (size_t k = 0; k < frames; ++k) {
        auto c = std::make_shared<cloud_t>();
        c->points.resize(n);
        for (size_t i = 0; i < n; ++i) { c->points[i].x = (float)i * 1e-5f; c->points[i].y = (float)k; c->points[i].z = 1.0f; }
        c->width = 640; c->height = 480; c->is_dense = true;
        host.push_back(c);
    }
    auto ctx = Context::Default();
    std::vector<std::unique_ptr<dcloud_t>> dev;
    for (size_t k = 0; k < frames; ++k) dev.emplace_back(new dcloud_t(ctx));
    for (int rep = 0; rep < 4; ++rep) {
        double t0 = now();
        for (size_t k = 0; k < frames; ++k) dev[k]->upload_deferred(*host[k]);
        for (size_t k = 0; k < frames; ++k) (void)rsreg_cloud_device_ptr(dev[k]->handle());   // settled = in HBM
        double t1 = now();
        std::fprintf(stderr, "run %d: 16 uploads            %.2f ms = %.2f ms a frame\n", rep, t1 - t0, (t1 - t0) / frames);
        {
            PointVector<rgb_point> result = uninitialized_points<rgb_point>(n * frames);
            t0 = now();
            for (size_t k = 0; k < frames; ++k) dev[k]->download_async(result.data() + k * n, n);
            ctx->wait_downloads();
            t1 = now();
            std::fprintf(stderr, "run %d: 16 downloads (fresh)  %.2f ms = %.2f ms a frame\n", rep, t1 - t0, (t1 - t0) / frames);
            t0 = now();
            for (size_t k = 0; k < frames; ++k) dev[k]->download_async(result.data() + k * n, n);
            ctx->wait_downloads();
            t1 = now();
            std::fprintf(stderr, "run %d: 16 downloads (touched) %.2f ms = %.2f ms a frame\n", rep, t1 - t0, (t1 - t0) / frames);
        }
        for (int advice : {-1, MADV_HUGEPAGE, MADV_NOHUGEPAGE}) {   // the destination's pages: as the system gives them / huge / small
            void *raw = nullptr;
            const size_t bytes = n * frames * sizeof(rgb_point);
            if (posix_memalign(&raw, 2u << 20, bytes)) return 1;
            if (advice >= 0) madvise(raw, bytes, advice);
            rgb_point *dst = static_cast<rgb_point *>(raw);
            t0 = now();
            for (size_t k = 0; k < frames; ++k) dev[k]->download_async(dst + k * n, n);
            ctx->wait_downloads();
            t1 = now();
            std::fprintf(stderr, "run %d: 16 downloads into fresh 2 MB-aligned memory, %s: %.2f ms = %.2f ms a frame\n", rep,
                         advice < 0 ? "no advice" : (advice == MADV_HUGEPAGE ? "MADV_HUGEPAGE" : "MADV_NOHUGEPAGE"), t1 - t0, (t1 - t0) / frames);
            free(raw);
        }
        {
            PointVector<rgb_point> result = uninitialized_points<rgb_point>(n * frames);
            t0 = now();
            for (size_t k = 0; k < frames; ++k) dev[k]->upload_deferred(*host[k]);
            for (size_t k = 0; k < frames; ++k) {
                (void)rsreg_cloud_device_ptr(dev[k]->handle());
                dev[k]->download_async(result.data() + k * n, n);
            }
            ctx->wait_downloads();
            t1 = now();
            std::fprintf(stderr, "run %d: up and home again     %.2f ms = %.2f ms a frame\n", rep, t1 - t0, (t1 - t0) / frames);
        }
    }
    return 0;
}
