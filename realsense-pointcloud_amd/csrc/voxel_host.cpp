// voxel_host.cpp — pcl::ApproximateVoxelGrid<PointXYZRGB>::filter on the host.
//
// Reference call sites: src/incremental_icp.hpp:54-55 (default 1 m leaf),
// src/icp_edge_based_registration.hpp:47,59-60,75-76, src/ndt_edge_based_registration.hpp:45,
// 57-58,68-69 (1 cm leaf, also in place).  The filter is a sequential stream over the points
// with a 512-slot hash history that flushes on collision: its output
// depends on the input ORDER and may hold several centroids per voxel, so a parallel version
// cannot reproduce it record for record.  It runs on ~30 k-point edge clouds and is O(N);
// it stays on the host for parity (DESIGN.md "what stays on the host").
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rsreg.h"

namespace {

struct Slot {
    int ix = 0, iy = 0, iz = 0, count = 0;
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};  // x y z rgb-as-float r g b
};

void emit(unsigned char *out, size_t &n_out, size_t stride, const Slot &s)
{
    const float n = static_cast<float>(s.count);
    float c[7];
    for (int k = 0; k < 7; ++k) c[k] = s.acc[k] / n;
    unsigned char *rec = out + n_out * stride;  // a default PointXYZRGB: zeros, data[3] = 1
    ++n_out;
    std::memset(rec, 0, stride);
    const float one = 1.0f;
    std::memcpy(rec, c, 12);
    std::memcpy(rec + 12, &one, 4);
    const int rgb = (static_cast<int>(c[4]) << 16) | (static_cast<int>(c[5]) << 8) | static_cast<int>(c[6]);
    std::memcpy(rec + 16, &rgb, 4);
}

}  // namespace

extern "C" int rsreg_approx_voxel_grid(const void *in, size_t n, size_t stride, const float leaf[3], void *out,
                                       size_t *n_out)
{
    if (!leaf || !n_out || (n && (!in || !out)) || stride < 20) return RSREG_ERR_INVALID_ARG;
    if (!(leaf[0] > 0) || !(leaf[1] > 0) || !(leaf[2] > 0)) return RSREG_ERR_INVALID_ARG;
    constexpr int kHist = 512;
    std::vector<Slot> hist(kHist);
    const float inv[3] = {1.0f / leaf[0], 1.0f / leaf[1], 1.0f / leaf[2]};
    // in == out is allowed, and the output can only fall behind the input (a record is emitted
    // after at least one more input record has been read) except in the final flush -- but the
    // slot an emission frees may still be read as input later, so an aliased call works aside
    std::vector<unsigned char> aside;
    unsigned char *result = static_cast<unsigned char *>(out);
    if (in == out) {
        aside.resize(n * stride);
        result = aside.data();
    }
    size_t count = 0;
    const unsigned char *src = static_cast<const unsigned char *>(in);
    for (size_t i = 0; i < n; ++i) {
        const unsigned char *rec = src + i * stride;
        float xyz[3], rgbf;
        unsigned char bgra[4];
        std::memcpy(xyz, rec, 12);
        std::memcpy(&rgbf, rec + 16, 4);
        std::memcpy(bgra, rec + 16, 4);
        if (!std::isfinite(xyz[0]) || !std::isfinite(xyz[1]) || !std::isfinite(xyz[2])) continue;
        const int ix = static_cast<int>(std::floor(xyz[0] * inv[0]));
        const int iy = static_cast<int>(std::floor(xyz[1] * inv[1]));
        const int iz = static_cast<int>(std::floor(xyz[2] * inv[2]));
        const unsigned h = static_cast<unsigned>((ix * 7171 + iy * 3079 + iz * 4231) & (kHist - 1));
        Slot &s = hist[h];
        if (s.count && (s.ix != ix || s.iy != iy || s.iz != iz)) {
            emit(result, count, stride, s);
            s = Slot();
        }
        s.ix = ix; s.iy = iy; s.iz = iz;
        ++s.count;
        const float add[7] = {xyz[0], xyz[1], xyz[2], rgbf, float(bgra[2]), float(bgra[1]), float(bgra[0])};
        for (int k = 0; k < 7; ++k) s.acc[k] += add[k];
    }
    for (const Slot &s : hist)
        if (s.count) emit(result, count, stride, s);
    if (in == out && count) std::memcpy(out, result, count * stride);
    *n_out = count;
    return RSREG_OK;
}
