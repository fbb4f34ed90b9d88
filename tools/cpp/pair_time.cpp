// pair_time.cpp -- what a C++ caller of the C ABI pays for one pair with the reference's parameters (setInputSource,
// setInputTarget, align with the aligned cloud; incremental_icp.hpp:57-63): no Python between the calls.
// usage: pair_time <target.f32> <source.f32> <n_target> <n_source> [host [copy]]   (32-byte PointXYZRGB records, raw)
// host: the literal call surface -- both clouds in HOST memory in, 4x4 and the aligned cloud in host memory out
// (rsreg_icp_set_source / rsreg_icp_set_target / rsreg_icp_align_records; INTEGRATION.md adaptor A), with the
// library's own account of where the host's time went (rsreg_ctx_host_timing); prints one JSON line.
// build: g++ -std=c++17 -O2 -I include tools/cpp/pair_time.cpp -o tools/_build/pair_time -L realsense-pointcloud_amd -lrsreg -Wl,-rpath,$PWD/realsense-pointcloud_amd
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rsreg.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { int rc_ = (x); if (rc_) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, rsreg_last_error(ctx)); return 1; } } while (0)

static std::vector<char> slurp(const char *path, size_t bytes)
{
    std::vector<char> v(bytes);
    FILE *f = std::fopen(path, "rb");
    if (!f || std::fread(v.data(), 1, bytes, f) != bytes) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    std::fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const size_t nt = std::strtoull(argv[3], nullptr, 10), ns = std::strtoull(argv[4], nullptr, 10);
    const std::vector<char> t = slurp(argv[1], nt * 32), s = slurp(argv[2], ns * 32);
    rsreg_ctx *ctx = nullptr;
    if (rsreg_ctx_create(0, nullptr, &ctx)) return 1;
    if (argc > 5 && argv[5][0] == 'h') {
        rsreg_icp_params prm;
        rsreg_icp_params_reference(&prm);
        rsreg_icp_result res;
        std::vector<char> aligned(ns * 32);
        const bool copy_outside = argc > 6 && argv[6][0] == 'c';
        const int reps = 30, skip = 5;
        std::vector<double> total, t_src, t_tgt, t_copy, t_align;
        std::vector<std::vector<double>> parts(7);
        for (int k = 0; k < reps; ++k) {
            CK(rsreg_ctx_synchronize(ctx));
            const double t0 = now_ms();
            CK(rsreg_icp_set_source(ctx, s.data(), ns, 32, 0));
            const double t1 = now_ms();
            CK(rsreg_icp_set_target(ctx, t.data(), nt, 32, 0, prm.max_correspondence_distance));
            const double t2 = now_ms();
            // PCL's align(output) starts from output = *input (colours and padding): rounds 1-5, the caller's own memcpy in front of
            // rsreg_icp_align (`copy` mode, one thread: 0.9 ms at 10^6 points); round 6, inside rsreg_icp_align_records
            if (copy_outside) std::memcpy(aligned.data(), s.data(), ns * 32);
            const double t3 = now_ms();
            if (copy_outside) CK(rsreg_icp_align(ctx, nullptr, &prm, &res, aligned.data(), 32));
            else CK(rsreg_icp_align_records(ctx, nullptr, &prm, &res, s.data(), aligned.data(), 32));
            const double t4 = now_ms();
            if (k < skip) continue;
            total.push_back(t4 - t0); t_src.push_back(t1 - t0); t_tgt.push_back(t2 - t1); t_copy.push_back(t3 - t2); t_align.push_back(t4 - t3);
            rsreg_host_timing ht;
            CK(rsreg_ctx_host_timing(ctx, &ht));
            const double v[7] = {ht.source_stage_wait, ht.source_pack, ht.target_stage_wait, ht.target_pack, ht.target_build, ht.align, ht.aligned_copy};
            for (int j = 0; j < 7; ++j) parts[j].push_back(v[j]);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        std::printf("{\"n_source\": %zu, \"n_target\": %zu, \"input_records_copied\": \"%s\", \"ms_per_pair\": %.4f, \"p10\": %.4f, \"p90\": %.4f, \"set_source\": %.4f, \"set_target\": %.4f, "
                    "\"copy_input_records\": %.4f, \"align\": %.4f, \"inside\": {\"source_stage_wait\": %.4f, \"source_pack\": %.4f, \"target_stage_wait\": %.4f, "
                    "\"target_pack\": %.4f, \"target_build\": %.4f, \"align_iterations\": %.4f, \"aligned_cloud_home\": %.4f}, \"iterations\": %d, "
                    "\"n_correspondences\": %llu, \"pairs_timed\": %zu}\n",
                    ns, nt, copy_outside ? "by the caller, in front of rsreg_icp_align" : "inside rsreg_icp_align_records", med(total), [&] { auto v = total; std::sort(v.begin(), v.end()); return v[v.size() / 10]; }(),
                    [&] { auto v = total; std::sort(v.begin(), v.end()); return v[v.size() * 9 / 10]; }(), med(t_src), med(t_tgt), med(t_copy), med(t_align),
                    med(parts[0]), med(parts[1]), med(parts[2]), med(parts[3]), med(parts[4]), med(parts[5]), med(parts[6]), res.iterations,
                    (unsigned long long)res.n_correspondences, total.size());
        rsreg_ctx_destroy(ctx);
        return 0;
    }
    rsreg_cloud *ct, *cs, *out;
    CK(rsreg_cloud_create(ctx, &ct)); CK(rsreg_cloud_create(ctx, &cs)); CK(rsreg_cloud_create(ctx, &out));
    CK(rsreg_cloud_upload(ct, t.data(), nt, 32, (uint32_t)nt, 1, 0));
    CK(rsreg_cloud_upload(cs, s.data(), ns, 32, (uint32_t)ns, 1, 0));
    rsreg_icp_params prm;
    rsreg_icp_params_reference(&prm);
    rsreg_icp_result res;
    for (int fresh = 0; fresh < 2; ++fresh) {
        std::vector<double> ms;
        for (int k = 0; k < 105; ++k) {
            if (fresh) {   // new records under the handles (outside the clock): boxes measured again, caches cold from the copy
                CK(rsreg_cloud_upload(ct, t.data(), nt, 32, (uint32_t)nt, 1, 0));
                CK(rsreg_cloud_upload(cs, s.data(), ns, 32, (uint32_t)ns, 1, 0));
            }
            CK(rsreg_ctx_synchronize(ctx));
            const double t0 = now_ms();
            CK(rsreg_icp_set_source_cloud(ctx, cs));
            CK(rsreg_icp_set_target_cloud(ctx, ct, prm.max_correspondence_distance));
            CK(rsreg_icp_align_cloud(ctx, nullptr, &prm, &res, out));
            CK(rsreg_ctx_synchronize(ctx));
            if (k >= 5) ms.push_back(now_ms() - t0);
        }
        std::sort(ms.begin(), ms.end());
        std::printf("C++ caller, %zu x %zu points, reference parameters, %s: median %.3f ms per pair (p10 %.3f, p90 %.3f), %d iteration(s), %llu correspondences\n",
                    ns, nt, fresh ? "new records under the handles every time" : "the same clouds again (boxes measured)", ms[ms.size() / 2], ms[ms.size() / 10],
                    ms[ms.size() * 9 / 10], res.iterations, (unsigned long long)res.n_correspondences);
    }
    rsreg_cloud_destroy(ct); rsreg_cloud_destroy(cs); rsreg_cloud_destroy(out);
    rsreg_ctx_destroy(ctx);
    return 0;
}
