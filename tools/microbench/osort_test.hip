// osort_test.hip — the library's own radix sort (csrc/osort.hpp) against std::stable_sort, 1 .. 3 x 10^6 pairs, 7 .. 32 key bits, with timings.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/osort_test.hip -o tools/_build/osort_test
#include "../../realsense-pointcloud_amd/csrc/osort.hpp"
#include <cstdio>
#include <vector>
#include <random>
#include <numeric>
using namespace rsreg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main()
{
    std::mt19937 rng(7);
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (size_t n : {1ul, 63ul, 4096ul, 4097ul, 36000ul, 300007ul, 1000000ul, 3000001ul}) {
        for (unsigned bits : {7u, 8u, 10u, 17u, 21u, 32u}) {
            std::vector<uint32_t> k(n), v(n);
            for (size_t i = 0; i < n; ++i) { k[i] = rng() & (bits == 32 ? 0xffffffffu : ((1u << bits) - 1u)); if (i % 7 == 0) k[i] &= 0xff; v[i] = (uint32_t)i; }
            uint32_t *ka, *kb, *va, *vb, *scr;
            const OsortPlan p = osort_plan(n, 0, bits);
            CHECK(hipMalloc(&ka, n * 4 + 4)); CHECK(hipMalloc(&kb, n * 4 + 4)); CHECK(hipMalloc(&va, n * 4 + 4)); CHECK(hipMalloc(&vb, n * 4 + 4));
            CHECK(hipMalloc(&scr, (size_t)p.words * 4 + 4));
            CHECK(hipMemcpy(ka, k.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(va, v.data(), n * 4, hipMemcpyHostToDevice));
            float best = 1e30f;
            bool first = true;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemcpy(ka, k.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(va, v.data(), n * 4, hipMemcpyHostToDevice));
                CHECK(hipMemsetAsync(scr, 0, (size_t)p.words * 4, st));
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipEventRecord(e0, st));
                CHECK(osort_pairs(p, scr, ka, kb, va, vb, n, 0, bits, st, &first));
                CHECK(hipEventRecord(e1, st));
                CHECK(hipStreamSynchronize(st));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
            }
            std::vector<uint32_t> ok(n), ov(n);
            CHECK(hipMemcpy(ok.data(), first ? ka : kb, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ov.data(), first ? va : vb, n * 4, hipMemcpyDeviceToHost));
            std::vector<uint32_t> idx(n); std::iota(idx.begin(), idx.end(), 0u);
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
            size_t bad = 0;
            for (size_t i = 0; i < n; ++i) if (ok[i] != k[idx[i]] || ov[i] != idx[i]) { if (!bad) printf("  first mismatch at %zu: key %u val %u, want key %u val %u\n", i, ok[i], ov[i], k[idx[i]], idx[i]); ++bad; }
            printf("n %8zu bits %2u passes %u: %s  %.1f us\n", n, bits, p.passes, bad ? "MISMATCH" : "ok", best * 1e3);
            hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(scr);
        }
    }
    return 0;
}
