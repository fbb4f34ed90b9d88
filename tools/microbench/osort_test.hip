// osort_test.hip — the library's own radix sort (csrc/osort.hpp: 32- and 64-bit keys) against std::stable_sort, 1 .. 3 x 10^6 pairs,
// 7 .. 64 key bits, and its own prefix sums (csrc/oscan.hpp: uint32 / uint64, exclusive / inclusive, in place) against a host loop; with timings.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/osort_test.hip -o tools/_build/osort_test
#include "../../realsense-pointcloud_amd/csrc/osort.hpp"
#include "../../realsense-pointcloud_amd/csrc/oscan.hpp"
#include <cstdio>
#include <vector>
#include <random>
#include <numeric>
using namespace rsreg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static int g_bad = 0;

template <typename K> int sort_case(size_t n, unsigned bits, std::mt19937_64 &rng, hipStream_t st)
{
    std::vector<K> k(n);
    std::vector<uint32_t> v(n);
    const K mask = bits == sizeof(K) * 8 ? ~(K)0 : (((K)1 << bits) - 1);
    for (size_t i = 0; i < n; ++i) { k[i] = (K)rng() & mask; if (i % 7 == 0) k[i] &= 0xff; v[i] = (uint32_t)i; }
    K *ka, *kb;
    uint32_t *va, *vb, *scr;
    const OsortPlan p = osort_plan<K>(n, 0, bits);
    CHECK(hipMalloc(&ka, n * sizeof(K) + 8)); CHECK(hipMalloc(&kb, n * sizeof(K) + 8)); CHECK(hipMalloc(&va, n * 4 + 4)); CHECK(hipMalloc(&vb, n * 4 + 4));
    CHECK(hipMalloc(&scr, (size_t)p.words * 4 + 4));
    float best = 1e30f;
    bool first = true;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemcpy(ka, k.data(), n * sizeof(K), hipMemcpyHostToDevice)); CHECK(hipMemcpy(va, v.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemsetAsync(scr, 0, (size_t)p.words * 4, st));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, st));
        CHECK(osort_pairs<K>(p, scr, ka, kb, va, vb, n, 0, bits, st, &first));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    if (first != osort_ends_in_first<K>(p, n)) { printf("  osort_ends_in_first disagrees with the sort\n"); ++g_bad; }
    std::vector<K> ok(n);
    std::vector<uint32_t> ov(n);
    CHECK(hipMemcpy(ok.data(), first ? ka : kb, n * sizeof(K), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ov.data(), first ? va : vb, n * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> idx(n); std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) if (ok[i] != k[idx[i]] || ov[i] != idx[i]) { if (!bad) printf("  first mismatch at %zu\n", i); ++bad; }
    printf("sort  %2zu-bit keys n %8zu bits %2u passes %u: %s  %.1f us\n", sizeof(K) * 8, n, bits, p.passes, bad ? "MISMATCH" : "ok", best * 1e3);
    if (bad) ++g_bad;
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(scr);
    return 0;
}

template <typename T, bool kInclusive> int scan_case(size_t n, bool in_place, std::mt19937_64 &rng, hipStream_t st)
{
    std::vector<T> h(n), want(n), got(n);
    for (size_t i = 0; i < n; ++i) h[i] = (T)(rng() & (sizeof(T) == 8 ? 0xffffffffffull : 0xfffull));
    const T init = 5;
    T run = init;
    for (size_t i = 0; i < n; ++i) { if (kInclusive) run += h[i]; want[i] = run; if (!kInclusive) run += h[i]; }
    T *din, *dout;
    void *scr;
    CHECK(hipMalloc(&din, n * sizeof(T) + 8)); CHECK(hipMalloc(&dout, n * sizeof(T) + 8)); CHECK(hipMalloc(&scr, oscan_scratch_bytes<T>(n) + 8));
    CHECK(hipMemset(scr, 0xff, oscan_scratch_bytes<T>(n)));   // (contents irrelevant)
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemcpy(din, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, st));
        CHECK((oscan<T, kInclusive>(din, in_place ? din : dout, n, init, scr, st)));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    CHECK(hipMemcpy(got.data(), in_place ? din : dout, n * sizeof(T), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) if (got[i] != want[i]) { if (!bad) printf("  first mismatch at %zu\n", i); ++bad; }
    printf("scan  %2zu-bit %s%s n %9zu: %s  %.1f us\n", sizeof(T) * 8, kInclusive ? "inclusive" : "exclusive", in_place ? " in place" : "", n, bad ? "MISMATCH" : "ok", best * 1e3);
    if (bad) ++g_bad;
    hipFree(din); hipFree(dout); hipFree(scr);
    return 0;
}

int main()
{
    std::mt19937_64 rng(7);
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (size_t n : {1ul, 63ul, 2048ul, 2049ul, 4096ul, 4097ul, 36000ul, 300007ul, 1000000ul, 3000001ul}) {
        for (unsigned bits : {7u, 8u, 10u, 17u, 21u, 32u})
            if (sort_case<uint32_t>(n, bits, rng, st)) return 1;
        for (unsigned bits : {9u, 33u, 40u, 49u, 64u})
            if (sort_case<unsigned long long>(n, bits, rng, st)) return 1;
    }
    for (size_t n : {1ul, 255ul, 1024ul, 1025ul, 36000ul, 1000000ul, 1048577ul, 7400000ul, 40000001ul}) {
        if (scan_case<uint32_t, false>(n, false, rng, st)) return 1;
        if (scan_case<uint32_t, false>(n, true, rng, st)) return 1;
        if (scan_case<uint32_t, true>(n, false, rng, st)) return 1;
        if (scan_case<unsigned long long, false>(n, false, rng, st)) return 1;
    }
    printf(g_bad ? "FAILED: %d case(s)\n" : "all sorts and scans ok\n", g_bad);
    return g_bad ? 1 : 0;
}
