// small_sort_parts.hip — dev probe: the two kernels of tools/microbench/small_sort.hpp in variants, timed apart (48 k pairs).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/small_sort_parts.hip -o tools/_build/small_sort_parts
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/block/block_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <unsigned B, unsigned IPT, unsigned Bits, rocprim::block_radix_rank_algorithm A>
__global__ __launch_bounds__(B) void k_runs(const uint32_t *keys, const uint32_t *vals, uint32_t n, unsigned end_bit, uint32_t *rk, uint32_t *rv)
{
    using Sort = rocprim::block_radix_sort<uint32_t, B, IPT, uint32_t, 1, 1, Bits, A>;
    __shared__ typename Sort::storage_type storage;
    const uint32_t base = blockIdx.x * B * IPT + threadIdx.x * IPT;
    uint32_t k[IPT], v[IPT];
#pragma unroll
    for (uint32_t j = 0; j < IPT; ++j) {
        const bool in = base + j < n;
        k[j] = in ? keys[base + j] : ~0u;
        v[j] = in ? vals[base + j] : 0u;
    }
    Sort().sort(k, v, storage, 0, end_bit);
#pragma unroll
    for (uint32_t j = 0; j < IPT; ++j)
        if (base + j < n) { rk[base + j] = k[j]; rv[base + j] = v[j]; }
}

// all other runs searched at once (R = runs at most), branch-free
template <unsigned RunBits, unsigned R, unsigned TB>
__global__ __launch_bounds__(TB) void k_place(const uint32_t *rk, const uint32_t *rv, uint32_t n, uint32_t *ok, uint32_t *ov)
{
    constexpr uint32_t Run = 1u << RunBits;
    const uint32_t e = blockIdx.x * TB + threadIdx.x;
    if (e >= n) return;
    const uint32_t run = e >> RunBits, n_runs = (n + Run - 1) >> RunBits;
    const uint32_t key = rk[e];
    uint32_t pos[R];
#pragma unroll
    for (uint32_t r = 0; r < R; ++r) pos[r] = 0;
#pragma unroll 1
    for (uint32_t step = Run; step; step >>= 1) {
#pragma unroll
        for (uint32_t r = 0; r < R; ++r) {
            const uint32_t base = r << RunBits;
            const uint32_t len = r < n_runs ? min(Run, n - base) : 0u;
            const uint32_t at = pos[r] + step;
            const bool ok_at = at <= len;
            const uint32_t other = rk[ok_at ? base + at - 1 : e];
            const bool take = ok_at && (r < run ? other <= key : other < key);
            pos[r] = take ? at : pos[r];
        }
    }
    uint32_t place = e & (Run - 1);
#pragma unroll
    for (uint32_t r = 0; r < R; ++r) place += (r == run) ? 0u : pos[r];
    ok[place] = key;
    ov[place] = rv[e];
}

template <class F>
float time_it(F f)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float sum = 0;
    for (int r = 0; r < 22; ++r) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        f();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) sum += ms;
    }
    return sum / 20 * 1e3f;
}

uint32_t *k0, *v0, *rk, *rv, *ok, *ov;
std::vector<uint32_t> hk, expect;

template <unsigned B, unsigned IPT, unsigned Bits, rocprim::block_radix_rank_algorithm A>
void runs(const char *name, uint32_t n, unsigned end_bit)
{
    const uint32_t per = B * IPT;
    float t = time_it([&] { hipLaunchKernelGGL((k_runs<B, IPT, Bits, A>), dim3((n + per - 1) / per), dim3(B), 0, 0, k0, v0, n, end_bit, rk, rv); });
    printf("runs  %-30s n %u end_bit %u: %.1f us\n", name, n, end_bit, t);
}

template <unsigned RunBits, unsigned R, unsigned TB>
void place(const char *name, uint32_t n)
{
    float t = time_it([&] { hipLaunchKernelGGL((k_place<RunBits, R, TB>), dim3((n + TB - 1) / TB), dim3(TB), 0, 0, rk, rv, n, ok, ov); });
    std::vector<uint32_t> got(n);
    CHECK(hipMemcpy(got.data(), ok, n * 4, hipMemcpyDeviceToHost));
    printf("place %-30s n %u: %.1f us %s\n", name, n, t, got == expect ? "sorted" : "WRONG");
}

int main()
{
    using A = rocprim::block_radix_rank_algorithm;
    for (uint32_t n : {36049u, 47851u, 65536u}) {
        hk.resize(n);
        std::mt19937 rng(n);
        for (auto &k : hk) k = rng();
        expect = hk;
        std::sort(expect.begin(), expect.end());
        std::vector<uint32_t> hv(n);
        for (uint32_t i = 0; i < n; ++i) hv[i] = i;
        CHECK(hipMalloc(&k0, n * 4)); CHECK(hipMalloc(&v0, n * 4)); CHECK(hipMalloc(&rk, n * 4)); CHECK(hipMalloc(&rv, n * 4)); CHECK(hipMalloc(&ok, n * 4)); CHECK(hipMalloc(&ov, n * 4));
        CHECK(hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
        runs<256, 4, 0, A::default_for_radix_sort>("256 x 4 default", n, 32);
        runs<256, 8, 0, A::default_for_radix_sort>("256 x 8 default", n, 32);
        runs<256, 16, 0, A::default_for_radix_sort>("256 x 16 default", n, 32);
        runs<512, 8, 0, A::default_for_radix_sort>("512 x 8 default", n, 32);
        runs<1024, 4, 0, A::default_for_radix_sort>("1024 x 4 default", n, 32);
        runs<256, 16, 8, A::match>("256 x 16, 8 bits match", n, 32);
        runs<512, 8, 8, A::match>("512 x 8, 8 bits match", n, 32);
        runs<1024, 4, 8, A::match>("1024 x 4, 8 bits match", n, 32);
        runs<256, 16, 6, A::basic_memoize>("256 x 16, 6 bits memoize", n, 32);
        runs<512, 8, 6, A::basic_memoize>("512 x 8, 6 bits memoize", n, 32);
        runs<1024, 4, 6, A::basic_memoize>("1024 x 4, 6 bits memoize", n, 32);
        runs<512, 8, 5, A::basic_memoize>("512 x 8, 5 bits memoize", n, 32);
        // the place kernel on 4096-item runs made by the last variant run with 4096 items
        runs<1024, 4, 0, A::default_for_radix_sort>("1024 x 4 default (for place)", n, 32);
        place<12, 16, 256>("run 4096, 16 at once, 256 thr", n);
        place<12, 16, 128>("run 4096, 16 at once, 128 thr", n);
        place<12, 16, 64>("run 4096, 16 at once, 64 thr", n);
        runs<256, 8, 0, A::default_for_radix_sort>("256 x 8 default (for place)", n, 32);
        place<11, 32, 256>("run 2048, 32 at once, 256 thr", n);
        place<11, 32, 64>("run 2048, 32 at once, 64 thr", n);
        CHECK(hipFree(k0)); CHECK(hipFree(v0)); CHECK(hipFree(rk)); CHECK(hipFree(rv)); CHECK(hipFree(ok)); CHECK(hipFree(ov));
    }
    return 0;
}
