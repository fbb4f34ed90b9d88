// radix32_configs.hip (tiling sweep of the digit passes; see radix32_small.hip)
// radix32_small.hip — the engine's own onesweep driver (csrc/radix32.hpp: rocPRIM's device functions, state cleared by one
// kernel, no memsets) against rocprim::radix_sort_pairs (merge-sort path below 65 536 items, onesweep above) on (uint32, uint32)
// pairs of the sizes and key widths the frame loops sort.  Outputs are compared.
// Build: hipcc --offload-arch=gfx950 -O3 -I realsense-pointcloud_amd/csrc tools/microbench/radix32_small.hip -o tools/_build/radix32_small
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "radix32.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace rsreg;

__global__ void k_fill(const uint32_t *src, uint32_t *keys, uint32_t *vals, uint32_t n, uint32_t mask, uint32_t *scratch, uint32_t words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (scratch) radix32_clear(scratch, words, i, gridDim.x * blockDim.x);
    if (i < n) { keys[i] = src[i] & mask; vals[i] = i; }
}

template <unsigned B, unsigned I>
float run_own(size_t n, unsigned bits, const uint32_t *src, uint32_t *k0, uint32_t *k1, uint32_t *v0, uint32_t *v1, uint32_t *scratch, std::vector<uint32_t> &out_k,
              std::vector<uint32_t> &out_v)
{
    const Radix32Plan plan = radix32_plan(n, 0, bits, B * I);
    const uint32_t mask = bits >= 32 ? 0xffffffffu : (1u << bits) - 1u;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float sum = 0;
    const int reps = 20;
    bool in_first = false;
    for (int r = 0; r < reps + 2; ++r) {
        k_fill<<<(unsigned)((n + 255) / 256), 256>>>(src, k0, v0, (uint32_t)n, mask, scratch, plan.words);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        CHECK((radix32_sort_pairs<B, I>(plan, scratch, k0, k1, v0, v1, n, 0, bits, 0, &in_first)));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) sum += ms;
    }
    out_k.resize(n); out_v.resize(n);
    CHECK(hipMemcpy(out_k.data(), in_first ? k0 : k1, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(out_v.data(), in_first ? v0 : v1, n * 4, hipMemcpyDeviceToHost));
    return sum / reps * 1e3f;
}

float run_rocprim(size_t n, unsigned bits, const uint32_t *src, uint32_t *k0, uint32_t *k1, uint32_t *v0, uint32_t *v1, std::vector<uint32_t> &out_k,
                  std::vector<uint32_t> &out_v)
{
    size_t bytes = 0;
    CHECK(rocprim::radix_sort_pairs<RadixCfg32>(nullptr, bytes, k0, k1, v0, v1, n, 0, bits, 0));
    void *tmp;
    CHECK(hipMalloc(&tmp, bytes + 256));
    const uint32_t mask = bits >= 32 ? 0xffffffffu : (1u << bits) - 1u;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
        k_fill<<<(unsigned)((n + 255) / 256), 256>>>(src, k0, v0, (uint32_t)n, mask, nullptr, 0);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        CHECK(rocprim::radix_sort_pairs<RadixCfg32>(tmp, bytes, k0, k1, v0, v1, n, 0, bits, 0));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) sum += ms;
    }
    out_k.resize(n); out_v.resize(n);
    CHECK(hipMemcpy(out_k.data(), k1, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(out_v.data(), v1, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipFree(tmp));
    return sum / reps * 1e3f;
}

template <unsigned B, unsigned I>
void one(size_t n, unsigned bits, const uint32_t *src, uint32_t *k0, uint32_t *k1, uint32_t *v0, uint32_t *v1, uint32_t *scratch, const std::vector<uint32_t> &rk,
         const std::vector<uint32_t> &rv)
{
    std::vector<uint32_t> ok, ov;
    const float t = run_own<B, I>(n, bits, src, k0, k1, v0, v1, scratch, ok, ov);
    printf("  %4ux%-2u %6.1f us %s", B, I, t, (ok == rk && ov == rv) ? "same" : "DIFFERENT");
}

int main()
{
    for (size_t n : {(size_t)307200, (size_t)1000000}) {
        std::vector<uint32_t> hk(n);
        std::mt19937 rng(5);
        for (auto &k : hk) k = rng();
        uint32_t *src, *k0, *k1, *v0, *v1, *scratch;
        CHECK(hipMalloc(&src, n * 4)); CHECK(hipMalloc(&k0, n * 4)); CHECK(hipMalloc(&k1, n * 4)); CHECK(hipMalloc(&v0, n * 4)); CHECK(hipMalloc(&v1, n * 4));
        CHECK(hipMalloc(&scratch, (size_t)radix32_plan(n, 0, 32, 256 * 4).words * 4 + 256));
        CHECK(hipMemcpy(src, hk.data(), n * 4, hipMemcpyHostToDevice));
        for (unsigned bits : {24u, 32u}) {
            std::vector<uint32_t> rk, rv;
            const float t_r = run_rocprim(n, bits, src, k0, k1, v0, v1, rk, rv);
            printf("n %7zu bits %2u: rocprim %6.1f us |", n, bits, t_r);
            one<1024, 4>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<1024, 2>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<1024, 3>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<1024, 6>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<1024, 8>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<512, 4>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<512, 6>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<512, 8>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<512, 12>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            one<256, 16>(n, bits, src, k0, k1, v0, v1, scratch, rk, rv);
            printf("\n");
        }
        CHECK(hipFree(src)); CHECK(hipFree(k0)); CHECK(hipFree(k1)); CHECK(hipFree(v0)); CHECK(hipFree(v1)); CHECK(hipFree(scratch));
    }
    return 0;
}
