// small_sort.hip — tools/microbench/small_sort.hpp against rocprim::radix_sort_pairs below 65 536 items: same permutation, time per sort.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/small_sort.hip -o tools/_build/small_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "small_sort.hpp"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class K>
void run(const char *what, uint32_t n, unsigned begin_bit, unsigned end_bit, unsigned key_mask_bits)
{
    std::vector<K> hk(n);
    std::vector<uint32_t> hv(n);
    std::mt19937_64 rng(n * 31 + end_bit);
    for (uint32_t i = 0; i < n; ++i) {
        hk[i] = (K)rng();
        if (key_mask_bits < sizeof(K) * 8) hk[i] &= ((K(1) << key_mask_bits) - 1);   // (few distinct keys: ties everywhere)
        hv[i] = i;
    }
    K *k0, *k1, *k2;
    uint32_t *v0, *v1, *v2;
    CHECK(hipMalloc(&k0, n * sizeof(K) + 16)); CHECK(hipMalloc(&k1, n * sizeof(K) + 16)); CHECK(hipMalloc(&k2, n * sizeof(K) + 16));
    CHECK(hipMalloc(&v0, n * 4 + 16)); CHECK(hipMalloc(&v1, n * 4 + 16)); CHECK(hipMalloc(&v2, n * 4 + 16));
    CHECK(hipMemcpy(k0, hk.data(), n * sizeof(K), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
    size_t bytes = 0;
    CHECK(rocprim::radix_sort_pairs(nullptr, bytes, k0, k1, v0, v1, n, begin_bit, end_bit, 0));
    void *tmp, *tmp2;
    CHECK(hipMalloc(&tmp, bytes + 256));
    CHECK(hipMalloc(&tmp2, rsreg::small_sort_tmp_bytes(n, sizeof(K), 4)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float t[2] = {0, 0};
    const int reps = 20;
    for (int which = 0; which < 2; ++which)
        for (int r = 0; r < reps + 2; ++r) {
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            if (which == 0) CHECK(rocprim::radix_sort_pairs(tmp, bytes, k0, k1, v0, v1, n, begin_bit, end_bit, 0));
            else CHECK(rsreg::small_sort_pairs(tmp2, k0, k2, v0, v2, n, begin_bit, end_bit, 0));
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) t[which] += ms;
        }
    std::vector<K> a(n), b(n);
    std::vector<uint32_t> va(n), vb(n);
    CHECK(hipMemcpy(a.data(), k1, n * sizeof(K), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), k2, n * sizeof(K), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(va.data(), v1, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(vb.data(), v2, n * 4, hipMemcpyDeviceToHost));
    const bool same = a == b && va == vb;
    printf("%-34s n %6u bits [%u, %u): rocPRIM %.1f us, two launches %.1f us, %s\n", what, n, begin_bit, end_bit, t[0] / reps * 1e3, t[1] / reps * 1e3,
           same ? "same permutation" : "DIFFERENT");
    CHECK(hipFree(k0)); CHECK(hipFree(k1)); CHECK(hipFree(k2)); CHECK(hipFree(v0)); CHECK(hipFree(v1)); CHECK(hipFree(v2)); CHECK(hipFree(tmp)); CHECK(hipFree(tmp2));
}

int main()
{
    for (uint32_t n : {1u, 63u, 4096u, 4097u, 12000u, 36049u, 47851u, 50000u, 65535u, 65536u}) {
        run<uint32_t>("u32 keys, all bits", n, 0, 32, 32);
        run<uint32_t>("u32 keys, 9 distinct bits", n, 0, 32, 9);
        run<uint32_t>("u32 keys, sorted by bits 0-10", n, 0, 10, 32);
        run<uint32_t>("u32 keys, sorted by bits 5-21", n, 5, 21, 32);
        run<uint64_t>("u64 keys, all bits", n, 0, 64, 64);
        run<uint64_t>("u64 keys, 40 bits", n, 0, 40, 64);
    }
    return 0;
}
