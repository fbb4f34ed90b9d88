// sort_small_n.hip — rocPRIM radix_sort_pairs of 3-6 x 10^4 (uint32, uint32) pairs: the merge-sort path it takes below its
// merge-sort limit (a block sort and log2(n / 1024) merge launches, whatever the key bits) against onesweep forced by a small
// limit (one histogram launch + one launch per 8-bit digit of the bits asked for).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/sort_small_n.hip -o tools/_build/sort_small_n
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class Cfg>
void run(const char *name, size_t n, unsigned end_bit, uint32_t *k0, uint32_t *k1, uint32_t *v0, uint32_t *v1)
{
    size_t bytes = 0;
    CHECK(rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, k0, k1, v0, v1, n, 0, end_bit, 0));
    void *tmp;
    CHECK(hipMalloc(&tmp, bytes + 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        CHECK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, k0, k1, v0, v1, n, 0, end_bit, 0));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) sum += ms;
    }
    printf("%-44s n %6zu bits %2u: %.1f us\n", name, n, end_bit, sum / reps * 1e3);
    CHECK(hipFree(tmp));
}

using namespace rocprim;
using Merge = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<1024, 4>, 8, block_radix_rank_algorithm::match>, 65536>;
template <unsigned B, unsigned IPT>
using One = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<B, IPT>, 8, block_radix_rank_algorithm::match>, 2048>;

int main()
{
    for (size_t n : {(size_t)12000, (size_t)36049, (size_t)47851, (size_t)65536}) {
        std::vector<uint32_t> hk(n);
        std::mt19937 rng(5);
        for (auto &k : hk) k = rng();
        uint32_t *k0, *k1, *v0, *v1;
        CHECK(hipMalloc(&k0, n * 4)); CHECK(hipMalloc(&k1, n * 4)); CHECK(hipMalloc(&v0, n * 4)); CHECK(hipMalloc(&v1, n * 4));
        CHECK(hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice));
        for (unsigned bits : {10u, 17u, 24u, 32u}) {
            run<Merge>("merge-sort path (limit 65536)", n, bits, k0, k1, v0, v1);
            run<One<1024, 4>>("onesweep forced, 1024 x 4", n, bits, k0, k1, v0, v1);
            run<One<512, 4>>("onesweep forced, 512 x 4", n, bits, k0, k1, v0, v1);
            run<One<256, 4>>("onesweep forced, 256 x 4", n, bits, k0, k1, v0, v1);
            run<One<256, 8>>("onesweep forced, 256 x 8", n, bits, k0, k1, v0, v1);
        }
        CHECK(hipFree(k0)); CHECK(hipFree(k1)); CHECK(hipFree(v0)); CHECK(hipFree(v1));
    }
    return 0;
}
