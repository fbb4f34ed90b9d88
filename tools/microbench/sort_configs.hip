// sort_configs.hip — rocPRIM radix_sort_pairs of 10^6 (uint32 key, uint32 value) pairs under different onesweep
// configurations: bits per pass, items per block, rank algorithm (dev probe for the index build).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/sort_configs.hip -o tools/_build/sort_configs
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class Cfg>
void run(const char *name, size_t n, unsigned end_bit, uint32_t *k0, uint32_t *k1, uint32_t *v0, uint32_t *v1, const std::vector<uint32_t> &hk)
{
    size_t bytes = 0;
    CHECK(rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, k0, k1, v0, v1, n, 0, end_bit, 0));
    void *tmp;
    CHECK(hipMalloc(&tmp, bytes + 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; ++r) {
        CHECK(hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        CHECK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, k0, k1, v0, v1, n, 0, end_bit, 0));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { best = std::min(best, ms); sum += ms; }
    }
    std::vector<uint32_t> out(n);
    CHECK(hipMemcpy(out.data(), k1, n * 4, hipMemcpyDeviceToHost));
    bool ok = std::is_sorted(out.begin(), out.end());
    printf("%-58s n %zu bits %u: mean %.1f us best %.1f us %s\n", name, n, end_bit, sum / reps * 1e3, best * 1e3, ok ? "sorted" : "NOT SORTED");
    CHECK(hipFree(tmp));
}

using namespace rocprim;
template <unsigned B, unsigned IPT, unsigned Bits, block_radix_rank_algorithm A = block_radix_rank_algorithm::default_algorithm>
using Cfg = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<B, IPT>, Bits, A>, 65536>;

int main()
{
    for (size_t n : {(size_t)1000000, (size_t)307200, (size_t)50000}) {
        std::vector<uint32_t> hk(n);
        std::mt19937 rng(5);
        for (auto &k : hk) k = rng();
        uint32_t *k0, *k1, *v0, *v1;
        CHECK(hipMalloc(&k0, n * 4)); CHECK(hipMalloc(&k1, n * 4)); CHECK(hipMalloc(&v0, n * 4)); CHECK(hipMalloc(&v1, n * 4));
        run<radix_sort_config<default_config, default_config, default_config, 65536>>("rocPRIM default onesweep", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<512, 8, 8, block_radix_rank_algorithm::match>>("512 x 8, 8 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<512, 4, 8, block_radix_rank_algorithm::match>>("512 x 4, 8 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<512, 6, 8, block_radix_rank_algorithm::match>>("512 x 6, 8 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<512, 12, 8, block_radix_rank_algorithm::match>>("512 x 12, 8 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<1024, 4, 8, block_radix_rank_algorithm::match>>("1024 x 4, 8 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<1024, 8, 8, block_radix_rank_algorithm::match>>("1024 x 8, 8 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<256, 8, 8, block_radix_rank_algorithm::match>>("256 x 8, 8 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<512, 8, 7, block_radix_rank_algorithm::match>>("512 x 8, 7 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<512, 8, 9, block_radix_rank_algorithm::match>>("512 x 8, 9 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<1024, 8, 9, block_radix_rank_algorithm::match>>("1024 x 8, 9 bits, match", n, 32, k0, k1, v0, v1, hk);
        run<Cfg<512, 8, 8, block_radix_rank_algorithm::match>>("512 x 8, 8 bits, match, 29 key bits", n, 29, k0, k1, v0, v1, hk);
        CHECK(hipFree(k0)); CHECK(hipFree(k1)); CHECK(hipFree(v0)); CHECK(hipFree(v1));
    }
    return 0;
}
