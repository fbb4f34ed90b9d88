// small_sort.hpp — a stable sort of at most 65 536 (key, value) pairs in TWO launches.  AN EXPERIMENT, not part of the library:
// same permutation as rocprim::radix_sort_pairs, but 64 us against rocPRIM's 41 us for 5 x 10^4 pairs as written
// (small_sort.hip, small_sort_parts.hip: the 4096-item block sort alone takes 25 us, the placement 30-40 us).
//
// The frame loops of the edge schemes sort five clouds of 30-50 k points per frame (the source of each alignment by its
// Morton key, the voxel filter's slots and runs, the new points of the grown target: icp_edge_based_registration.hpp:75-120
// through icp.hip / voxel.hip / ndt.hip).  Below 65 536 items rocPRIM's radix_sort_pairs is a block sort followed by
// log2(n / 1024) merge passes: seven launches and 46 us for 5 x 10^4 pairs (profiles/r03_sort_configs.txt), every one of
// them a dependent launch the GPU idles before.  Here:
//   k_small_sort_runs   sorts runs of 4096 items in LDS (rocprim::block_radix_sort: stable, the bits asked for only);
//   k_small_sort_place  gives every item its final place: its position in its run + the number of items of every EARLIER
//                       run that are <= it + the number of items of every LATER run that are < it (one branch-free binary
//                       search per other run, <= 15 of them), i.e. the stable merge of all runs at once.
// Same permutation as rocprim::radix_sort_pairs (both are stable on the same bits).
#pragma once

#include <hip/hip_runtime.h>
#include <rocprim/block/block_load.hpp>
#include <rocprim/block/block_radix_sort.hpp>

#include <cstddef>
#include <cstdint>

namespace rsreg {

constexpr uint32_t kSmallSortMax = 65536;   // items at most
constexpr uint32_t kSmallSortBlock = 1024, kSmallSortItems = 4, kSmallSortRun = kSmallSortBlock * kSmallSortItems;
constexpr uint32_t kSmallSortRunBits = 12;  // log2(kSmallSortRun)
constexpr uint32_t kSmallSortMaxRuns = kSmallSortMax / kSmallSortRun;

template <class K>
__device__ __forceinline__ K small_sort_bits(K k, unsigned begin_bit, unsigned end_bit)
{
    const unsigned w = end_bit - begin_bit;
    k >>= begin_bit;
    return w >= sizeof(K) * 8 ? k : (k & ((K(1) << w) - K(1)));
}

// Runs of kSmallSortRun consecutive items, each sorted by bits [begin_bit, end_bit) of its keys (stable).
template <class K, class V>
__global__ __launch_bounds__(kSmallSortBlock) void k_small_sort_runs(const K *keys, const V *vals, uint32_t n, unsigned begin_bit, unsigned end_bit,
                                                                       K *run_keys, V *run_vals)
{
    using Sort = rocprim::block_radix_sort<K, kSmallSortBlock, kSmallSortItems, V>;
    __shared__ typename Sort::storage_type storage;
    const uint32_t base = blockIdx.x * kSmallSortRun + threadIdx.x * kSmallSortItems;
    K k[kSmallSortItems];
    V v[kSmallSortItems];
#pragma unroll
    for (uint32_t j = 0; j < kSmallSortItems; ++j) {
        const bool in = base + j < n;
        // beyond the end: all ones in the sorted bits, and behind every real item of the run (the sort is stable)
        k[j] = in ? keys[base + j] : ~K(0);
        v[j] = in ? vals[base + j] : V();
    }
    Sort().sort(k, v, storage, begin_bit, end_bit);
#pragma unroll
    for (uint32_t j = 0; j < kSmallSortItems; ++j)
        if (base + j < n) {
            run_keys[base + j] = k[j];
            run_vals[base + j] = v[j];
        }
}

// Every item to its place in the stable merge of all runs.
template <class K, class V>
__global__ __launch_bounds__(256) void k_small_sort_place(const K *run_keys, const V *run_vals, uint32_t n, unsigned begin_bit, unsigned end_bit,
                                                          K *out_keys, V *out_vals)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint32_t run = e >> kSmallSortRunBits, n_runs = (n + kSmallSortRun - 1) >> kSmallSortRunBits;
    const K key = run_keys[e], bits = small_sort_bits(key, begin_bit, end_bit);
    uint32_t place = e & (kSmallSortRun - 1);
    for (uint32_t r0 = 0; r0 < n_runs; r0 += 4) {   // four searches at a time: their loads are independent
        uint32_t pos[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t step = kSmallSortRun; step; step >>= 1) {   // (first step: is the whole run in front?)
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t r = r0 + q;
                if (r >= n_runs || r == run) continue;
                const uint32_t base = r << kSmallSortRunBits, len = min(kSmallSortRun, n - base), at = pos[q] + step;
                if (at > len) continue;
                const K other = small_sort_bits(run_keys[base + at - 1], begin_bit, end_bit);
                if (r < run ? other <= bits : other < bits) pos[q] = at;
            }
        }
        place += pos[0] + pos[1] + pos[2] + pos[3];
    }
    out_keys[place] = key;
    out_vals[place] = run_vals[e];
}

inline size_t small_sort_tmp_bytes(size_t n, size_t key_bytes, size_t val_bytes) { return ((n * key_bytes + 255) & ~size_t(255)) + n * val_bytes + 256; }

// keys_out / vals_out <- the pairs sorted by bits [begin_bit, end_bit) of the keys, stable.  n <= kSmallSortMax;
// tmp: small_sort_tmp_bytes(n, sizeof(K), sizeof(V)) bytes.  The inputs are left as they were.
template <class K, class V>
hipError_t small_sort_pairs(void *tmp, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, uint32_t n, unsigned begin_bit, unsigned end_bit,
                            hipStream_t st)
{
    if (!n) return hipSuccess;
    if (n > kSmallSortMax || end_bit <= begin_bit || end_bit > sizeof(K) * 8) return hipErrorInvalidValue;
    K *run_keys = static_cast<K *>(tmp);
    V *run_vals = reinterpret_cast<V *>(static_cast<char *>(tmp) + ((size_t(n) * sizeof(K) + 255) & ~size_t(255)));
    const uint32_t n_runs = (n + kSmallSortRun - 1) / kSmallSortRun;
    if (n_runs == 1) {   // one run: it is the result
        hipLaunchKernelGGL((k_small_sort_runs<K, V>), dim3(1), dim3(kSmallSortBlock), 0, st, keys_in, vals_in, n, begin_bit, end_bit, keys_out, vals_out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((k_small_sort_runs<K, V>), dim3(n_runs), dim3(kSmallSortBlock), 0, st, keys_in, vals_in, n, begin_bit, end_bit, run_keys, run_vals);
    hipLaunchKernelGGL((k_small_sort_place<K, V>), dim3((n + 255) / 256), dim3(256), 0, st, run_keys, run_vals, n, begin_bit, end_bit, keys_out, vals_out);
    return hipGetLastError();
}

}  // namespace rsreg
