// radix32.hpp — the onesweep radix sort of (uint32 key, uint32 value) pairs, driven without a memset.
//
// The index build (the analogue of the kd-tree PCL builds on every setInputTarget, incremental_icp.hpp:58) and the source load
// each sort ~10^6 pairs by a 32-bit key.  rocPRIM's driver queues, in front of every one of the four digit passes, two
// hipMemsetAsync (the pass's look-back states and its block counter) and one more for the histograms: 9 fills per sort, 18
// of the 20 `fillBufferAligned` launches of a bench step (72 us of GPU time, and as many gaps in a chain that is bound by
// its launch count).  Here the device functions of rocPRIM's onesweep (histograms, histogram scan, one digit pass) are
// wrapped in kernels of this library and given SEPARATE state for every pass, all of it in one scratch block that the
// kernel in front of the sort (the one that writes the keys) clears on its way: a sort is 2 + passes launches and nothing
// else.  The passes ping-pong between the caller's two buffer pairs (no third copy): the result lies in the second
// pair after an odd number of passes, in the first after an even number.  Same kernels, same configuration (1024 x 4
// items, 8 bits, match-based rank: sort_cfg.hpp), same stable order as rocprim::radix_sort_pairs<RadixCfg32>.
#pragma once

#include <cstdint>
#include <cstdlib>

#include <rocprim/rocprim_version.hpp>

#include "sort_cfg.hpp"

// The three kernels below wrap device functions of rocPRIM's onesweep that live in its `detail` namespace: an internal
// interface, checked here against the one release it was written for.  On any other rocPRIM every sort goes through the
// public rocprim::radix_sort_pairs (radix32_pays() is false, the callers' rocPRIM branch runs): slower by nine memsets per
// sort, never wrong.
#if ROCPRIM_VERSION == 400200
#define RSREG_RADIX32_DETAIL 1
#else
#define RSREG_RADIX32_DETAIL 0
#endif

namespace rsreg {

constexpr unsigned kR32SortBlock = 1024, kR32SortItems = 4, kR32Bits = 8, kR32HistBlock = 256, kR32HistItems = 12;
constexpr unsigned kR32MaxPlaces = 4;

#if RSREG_RADIX32_DETAIL
using R32Bid = rocprim::detail::block_id_wrapper<unsigned int, true>;
using R32State = rocprim::detail::onesweep_lookback_state;
#endif

struct Radix32Plan {
    uint32_t places = 0, blocks = 0, hist_blocks = 0;
    uint32_t words = 0;   // of the scratch block, all to be zero when the sort starts
    // word offsets into the scratch block
    uint32_t off_digits = 0, off_tmp = 0, off_bid = 0, off_states = 0;
};

// (items_per_block: of the digit passes -- 1024 x 4 unless radix32_sort_pairs is instantiated otherwise)
inline Radix32Plan radix32_plan(size_t n, unsigned begin_bit, unsigned end_bit, unsigned items_per_block = kR32SortBlock * kR32SortItems)
{
    Radix32Plan p;
    p.places = (end_bit - begin_bit + kR32Bits - 1) / kR32Bits;
    p.blocks = (uint32_t)((n + items_per_block - 1) / items_per_block);
    p.hist_blocks = (uint32_t)((n + kR32HistBlock * kR32HistItems - 1) / (kR32HistBlock * kR32HistItems));
    p.off_digits = 0;
    p.off_tmp = p.places << kR32Bits;
    p.off_bid = p.off_tmp + (1u << kR32Bits);
    p.off_states = p.off_bid + 16;
    p.words = p.off_states + p.places * (p.blocks << kR32Bits);
    return p;
}

// When this driver is the faster one (profiles/r04_sort_driver.txt): from 65 536 pairs on whatever the bits (below that
// rocPRIM merge-sorts: 27-43 us), and from 8 192 pairs on for keys of at most 16 bits (two digit passes: 24 us at 36 k pairs
// where the merge path takes 42).  RSREG_ROCPRIM_SORT=1 hands every sort back to rocPRIM's driver.
inline bool radix32_pays(size_t n, unsigned bits)
{
    static const bool off = std::getenv("RSREG_ROCPRIM_SORT") && std::getenv("RSREG_ROCPRIM_SORT")[0] == '1';
    if (!RSREG_RADIX32_DETAIL || off || bits == 0 || bits > 32) return false;
    if (n >= (1ull << 30)) return false;   // (radix32_sort_pairs counts in 32 bits with room to spare: larger sorts are rocPRIM's)
    return n >= 65536 || (n >= 8192 && bits <= 16);
}

// what the kernel in front of a sort does on its way: thread `t` of `threads` clears its share of the scratch block
__device__ __forceinline__ void radix32_clear(uint32_t *scratch, uint32_t words, uint32_t t, uint32_t threads)
{
    for (uint32_t w = t; w < words; w += threads) scratch[w] = 0u;
}

#if RSREG_RADIX32_DETAIL
// (templates, so that the several translation units of the library that include this header share one definition)
template <int kDummy = 0>
__global__ __launch_bounds__(kR32HistBlock) void k_r32_histograms(const uint32_t *keys, uint32_t *digits, uint32_t n, uint32_t full_blocks,
                                                                  uint32_t begin_bit, uint32_t end_bit)
{
    rocprim::detail::onesweep_histograms<kR32HistBlock, kR32HistItems, kR32Bits, false>(keys, digits, n, full_blocks, rocprim::identity_decomposer{},
                                                                                         begin_bit, end_bit);
}

template <int kDummy = 0>
__global__ __launch_bounds__(kR32HistBlock) void k_r32_scan_histograms(uint32_t *digits)
{
    rocprim::detail::onesweep_scan_histograms<kR32HistBlock, kR32Bits>(digits);
}

template <unsigned kBlockT, unsigned kItemsT>
__global__ __launch_bounds__(kBlockT) void k_r32_pass(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                                                      uint32_t n, uint32_t *digits_in, uint32_t *digits_out, R32State *states, uint32_t bit,
                                                      uint32_t bits, uint32_t full_blocks, R32Bid bid)
{
    rocprim::detail::onesweep_iteration<kBlockT, kItemsT, kR32Bits, false, rocprim::block_radix_rank_algorithm::match>(
        keys_in, keys_out, vals_in, vals_out, n, digits_in, digits_out, states, rocprim::identity_decomposer{}, bit, bits, full_blocks, bid);
}

// Sorts n pairs by bits [begin_bit, end_bit) of the key.  `scratch` (plan.words words) must be all zero when the first
// kernel starts and is dirty afterwards.  Returns through *in_first whether the sorted pairs lie in (keys_a, vals_a)
// (true) or in (keys_b, vals_b); the other pair is overwritten too.
template <unsigned kBlockT = kR32SortBlock, unsigned kItemsT = kR32SortItems>
inline hipError_t radix32_sort_pairs(const Radix32Plan &p, uint32_t *scratch, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b,
                                     size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st, bool *in_first)
{
    *in_first = true;
    if (n == 0 || p.places == 0) return hipSuccess;
    // (every check before the first launch: an error return leaves nothing queued on a dirty scratch block)
    if (p.places > kR32MaxPlaces || n >= (1ull << 30)) return hipErrorInvalidValue;
    const uint32_t per = kBlockT * kItemsT;
    if (p.blocks != (uint32_t)((n + per - 1) / per)) return hipErrorInvalidValue;   // (the plan was made for another tiling)
    uint32_t *digits = scratch + p.off_digits;
    {
        const uint32_t hper = kR32HistBlock * kR32HistItems;
        const uint32_t hfull = (uint32_t)(n % hper == 0 ? p.hist_blocks : p.hist_blocks - 1);
        k_r32_histograms<0><<<p.hist_blocks, kR32HistBlock, 0, st>>>(keys_a, digits, (uint32_t)n, hfull, begin_bit, end_bit);
        k_r32_scan_histograms<0><<<p.places, kR32HistBlock, 0, st>>>(digits);
    }
    const uint32_t full = (uint32_t)(n % per == 0 ? p.blocks : p.blocks - 1);
    bool from_a = true;
    unsigned bit = begin_bit;
    for (uint32_t place = 0; place < p.places; ++place, bit += kR32Bits) {
        const uint32_t bits = std::min(kR32Bits, end_bit - bit);
        R32Bid bid = R32Bid::create(scratch + p.off_bid + place);
        auto *states = reinterpret_cast<R32State *>(scratch + p.off_states + (size_t)place * (p.blocks << kR32Bits));
        k_r32_pass<kBlockT, kItemsT><<<p.blocks, kBlockT, 0, st>>>(from_a ? keys_a : keys_b, from_a ? keys_b : keys_a, from_a ? vals_a : vals_b,
                                                        from_a ? vals_b : vals_a, (uint32_t)n, digits + (place << kR32Bits), scratch + p.off_tmp, states,
                                                        bit, bits, full, bid);
        from_a = !from_a;
    }
    *in_first = from_a;
    return hipGetLastError();
}
#else
template <unsigned kBlockT = kR32SortBlock, unsigned kItemsT = kR32SortItems>
inline hipError_t radix32_sort_pairs(const Radix32Plan &, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t *, size_t, unsigned, unsigned,
                                     hipStream_t, bool *in_first)
{
    *in_first = true;
    return hipErrorNotSupported;   // (never reached: radix32_pays() is false on this rocPRIM)
}
#endif

}  // namespace rsreg
