// radix32.hpp — the names the callers of the library's radix sort use (the sort itself: osort.hpp).
//
// Rounds 3-4 drove rocPRIM's onesweep device functions from here (its `detail` namespace: an internal interface); round 5
// replaced them by this library's own kernels.  What is left is the vocabulary of the call sites: a plan (scratch words to
// be zero when the sort starts, cleared by the kernel that writes the keys on its way), whether the own sort is the one to
// use, and the sort.
#pragma once

#include <cstdint>
#include <cstdlib>

#include "osort.hpp"
#include "tunables.hpp"

namespace rsreg {

struct Radix32Plan : OsortPlan {
    uint32_t places = 0;   // digit passes (an even number ends in the buffer pair it started from, unless one workgroup sorts it all)
    bool ends_in_first = true;
};

inline Radix32Plan radix32_plan(size_t n, unsigned begin_bit, unsigned end_bit)
{
    Radix32Plan p;
    static_cast<OsortPlan &>(p) = osort_plan(n, begin_bit, end_bit);
    p.places = p.passes;
    p.ends_in_first = osort_ends_in_first(p, n);
    return p;
}

// Every sort of 32-bit keys is the library's own (RSREG_ROCPRIM_SORT=1: rocPRIM's public radix_sort_pairs instead, for A/B runs).
inline bool radix32_pays(size_t n, unsigned bits)
{
    return !tunables().rocprim_sort && bits > 0 && bits <= 32 && n < (1ull << 30);
}

__device__ __forceinline__ void radix32_clear(uint32_t *scratch, uint32_t words, uint32_t t, uint32_t threads) { osort_clear(scratch, words, t, threads); }

inline hipError_t radix32_sort_pairs(const Radix32Plan &p, uint32_t *scratch, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n,
                                     unsigned begin_bit, unsigned end_bit, hipStream_t st, bool *in_first, const uint32_t *hist_ready = nullptr)
{
    return osort_pairs(p, scratch, keys_a, keys_b, vals_a, vals_b, n, begin_bit, end_bit, st, in_first, hist_ready);
}

}  // namespace rsreg
