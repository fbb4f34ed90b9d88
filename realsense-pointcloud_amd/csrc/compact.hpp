// compact.hpp — "flag, scan, scatter" of a sorted array in ONE launch.
//
// The index build ends its radix sort by flagging the sorted records (kept / first of its cell), scanning the flags and
// scattering the records by the scan -- as separate launches that was a flag kernel, rocPRIM's scan (an init kernel and
// the scan proper) and a scatter kernel: four launches, 57 us of a 10^6-point index build and as many gaps (the analogue
// of the kd-tree PCL builds in setInputTarget, incremental_icp.hpp:58).  Here a workgroup flags its 4 096 records, scans
// them in registers and LDS, learns what lies in front of it through a decoupled look-back over one 64-bit word per
// workgroup (the two running counts and a status, so a word is complete the moment it is visible: no fence), and
// scatters.  Workgroups number themselves by a ticket, so a workgroup only ever waits for workgroups that are already
// running.  The words and the ticket must be zero when the kernel starts: the kernel in front of the sort clears them on
// its way with the sort's own state (radix32.hpp).  The block scan and the look-back are here; the kernels that use them are
// k_dense_compact (icp_dense.hpp: the sort-based index build) and k_edge_compact (edges.hip: the edge points' records).  (The same form for the source load's tail was measured and is level with rocPRIM's
// scan there: profiles/r04_experiments/README.md.)
#pragma once

#include <cstdint>

namespace rsreg {

constexpr unsigned kCompactBlock = 1024, kCompactItems = 4;
constexpr unsigned long long kCsPartial = 1ull << 62, kCsInclusive = 2ull << 62, kCsValue = (1ull << 62) - 1ull;

// words of scratch (uint32) behind `at` (rounded up to an 8-byte boundary): the look-back words, then the ticket
struct CompactPlan {
    uint32_t blocks = 0, off_state = 0, off_ticket = 0, end = 0;
};

inline CompactPlan compact_plan(size_t n, uint32_t at)
{
    CompactPlan p;
    p.blocks = (uint32_t)((n + kCompactBlock * kCompactItems - 1) / (kCompactBlock * kCompactItems));
    p.off_state = (at + 1u) & ~1u;
    p.off_ticket = p.off_state + 2u * p.blocks;
    p.end = p.off_ticket + 2u;
    return p;
}

// the two counts of a record travel as one 64-bit sum: low word | high word << 31 inside a look-back word (both below
// 2^31: n < 2^31), low | high << 32 everywhere else
__device__ __forceinline__ unsigned long long cs_pack(unsigned long long v) { return (v & 0x7fffffffull) | ((v >> 32) << 31); }
__device__ __forceinline__ unsigned long long cs_unpack(unsigned long long w) { return (w & 0x7fffffffull) | (((w & kCsValue) >> 31) << 32); }

// exclusive scan of v over the workgroup (in thread order); total = the workgroup's sum
__device__ __forceinline__ unsigned long long compact_block_scan(unsigned long long v, unsigned long long *s_wave, unsigned long long &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = (unsigned long long)__shfl_up((long long)inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned long long before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < (int)(kCompactBlock / 64); ++w) {
        const unsigned long long t = s_wave[w];
        if (w < wave) before += t;
        tot += t;
    }
    total = tot;
    return before + inc - v;
}

// what lies in front of workgroup `bid` (the sum of the totals of workgroups 0 .. bid - 1); called by all threads, ends in
// a barrier.  Wave 0 publishes this workgroup's total at once (so nobody behind it waits longer than that), then reads
// its predecessors' words 64 at a time, nearest first, up to the first one that already holds an inclusive sum.
__device__ __forceinline__ unsigned long long compact_lookback(unsigned long long *state, uint32_t bid, unsigned long long total,
                                                               unsigned long long *s_excl)
{
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        unsigned long long excl = 0;
        if (lane == 0)
            __hip_atomic_store(&state[bid], (bid == 0 ? kCsInclusive : kCsPartial) | cs_pack(total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bid > 0) {
            int look = (int)bid - 1;
            for (;;) {
                const int idx = look - lane;
                unsigned long long w;
                for (;;) {
                    w = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kCsInclusive;
                    if (!__any((w >> 62) == 0ull)) break;   // (a predecessor that has not published yet: it is running, by its ticket)
                    __builtin_amdgcn_s_sleep(1);
                }
                const unsigned long long incl = __ballot((w >> 62) == 2ull);
                const int first = incl ? __ffsll((long long)incl) - 1 : 64;
                unsigned long long c = lane <= first ? cs_unpack(w) : 0ull;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) c += (unsigned long long)__shfl_down((long long)c, off);
                excl += c;   // (lane 0 holds the sum)
                if (incl) break;
                look -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&state[bid], kCsInclusive | cs_pack(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) *s_excl = excl;
    }
    __syncthreads();
    return *s_excl;
}

}  // namespace rsreg
