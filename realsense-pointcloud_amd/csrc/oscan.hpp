// oscan.hpp — this library's own device-wide prefix sums (uint32 / uint64 values, exclusive or inclusive, in place allowed).
//
// What it scans: the keep flags of the source load (-> positions of the distinct points), the run flags of the voxel filter,
// the leaf-start flags of the NDT grid, the weights of the trimmed rejector, and -- on the fallback builds of the target
// index only -- the flag words and the cell-count table.  10^4 .. 10^6 values as a rule, up to 2^28 table entries.
//
// Reduce, then scan: k_oscan_sums (a workgroup adds its 1 024 values), k_oscan_apply (a workgroup adds up the sums in front of it --
// at most 4 096 of them; beyond that k_oscan_top, ONE workgroup, turns the sums into what lies in front of each first -- and
// scans its 1 024 values from there).  Two launches (three beyond 4 M values), the values
// read twice, NO state that has to be zero beforehand and no workgroup that waits for another one -- which is what this
// library had rocPRIM's scan for until round 6, at the price of its ~700 kernel instantiations per translation unit in the
// code object (every architecture's tuning of every algorithm: 100 ms of code-object loading in the first registration() of
// a process, profiles/r06_cold_run.txt).  The sums are integer: any order of addition gives the same bits.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace rsreg {

constexpr unsigned kScBlock = 256, kScItems = 4, kScTile = kScBlock * kScItems, kScTopBlock = 1024, kScSelfTop = 4096;   // (4 consecutive values a thread: one 16-byte load of uint32s)

inline uint32_t oscan_blocks(size_t n) { return (uint32_t)((n + kScTile - 1) / kScTile); }
// bytes of scratch a scan of n values of type T needs (the workgroups' sums)
template <typename T> inline size_t oscan_scratch_bytes(size_t n) { return ((size_t)oscan_blocks(n) + 1) * sizeof(T); }

template <typename T> __device__ __forceinline__ T oscan_wave_incl(T v, uint32_t lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(v, off);
        if ((int)lane >= off) v += o;
    }
    return v;
}

// exclusive prefix of `v` over the workgroup's threads (kThreads, a multiple of 64) + the workgroup's total in *total
template <typename T, unsigned kThreads> __device__ __forceinline__ T oscan_block_excl(T v, T *s_part /* kThreads / 64 */, T *total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const T incl = oscan_wave_incl(v, lane);
    if (lane == 63u) s_part[wave] = incl;
    __syncthreads();
    T before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < kThreads / 64; ++w) {
        const T p = s_part[w];
        if (w < wave) before += p;
        all += p;
    }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

template <typename T> __global__ __launch_bounds__(kScBlock) void k_oscan_sums(const T *in, size_t n, T *part)
{
    __shared__ T s_part[kScBlock / 64];
    const size_t base = (size_t)blockIdx.x * kScTile;
    T v = 0;
    // thread t owns the kScItems CONSECUTIVE values from base + t * kScItems on (k_oscan_apply: the same ownership)
    const size_t at = base + (size_t)threadIdx.x * kScItems;
    if (at + kScItems <= n) {
#pragma unroll
        for (uint32_t j = 0; j < kScItems; ++j) v += in[at + j];
    } else {
#pragma unroll
        for (uint32_t j = 0; j < kScItems; ++j)
            if (at + j < n) v += in[at + j];
    }
    T total;
    (void)oscan_block_excl<T, kScBlock>(v, s_part, &total);
    if (threadIdx.x == 0) part[blockIdx.x] = total;
}

// part[b] <- part[0] + ... + part[b - 1] (one workgroup; `blocks` sums in chunks of 1 024 with a carry); part[blocks] <- the grand total
template <typename T> __global__ __launch_bounds__(kScTopBlock) void k_oscan_top(T *part, uint32_t blocks)
{
    __shared__ T s_part[kScTopBlock / 64];
    T carry = 0;
    for (uint32_t b0 = 0; b0 < blocks; b0 += kScTopBlock) {
        const uint32_t b = b0 + threadIdx.x;
        const T v = b < blocks ? part[b] : (T)0;
        T total;
        const T excl = oscan_block_excl<T, kScTopBlock>(v, s_part, &total);
        if (b < blocks) part[b] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) part[blocks] = carry;
}

// kTopDone: part[b] already holds what lies in front of workgroup b (k_oscan_top has run); otherwise part[] are the workgroups'
// own sums and every workgroup adds up the ones in front of it itself (at most kScSelfTop of them: one launch less)
template <typename T, bool kInclusive, bool kTopDone> __global__ __launch_bounds__(kScBlock) void k_oscan_apply(const T *in, T *out, size_t n, T init, const T *part)
{
    __shared__ T s_part[kScBlock / 64];
    T front = 0;
    if (kTopDone) {
        front = part[blockIdx.x];
    } else {
        T mine = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kScBlock) mine += part[b];
        (void)oscan_block_excl<T, kScBlock>(mine, s_part, &front);   // (front <- the sum over the workgroup's threads)
    }
    const size_t at = (size_t)blockIdx.x * kScTile + (size_t)threadIdx.x * kScItems;
    T x[kScItems], v = 0;
    if (at + kScItems <= n) {
#pragma unroll
        for (uint32_t j = 0; j < kScItems; ++j) x[j] = in[at + j];
    } else {
#pragma unroll
        for (uint32_t j = 0; j < kScItems; ++j) x[j] = at + j < n ? in[at + j] : (T)0;
    }
#pragma unroll
    for (uint32_t j = 0; j < kScItems; ++j) v += x[j];
    T total;
    T run = init + front + oscan_block_excl<T, kScBlock>(v, s_part, &total);
#pragma unroll
    for (uint32_t j = 0; j < kScItems; ++j) {
        if (kInclusive) run += x[j];
        if (at + j < n) out[at + j] = run;
        if (!kInclusive) run += x[j];
    }
}

// out[i] = init + in[0] + ... + in[i - 1] (exclusive) or ... + in[i] (inclusive); out may be in.  `scratch`: oscan_scratch_bytes<T>(n)
// bytes, contents irrelevant.  n == 0: nothing is launched.
template <typename T, bool kInclusive = false> inline hipError_t oscan(const T *in, T *out, size_t n, T init, void *scratch, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const uint32_t blocks = oscan_blocks(n);
    T *part = static_cast<T *>(scratch);
    k_oscan_sums<T><<<blocks, kScBlock, 0, st>>>(in, n, part);
    if (blocks <= kScSelfTop) {   // (10^6 values: 977 sums, four loads a thread)
        k_oscan_apply<T, kInclusive, false><<<blocks, kScBlock, 0, st>>>(in, out, n, init, part);
    } else {
        k_oscan_top<T><<<1, kScTopBlock, 0, st>>>(part, blocks);
        k_oscan_apply<T, kInclusive, true><<<blocks, kScBlock, 0, st>>>(in, out, n, init, part);
    }
    return hipGetLastError();
}

}  // namespace rsreg
