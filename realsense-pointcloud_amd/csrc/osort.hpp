// osort.hpp — this library's own radix sort of (key, uint32 value) pairs, keys of 32 or 64 bits: least significant digit
// first, 8 bits a pass, stable, one sweep over the data per pass (gfx950, wave64).  The only sort of the library since round 6
// (rounds 1-5 kept rocPRIM's radix_sort_pairs for 64-bit keys and as an A/B fallback: its ~700 kernel instantiations per
// translation unit were 100 ms of code-object loading in the first registration() of a process).
//
// What it sorts: the source's Morton keys (the spatial order the lanes of a search wave want), the voxel filter's hash slots
// and emission positions, the NDT leaf keys, the trimmed rejector's distances -- 10^4 .. 10^6 pairs, 10 .. 32 key bits
// (the target's index needs no sort any more: cellsort.hpp).  Launches per sort: one histogram kernel + one kernel per digit.
//   k_os_hist   every workgroup counts all the digits of its keys in LDS and adds the counts to the global histograms
//               (passes x 256 words);
//   k_os_pass   a workgroup takes 4 096 consecutive pairs (a wave 256 of them, 64 at a time, in order).  Ranks inside a wave
//               by ballots: the lanes holding my digit are the AND over the 8 digit bits of "ballot of that bit, or its
//               complement"; the first of them bumps the wave's counter of that digit in LDS, the others sit behind it in
//               lane order.  The waves' counters are scanned per digit, the workgroup's digit totals are published and what
//               the workgroups in front hold of every digit is collected by a decoupled look-back (one chain per digit,
//               256 threads at once; a workgroup numbers itself by a ticket, so it only waits for workgroups that run),
//               the pairs are put into their sorted order in LDS and written out digit segment by digit segment, so that
//               neighbouring threads write neighbouring addresses.
// State (histograms, look-back words, tickets) lives in one scratch block that must be ZERO when the histogram kernel
// starts: the kernel that writes the keys clears it on its way (osort_clear), so a sort queues no memset.
// The passes ping-pong between the caller's two buffer pairs.  At most one tile of pairs is sorted by ONE workgroup in one launch
// (k_os_small: the same passes, in LDS).  A tile is 4 096 pairs of 32-bit keys, 2 048 of 64-bit keys (the keys of a tile sit in LDS).
// The vocabulary of the call sites (rounds 3-5: csrc/radix32.hpp) lives at the end of this file.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace rsreg {

constexpr unsigned kOsBlock = 1024, kOsBits = 8, kOsDigits = 1u << kOsBits;
constexpr unsigned kOsHistBlock = 256, kOsHistItems = 16;
// per key type: pairs a thread holds (a tile = 1 024 threads x that), digit passes at most
template <typename K> struct OsKey;
template <> struct OsKey<uint32_t> { static constexpr unsigned items = 4, max_passes = 4; };
template <> struct OsKey<unsigned long long> { static constexpr unsigned items = 2, max_passes = 8; };
template <typename K> constexpr unsigned os_tile() { return kOsBlock * OsKey<K>::items; }
constexpr unsigned kOsTile = kOsBlock * 4;   // (of 32-bit keys: what the call sites of 32-bit sorts size their buffers by)

struct OsortPlan {
    uint32_t passes = 0, blocks = 0, hist_blocks = 0;
    uint32_t words = 0;   // of the scratch block, all to be zero when the sort starts
    uint32_t off_hist = 0, off_ticket = 0, off_state = 0;   // word offsets: passes x 256 counts | a ticket per pass | passes x blocks x 256 look-back words
};

template <typename K = uint32_t> inline OsortPlan osort_plan(size_t n, unsigned begin_bit, unsigned end_bit)
{
    OsortPlan p;
    p.passes = (end_bit - begin_bit + kOsBits - 1) / kOsBits;
    p.blocks = (uint32_t)((n + os_tile<K>() - 1) / os_tile<K>());
    p.hist_blocks = (uint32_t)((n + kOsHistBlock * kOsHistItems - 1) / (kOsHistBlock * kOsHistItems));
    p.off_hist = 0;
    p.off_ticket = p.passes * kOsDigits;
    p.off_state = p.off_ticket + 8;
    p.words = p.off_state + p.passes * p.blocks * kOsDigits;
    return p;
}

// what the kernel in front of a sort does on its way: thread `t` of `threads` clears its share of the scratch block
__device__ __forceinline__ void osort_clear(uint32_t *scratch, uint32_t words, uint32_t t, uint32_t threads)
{
    for (uint32_t w = t; w < words; w += threads) scratch[w] = 0u;
}

template <typename K>
__global__ __launch_bounds__(kOsHistBlock) void k_os_hist(const K *keys, uint32_t n, uint32_t begin_bit, uint32_t end_bit, uint32_t passes, uint32_t *hist)
{
    __shared__ uint32_t s_h[OsKey<K>::max_passes * kOsDigits];
    for (uint32_t k = threadIdx.x; k < passes * kOsDigits; k += kOsHistBlock) s_h[k] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * kOsHistBlock * kOsHistItems;
#pragma unroll 4
    for (uint32_t j = 0; j < kOsHistItems; ++j) {
        const uint32_t i = base + j * kOsHistBlock + threadIdx.x;
        if (i < n) {
            const K k = keys[i];
            for (uint32_t p = 0; p < passes; ++p) {
                const uint32_t bit = begin_bit + p * kOsBits, bits = min(kOsBits, end_bit - bit);
                atomicAdd(&s_h[p * kOsDigits + ((uint32_t)(k >> bit) & ((1u << bits) - 1u))], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < passes * kOsDigits; k += kOsHistBlock)
        if (s_h[k]) atomicAdd(&hist[k], s_h[k]);
}

// Ranks of a wave's 4 x 64 pairs among the pairs of the same digit earlier in the wave (order: round, lane), and the wave's
// count per digit in cnt[256] (LDS, zero on entry).  The lanes holding my digit are the AND over the digit's bits of "ballot
// of that bit, or its complement"; the first of them bumps the counter, the others sit behind it in lane order.
template <typename K, unsigned kItems>
__device__ __forceinline__ void os_wave_ranks(const K (&key)[kItems], uint32_t bit, uint32_t mask, uint32_t *cnt, uint32_t (&rank)[kItems])
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t r = 0; r < kItems; ++r) {
        const uint32_t d = (uint32_t)(key[r] >> bit) & mask;
        unsigned long long peers = ~0ull;
#pragma unroll
        for (uint32_t b = 0; b < kOsBits; ++b) {
            const bool one = (d >> b) & 1u;
            const unsigned long long bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        const int leader = __ffsll((long long)peers) - 1;
        uint32_t prev = 0;
        if ((int)lane == leader) {
            prev = cnt[d];
            cnt[d] = prev + (uint32_t)__popcll(peers);
        }
        prev = __shfl(prev, leader);
        rank[r] = prev + (uint32_t)__popcll(peers & lt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// At most 4 096 pairs: the whole sort in one workgroup, every pass in LDS (one launch, no scratch).
template <typename K>
__global__ __launch_bounds__(kOsBlock) void k_os_small(const K *keys_in, K *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n,
                                                       uint32_t begin_bit, uint32_t end_bit)
{
    constexpr unsigned kOsItems = OsKey<K>::items, kOsTile = os_tile<K>();
    __shared__ uint32_t s_cnt[kOsBlock / 64][kOsDigits];
    __shared__ K s_keys[kOsTile];
    __shared__ uint32_t s_vals[kOsTile];
    __shared__ uint32_t s_start[kOsDigits], s_part[kOsDigits / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, base = wave * (kOsItems * 64u);
    K key[kOsItems];
    uint32_t val[kOsItems], rank[kOsItems];
#pragma unroll
    for (uint32_t r = 0; r < kOsItems; ++r) {
        const uint32_t i = base + r * 64u + lane;
        key[r] = i < n ? keys_in[i] : ~(K)0;   // (padding: the last digit of every pass, behind every real pair)
        val[r] = i < n ? vals_in[i] : 0u;
    }
    for (uint32_t bit = begin_bit; bit < end_bit; bit += kOsBits) {
        const uint32_t mask = (1u << min(kOsBits, end_bit - bit)) - 1u;
        for (uint32_t k = t; k < (kOsBlock / 64) * kOsDigits; k += kOsBlock) (&s_cnt[0][0])[k] = 0u;
        __syncthreads();
        os_wave_ranks<K, kOsItems>(key, bit, mask, s_cnt[wave], rank);
        __syncthreads();
        uint32_t total = 0, excl = 0;
        if (t < kOsDigits) {
#pragma unroll
            for (uint32_t w = 0; w < kOsBlock / 64; ++w) {
                const uint32_t c = s_cnt[w][t];
                s_cnt[w][t] = total;
                total += c;
            }
            uint32_t incl = total;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off);
                if ((int)lane >= off) incl += o;
            }
            if (lane == 63u) s_part[wave] = incl;
            excl = incl - total;
        }
        __syncthreads();
        if (t < kOsDigits) {
            for (uint32_t w = 0; w < wave; ++w) excl += s_part[w];
            s_start[t] = excl;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < kOsItems; ++r) {
            const uint32_t d = (uint32_t)(key[r] >> bit) & mask;
            const uint32_t p = s_start[d] + s_cnt[wave][d] + rank[r];
            s_keys[p] = key[r];
            s_vals[p] = val[r];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < kOsItems; ++r) {
            key[r] = s_keys[base + r * 64u + lane];
            val[r] = s_vals[base + r * 64u + lane];
        }
        __syncthreads();
    }
#pragma unroll
    for (uint32_t r = 0; r < kOsItems; ++r) {
        const uint32_t i = base + r * 64u + lane;
        if (i < n) { keys_out[i] = key[r]; vals_out[i] = val[r]; }
    }
}

// look-back word of (workgroup, digit): status << 30 | count; 0 = nothing yet, 1 = the workgroup's own count, 2 = inclusive
constexpr uint32_t kOsPartial = 1u << 30, kOsInclusive = 2u << 30, kOsValue = (1u << 30) - 1u;

template <typename K>
__global__ __launch_bounds__(kOsBlock) void k_os_pass(const K *keys_in, K *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint32_t n,
                                                      const uint32_t *hist /* this pass's 256 counts */, uint32_t *state, uint32_t *ticket, uint32_t bit,
                                                      uint32_t bits)
{
    constexpr unsigned kOsItems = OsKey<K>::items, kOsTile = os_tile<K>();
    __shared__ uint32_t s_cnt[kOsBlock / 64][kOsDigits];   // per wave and digit: count, then (in place) what the waves before hold
    __shared__ K s_keys[kOsTile];
    __shared__ uint32_t s_vals[kOsTile];
    __shared__ uint32_t s_start[kOsDigits];                // first local sorted position of the digit
    __shared__ int s_gpos[kOsDigits];                      // global position of local sorted position 0 of the digit's segment, minus s_start
    __shared__ uint32_t s_part[kOsDigits / 64], s_gpart[kOsDigits / 64];
    __shared__ uint32_t s_bid;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, mask = (1u << bits) - 1u;
    if (t == 0) s_bid = atomicAdd(ticket, 1u);
    for (uint32_t k = t; k < (kOsBlock / 64) * kOsDigits; k += kOsBlock) (&s_cnt[0][0])[k] = 0u;
    __syncthreads();
    const uint32_t bid = s_bid, base = bid * kOsTile + wave * (kOsItems * 64u);
    K key[kOsItems];
    uint32_t val[kOsItems], rank[kOsItems];
#pragma unroll
    for (uint32_t r = 0; r < kOsItems; ++r) {
        const uint32_t i = base + r * 64u + lane;
        key[r] = i < n ? keys_in[i] : ~(K)0;   // (padding: the last digit, behind every real pair of the last workgroup)
        val[r] = i < n ? vals_in[i] : 0u;
    }
    os_wave_ranks<K, kOsItems>(key, bit, mask, s_cnt[wave], rank);
    __syncthreads();
    // per digit: what the waves before hold, the workgroup's total (published at once: nobody behind waits longer than
    // that); the digits' first local positions and the first global position of every digit's segment are two exclusive
    // scans over the 256 digits (threads 0 .. 255, one digit each)
    uint32_t total = 0, local_excl = 0, global_excl = 0;
    if (t < kOsDigits) {
#pragma unroll
        for (uint32_t w = 0; w < kOsBlock / 64; ++w) {
            const uint32_t c = s_cnt[w][t];
            s_cnt[w][t] = total;
            total += c;
        }
        __hip_atomic_store(&state[bid * kOsDigits + t], (bid == 0 ? kOsInclusive : kOsPartial) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t g = hist[t];
        uint32_t incl = total, gincl = g;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off), go = __shfl_up(gincl, off);
            if ((int)lane >= off) { incl += o; gincl += go; }
        }
        if (lane == 63u) { s_part[wave] = incl; s_gpart[wave] = gincl; }
        local_excl = incl - total;
        global_excl = gincl - g;
    }
    __syncthreads();
    if (t < kOsDigits) {
        for (uint32_t w = 0; w < wave; ++w) { local_excl += s_part[w]; global_excl += s_gpart[w]; }
        // what the workgroups in front hold of this digit: a chain per digit, walked back to the first inclusive word
        uint32_t excl = 0;
        if (bid > 0) {
            int look = (int)bid - 1;
            for (;;) {
                uint32_t w;
                for (;;) {
                    w = __hip_atomic_load(&state[(uint32_t)look * kOsDigits + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (w >> 30) break;   // (a predecessor that has not published yet: it is running, by its ticket)
                    __builtin_amdgcn_s_sleep(1);
                }
                excl += w & kOsValue;
                if ((w >> 30) == 2u || look == 0) break;
                --look;
            }
            __hip_atomic_store(&state[bid * kOsDigits + t], kOsInclusive | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_start[t] = local_excl;
        s_gpos[t] = (int)(global_excl + excl) - (int)local_excl;
    }
    __syncthreads();
    // into sorted order in LDS
#pragma unroll
    for (uint32_t r = 0; r < kOsItems; ++r) {
        const uint32_t d = (uint32_t)(key[r] >> bit) & mask;
        const uint32_t p = s_start[d] + s_cnt[wave][d] + rank[r];
        s_keys[p] = key[r];
        s_vals[p] = val[r];
    }
    __syncthreads();
    const uint32_t valid = min(kOsTile, n - min(n, bid * kOsTile));
#pragma unroll
    for (uint32_t j = 0; j < kOsItems; ++j) {
        const uint32_t p = j * kOsBlock + t;
        if (p < valid) {
            const K k = s_keys[p];
            const uint32_t to = (uint32_t)(s_gpos[(uint32_t)(k >> bit) & mask] + (int)p);
            keys_out[to] = k;
            vals_out[to] = s_vals[p];
        }
    }
}

// Sorts n pairs by bits [begin_bit, end_bit) of the key.  `scratch` (plan.words words) must be all zero when the first
// kernel starts and is dirty afterwards.  Returns through *in_first whether the sorted pairs lie in (keys_a, vals_a)
// (true) or in (keys_b, vals_b); the other pair is overwritten too.
// whether osort_pairs leaves the result in the pair it started from
template <typename K = uint32_t> inline bool osort_ends_in_first(const OsortPlan &p, size_t n) { return n == 0 || p.passes == 0 || (n > os_tile<K>() && p.passes % 2 == 0); }

// `hist_ready`: the digit histograms (passes x 256 counts from begin_bit on, as k_os_hist leaves them) if the kernel that wrote the
// keys has counted them on its way (k_source_keys_hist): no histogram launch then; nullptr: k_os_hist counts them in the scratch block.
template <typename K>
inline hipError_t osort_pairs(const OsortPlan &p, uint32_t *scratch, K *keys_a, K *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n,
                              unsigned begin_bit, unsigned end_bit, hipStream_t st, bool *in_first, const uint32_t *hist_ready = nullptr)
{
    constexpr unsigned kOsTile = os_tile<K>();
    *in_first = true;
    if (n == 0 || p.passes == 0) return hipSuccess;
    // (every check before the first launch: an error return leaves nothing queued on a dirty scratch block)
    if (p.passes > OsKey<K>::max_passes || end_bit > sizeof(K) * 8 || n >= (1ull << 30)) return hipErrorInvalidValue;
    if (p.blocks != (uint32_t)((n + kOsTile - 1) / kOsTile)) return hipErrorInvalidValue;
    if (n <= kOsTile) {   // one workgroup, one launch; the result in the second pair
        k_os_small<K><<<1, kOsBlock, 0, st>>>(keys_a, keys_b, vals_a, vals_b, (uint32_t)n, begin_bit, end_bit);
        *in_first = false;
        return hipGetLastError();
    }
    const uint32_t *hist = hist_ready ? hist_ready : scratch + p.off_hist;
    if (!hist_ready) k_os_hist<K><<<p.hist_blocks, kOsHistBlock, 0, st>>>(keys_a, (uint32_t)n, begin_bit, end_bit, p.passes, scratch + p.off_hist);
    bool from_a = true;
    unsigned bit = begin_bit;
    for (uint32_t pass = 0; pass < p.passes; ++pass, bit += kOsBits) {
        const uint32_t bits = std::min(kOsBits, end_bit - bit);
        k_os_pass<K><<<p.blocks, kOsBlock, 0, st>>>(from_a ? keys_a : keys_b, from_a ? keys_b : keys_a, from_a ? vals_a : vals_b, from_a ? vals_b : vals_a,
                                                    (uint32_t)n, hist + pass * kOsDigits, scratch + p.off_state + (size_t)pass * p.blocks * kOsDigits,
                                                    scratch + p.off_ticket + pass, bit, bits);
        from_a = !from_a;
    }
    *in_first = from_a;
    return hipGetLastError();
}

// ---- the vocabulary of the call sites (rounds 3-5: csrc/radix32.hpp).  A plan: the scratch words that must be zero when the sort
// starts (cleared by the kernel that writes the keys on its way: osort_clear), the digit passes, where the result ends.
struct Radix32Plan : OsortPlan {
    uint32_t places = 0;   // digit passes (an even number ends in the buffer pair it started from, unless one workgroup sorts it all)
    bool ends_in_first = true;
};

template <typename K = uint32_t> inline Radix32Plan radix32_plan(size_t n, unsigned begin_bit, unsigned end_bit)
{
    Radix32Plan p;
    static_cast<OsortPlan &>(p) = osort_plan<K>(n, begin_bit, end_bit);
    p.places = p.passes;
    p.ends_in_first = osort_ends_in_first<K>(p, n);
    return p;
}

__device__ __forceinline__ void radix32_clear(uint32_t *scratch, uint32_t words, uint32_t t, uint32_t threads) { osort_clear(scratch, words, t, threads); }

template <typename K>
inline hipError_t radix32_sort_pairs(const Radix32Plan &p, uint32_t *scratch, K *keys_a, K *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n,
                                     unsigned begin_bit, unsigned end_bit, hipStream_t st, bool *in_first, const uint32_t *hist_ready = nullptr)
{
    return osort_pairs<K>(p, scratch, keys_a, keys_b, vals_a, vals_b, n, begin_bit, end_bit, st, in_first, hist_ready);
}

// A sort whose keys were NOT written by a kernel of this library that clears the scratch block on its way: the block is cleared by a
// memset in front (one fill launch).  `scratch` holds osort_plan<K>(n, begin_bit, end_bit).words words.
template <typename K>
inline hipError_t osort_pairs_cleared(uint32_t *scratch, K *keys_a, K *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n, unsigned begin_bit,
                                      unsigned end_bit, hipStream_t st, bool *in_first)
{
    const OsortPlan p = osort_plan<K>(n, begin_bit, end_bit);
    if (n > os_tile<K>()) {
        const hipError_t e = hipMemsetAsync(scratch, 0, (size_t)p.words * 4, st);
        if (e != hipSuccess) return e;
    }
    return osort_pairs<K>(p, scratch, keys_a, keys_b, vals_a, vals_b, n, begin_bit, end_bit, st, in_first);
}

}  // namespace rsreg
