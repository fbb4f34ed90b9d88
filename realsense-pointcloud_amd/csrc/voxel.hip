// voxel.hip — pcl::ApproximateVoxelGrid<PointXYZRGB>::filter on the GPU, record for record.
//
// Reference call sites: src/incremental_icp.hpp:54-55, src/icp_edge_based_registration.hpp:47,
// 59-60,75-76, src/ndt_edge_based_registration.hpp:45,57-58,68-69.  PCL's filter (SURVEY.md
// App. A.5; sequential restatement: voxel_host.cpp) streams the points through a 512-slot hash
// history: a point whose slot holds a different voxel flushes that voxel's centroid to the
// output and takes the slot; at the end the occupied slots are flushed in slot order.  The
// output therefore depends on the input order -- but not on anything a parallel machine cannot
// reconstruct:
//   * the points that hash to one slot form an independent stream; inside it, every maximal run
//     of consecutive points with the same voxel yields exactly one centroid;
//   * a run that is followed by another run in its slot is emitted when that next run's first
//     point arrives, i.e. at that point's position in the input; the last run of every slot is
//     emitted at the end, in slot order;
//   * a centroid is a float sum in input order divided by the count.
// So: stable-sort the point indices by slot, cut the runs, add up each run in order (same float
// additions as PCL: a thread per short run, a wave per run of 48 points and more, and for the runs
// of 1 024 points and more -- PCL's default 1 m leaf -- a workgroup that gets the same bits from a
// scan of rounding steps, k_vox_huge_runs below), and sort the runs by their emission position.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <algorithm>
#include <cmath>

#include "rsreg_ctx.hpp"
#include "osort.hpp"
#include "oscan.hpp"

namespace rsreg {
namespace {

constexpr int kVBlock = 256;
constexpr uint32_t kHist = 512;   // PCL's histsize_

struct VoxelOf {
    int ix, iy, iz;
    bool ok;
};

__device__ __forceinline__ VoxelOf voxel_of(const char *rec, float ivx, float ivy, float ivz)
{
    const float *p = reinterpret_cast<const float *>(rec);
    const float x = p[0], y = p[1], z = p[2];
    VoxelOf v;
    v.ok = isfinite(x) && isfinite(y) && isfinite(z);
    v.ix = (int)floorf(__fmul_rn(x, ivx));
    v.iy = (int)floorf(__fmul_rn(y, ivy));
    v.iz = (int)floorf(__fmul_rn(z, ivz));
    return v;
}

__device__ __forceinline__ uint32_t slot_of(const VoxelOf &v)
{
    return (uint32_t)(v.ix * 7171 + v.iy * 3079 + v.iz * 4231) & (kHist - 1);
}

// key = slot (non-finite points: kHist, sorted behind everything), value = input position; the filter's four counters
// start at zero (this is its first kernel: no launch of a memset for 16 bytes)
__global__ __launch_bounds__(kVBlock) void k_vox_keys(const char *recs, size_t stride, uint32_t n, float ivx, float ivy, float ivz,
                                                      uint32_t *keys, uint32_t *vals, uint32_t *stats, uint32_t *sort_scratch,
                                                      uint32_t sort_scratch_words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4) stats[i] = 0u;
    if (sort_scratch) rsreg::radix32_clear(sort_scratch, sort_scratch_words, i, gridDim.x * blockDim.x);   // (the state of the sort that follows: radix32.hpp)
    if (i >= n) return;
    const VoxelOf v = voxel_of(recs + (size_t)i * stride, ivx, ivy, ivz);
    keys[i] = v.ok ? slot_of(v) : kHist;
    vals[i] = i;
}

// p = position in the slot-sorted order; flag[p] = 1 when a run starts there
__global__ __launch_bounds__(kVBlock) void k_vox_flags(const char *recs, size_t stride, uint32_t n, float ivx, float ivy, float ivz,
                                                       const uint32_t *skeys, const uint32_t *svals, uint32_t *flag)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t f = 0;
    if (skeys[p] < kHist) {
        f = 1;
        if (p > 0 && skeys[p - 1] == skeys[p]) {
            const VoxelOf a = voxel_of(recs + (size_t)svals[p] * stride, ivx, ivy, ivz);
            const VoxelOf b = voxel_of(recs + (size_t)svals[p - 1] * stride, ivx, ivy, ivz);
            if (a.ix == b.ix && a.iy == b.iy && a.iz == b.iz) f = 0;
        }
    }
    flag[p] = f;
}

// run r starts at sorted position start[r]; stats[0] = number of runs, stats[1] = number of finite points
__global__ __launch_bounds__(kVBlock) void k_vox_starts(const uint32_t *skeys, const uint32_t *flag, const uint32_t *rid, uint32_t n,
                                                        uint32_t *start, uint32_t *stats)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (flag[p]) start[rid[p]] = p;
    if (p == n - 1) stats[0] = rid[p] + flag[p];
    if (skeys[p] < kHist && (p == n - 1 || skeys[p + 1] >= kHist)) stats[1] = p + 1;
}

constexpr uint32_t kLongRun = 48;   // runs from this length on are summed by a whole wave (k_vox_long_runs)
constexpr uint32_t kHugeRun = 1024; // and from this length on by a workgroup that scans instead of adding (k_vox_huge_runs)

__device__ __forceinline__ void vox_store_run(const float acc[7], uint32_t a, uint32_t b, uint32_t n, uint32_t nfin, uint32_t r,
                                              const uint32_t *skeys, const uint32_t *svals, float *cent, uint32_t *ekey, uint32_t *erun)
{
    const float cnt = (float)(b - a);
    float *o = cent + (size_t)r * 8;
    for (int k = 0; k < 7; ++k) o[k] = __fdiv_rn(acc[k], cnt);
    const uint32_t slot = skeys[a];
    const bool last_of_slot = (b >= nfin) || skeys[b] != slot;
    ekey[r] = last_of_slot ? n + slot : svals[b];
    erun[r] = r;
}

// the seven values PCL accumulates for a point: x y z, the rgb field read as a float, r g b
__device__ __forceinline__ void vox_terms(const char *rec, float t[7])
{
    const float *f = reinterpret_cast<const float *>(rec);
    const unsigned char *c = reinterpret_cast<const unsigned char *>(rec + 16);
    t[0] = f[0]; t[1] = f[1]; t[2] = f[2]; t[3] = f[4];
    t[4] = (float)c[2]; t[5] = (float)c[1]; t[6] = (float)c[0];
}

// One thread per run: the centroid record (sums in input order, like PCL) and the run's
// emission key: the input position of the first point of the next run in the same slot, or
// n + slot for the last run of a slot.  Long runs are only listed here (stats[2] counts them).
__global__ __launch_bounds__(kVBlock) void k_vox_runs(const char *recs, size_t stride, uint32_t n, const uint32_t *skeys,
                                                      const uint32_t *svals, const uint32_t *start, uint32_t *stats,
                                                      float *cent /* 8 floats per run */, uint32_t *ekey, uint32_t *erun,
                                                      uint32_t *long_runs, uint32_t *huge_runs)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nr = stats[0], nfin = stats[1];
    if (r >= nr) return;
    const uint32_t a = start[r], b = (r + 1 < nr) ? start[r + 1] : nfin;
    if (b - a >= kHugeRun) {
        huge_runs[atomicAdd(&stats[3], 1u)] = r;
        return;
    }
    if (b - a >= kLongRun) {
        long_runs[atomicAdd(&stats[2], 1u)] = r;
        return;
    }
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (uint32_t p = a; p < b; ++p) {
        float t[7];
        vox_terms(recs + (size_t)svals[p] * stride, t);
        for (int k = 0; k < 7; ++k) acc[k] = __fadd_rn(acc[k], t[k]);
    }
    vox_store_run(acc, a, b, n, nfin, r, skeys, svals, cent, ekey, erun);
}

// The seven terms of a point from two loads when the records allow it (32-byte PointXYZRGB records, 16-byte aligned):
// xyz in one 16-byte load, the rgb word once (it is both the float PCL adds and the three bytes).
struct VoxRaw {
    float x, y, z;
    uint32_t rgb;
};
__device__ __forceinline__ VoxRaw vox_load(const char *rec, bool vec)
{
    VoxRaw r;
    if (vec) {
        const float4 a = *reinterpret_cast<const float4 *>(rec);
        r.x = a.x; r.y = a.y; r.z = a.z;
    } else {
        const float *f = reinterpret_cast<const float *>(rec);
        r.x = f[0]; r.y = f[1]; r.z = f[2];
    }
    r.rgb = *reinterpret_cast<const uint32_t *>(rec + 16);
    return r;
}

// One wave per long run (48 .. 1 023 points since round 6; longer ones go to k_vox_huge_runs: a 1 m leaf puts 10^4..10^5
// points in a run, and such a run WAS the filter's run time here: its seven sums are chains of dependent float additions in
// input order, like PCL's, ~8 cycles a link).  64 lanes fetch 64
// points at a time and pass them through LDS (component-major); lanes 0..6 each add one component in input order.
// Nothing but the chain may be on the critical path: the point indices are fetched eight chunks ahead and the records
// four chunks ahead (a gather from HBM takes longer than four chunks of additions), and chunk k + 1 is written to the
// other half of the LDS buffer before chunk k is added.  Positions past the end of the run contribute +0.0f, which
// leaves a float sum unchanged.
__global__ __launch_bounds__(kVBlock) void k_vox_long_runs(const char *recs, size_t stride, uint32_t n, const uint32_t *skeys,
                                                           const uint32_t *svals, const uint32_t *start, const uint32_t *stats,
                                                           float *cent, uint32_t *ekey, uint32_t *erun, const uint32_t *long_runs)
{
    constexpr int kAhead = 4;
    __shared__ __attribute__((aligned(16))) float sh[kVBlock / 64][2][8][64];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t nr = stats[0], nfin = stats[1], n_long = stats[2];
    const bool vec = (stride % 16 == 0) && ((reinterpret_cast<size_t>(recs) & 15) == 0);
    for (uint32_t k = wave; k < n_long; k += n_waves) {
        const uint32_t r = long_runs[k];
        const uint32_t a = start[r], b = (r + 1 < nr) ? start[r + 1] : nfin;
        const uint32_t n_chunks = (b - a + 63) / 64;
        // chunk c covers sorted positions a + 64 c + lane; idx[j] / raw[j]: what is in flight for chunk == j (mod kAhead)
        uint32_t idx[kAhead];
        VoxRaw raw[kAhead];
        auto fetch_idx = [&](uint32_t c) -> uint32_t {
            const uint32_t p = a + 64 * c + lane;
            return p < b ? svals[p] : 0xffffffffu;
        };
        auto fetch_raw = [&](uint32_t i) -> VoxRaw {
            if (i == 0xffffffffu) return VoxRaw{0.0f, 0.0f, 0.0f, 0u};
            return vox_load(recs + (size_t)i * stride, vec);
        };
        auto stage = [&](int half, const VoxRaw &v, uint32_t i) {   // a lane's point into its column of the LDS half
            float *col = &sh[w][half][0][lane];
            const bool in = i != 0xffffffffu;
            col[0 * 64] = v.x; col[1 * 64] = v.y; col[2 * 64] = v.z;
            col[3 * 64] = __uint_as_float(v.rgb);
            col[4 * 64] = in ? (float)((v.rgb >> 16) & 0xffu) : 0.0f;
            col[5 * 64] = in ? (float)((v.rgb >> 8) & 0xffu) : 0.0f;
            col[6 * 64] = in ? (float)(v.rgb & 0xffu) : 0.0f;
        };
        uint32_t was[kAhead];   // the index a staged record was fetched with (decides the padding)
#pragma unroll
        for (int j = 0; j < kAhead; ++j) {
            was[j] = fetch_idx((uint32_t)j);
            raw[j] = fetch_raw(was[j]);
        }
#pragma unroll
        for (int j = 0; j < kAhead; ++j) idx[j] = fetch_idx((uint32_t)(kAhead + j));
        stage(0, raw[0], was[0]);
        float acc = 0.0f;
        // one step: chunk c is added; J = c mod kAhead, spelled out so that every register array index is a constant
#define RSREG_VOX_STEP(J)                                                                                               \
    {                                                                                                                   \
        const uint32_t c = c0 + (J);                                                                                    \
        if (c >= n_chunks) break;                                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                          \
        __builtin_amdgcn_wave_barrier();                                                                                \
        /* chunk c + 1 into the other half (its records were asked for three steps ago); then these registers take     \
           chunk c + kAhead (indices asked for four steps ago) and the indices of chunk c + 2 kAhead go out */         \
        stage((int)((c + 1) & 1u), raw[((J) + 1) % kAhead], was[((J) + 1) % kAhead]);                                   \
        was[(J)] = idx[(J)];                                                                                            \
        raw[(J)] = fetch_raw(idx[(J)]);                                                                                 \
        idx[(J)] = fetch_idx(c + 2 * kAhead);                                                                           \
        if (lane < 7) {                                                                                                 \
            const float4 *v = reinterpret_cast<const float4 *>(&sh[w][c & 1u][lane][0]);                                \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                              \
            {                                                                                                           \
                const float4 q = v[i];                                                                                  \
                acc = __fadd_rn(acc, q.x);                                                                              \
                acc = __fadd_rn(acc, q.y);                                                                              \
                acc = __fadd_rn(acc, q.z);                                                                              \
                acc = __fadd_rn(acc, q.w);                                                                              \
            }                                                                                                           \
        }                                                                                                               \
    }
        static_assert(kAhead == 4, "the steps below are spelled out for four chunks in flight");
        for (uint32_t c0 = 0; c0 < n_chunks; c0 += kAhead) {
            RSREG_VOX_STEP(0)
            RSREG_VOX_STEP(1)
            RSREG_VOX_STEP(2)
            RSREG_VOX_STEP(3)
        }
#undef RSREG_VOX_STEP
        __builtin_amdgcn_wave_barrier();
        float all[7];
        for (int c = 0; c < 7; ++c) all[c] = __shfl(acc, c);
        if (lane == 0) vox_store_run(all, a, b, n, nfin, r, skeys, svals, cent, ekey, erun);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Huge runs (PCL's default 1 m leaf puts 10^4..10^5 points of a frame into ONE run): a float sum in input order, bit for bit,
// WITHOUT adding the floats one after the other.
//
// While a running sum s stays inside one binade [2^E, 2^(E+1)), s = m u with u = 2^(E-23) and an integer m, and s (+) t =
// round-to-nearest-even(m + t/u) u: the step moves m by rndne(t/u) whatever m is -- except when t/u lies exactly half way
// between two integers, where the result is the even neighbour, i.e. depends on the PARITY of m only.  So every step, and every
// sequence of steps, is a function m -> m + (m odd ? A1 : A0); two such functions compose into another (f then g: h_p = f_p +
// g_[(p + f_p) & 1]), the composition is associative, and a wave can scan it: lane l composes its own kHL points, the wave
// scans the 64 functions (plain integer additions unless somebody met a half-way case), and every lane learns the sum it would
// START from.
//
// The binade is only an assumption (the sum doubles ~17 times on its way to 10^5 points, a coordinate may cancel, a colour sum
// may pass 2^24).  It is not argued away, it is CHECKED: every lane adds its own points to its predicted start with real float
// additions (16 of them, all lanes at once) and compares the bits with its predicted end.  Lane 0 starts from the true sum; if
// lane i started from the true sum and its real additions end where the prediction said, lane i + 1 started from the true sum
// too.  So everything before the first lane that disagrees is exact whatever the prediction was made of, that lane's REAL end is
// the true sum there, and the wave goes round again from the next lane with the binade of that sum.  A round costs what ~40
// additions cost and settles up to 1 024 points; a run of 37 000 points (the longest of a 307 k-point frame) takes ~50 rounds.
//
// One workgroup per run.  Two of its eight waves fetch the run's records -- 4 096 points at a time, in two halves of 2 048 (one
// half's records and the other's indices in flight in registers between two steps) -- and lay them out in one half of the LDS
// by component, while the other six sum ONE component each (x y z r g b) out of the other half, window after window of
// 1 024 points, every wave at its own pace: no wave waits for another's rounds, and the gathers are issued beside the rounds,
// not in front of them (with all eight waves fetching 8 192 points and then six of them summing, a quarter of the kernel was
// issuing loads: 63 -> 49 us for the runs of a 307 k frame; the two fetching waves are now what a step waits for -- they hold
// 222 registers for sixteen records in flight a lane, and a second set would spill).  (PCL also accumulates the rgb word read
// as a float -- usually a NaN -- and never reads it back: the centroid's fourth component is not part of the output record, and
// this kernel leaves it out.)
constexpr int kHB = 512;                          // threads of a workgroup
constexpr int kHL = 16;                           // points of a lane in a window
constexpr uint32_t kHWin = 64u * kHL;             // points of a window (one wave, one round at least)
constexpr uint32_t kHWins = 4;                    // windows in one half of the LDS
constexpr uint32_t kHSuper = kHWin * kHWins;      // = points fetched at a time
constexpr int kHLoaders = 128;                    // threads that fetch (the last two waves); the other six waves sum
constexpr uint32_t kHHalf = kHSuper / 2;          // the fetching threads take the 4 096 in two halves (registers)
constexpr int kHPer = (int)(kHHalf / kHLoaders);  // ... each of them this many points of a half
constexpr uint32_t kHPitch = 68;                  // a window in the LDS: 16 rows (a lane's i-th point) of 64 values, 68 words apart:
                                                  //   64 consecutive points written (16 rows x 4 columns) hit 64 different banks, a row read does too
static_assert(kHL == 16 && kHB - kHLoaders == 6 * 64 && 2 * kHPer * kHLoaders == (int)kHSuper, "six summing waves, windows of 16 points a lane");

struct HFn {   // m -> m + (m odd ? a1 : a0), modulo 2^32
    uint32_t a0, a1;
};
__device__ __forceinline__ HFn h_then(const HFn f, const HFn g)
{
    HFn h;
    h.a0 = f.a0 + ((f.a0 & 1u) ? g.a1 : g.a0);
    h.a1 = f.a1 + ((f.a1 & 1u) ? g.a0 : g.a1);
    return h;
}
struct HBase {   // a sum as (sign, exponent field >= 1, integer mantissa): value = (-1)^neg m 2^(e - 150)
    uint32_t e, m, neg;
};
__device__ __forceinline__ HBase h_base(uint32_t bits)
{
    HBase b;
    const uint32_t ef = (bits >> 23) & 0xffu;
    b.e = ef ? ef : 1u;
    b.m = (bits & 0x7fffffu) | (ef ? 0x800000u : 0u);
    b.neg = bits >> 31;
    return b;
}
__device__ __forceinline__ uint32_t h_bits(const HBase &b, uint32_t m) { return ((((b.e - 1u) << 23) + m) & 0x7fffffffu) | (b.neg << 31); }

// inclusive sum over the 64 lanes of a wave in six DPP steps (rows of 16 lanes, then the rows' last lanes handed on)
__device__ __forceinline__ uint32_t h_wave_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// One window of one component: t[i] = the lane's i-th term (a float; bytes already converted), s = the sum so far (bits, the
// same in every lane).  Returns the sum after the window.  kBytes: the terms are integers 0..255 (a colour).
template <bool kBytes>
__device__ __forceinline__ uint32_t h_window(const float (&t)[kHL], uint32_t s, uint32_t lane)
{
    uint32_t lo = 0;   // lanes below lo are settled; s is the true sum in front of lane lo
    for (;;) {
        const HBase bs = h_base(s);
        const int sh = 150 - (int)bs.e;
        HFn own{0u, 0u};
        if (lane >= lo) {
            if (kBytes && sh >= 0 && sh < 20) {
                // a colour sum well below 2^24: bytes on a grid of 2^-sh, nothing to round (should the sum leave that range
                // inside the window, the check below notices, like everything else)
                uint32_t sum = 0;
#pragma unroll
                for (int i = 0; i < kHL; ++i) sum += (uint32_t)t[i];
                sum <<= sh;
                own.a0 = own.a1 = bs.neg ? 0u - sum : sum;
            } else {
                // all 16 steps as plain ones first (no branch between them: they overlap in the pipeline) ...
                bool tie_any = false;
#pragma unroll
                for (int i = 0; i < kHL; ++i) {
                    float q = ldexpf(t[i], sh);
                    if (bs.neg) q = -q;
                    const float rn = rintf(q);
                    tie_any |= fabsf(q - rn) == 0.5f;
                    own.a0 += (uint32_t)(int)rn;
                }
                own.a1 = own.a0;
                if (__builtin_amdgcn_ballot_w64(tie_any)) {   // (uniform, rare) ... again with the parity where a step needs it
                    own = HFn{0u, 0u};
#pragma unroll
                    for (int i = 0; i < kHL; ++i) {
                        float q = ldexpf(t[i], sh);
                        if (bs.neg) q = -q;
                        const float rn = rintf(q);
                        const bool tie = fabsf(q - rn) == 0.5f;
                        const uint32_t st = (uint32_t)(int)(tie ? floorf(q) : rn);
                        const uint32_t x0 = own.a0 + st, x1 = own.a1 + st;
                        own.a0 = x0 + (tie ? (x0 & 1u) : 0u);
                        own.a1 = x1 + (tie ? (~x1 & 1u) : 0u);
                    }
                }
            }
        }
        HFn upto, before;   // the lanes up to and including this one / before it
        if (__builtin_amdgcn_ballot_w64(own.a0 != own.a1)) {   // (uniform, rare) a step of this window depends on the parity
            upto = own;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                HFn f;
                f.a0 = __shfl_up(upto.a0, d);
                f.a1 = __shfl_up(upto.a1, d);
                if (lane >= (uint32_t)d) upto = h_then(f, upto);
            }
            before.a0 = __shfl_up(upto.a0, 1);
            before.a1 = __shfl_up(upto.a1, 1);
            if (lane == 0u) before = HFn{0u, 0u};
        } else {   // plain steps compose by adding
            upto.a0 = upto.a1 = h_wave_scan_add(own.a0);
            before.a0 = before.a1 = upto.a0 - own.a0;
        }
        const uint32_t p0 = bs.m & 1u;
        float v = __uint_as_float(h_bits(bs, bs.m + (p0 ? before.a1 : before.a0)));
#pragma unroll
        for (int i = 0; i < kHL; ++i) v = __fadd_rn(v, t[i]);
        const uint32_t want = h_bits(bs, bs.m + (p0 ? upto.a1 : upto.a0));
        const unsigned long long bad = __builtin_amdgcn_ballot_w64(lane >= lo && __float_as_uint(v) != want);
        if (!bad) return (uint32_t)__builtin_amdgcn_readlane((int)want, 63);
        const uint32_t first = (uint32_t)__builtin_ctzll(bad);
        s = (uint32_t)__shfl((int)__float_as_uint(v), (int)first);
        s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
        if (first == 63u) return s;
        lo = first + 1u;
    }
}

__global__ __launch_bounds__(kHB) void k_vox_huge_runs(const char *recs, size_t stride, uint32_t n, const uint32_t *skeys,
                                                       const uint32_t *svals, const uint32_t *start, const uint32_t *stats,
                                                       float *cent, uint32_t *ekey, uint32_t *erun, const uint32_t *huge_runs)
{
    // [half][component: x y z rgb][window][row][column]
    __shared__ uint32_t sh[2][4][kHWins][kHL * kHPitch];
    __shared__ uint32_t sh_s[6];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool loader = tid >= (uint32_t)(kHB - kHLoaders);
    const uint32_t tl = tid - (uint32_t)(kHB - kHLoaders);   // (loaders) 0 .. 127
    const uint32_t nr = stats[0], nfin = stats[1], n_huge = stats[3];
    const bool vec = (stride % 16 == 0) && ((reinterpret_cast<size_t>(recs) & 15) == 0);
    for (uint32_t k = blockIdx.x; k < n_huge; k += gridDim.x) {
        const uint32_t r = huge_runs[k];
        const uint32_t a = start[r], b = (r + 1 < nr) ? start[r + 1] : nfin;
        // loader tl fetches the points base + tl + 128 j of a half (2 048 points) of the 4 096 (consecutive threads, consecutive
        // indices).  Every load is unconditional -- a position past the end of the run reads SOME record of the cloud and is set
        // to zero where it is written to the LDS -- so that nothing waits for a load before the next one is issued.
        auto fetch_idx = [&](uint32_t base, uint32_t idx[kHPer]) {
#pragma unroll
            for (int j = 0; j < kHPer; ++j) idx[j] = svals[min(base + tl + (uint32_t)kHLoaders * (uint32_t)j, n - 1u)];
        };
        auto fetch_recs = [&](const uint32_t idx[kHPer], VoxRaw rec[kHPer]) {
#pragma unroll
            for (int j = 0; j < kHPer; ++j) rec[j] = vox_load(recs + (size_t)idx[j] * stride, vec);
        };
        // (half_no: which half of the 4 096 these records are; base: the run position of the 4 096's first point)
        auto lay_out = [&](uint32_t lds_half, uint32_t base, uint32_t half_no, const VoxRaw rec[kHPer]) {
#pragma unroll
            for (int j = 0; j < kHPer; ++j) {
                const uint32_t e = half_no * kHHalf + tl + (uint32_t)kHLoaders * (uint32_t)j;   // point e of these 4 096
                const uint32_t w = e / kHWin, ew = e % kHWin;                 // window, point of the window: lane ew / 16, its point ew % 16
                const uint32_t at = (ew % (uint32_t)kHL) * kHPitch + ew / (uint32_t)kHL;
                const bool in = base + e < b;   // (past the end of the run: +0.0f and a black byte leave every sum as it is)
                sh[lds_half][0][w][at] = in ? __float_as_uint(rec[j].x) : 0u;
                sh[lds_half][1][w][at] = in ? __float_as_uint(rec[j].y) : 0u;
                sh[lds_half][2][w][at] = in ? __float_as_uint(rec[j].z) : 0u;
                sh[lds_half][3][w][at] = in ? rec[j].rgb : 0u;
            }
        };
        // in flight between two steps: the records of the first half of the next 4 096 (rec) and the indices of their second half (idx_b)
        uint32_t idx_a[kHPer], idx_b[kHPer];
        VoxRaw rec[kHPer];
        __syncthreads();   // (the LDS has been read: the previous run's last points)
        if (loader) {
            fetch_idx(a, idx_a);
            fetch_idx(a + kHHalf, idx_b);
            fetch_recs(idx_a, rec);
            fetch_idx(a + kHSuper, idx_a);
            lay_out(0u, a, 0u, rec);
            fetch_recs(idx_b, rec);
            fetch_idx(a + kHSuper + kHHalf, idx_b);
            lay_out(0u, a, 1u, rec);
            fetch_recs(idx_a, rec);   // (the first half of the second 4 096; requested even past the end of the run: never laid out then)
        }
        __syncthreads();
        uint32_t s = 0u;   // (waves 0..5) the wave's component so far: +0.0f
        uint32_t half = 0u;
        for (uint32_t base = a; base < b; base += kHSuper, half ^= 1u) {
            if (loader) {
                if (base + kHSuper < b) {   // the next 4 096 points into the other half of the LDS, beside the sums of this one
                    const uint32_t nb = base + kHSuper;
                    lay_out(half ^ 1u, nb, 0u, rec);
                    fetch_recs(idx_b, rec);              // their second half: waited for below, by these two waves only
                    fetch_idx(nb + kHSuper, idx_a);
                    fetch_idx(nb + kHSuper + kHHalf, idx_b);
                    lay_out(half ^ 1u, nb, 1u, rec);
                    fetch_recs(idx_a, rec);              // the first half of the 4 096 after them
                }
            } else {
                const uint32_t n_win = min(kHWins, (b - base + kHWin - 1u) / kHWin);
                for (uint32_t w = 0; w < n_win; ++w) {
                    float t[kHL];
                    if (wave < 3u) {
#pragma unroll
                        for (int i = 0; i < kHL; ++i) t[i] = __uint_as_float(sh[half][wave][w][(uint32_t)i * kHPitch + lane]);
                        s = h_window<false>(t, s, lane);
                    } else {
                        const uint32_t shift = 8u * (5u - wave);
#pragma unroll
                        for (int i = 0; i < kHL; ++i) t[i] = (float)((sh[half][3][w][(uint32_t)i * kHPitch + lane] >> shift) & 0xffu);
                        s = h_window<true>(t, s, lane);
                    }
                }
            }
            __syncthreads();   // (this half has been summed, the other one is laid out)
        }
        if (!loader && lane == 0u) sh_s[wave] = s;
        __syncthreads();
        if (tid == 0) {
            float all[7];
            all[0] = __uint_as_float(sh_s[0]); all[1] = __uint_as_float(sh_s[1]); all[2] = __uint_as_float(sh_s[2]);
            all[3] = 0.0f;   // (the sum PCL never reads)
            all[4] = __uint_as_float(sh_s[3]); all[5] = __uint_as_float(sh_s[4]); all[6] = __uint_as_float(sh_s[5]);
            vox_store_run(all, a, b, n, nfin, r, skeys, svals, cent, ekey, erun);
        }
    }
}

// output record j = centroid of run order[j]: a default PointXYZRGB with xyz and packed rgb set
__global__ __launch_bounds__(kVBlock) void k_vox_emit(const float *cent, const uint32_t *order, const uint32_t *stats, size_t stride,
                                                      char *out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= stats[0]) return;
    const float *c = cent + (size_t)order[j] * 8;
    char *rec = out + (size_t)j * stride;
    for (size_t k = 0; k < stride; k += 4) *reinterpret_cast<uint32_t *>(rec + k) = 0u;
    float *f = reinterpret_cast<float *>(rec);
    f[0] = c[0];
    f[1] = c[1];
    f[2] = c[2];
    f[3] = 1.0f;
    const int rgb = ((int)c[4] << 16) | ((int)c[5] << 8) | (int)c[6];
    *reinterpret_cast<int *>(rec + 16) = rgb;
}

inline uint32_t div_up_u(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

}  // namespace
}  // namespace rsreg

using namespace rsreg;

namespace rsreg {

// The filter on records already in HBM (d_in, N records of `stride` bytes); the filtered records
// land in the scratch set's `out`, *n_out of them.  One host synchronisation (the number of runs); what follows it
// (the runs' sums, their order, the output records) is only queued.  `side` >= 0: that scratch set of the context and its
// stream (rsreg_cloud_filter_async: the filters of the next frames under the alignment of this one); -1: the main set.
int voxel_filter_device(rsreg_ctx *ctx, const char *d_in, uint32_t N, size_t stride, const float leaf[3], uint32_t *n_out, int side_set)
{
    *n_out = 0;
    if (N == 0) return RSREG_OK;
    const bool side = side_set >= 0;
    rsreg_ctx::SideSet &ss = ctx->side_sets[side ? side_set : 0];
    hipStream_t st = side ? ss.stream : ctx->stream;
    const size_t n = N;
    const float ivx = 1.0f / leaf[0], ivy = 1.0f / leaf[1], ivz = 1.0f / leaf[2];
    // buffers (the main set is shared with the ICP index build: nothing on the main stream overlaps an ICP call in
    // flight; the side set is the side stream's own)
    DevBuf &b_out = side ? ss.out : ctx->d_vox_out, &b_keys = side ? ss.keys : ctx->d_keys,
           &b_keys_alt = side ? ss.keys_alt : ctx->d_keys_alt, &b_vals = side ? ss.vals : ctx->d_vals,
           &b_vals_alt = side ? ss.vals_alt : ctx->d_vals_alt, &b_flags = side ? ss.flags : ctx->d_flags,
           &b_scan = side ? ss.scan : ctx->d_scan, &b_cent = side ? ss.cent : ctx->d_vox_cent,
           &b_misc = side ? ss.misc : ctx->d_misc, &b_tmp = side ? ss.tmp : ctx->d_tmp;
    PinnedBuf &b_host = side ? ss.host : ctx->h_sums;
    RSREG_HIP(ctx, b_out.reserve(n * stride));
    RSREG_HIP(ctx, b_keys.reserve(n * 8));
    RSREG_HIP(ctx, b_keys_alt.reserve(n * 8));
    RSREG_HIP(ctx, b_vals.reserve(n * 8));
    RSREG_HIP(ctx, b_vals_alt.reserve(n * 8));
    RSREG_HIP(ctx, b_flags.reserve(n * 8));
    RSREG_HIP(ctx, b_scan.reserve(n * 8));
    RSREG_HIP(ctx, b_cent.reserve(n * 32));
    RSREG_HIP(ctx, b_misc.reserve(256));
    RSREG_HIP(ctx, b_host.reserve(2048));
    char *d_out = b_out.as<char>();
    uint32_t *keys = b_keys.as<uint32_t>(), *skeys = b_keys_alt.as<uint32_t>();
    uint32_t *vals = b_vals.as<uint32_t>(), *svals = b_vals_alt.as<uint32_t>();
    uint32_t *flag = b_flags.as<uint32_t>(), *rid = b_scan.as<uint32_t>();
    uint32_t *start = keys;                 // keys are dead once sorted
    uint32_t *ekey = vals, *erun = flag;    // vals dead once sorted, flag dead after the starts
    uint32_t *ekey2 = keys + N, *order = vals + N;
    uint32_t *long_runs = rid;              // the run ids are dead once the starts are written
    uint32_t *huge_runs = rid + N;          // (at most N / kLongRun long runs; the buffer holds 2 N words)
    float *cent = b_cent.as<float>();
    uint32_t *stats = b_misc.as<uint32_t>() + 32;
    const uint32_t nb = div_up_u(N, kVBlock);
    // both sorts are the library's own (osort.hpp): the slot sort (10 bits, stable) and, further down, the sort of the runs
    // by emission position (as many bits as n + 512 has).  Their state lies in one scratch block that the keys kernel clears
    // on its way: [slot sort | emission sort, sized for n runs | the scan's sums (oscan.hpp)]
    unsigned ebits = 1;
    while ((1ull << ebits) < (unsigned long long)n + kHist + 1ull) ++ebits;
    const Radix32Plan plan = radix32_plan(n, 0, 10), plan2max = radix32_plan(n, 0, ebits);
    const size_t sort_bytes = (size_t)plan.words * 4, sort2_bytes = (size_t)plan2max.words * 4, scan_bytes = oscan_scratch_bytes<uint32_t>(n);
    const size_t off_sort2 = (sort_bytes + 255) & ~(size_t)255, off_scan = (off_sort2 + sort2_bytes + 255) & ~(size_t)255;
    RSREG_HIP(ctx, b_tmp.reserve(off_scan + scan_bytes + 256));
    char *tmp = b_tmp.as<char>();
    // (the slot sort ends in the pair (skeys, svals): the keys are written into whichever pair that takes)
    uint32_t *k_a = !plan.ends_in_first ? keys : skeys, *v_a = !plan.ends_in_first ? vals : svals;
    uint32_t *k_b = k_a == keys ? skeys : keys, *v_b = v_a == vals ? svals : vals;
    k_vox_keys<<<nb, kVBlock, 0, st>>>(d_in, stride, N, ivx, ivy, ivz, k_a, v_a, stats, b_tmp.as<uint32_t>(), (uint32_t)((off_sort2 + sort2_bytes) / 4));
    RSREG_HIP(ctx, hipGetLastError());
    {
        bool in_first = false;
        RSREG_HIP(ctx, radix32_sort_pairs<uint32_t>(plan, b_tmp.as<uint32_t>(), k_a, k_b, v_a, v_b, n, 0, 10, st, &in_first));
        if ((in_first ? k_a : k_b) != skeys) return fail(ctx, RSREG_ERR_STATE, "osort: the sorted pairs are not where they belong");
    }
    k_vox_flags<<<nb, kVBlock, 0, st>>>(d_in, stride, N, ivx, ivy, ivz, skeys, svals, flag);
    RSREG_HIP(ctx, hipGetLastError());
    RSREG_HIP(ctx, (oscan<uint32_t>(flag, rid, n, 0u, tmp + off_scan, st)));
    k_vox_starts<<<nb, kVBlock, 0, st>>>(skeys, flag, rid, N, start, stats);
    RSREG_HIP(ctx, hipGetLastError());
    uint32_t *h = b_host.as<uint32_t>();
    RSREG_HIP(ctx, hipMemcpyAsync(h, stats, 8, hipMemcpyDeviceToHost, st));
    RSREG_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t nr = h[0];
    if (nr == 0) return RSREG_OK;
    k_vox_runs<<<div_up_u(nr, kVBlock), kVBlock, 0, st>>>(d_in, stride, N, skeys, svals, start, stats, cent, ekey, erun, long_runs, huge_runs);
    RSREG_HIP(ctx, hipGetLastError());
    k_vox_long_runs<<<std::min(div_up_u(nr, 4u), 2048u), kVBlock, 0, st>>>(d_in, stride, N, skeys, svals, start, stats, cent, ekey, erun,
                                                                           long_runs);
    RSREG_HIP(ctx, hipGetLastError());
    if (const uint32_t huge_at_most = (N - std::min(N, nr)) / (kHugeRun - 1u)) {   // (nr runs hold at least one point each)
        k_vox_huge_runs<<<std::min(huge_at_most, 512u), kHB, 0, st>>>(d_in, stride, N, skeys, svals, start, stats, cent, ekey, erun, huge_runs);
        RSREG_HIP(ctx, hipGetLastError());
    }
    const uint32_t *emit_order = order;
    {
        const Radix32Plan plan2 = radix32_plan(nr, 0, ebits);   // (nr <= n: its state fits the block cleared for n runs)
        bool in_first = false;
        RSREG_HIP(ctx, radix32_sort_pairs<uint32_t>(plan2, reinterpret_cast<uint32_t *>(tmp + off_sort2), ekey, ekey2, erun, order, nr, 0, ebits, st, &in_first));
        emit_order = in_first ? erun : order;
    }
    k_vox_emit<<<div_up_u(nr, kVBlock), kVBlock, 0, st>>>(cent, emit_order, stats, stride, d_out);
    RSREG_HIP(ctx, hipGetLastError());
    *n_out = nr;
    return RSREG_OK;
}

}  // namespace rsreg

extern "C" int rsreg_approx_voxel_grid_gpu(rsreg_ctx *ctx, const void *in, size_t n, size_t stride, const float leaf[3], void *out,
                                           size_t *n_out)
{
    if (!ctx || !leaf || !n_out || (n && (!in || !out)) || stride < 20 || (stride & 3)) return RSREG_ERR_INVALID_ARG;
    if (!(leaf[0] > 0) || !(leaf[1] > 0) || !(leaf[2] > 0)) return RSREG_ERR_INVALID_ARG;
    if (n > 0x7ffffff0ull) return fail(ctx, RSREG_ERR_INVALID_ARG, "cloud too large");
    *n_out = 0;
    if (n == 0) return RSREG_OK;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    RSREG_HIP(ctx, ctx->d_vox_in.reserve(n * stride));
    // through pinned staging, copied by a few threads: a pageable hipMemcpy of 10-30 MB is several times slower
    RSREG_HIP(ctx, ctx->h_stage.reserve(n * stride));
    {
        char *stage = ctx->h_stage.as<char>();
        const char *src = static_cast<const char *>(in);
        host_parallel_for(n, [=](size_t lo, size_t hi) { rsreg::stream_copy(stage + lo * stride, src + lo * stride, (hi - lo) * stride); });
    }
    RSREG_HIP(ctx, hipMemcpyAsync(ctx->d_vox_in.ptr, ctx->h_stage.ptr, n * stride, hipMemcpyHostToDevice, st));
    uint32_t nr = 0;
    int rc = voxel_filter_device(ctx, ctx->d_vox_in.as<char>(), (uint32_t)n, stride, leaf, &nr, -1);
    if (rc || nr == 0) return rc;
    RSREG_HIP(ctx, hipMemcpyAsync(ctx->h_stage.ptr, ctx->d_vox_out.ptr, (size_t)nr * stride, hipMemcpyDeviceToHost, st));
    RSREG_HIP(ctx, hipStreamSynchronize(st));
    {
        const char *stage = ctx->h_stage.as<char>();
        char *dst = static_cast<char *>(out);
        host_parallel_for(nr, [=](size_t lo, size_t hi) { std::memcpy(dst + lo * stride, stage + lo * stride, (hi - lo) * stride); });
    }
    *n_out = nr;
    return RSREG_OK;
}
