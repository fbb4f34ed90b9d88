// ndt_kernels.hpp — device code of the NDT path (gfx950, wave64).  Included by ndt.hip only.
//
// Second kernel set of the engine (north star): voxel-grid binning + per-voxel mean and
// covariance of the target (pcl::VoxelGridCovariance::applyFilter) and the
// per-point score / gradient / Hessian pass of pcl::NormalDistributionsTransform
// (computeDerivatives / updateDerivatives / computeHessian, App. A.7).
// Reference call sites: src/ndt_edge_based_registration.hpp:38-43,71-72,83,92.
// All statistics and derivatives are f64 like PCL's; none of this is a dense contraction, so
// MFMA is unused: the pass is f64-VALU work over an LDS-resident voxel table.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>

#include "osort.hpp"
#include <cstdint>

#include "records.hpp"
#include "rsreg_ctx.hpp"

#include "ndt_math.hpp"

namespace rsreg {

constexpr int kNdtBlock = 256;
constexpr int kNdtAcc = 28;        // score + 6 gradient + 21 upper-triangle Hessian entries
constexpr int kNdtVoxChunk = 64;   // voxels staged through LDS per pass over the table (8 KB: many workgroups per CU)
constexpr int kNdtStatBlocks = 1;  // (per voxel) one block reduces one voxel's points

struct NdtVoxel {        // 128 B, device voxel table entry
    double mean[3];
    double icov[9];
    float centroid[3];
    float pad;
    double pad2;
};

struct NdtBinParams {
    float inv_leaf;
    int min_b[3];
    long long mul[3];
};

struct NdtPassParams {
    float M[12];          // rows of the 3x4 pose matrix (f32, like PCL's transformPointCloud)
    double jang[8][3];    // j_ang_a .. j_ang_h
    double hang[15][3];   // h_ang_a2 .. h_ang_f3
    double d1, d2;
    float r2;             // resolution^2 as float (radiusSearch)
    int n_vox;
    int mode;             // 0 score+gradient+Hessian, 1 score+gradient, 2 Hessian only
};

__device__ __forceinline__ bool ndt_finite3(float x, float y, float z)
{
    return isfinite(x) && isfinite(y) && isfinite(z);
}

// leaf key = ijk0 + ijk1*div0 + ijk2*div0*div1, ijk = floor(p*inv_leaf) - min_b (PCL formula)
// (non-finite points get `invalid_key`, one above every leaf's: they sort last, and the sort only has to look at its bits)
template <typename KeyT>
__global__ __launch_bounds__(kNdtBlock) void k_ndt_keys(const char *pts, size_t stride, uint32_t n, NdtBinParams bp, KeyT invalid_key,
                                                        KeyT *keys, uint32_t *vals, uint32_t *sort_scratch, uint32_t sort_scratch_words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (sort_scratch) radix32_clear(sort_scratch, sort_scratch_words, i, gridDim.x * blockDim.x);   // (the state of the sort that follows: radix32.hpp)
    if (i >= n) return;
    const float *p = reinterpret_cast<const float *>(pts + (size_t)i * stride);
    const float x = p[0], y = p[1], z = p[2];
    KeyT key = invalid_key;
    if (ndt_finite3(x, y, z)) {
        const int i0 = (int)(floorf(__fmul_rn(x, bp.inv_leaf)) - (float)bp.min_b[0]);
        const int i1 = (int)(floorf(__fmul_rn(y, bp.inv_leaf)) - (float)bp.min_b[1]);
        const int i2 = (int)(floorf(__fmul_rn(z, bp.inv_leaf)) - (float)bp.min_b[2]);
        key = (KeyT)((long long)i0 * bp.mul[0] + (long long)i1 * bp.mul[1] + (long long)i2 * bp.mul[2]);
    }
    keys[i] = key;
    vals[i] = i;
}

template <typename KeyT>
__global__ __launch_bounds__(kNdtBlock) void k_ndt_flag_starts(const KeyT *keys, uint32_t nfin, uint32_t *start)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    start[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(kNdtBlock) void k_ndt_seg_offsets(const uint32_t *start, const uint32_t *sid, uint32_t nfin,
                                                               uint32_t *seg_begin, uint32_t *counts /*[1]*/)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfin) return;
    if (start[i]) seg_begin[sid[i]] = i;
    if (i == nfin - 1) {
        const uint32_t ns = sid[i] + start[i];
        counts[0] = ns;
        seg_begin[ns] = nfin;
    }
}

// `parts` blocks per voxel (16 when the voxels are few and large, else 1), each over a contiguous share of its points: n, sum p (3), sum p p^T (6 unique) in f64,
// fixed reduction order; the host adds the parts of a voxel in order.  (One block per voxel was 86 us for the two dozen
// 1 m voxels of a 5 x 10^5-point edge target: two dozen blocks on 256 CUs, each gathering 2 x 10^4 records.)
// out[(v * parts + part) * 10 + k]: 0 n, 1..3 sum, 4..9 xx xy xz yy yz zz
__global__ __launch_bounds__(kNdtBlock) void k_ndt_voxel_stats(const uint32_t *vals, const uint32_t *seg_begin,
                                                               const char *pts, size_t stride, uint32_t parts, double *out,
                                                               const uint32_t *n_segments /* launched for more voxels than there may be: those beyond leave */)
{
    const uint32_t v = blockIdx.x / parts, part = blockIdx.x % parts;
    if (n_segments && v >= *n_segments) return;
    const uint32_t vb = seg_begin[v], ve = seg_begin[v + 1];
    const uint32_t share = (ve - vb + parts - 1) / parts;
    const uint32_t b = min(ve, vb + part * share), e = min(ve, b + share);
    double a[9];
    for (int k = 0; k < 9; ++k) a[k] = 0.0;
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const float *p = reinterpret_cast<const float *>(pts + (size_t)vals[i] * stride);
        const double x = p[0], y = p[1], z = p[2];
        a[0] += x; a[1] += y; a[2] += z;
        a[3] += x * x; a[4] += x * y; a[5] += x * z;
        a[6] += y * y; a[7] += y * z; a[8] += z * z;
    }
    __shared__ double sh[kNdtBlock / 64][9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 9; ++k) {
        double s = a[k];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) sh[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double s = sh[0][threadIdx.x];
        for (int w = 1; w < kNdtBlock / 64; ++w) s += sh[w][threadIdx.x];
        out[(size_t)blockIdx.x * 10 + 1 + threadIdx.x] = s;
    }
    if (threadIdx.x == 0) out[(size_t)blockIdx.x * 10] = (double)(e - b);
}

// PCL's voxel centroid (VoxelGridCovariance::applyFilter: leaf.centroid += pt for every point in input order, all in
// float; leaf.centroid /= float(n)): a sequential float chain per voxel and component.  One wave per voxel: the wave
// stages 64 points at a time in LDS (the sort by voxel is stable, so a segment lists its points in input order),
// lanes 0-2 add one component each.  out[v*3 + k] = centroid component k.
__global__ __launch_bounds__(64) void k_ndt_voxel_csum(const uint32_t *vals, const uint32_t *seg_begin, const char *pts, size_t stride,
                                                       float *out)
{
    const uint32_t v = blockIdx.x, lane = threadIdx.x;
    const uint32_t b = seg_begin[v], e = seg_begin[v + 1];
    __shared__ float sh[2][3][64];
    float s = 0.0f;
    int buf = 0;
    if (b + lane < e) {
        const float *p = reinterpret_cast<const float *>(pts + (size_t)vals[b + lane] * stride);
        sh[0][0][lane] = p[0]; sh[0][1][lane] = p[1]; sh[0][2][lane] = p[2];
    }
    for (uint32_t i0 = b; i0 < e; i0 += 64, buf ^= 1) {
        __syncthreads();
        if (i0 + 64 + lane < e) {   // the next chunk while this one is added
            const float *p = reinterpret_cast<const float *>(pts + (size_t)vals[i0 + 64 + lane] * stride);
            sh[buf ^ 1][0][lane] = p[0]; sh[buf ^ 1][1][lane] = p[1]; sh[buf ^ 1][2][lane] = p[2];
        }
        if (lane < 3) {
            const uint32_t cnt = min(64u, e - i0);
            for (uint32_t k = 0; k < cnt; ++k) s = __fadd_rn(s, sh[buf][lane][k]);
        }
    }
    if (lane < 3) out[(size_t)v * 3 + lane] = __fdiv_rn(s, (float)(e - b));
}

__device__ __forceinline__ double dot3d(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// f32 transform of a source record by the pose, PCL's operation order
__device__ __forceinline__ void ndt_transform(const NdtPassParams &pp, const float4 &s4, float &tx, float &ty, float &tz)
{
    tx = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(pp.M[0], s4.x), __fmul_rn(pp.M[1], s4.y)), __fmul_rn(pp.M[2], s4.z)), pp.M[3]);
    ty = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(pp.M[4], s4.x), __fmul_rn(pp.M[5], s4.y)), __fmul_rn(pp.M[6], s4.z)), pp.M[7]);
    tz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(pp.M[8], s4.x), __fmul_rn(pp.M[9], s4.y)), __fmul_rn(pp.M[10], s4.z)), pp.M[11]);
}

// score / gradient / Hessian contribution of one (point, voxel) pair whose centroid passed the radius test
__device__ __forceinline__ void ndt_pair(const NdtPassParams &pp, const float4 &s4, float tx, float ty, float tz,
                                         const NdtVoxel &vx, double *acc)
{
    const double x[3] = {s4.x, s4.y, s4.z};
    // point gradient columns 3..5 (columns 0..2 are the identity)
    const double g13 = dot3d(x, pp.jang[0]), g23 = dot3d(x, pp.jang[1]);
    const double g04 = dot3d(x, pp.jang[2]), g14 = dot3d(x, pp.jang[3]), g24 = dot3d(x, pp.jang[4]);
    const double g05 = dot3d(x, pp.jang[5]), g15 = dot3d(x, pp.jang[6]), g25 = dot3d(x, pp.jang[7]);
    const double J[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, g13, g23}, {g04, g14, g24}, {g05, g15, g25}};
    const double xm[3] = {(double)tx - vx.mean[0], (double)ty - vx.mean[1], (double)tz - vx.mean[2]};
    const double *ci = vx.icov;
    const double cx[3] = {ci[0] * xm[0] + ci[1] * xm[1] + ci[2] * xm[2],
                          ci[3] * xm[0] + ci[4] * xm[1] + ci[5] * xm[2],
                          ci[6] * xm[0] + ci[7] * xm[1] + ci[8] * xm[2]};
    double e = exp(-pp.d2 * dot3d(xm, cx) / 2);
    const double score_inc = -pp.d1 * e;
    e = pp.d2 * e;
    if (e > 1 || e < 0 || e != e) return;
    if (pp.mode != 2) acc[0] += score_inc;
    e *= pp.d1;
    double cg[6][3], xcg[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int r = 0; r < 3; ++r) cg[a][r] = ci[r * 3] * J[a][0] + ci[r * 3 + 1] * J[a][1] + ci[r * 3 + 2] * J[a][2];
        xcg[a] = dot3d(xm, cg[a]);
        if (pp.mode != 2) acc[1 + a] += xcg[a] * e;
    }
    if (pp.mode == 1) return;
    double ha[3], hb[3], hc[3], hd[3], he[3], hf[3];
    ha[0] = 0; ha[1] = dot3d(x, pp.hang[0]); ha[2] = dot3d(x, pp.hang[1]);
    hb[0] = 0; hb[1] = dot3d(x, pp.hang[2]); hb[2] = dot3d(x, pp.hang[3]);
    hc[0] = 0; hc[1] = dot3d(x, pp.hang[4]); hc[2] = dot3d(x, pp.hang[5]);
    hd[0] = dot3d(x, pp.hang[6]); hd[1] = dot3d(x, pp.hang[7]); hd[2] = dot3d(x, pp.hang[8]);
    he[0] = dot3d(x, pp.hang[9]); he[1] = dot3d(x, pp.hang[10]); he[2] = dot3d(x, pp.hang[11]);
    hf[0] = dot3d(x, pp.hang[12]); hf[1] = dot3d(x, pp.hang[13]); hf[2] = dot3d(x, pp.hang[14]);
    int h = 7;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = a; b < 6; ++b) {
            // second derivative of the transformed point w.r.t. (p_a, p_b)
            double xh = 0.0;
            if (a >= 3) {
                const double *hv = (a == 3) ? (b == 3 ? ha : (b == 4 ? hb : hc))
                                 : (a == 4) ? (b == 4 ? hd : he)
                                            : hf;
                // x'^T Sigma^-1 h  (PCL: x_trans.dot(c_inv * block)); Sigma^-1 symmetric
                const double ch[3] = {ci[0] * hv[0] + ci[1] * hv[1] + ci[2] * hv[2],
                                      ci[3] * hv[0] + ci[4] * hv[1] + ci[5] * hv[2],
                                      ci[6] * hv[0] + ci[7] * hv[1] + ci[8] * hv[2]};
                xh = dot3d(xm, ch);
            }
            const double jc = J[b][0] * cg[a][0] + J[b][1] * cg[a][1] + J[b][2] * cg[a][2];
            acc[h++] += e * (-pp.d2 * xcg[a] * xcg[b] + xh + jc);
        }
    }
}

// One derivative pass: transform each source point by the pose (f32, PCL op order), find the
// voxels whose centroid is within the resolution (f32 L2_Simple, strict <), accumulate
// score / gradient / Hessian in f64.  Voxel table chunks are staged through LDS.
// partials[block][28]: 0 score, 1..6 gradient, 7..27 Hessian upper triangle (row-major i<=j)
//
// Work items are (point, voxel) pairs, point-major; each wave takes a contiguous quarter of its workgroup's items.  About
// a third of the pairs pass the radius test, and a wave pays for the f64 derivative block whenever any of its lanes has
// one: a wave first packs the passing pairs into a list in LDS (ballot + prefix count, item order kept) and runs the
// block on 64 listed pairs at a time, every lane busy.  With one point per lane a pass took one point's walk over all
// its voxels (7-14 us of dependent f64 work whether the cloud had 2 k or 36 k points); with pairs over four times as many
// workgroups a lane holds one or two.  Which lane adds which pair depends on n, the pose and the table only, and the
// sums are reduced in the fixed tree below.
// the body of a derivative pass (k_ndt_pass: one pass per launch; k_ndt_line_search: the passes of a whole line search in one
// launch, their sums handed to other workgroups inside it)
template <bool kCoherent>
__device__ __forceinline__ void ndt_pass_body(const float4 *src, uint32_t n, const NdtVoxel *vox, const NdtPassParams &pp, float *trans_out, double *partials)
{
    __shared__ NdtVoxel sv[kNdtVoxChunk];
    __shared__ double sh[kNdtBlock / 64][kNdtAcc];
    __shared__ uint32_t pend[kNdtBlock / 64][128];
    // the workgroup's source points (71 of a 36 k cloud's): every (point, voxel) pair reads its point twice -- for the radius
    // test and again when its derivative block runs -- behind fences and barriers the loads of the next trip cannot pass, so
    // each was a round trip to the L2 on the pass's critical path (round 6: 14.7 -> 11.8 us per pass of the 36 k pair)
    constexpr uint32_t kSrcCap = 512;
    __shared__ float4 s_src[kSrcCap];
    double acc[kNdtAcc];
    for (int k = 0; k < kNdtAcc; ++k) acc[k] = 0.0;

    const uint32_t per_block = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = min(n, blockIdx.x * per_block);
    const uint32_t hi = min(n, lo + per_block);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *list = pend[wave];
    const bool staged = hi - lo <= kSrcCap;   // (uniform; larger clouds read their points where they lie, as before)
    if (staged)
        for (uint32_t k = threadIdx.x; k < hi - lo; k += blockDim.x) s_src[k] = src[lo + k];   // (visible behind the first chunk's barriers)
    auto point = [&](uint32_t pi) -> float4 { return staged ? s_src[pi] : src[lo + pi]; };

    auto run = [&](uint32_t it, int cn, int c0) {
        const uint32_t pi = it / (uint32_t)cn;
        const float4 s4 = point(pi);
        float tx, ty, tz;
        ndt_transform(pp, s4, tx, ty, tz);
        ndt_pair(pp, s4, tx, ty, tz, sv[it - pi * (uint32_t)cn], acc);
    };

    for (int c0 = 0; c0 < pp.n_vox; c0 += kNdtVoxChunk) {
        const int cn = min(kNdtVoxChunk, pp.n_vox - c0);
        __syncthreads();
        {   // cooperative copy of the chunk (16 doubles per voxel)
            const double *g = reinterpret_cast<const double *>(vox + c0);
            double *s = reinterpret_cast<double *>(sv);
            for (int k = threadIdx.x; k < cn * 16; k += blockDim.x) s[k] = g[k];
        }
        __syncthreads();
        const uint32_t items = (hi - lo) * (uint32_t)cn;
        const uint32_t per_wave = (items + kNdtBlock / 64 - 1) / (kNdtBlock / 64);
        const uint32_t w_lo = min(items, (uint32_t)wave * per_wave), w_hi = min(items, w_lo + per_wave);
        uint32_t held = 0;   // listed pairs not yet run (wave-uniform)
        for (uint32_t it0 = w_lo; it0 < w_hi; it0 += 64) {
            const uint32_t it = it0 + lane;
            bool pass = false;
            if (it < w_hi) {
                const uint32_t pi = it / (uint32_t)cn;
                const int v = (int)(it - pi * (uint32_t)cn);
                const uint32_t i = lo + pi;
                const float4 s4 = point(pi);
                float tx, ty, tz;
                ndt_transform(pp, s4, tx, ty, tz);
                if (trans_out && c0 == 0 && v == 0) { trans_out[3 * i] = tx; trans_out[3 * i + 1] = ty; trans_out[3 * i + 2] = tz; }
                const NdtVoxel &vx = sv[v];
                const float dx = __fsub_rn(tx, vx.centroid[0]), dy = __fsub_rn(ty, vx.centroid[1]), dz = __fsub_rn(tz, vx.centroid[2]);
                const float dd = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                pass = s4.w != 0.0f && dd < pp.r2;
            }
            const uint64_t m = __ballot(pass);
            if (pass) list[held + __popcll(m & ((1ull << lane) - 1ull))] = it;
            held += (uint32_t)__popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (held >= 64) {
                run(list[lane], cn, c0);
                const uint32_t moved = list[64 + lane];
                __builtin_amdgcn_wave_barrier();
                list[lane] = moved;
                held -= 64;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if ((uint32_t)lane < held) run(list[lane], cn, c0);
        __builtin_amdgcn_wave_barrier();
    }
    // 28 sums over the 64 lanes by recursive halving (records.hpp: halve_sums): at each of the six steps a lane keeps half
    // of the sums it still holds and hands the other half to its partner, 14 + 7 + 4 + 2 + 1 + 1 = 29 doubles moved
    // instead of 28 x 6 = 168 for 28 separate shuffle trees -- which were most of what a pass took (24 us whether the
    // cloud had 2 k or 36 k points).  The tree is fixed: the result depends on n and the voxel table only.
    {
        int base = 0, cnt = kNdtAcc;
        double v14[14], v7[7], v4[4], v2[2], v1[1], v0[1];
        rsreg::halve_sums<kNdtAcc>(acc, v14, lane, 32, base, cnt);
        rsreg::halve_sums<14>(v14, v7, lane, 16, base, cnt);
        rsreg::halve_sums<7>(v7, v4, lane, 8, base, cnt);
        rsreg::halve_sums<4>(v4, v2, lane, 4, base, cnt);
        rsreg::halve_sums<2>(v2, v1, lane, 2, base, cnt);
        rsreg::halve_sums<1>(v1, v0, lane, 1, base, cnt);
        if (cnt >= 1) sh[wave][base] = v0[0];   // exactly one lane ends up owning each of the 28 sums
    }
    __syncthreads();
    if (threadIdx.x < kNdtAcc) {
        double s = sh[0][threadIdx.x];
        for (int w = 1; w < kNdtBlock / 64; ++w) s += sh[w][threadIdx.x];
        if (kCoherent) {   // (read by workgroups of other XCDs inside the same launch: written through)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(partials) + (size_t)blockIdx.x * kNdtAcc + threadIdx.x,
                               (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            partials[(size_t)blockIdx.x * kNdtAcc + threadIdx.x] = s;
        }
    }
}

__global__ __launch_bounds__(kNdtBlock) void k_ndt_pass(const float4 *src, uint32_t n, const NdtVoxel *vox, NdtPassParams pp,
                                                        float *trans_out, double *partials)
{
    ndt_pass_body<false>(src, n, vox, pp, trans_out, partials);
}

// The pass and its final reduce in ONE launch (round 6; RSREG_NDT_ONE_LAUNCH=1 -- measured SLOWER, 0.72-0.74 against 0.57-0.61 ms per
// alignment of 17 passes, so the launch pair stays the default): the workgroup that finishes last adds the slabs up.  Every workgroup
// writes its 28 sums through (agent-scope stores), waits for their acknowledgement and takes a ticket; the last one -- 256
// threads -- loads slab t and slab t + 256 of all 28 sums (56 loads in flight per thread), adds them in k_ndt_final_reduce's
// order ((0 + x[t]) + x[t + 256] per thread, the shuffle tree over a wave, the waves in order) and writes the sums and the pass
// number where the host watches for it.  The 28 shuffle trees of 256 threads are ONE recursive-halving exchange
// (records.hpp: halve_sums pairs lane l with l + 32, then l with l + 16, ... exactly as `v += __shfl_down(v, off)` does, and
// an addition does not care which of the two lanes performs it): the same tree, hence the same bits as the two-launch form
// (k_ndt_pass + k_ndt_final_reduce, the default) -- with 28 x 6 shuffles replaced by 29 moved doubles.
// (Round 2 measured a last-workgroup reduce slower with release fences and sum-by-sum adds; this form -- no fence, one exchange --
// is slower too: every workgroup's written-through sums with their acknowledgement, 512 tickets on one word, and a tail in which ONE
// workgroup fetches 114 KB past the L2s cost more than a second launch of 28 workgroups behind a kernel boundary.)  Requires gridDim.x == 2 * kNdtBlock (the fixed 512 workgroups of a pass).
__global__ __launch_bounds__(kNdtBlock) void k_ndt_pass_reduce(const float4 *src, uint32_t n, const NdtVoxel *vox, NdtPassParams pp, float *trans_out,
                                                               double *partials, double *out, uint32_t *ticket, uint64_t *flag, uint64_t seq)
{
    ndt_pass_body<true>(src, n, vox, pp, trans_out, partials);
    __shared__ uint32_t s_last;
    __shared__ double sh2[kNdtBlock / 64][kNdtAcc];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this workgroup's partial sums have arrived device-wide)
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[kNdtAcc];
    {
        const unsigned long long *p0 = reinterpret_cast<const unsigned long long *>(partials) + (size_t)threadIdx.x * kNdtAcc;
        const unsigned long long *p1 = p0 + (size_t)kNdtBlock * kNdtAcc;
        unsigned long long x0[kNdtAcc], x1[kNdtAcc];
#pragma unroll
        for (int k = 0; k < kNdtAcc; ++k) x0[k] = __hip_atomic_load(const_cast<unsigned long long *>(p0 + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < kNdtAcc; ++k) x1[k] = __hip_atomic_load(const_cast<unsigned long long *>(p1 + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < kNdtAcc; ++k) {
            double v = 0.0;
            v += __longlong_as_double((long long)x0[k]);
            v += __longlong_as_double((long long)x1[k]);
            acc[k] = v;
        }
    }
    {
        int base = 0, cnt = kNdtAcc;
        double v14[14], v7[7], v4[4], v2[2], v1[1], v0[1];
        rsreg::halve_sums<kNdtAcc>(acc, v14, lane, 32, base, cnt);
        rsreg::halve_sums<14>(v14, v7, lane, 16, base, cnt);
        rsreg::halve_sums<7>(v7, v4, lane, 8, base, cnt);
        rsreg::halve_sums<4>(v4, v2, lane, 4, base, cnt);
        rsreg::halve_sums<2>(v2, v1, lane, 2, base, cnt);
        rsreg::halve_sums<1>(v1, v0, lane, 1, base, cnt);
        if (cnt >= 1) sh2[wave][base] = v0[0];
    }
    __syncthreads();
    if (threadIdx.x < kNdtAcc) {
        double t = sh2[0][threadIdx.x];
        for (int w = 1; w < kNdtBlock / 64; ++w) t += sh2[w][threadIdx.x];
        out[threadIdx.x] = t;
        __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *ticket = 0;
        if (flag) {
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// One workgroup per sum: fixed order (each thread its share of the partials in ascending order, shuffle tree, waves in
// order).  With a flag (one GPU, sums written straight to pinned host memory) the workgroup that finishes last stamps the
// pass number behind the sums, and the host watches for it instead of going through a stream synchronisation.
// (Finishing the sums inside k_ndt_pass -- last workgroup, or last of each 32 and then the last group -- made a pass
// longer, not shorter: every hand-over between workgroups is a trip to memory and back, profiles/r02_experiments.)
__global__ __launch_bounds__(kNdtBlock) void k_ndt_final_reduce(const double *partials, uint32_t nblocks, double *out,
                                                                uint32_t *ticket, uint64_t *flag, uint64_t seq)
{
    __shared__ double shw[kNdtBlock / 64];
    const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v = 0.0;
    for (uint32_t b = threadIdx.x; b < nblocks; b += kNdtBlock * 4) {   // four loads in flight, added in order
        double x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = b + j * kNdtBlock < nblocks ? partials[(size_t)(b + j * kNdtBlock) * kNdtAcc + k] : 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (b + j * kNdtBlock < nblocks) v += x[j];
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) shw[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = shw[0];
        for (int w = 1; w < kNdtBlock / 64; ++w) t += shw[w];
        out[k] = t;
        if (flag) {
            __threadfence_system();
            if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
                *ticket = 0;
                __threadfence_system();
                __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---- a whole line search in one launch ---------------------------------------------------------------------------------
// computeStepLengthMT evaluates the score and its gradient at a sequence of trial steps, each a derivative pass over the
// source cloud, each depending on the one before.  With a launch pair and the host between any two of them a pass is 19 us
// of kernels and 17 us of round trip (17 passes per alignment of a 36 k-point edge cloud).  k_ndt_line_search runs ALL the
// passes of one line search: its 512 workgroups are resident together (2 per CU), a pass ends in a grid-wide count
// (16 group counters and one on top: same-address atomics queue up), workgroups 0 .. 27 then add one sum each over the 512
// partials -- the order k_ndt_final_reduce adds them in --, workgroup 0's first thread takes the 28 sums through the state
// machine of ndt_math.hpp (the host loop's own source: same bits) and either leaves the next pass's pose, matrix and angle
// terms and lets the workgroups go on, or leaves the search's outcome in pinned host memory and lets them end.  What the
// workgroups hand each other inside the launch is written through and read past the XCDs' L2s (agent-scope atomics); every
// wait is bounded (a workgroup that never comes would otherwise hang the queue): on a time-out the launch ends with
// `error` set and the host runs the search pass by pass.
struct NdtLsCtl {
    NdtLs ls;                       // the state machine (workgroup 0's)
    NdtPassParams pp;               // the pass every workgroup runs next
    unsigned long long release;     // passes the workgroups may run so far (from 1 on) | kNdtLsStop: end
    uint32_t arrive[16 + 1];        // workgroups that have finished pass k, by group, and groups complete (both keep counting)
    uint32_t reduced;               // sums added up (keeps counting)
    uint32_t error;
    uint32_t pad;
    double out[kNdtAcc];
};
constexpr unsigned long long kNdtLsStop = 1ull << 63;
constexpr uint32_t kNdtLsSpins = 1u << 22;   // bounded waits: ~1 s of s_sleep

__host__ __device__ inline void ndt_fill_pass(NdtPassParams &pp, const NdtLs &ls)
{
    const Mat4f M = ndt_pose_matrix(ls.x_t);
    for (int row = 0; row < 3; ++row)
        for (int c = 0; c < 4; ++c) pp.M[row * 4 + c] = M(row, c);
    ndt_angle_terms(ls.x_t, pp.jang, pp.hang);
    pp.mode = ls.next_mode;
}

__device__ __forceinline__ unsigned long long ndt_load64(const void *p)
{
    return __hip_atomic_load(reinterpret_cast<unsigned long long *>(const_cast<void *>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ndt_store64(void *p, unsigned long long v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ndt_load32(const uint32_t *p)
{
    return __hip_atomic_load(const_cast<uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kNdtBlock) void k_ndt_line_search(const float4 *src, uint32_t n, const NdtVoxel *vox, NdtLsCtl *ctl, float *trans_out,
                                                               double *partials, NdtLs *host_out, uint64_t *host_flag, uint64_t seq)
{
    __shared__ NdtPassParams s_pp;
    __shared__ NdtLs s_ls;
    __shared__ double s_sums[kNdtAcc];
    __shared__ double s_shw[kNdtBlock / 64];
    __shared__ unsigned long long s_cmd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per_group = gridDim.x / 16u;
    if (blockIdx.x == 0) {   // the state the host has left
        const unsigned long long *g = reinterpret_cast<const unsigned long long *>(&ctl->ls);
        unsigned long long *l = reinterpret_cast<unsigned long long *>(&s_ls);
        for (uint32_t w = threadIdx.x; w < sizeof(NdtLs) / 8; w += kNdtBlock) l[w] = g[w];
    }
    for (uint32_t pass = 1;; ++pass) {
        if (threadIdx.x == 0) {
            unsigned long long c = 0;
            for (uint32_t spin = 0; spin < kNdtLsSpins; ++spin) {
                c = ndt_load64(&ctl->release);
                if ((c & kNdtLsStop) || c >= pass) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!(c & kNdtLsStop) && c < pass) { c = kNdtLsStop; atomicExch(&ctl->error, 1u); }
            s_cmd = c;
        }
        __syncthreads();
        if (s_cmd & kNdtLsStop) return;
        {
            unsigned long long *l = reinterpret_cast<unsigned long long *>(&s_pp);
            for (uint32_t w = threadIdx.x; w < sizeof(NdtPassParams) / 8; w += kNdtBlock) l[w] = ndt_load64(reinterpret_cast<const unsigned long long *>(&ctl->pp) + w);
        }
        __syncthreads();
        ndt_pass_body<true>(src, n, vox, s_pp, s_pp.mode != 2 ? trans_out : nullptr, partials);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (this workgroup's partial sums have arrived device-wide)
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t before = __hip_atomic_fetch_add(&ctl->arrive[blockIdx.x & 15u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((before + 1u) % per_group == 0u) __hip_atomic_fetch_add(&ctl->arrive[16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (blockIdx.x < (uint32_t)kNdtAcc) {
            // one sum: k_ndt_final_reduce's order (each thread its partials in ascending order, shuffle tree, waves in order)
            if (threadIdx.x == 0) {
                uint32_t spin = 0;
                while (ndt_load32(&ctl->arrive[16]) < 16u * pass && ++spin < kNdtLsSpins) __builtin_amdgcn_s_sleep(1);
                s_cmd = spin < kNdtLsSpins ? 0ull : kNdtLsStop;
            }
            __syncthreads();
            if (s_cmd & kNdtLsStop) {   // (a workgroup never came: end the launch)
                if (threadIdx.x == 0) { atomicExch(&ctl->error, 1u); ndt_store64(&ctl->release, kNdtLsStop); }
                return;
            }
            const int k = (int)blockIdx.x;
            double v = 0.0;
            for (uint32_t b = threadIdx.x; b < gridDim.x; b += kNdtBlock * 4) {
                double x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    x[j] = b + j * kNdtBlock < gridDim.x ? __longlong_as_double((long long)ndt_load64(partials + (size_t)(b + j * kNdtBlock) * kNdtAcc + k)) : 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (b + j * kNdtBlock < gridDim.x) v += x[j];
            }
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) s_shw[wave] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = s_shw[0];
                for (int w = 1; w < kNdtBlock / 64; ++w) t += s_shw[w];
                ndt_store64(&ctl->out[k], (unsigned long long)__double_as_longlong(t));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&ctl->reduced, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (blockIdx.x == 0 && wave == 0) {
            // the controller (workgroup 0's first wave): the 28 sums through the state machine, then the next pass -- or the
            // outcome.  Its lanes fetch the sums and send the next pass's parameters side by side (one thread doing either word
            // by word waits for every trip to memory in turn); lane 0 alone runs the state machine.
            uint32_t spin = 0;
            if (lane == 0)
                while (ndt_load32(&ctl->reduced) < (uint32_t)kNdtAcc * pass && ++spin < kNdtLsSpins) __builtin_amdgcn_s_sleep(1);
            spin = __shfl(spin, 0);
            if (spin >= kNdtLsSpins) {
                if (lane == 0) { atomicExch(&ctl->error, 1u); ndt_store64(&ctl->release, kNdtLsStop); }
            } else {
                if (lane < kNdtAcc) s_sums[lane] = __longlong_as_double((long long)ndt_load64(&ctl->out[lane]));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                int done = 0;
                if (lane == 0) {
                    ndt_ls_consume(s_ls, s_sums);
                    done = s_ls.phase == kNdtLsDone ? 1 : 0;
                    if (!done) ndt_fill_pass(s_pp, s_ls);   // (d1, d2, r2, n_vox stay)
                }
                done = __shfl(done, 0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (!done) {
                    const unsigned long long *l = reinterpret_cast<const unsigned long long *>(&s_pp);
                    for (uint32_t w = lane; w < sizeof(NdtPassParams) / 8; w += 64u) ndt_store64(reinterpret_cast<unsigned long long *>(&ctl->pp) + w, l[w]);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) ndt_store64(&ctl->release, (unsigned long long)pass + 1ull);
                } else {
                    const unsigned long long *l = reinterpret_cast<const unsigned long long *>(&s_ls);
                    unsigned long long *h = reinterpret_cast<unsigned long long *>(host_out);
                    for (uint32_t w = lane; w < sizeof(NdtLs) / 8; w += 64u) h[w] = l[w];
                    __threadfence_system();
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) {
                        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        ndt_store64(&ctl->release, kNdtLsStop);
                    }
                }
            }
        }
        __syncthreads();   // (workgroup 0: s_pp is the controller's scratch until here)
    }
}

__global__ __launch_bounds__(kNdtBlock) void k_ndt_load_source(const char *raw, size_t stride, uint32_t n, float4 *src)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = reinterpret_cast<const float *>(raw + (size_t)i * stride);
    const float x = p[0], y = p[1], z = p[2];
    src[i] = make_float4(x, y, z, ndt_finite3(x, y, z) ? 1.0f : 0.0f);
}

// the aligned cloud of ndt.align() as records in HBM: the source record, xyz at the last evaluated pose
// (a non-finite point keeps its coordinates), data[3] = 1; in and out may be the same array
__global__ __launch_bounds__(kNdtBlock) void k_ndt_write_aligned(const char *in, size_t stride, uint32_t n, const float *xyz, char *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(in + (size_t)i * stride);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)i * stride);
    const float x = __uint_as_float(src[0]), y = __uint_as_float(src[1]), z = __uint_as_float(src[2]);
    const bool ok = ndt_finite3(x, y, z);
    if (out != in)
        for (uint32_t k = 3; k < stride / 4; ++k) dst[k] = src[k];
    dst[0] = ok ? __float_as_uint(xyz[3 * i]) : src[0];
    dst[1] = ok ? __float_as_uint(xyz[3 * i + 1]) : src[1];
    dst[2] = ok ? __float_as_uint(xyz[3 * i + 2]) : src[2];
    if (stride >= 16) dst[3] = __float_as_uint(1.0f);
}

}  // namespace rsreg
