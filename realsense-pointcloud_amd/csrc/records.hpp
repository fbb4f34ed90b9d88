// records.hpp — small device helpers shared by the translation units that touch point records
// (icp.hip through icp_kernels.hpp, cloud.hip): the 3x4 transform in PCL's operation order, record access.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "rsreg_ctx.hpp"

namespace rsreg {

constexpr int kBlock = 256;            // 4 waves of 64

struct Mat34 {  // rows of the 3x4 part of a column-major Mat4f, passed by value to kernels
    float r0[4], r1[4], r2[4];
};

RSREG_HD inline Mat34 to_mat34(const Mat4f &T)
{
    Mat34 m;
    for (int c = 0; c < 4; ++c) { m.r0[c] = T(0, c); m.r1[c] = T(1, c); m.r2[c] = T(2, c); }
    return m;
}

__device__ __forceinline__ float3 xform(const Mat34 &m, float x, float y, float z)
{
    float ox = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m.r0[0], x), __fmul_rn(m.r0[1], y)), __fmul_rn(m.r0[2], z)), m.r0[3]);
    float oy = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m.r1[0], x), __fmul_rn(m.r1[1], y)), __fmul_rn(m.r1[2], z)), m.r1[3]);
    float oz = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m.r2[0], x), __fmul_rn(m.r2[1], y)), __fmul_rn(m.r2[2], z)), m.r2[3]);
    return make_float3(ox, oy, oz);
}

__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return isfinite(x) && isfinite(y) && isfinite(z);
}

__device__ __forceinline__ const float *rec_xyz(const char *base, size_t stride, size_t i)
{
    return reinterpret_cast<const float *>(base + i * stride);
}

// A record of the sorted target array (both index kinds): x, y, ORIGINAL INDEX, z.  The index sits in the third
// word so that a 128-bit load leaves (index, z) in an aligned register pair: the dense search overwrites z with the
// squared distance and compares (distance, index) as one 64-bit key without moving anything.
__device__ __forceinline__ float4 tgt_rec(float x, float y, float z, uint32_t idx) { return make_float4(x, y, __uint_as_float(idx), z); }
__device__ __forceinline__ float tgt_z(const float4 &t) { return t.w; }
__device__ __forceinline__ uint32_t tgt_idx(const float4 &t) { return __float_as_uint(t.z); }

// one step of the recursive-halving reduction (tile_reduce_store): the lane holds the sums
// [base, base + cnt) in v[0..N); it keeps the lower or the upper half according to its bit
// `mask` and adds its partner's copy of that half
// gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper half (the odd 16-lane rows) of one register with the
// lower half (the even rows) of another: exactly the "keep one half of the sums, hand the other half to the partner"
// of a halving step across lane bit 5 / bit 4.  After the swap a = {own lower-half sum in the lanes that keep it | the
// partner's upper-half sum} and b the other two, so a + b is the step's result in every lane -- the same two operands
// as `mine + partner's`, without the selects and the cross-lane shuffle.
template <int kMask>
__device__ __forceinline__ void swap_lane_halves(double &a, double &b)
{
    const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);
    const uint32_t al = (uint32_t)ua, ah = (uint32_t)(ua >> 32), bl = (uint32_t)ub, bh = (uint32_t)(ub >> 32);
    const auto r0 = kMask == 32 ? __builtin_amdgcn_permlane32_swap(al, bl, false, false) : __builtin_amdgcn_permlane16_swap(al, bl, false, false);
    const auto r1 = kMask == 32 ? __builtin_amdgcn_permlane32_swap(ah, bh, false, false) : __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
    a = __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0]));
    b = __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
}

template <int N>
__device__ __forceinline__ void halve_sums(const double (&v)[N], double (&out)[(N + 1) / 2], int lane, int mask, int &base, int &cnt)
{
    constexpr int H = (N + 1) / 2;
    const bool up = (lane & mask) != 0;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        double lo = v[j], hi = (H + j < N) ? v[H + j] : 0.0;
        if (mask == 32 || mask == 16) {   // (compile-time after inlining: tile_reduce_store passes literals)
            if (mask == 32) swap_lane_halves<32>(lo, hi); else swap_lane_halves<16>(lo, hi);
            out[j] = lo + hi;   // = mine + the partner's (an IEEE sum does not depend on the order of its two operands)
            continue;
        }
        const double mine = up ? hi : lo, send = up ? lo : hi;
        out[j] = mine + __shfl_xor(send, mask);
    }
    base = up ? base + H : base;
    cnt = up ? max(cnt - H, 0) : min(cnt, H);
}

}  // namespace rsreg
