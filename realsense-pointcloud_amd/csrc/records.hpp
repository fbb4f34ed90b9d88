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

}  // namespace rsreg
