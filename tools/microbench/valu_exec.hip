// valu_exec.hip — does a VALU wave-instruction cost a gfx950 SIMD less when part of the wave is masked off?
// k_icp_fused_dense runs at a lane utilisation of 0.36-0.38 (a wave pays for its slowest lane): if the SIMD skipped the
// 16-lane groups of a wave64 instruction whose lanes are all inactive, packing the busy lanes of a wave together would
// cut its issue time; if every instruction takes its four cycles whatever EXEC says, only fewer instructions help.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_exec.hip -o tools/_build/valu_exec
// Every SIMD holds 8 waves; each runs kIters trips of 64 v_add_f32 / v_pk_mul_f32 / v_cmp+v_cndmask over 8 independent
// chains under the given EXEC mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kIters = 4000;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ __launch_bounds__(1024) void k_exec(float *out, float seed, unsigned long long mask)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
    const float c = seed * 1.0001f;
    const unsigned lane = threadIdx.x & 63u;
    if ((mask >> lane) & 1ull) {
#pragma unroll 1
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz; 8 waves per SIMD, %d x 64 v_add_f32 per wave\n", prop.name, cus, ghz, kIters);
    float *out;
    const int blocks = cus * 2;   // 2 x 1024 threads per CU = 32 waves = 8 per SIMD
    CHECK(hipMalloc(&out, (size_t)blocks * 1024 * 4));
    struct Case { const char *name; unsigned long long mask; };
    const Case cases[] = {
        {"all 64 lanes", ~0ull},
        {"lanes 0-47", (1ull << 48) - 1},
        {"lanes 0-31", (1ull << 32) - 1},
        {"lanes 0-15", (1ull << 16) - 1},
        {"lanes 16-31", ((1ull << 16) - 1) << 16},
        {"lanes 0-7", 0xffull},
        {"lane 0", 1ull},
        {"every 4th lane (16 lanes, one in each group of 4)", 0x1111111111111111ull},
        {"lanes 0-15 and 32-47", 0x0000ffff0000ffffull},
        {"lanes 0-3 of every 16", 0x000f000f000f000full},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Case &c : cases) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            k_exec<<<blocks, 1024>>>(out, 1.5f, c.mask);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double insts_per_simd = 8.0 * kIters * 64;   // wave-instructions one SIMD issues
        printf("  %-52s %8.3f ms   %.2f ns = %.2f clocks per wave-instruction per SIMD\n", c.name, best, best * 1e6 / insts_per_simd,
               best * 1e6 / insts_per_simd * ghz);
    }
    return 0;
}
