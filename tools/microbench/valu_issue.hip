// valu_issue.hip — what one VALU wave-instruction costs a gfx950 SIMD, by instruction kind and by how many waves
// share the SIMD.  Settles the bound k_icp_fused_dense is priced against.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue.hip -o tools/_build/valu_issue
// Each wave runs kIters trips of a 64-instruction unrolled body over 8 independent register chains and stamps
// s_memtime (core clock) around the loop; per SIMD cost = wave cycles * 1 / (waves on the SIMD * instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kIters = 2000;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int kOp>
__global__ __launch_bounds__(1024) void k_issue(unsigned long long *out, float seed)
{
    float a[8], b[8];
    unsigned long long k[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; b[i] = seed * 0.5f + i; k[i] = threadIdx.x * 77ull + i; }
    float c = seed * 1.0001f;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v qxy = {seed * 3.0f, seed * 5.0f};
    float bd = 1e30f;
    unsigned bi = 0xffffffffu;
    unsigned int sel = 0;
    __builtin_amdgcn_s_barrier();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (kOp == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 2) {
#define X(i) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[i], b[i]}; f2 cc = {c, c}; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(cc)); a[i] = v.x; b[i] = v.y; }
                REP8(X)
#undef X
            } else if (kOp == 3) {
#define X(i) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[i], b[i]}; f2 cc = {c, c}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(cc)); a[i] = v.x; b[i] = v.y; }
                REP8(X)
#undef X
            } else if (kOp == 4) {   // 64-bit compare into vcc + one select (the tie-break of dconsider)
#define X(i) asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(k[i]), "v"(k[(i + 1) & 7]), "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (kOp == 5) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 6) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b[i]), "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (kOp == 7) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(sel) : "v"(threadIdx.x)); asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if (kOp == 8) {   // f64 add (the 17 sums)
#define X(i) { double d = __longlong_as_double((long long)k[i]); asm volatile("v_add_f64 %0, %0, %0" : "+v"(d)); k[i] = (unsigned long long)__double_as_longlong(d); }
                REP8(X)
#undef X
            } else if (kOp == 9) {   // plain scalar chain beside nothing: s_add_u32
#define X(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sel));
                REP8(X)
#undef X
            } else if (kOp == 10) {  // v_min3_f32
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                REP8(X)
#undef X
            } else if (kOp == 12) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 13) {
#define X(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 14) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 15) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (kOp == 16) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 17) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                REP8(X)
#undef X
            } else if (kOp == 18) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 19) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (kOp == 20) {
#define X(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(k[i]), "v"(k[(i + 1) & 7]) : "vcc");
                REP8(X)
#undef X
            } else if (kOp == 21) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
                REP8(X)
#undef X
            } else if (kOp == 22) {   // one candidate as icp_dense.hpp: dconsider scores it (packed x/y, 64-bit key): 9 VALU
#define X(i) { typedef float f2 __attribute__((ext_vector_type(2))); \
               asm volatile("" : "+v"(a[i]), "+v"(b[i]), "+v"(k[i]));   /* opaque: a fresh candidate every time */ \
               const f2 txy = {a[i], b[i]}; const f2 dxy = qxy - txy; const f2 sq = dxy * dxy; \
               const float dz = __fsub_rn(c, __uint_as_float((unsigned)(k[i] >> 32))); \
               const float d = __fadd_rn(__fadd_rn(sq.x, sq.y), __fmul_rn(dz, dz)); \
               const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)k[i]; \
               const unsigned long long bkey = ((unsigned long long)__float_as_uint(bd) << 32) | bi; \
               const bool better = key < bkey; bd = better ? d : bd; bi = better ? (unsigned)k[i] : bi; }
                REP8(X)
#undef X
            } else if (kOp == 23) {   // the same candidate with scalar f32 ops and a float compare (no tie-break): 11 VALU
#define X(i) { asm volatile("" : "+v"(a[i]), "+v"(b[i]), "+v"(k[i])); \
               const float dx = __fsub_rn(qxy.x, a[i]), dy = __fsub_rn(qxy.y, b[i]), dz = __fsub_rn(c, __uint_as_float((unsigned)(k[i] >> 32))); \
               const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)); \
               const bool better = d < bd; bd = better ? d : bd; bi = better ? (unsigned)k[i] : bi; }
                REP8(X)
#undef X
            } else if (kOp == 11) {  // v_pk_fma_f32
#define X(i) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[i], b[i]}; f2 cc = {c, c}; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(cc)); a[i] = v.x; b[i] = v.y; }
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i] + (float)k[i];
    if (s + bd + (float)bi == 123.456f) out[0] = sel;   // keep everything alive
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1 + 256 * 32 * 2] = r1 - r0;
}

static const char *kNames[] = {"v_add_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_cmp_lt_u64+v_cndmask", "v_min_f32",
                               "v_cmp_lt_f32+v_cndmask", "v_add_u32+v_lshlrev", "v_add_f64", "s_add_u32", "v_min3_f32", "v_pk_fma_f32", "v_mul_f32", "v_sub_f32", "v_fmac_f32", "v_cndmask_b32", "v_and_b32", "v_mov_b32", "v_max_f32", "v_add_u32", "v_cmp_lt_u64", "v_cmp_lt_f32", "CANDIDATE packed+u64key (9 VALU)", "CANDIDATE scalar+f32cmp (11 VALU)"};
static const int kInstPerRep[] = {1, 1, 1, 1, 2, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

template <int kOp>
void run(unsigned long long *d_out, int waves_per_simd)
{
    // waves_per_simd waves on each of the 4 SIMDs of every CU: blocks of 256 * min(w, 4) threads, w / min(w, 4) blocks per CU
    const int per_block = std::min(waves_per_simd, 4), blocks_per_cu = waves_per_simd / per_block;
    const int threads = 256 * per_block, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_issue<kOp><<<grid, threads>>>(d_out, 1.0f);   // warm-up
    CHECK(hipEventRecord(e0));
    k_issue<kOp><<<grid, threads>>>(d_out, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int n_waves = grid * threads / 64;
    std::vector<unsigned long long> h(1 + n_waves);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long real = 0;   // 100 MHz ticks of wave 0 over the same loop
    CHECK(hipMemcpy(&real, d_out + 1 + 256 * 32 * 2, 8, hipMemcpyDeviceToHost));
    const double ghz = (double)h[1] / ((double)real * 10.0);   // s_memtime ticks per ns (block 0 wave 0 is h[1] before the sort)
    std::sort(h.begin() + 1, h.end());
    const double med = (double)h[1 + n_waves / 2], mx = (double)h[n_waves];
    const double insts = (double)kIters * 64 * kInstPerRep[kOp];
    // s_memtime ticks at the 100 MHz reference on gfx950?  Report raw ticks per instruction AND time-based cycles at 2.4 GHz.
    printf("%-32s waves/SIMD %d: s_memtime ticks per wave-inst median %.3f max %.3f (%.3f ticks/ns) | per SIMD %.3f ticks/inst | kernel %.1f us = %.3f ns per inst per SIMD\n",
           kNames[kOp], waves_per_simd, med / insts, mx / insts, ghz, med / insts / waves_per_simd, ms * 1e3,
           ms * 1e6 / (insts * waves_per_simd));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main()
{
    unsigned long long *d_out;
    CHECK(hipMalloc(&d_out, 8 * (2 + 256 * 32 * 2)));
    const int ws[] = {1, 2, 4, 8};
    for (int w : ws) {
        run<0>(d_out, w); run<12>(d_out, w); run<13>(d_out, w); run<1>(d_out, w); run<14>(d_out, w); run<2>(d_out, w); run<3>(d_out, w); run<11>(d_out, w);
        run<5>(d_out, w); run<18>(d_out, w); run<10>(d_out, w); run<15>(d_out, w); run<16>(d_out, w); run<17>(d_out, w); run<19>(d_out, w); run<20>(d_out, w); run<21>(d_out, w);
        run<4>(d_out, w); run<6>(d_out, w); run<7>(d_out, w); run<8>(d_out, w); run<9>(d_out, w); run<22>(d_out, w); run<23>(d_out, w);
    }
    return 0;
}
