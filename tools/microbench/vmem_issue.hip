// vmem_issue.hip — what one vector-memory wave-instruction (a raw buffer load, the search kernel's kind) costs a gfx950 CU,
// by load width (4 / 8 / 12 / 16 bytes per lane), by how many DISTINCT addresses the 64 lanes ask for (1 / 4 / 16 / 64
// groups of lanes, each group one address in a line of its own) and by where the lines live (a working set that fits the
// CU's L1, an XCD's L2, the Infinity Cache).  Second part: the latency of one dependent load (pointer chase) at the same
// three working sets, one wave alone and with the CU full.
// Settles what a narrower candidate record can buy k_icp_fused_dense.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/vmem_issue.hip -o tools/_build/vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kIters = 512;     // trips; 8 loads in flight per trip
constexpr int kBlock = 256;     // 4 waves; 8 workgroups per CU fill 8 waves per SIMD

template <int W>
__device__ __forceinline__ unsigned load_w(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    if (W == 4) return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    if (W == 8) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); return v.x ^ v.y; }
    if (W == 12) { const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0); return v.x ^ v.y ^ v.z; }
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return v.x ^ v.y ^ v.z ^ v.w;
}

// G groups of 64 / G lanes; every group asks for one address per load, in a 128-byte line of its own (random in the
// working set); kActive: how many lanes of the wave take part at all (the others are masked off)
template <int W, int G, int kActive>
__global__ __launch_bounds__(kBlock) void k_tp(const unsigned *buf, unsigned ws_mask, unsigned *out)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(buf), 0, ws_mask + 1u + 64u, 0x00020000);
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned grp = lane / (64 / G);
    unsigned a = (wave * 0x9E3779B1u + grp * 0x85EBCA6Bu) | 1u;
    unsigned acc = 0;
    if (lane < (unsigned)kActive) {
#pragma unroll 1
        for (int it = 0; it < kIters; ++it) {
            unsigned o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a = a * 1664525u + 1013904223u;
                o[u] = (a >> 4) & ws_mask & ~127u;
            }
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load_w<W>(r, o[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

// the search kernel's own shape: every lane reads 4 consecutive 16-byte records (one 64-byte chunk) with 4 instructions;
// G groups of lanes, one chunk per group
template <int G>
__global__ __launch_bounds__(kBlock) void k_tp_chunk(const unsigned *buf, unsigned ws_mask, unsigned *out)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(buf), 0, ws_mask + 1u + 64u, 0x00020000);
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const unsigned grp = lane / (64 / G);
    unsigned a = (wave * 0x9E3779B1u + grp * 0x85EBCA6Bu) | 1u;
    unsigned acc = 0;
#pragma unroll 1
    for (int it = 0; it < kIters; ++it) {
        unsigned o[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            a = a * 1664525u + 1013904223u;
            o[u] = (a >> 4) & ws_mask & ~63u;
        }
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = load_w<16>(r, o[u >> 2] + 16u * (u & 3));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    if (acc == 0x12345u) out[0] = acc;
}

// latency: lane 0 of every wave chases a chain of 16-byte loads
__global__ __launch_bounds__(64) void k_lat(const unsigned *buf, unsigned ws_mask, int steps, unsigned long long *ticks, unsigned *out)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(buf), 0, ws_mask + 1u + 64u, 0x00020000);
    unsigned o = (blockIdx.x * 0x9E3779B1u) & ws_mask & ~127u;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0);
        o = v.x & ws_mask & ~127u;   // the buffer holds random offsets
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (steps == 0x7fffffff) out[0] = o;   // (keeps the chain alive)
}

static double time_kernel(void (*launch)(hipStream_t), hipStream_t s)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(s);
    CHECK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, s));
        launch(s);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

static const unsigned *g_buf; static unsigned g_mask; static unsigned *g_out; static int g_grid;
template <int W, int G, int A> static void launch_tp(hipStream_t s) { hipLaunchKernelGGL((k_tp<W, G, A>), dim3(g_grid), dim3(kBlock), 0, s, g_buf, g_mask, g_out); }
template <int G> static void launch_chunk(hipStream_t s) { hipLaunchKernelGGL((k_tp_chunk<G>), dim3(g_grid), dim3(kBlock), 0, s, g_buf, g_mask, g_out); }

template <int W, int G, int A> static void run_tp(hipStream_t s, const char *ws_name, int cus, double ghz)
{
    const double ms = time_kernel(launch_tp<W, G, A>, s);
    const double instr = (double)g_grid * (kBlock / 64) * kIters * 8.0;
    printf("  %-6s width %2d B, %2d addresses / instr, %2d lanes active: %7.3f ms  %6.1f clk per wave-load per CU  (%5.2f B/clk/CU useful)\n", ws_name, W, G, A, ms,
           ms * 1e-3 * ghz * 1e9 * cus / instr, instr * A * W / (ms * 1e-3 * ghz * 1e9 * cus));
}
template <int G> static void run_chunk(hipStream_t s, const char *ws_name, int cus, double ghz)
{
    const double ms = time_kernel(launch_chunk<G>, s);
    const double instr = (double)g_grid * (kBlock / 64) * kIters * 8.0;
    printf("  %-6s 64-byte chunks by 4 x 16 B, %2d chunks / wave: %7.3f ms  %6.1f clk per wave-load per CU\n", ws_name, G, ms, ms * 1e-3 * ghz * 1e9 * cus / instr);
}

int main(int argc, char **argv)
{
    const bool lat_only = argc > 1 && argv[1][0] == 'l';
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", p.name, cus, ghz);
    hipStream_t s; CHECK(hipStreamCreate(&s));
    const size_t bytes = 256u << 20;
    unsigned *buf; CHECK(hipMalloc(&buf, bytes + 4096));
    std::vector<unsigned> h(bytes / 4 + 1024);
    unsigned x = 12345u;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x >> 3; }
    CHECK(hipMemcpy(buf, h.data(), bytes + 4096, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&g_out, 64));
    g_buf = buf;
    g_grid = cus * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    struct { const char *name; unsigned mask; } sets[] = {{"L1", (16u << 10) - 1}, {"L2", (2u << 20) - 1}, {"MALL", (128u << 20) - 1}};
    for (auto &ws : sets) {
        if (lat_only) break;
        g_mask = ws.mask;
        printf("working set %s (%u KB), CU full (8 waves / SIMD)\n", ws.name, (ws.mask + 1) >> 10);
        run_tp<16, 64, 64>(s, ws.name, cus, ghz); run_tp<12, 64, 64>(s, ws.name, cus, ghz); run_tp<8, 64, 64>(s, ws.name, cus, ghz); run_tp<4, 64, 64>(s, ws.name, cus, ghz);
        run_tp<16, 16, 64>(s, ws.name, cus, ghz); run_tp<8, 16, 64>(s, ws.name, cus, ghz); run_tp<4, 16, 64>(s, ws.name, cus, ghz);
        run_tp<16, 4, 64>(s, ws.name, cus, ghz); run_tp<8, 4, 64>(s, ws.name, cus, ghz); run_tp<4, 4, 64>(s, ws.name, cus, ghz);
        run_tp<16, 1, 64>(s, ws.name, cus, ghz); run_tp<8, 1, 64>(s, ws.name, cus, ghz); run_tp<4, 1, 64>(s, ws.name, cus, ghz);
        run_tp<16, 64, 16>(s, ws.name, cus, ghz); run_tp<16, 64, 4>(s, ws.name, cus, ghz); run_tp<16, 64, 1>(s, ws.name, cus, ghz);
        run_tp<8, 64, 16>(s, ws.name, cus, ghz); run_tp<4, 64, 16>(s, ws.name, cus, ghz);
        run_chunk<64>(s, ws.name, cus, ghz); run_chunk<16>(s, ws.name, cus, ghz); run_chunk<4>(s, ws.name, cus, ghz);
    }
    // latency
    unsigned long long *ticks; CHECK(hipMalloc(&ticks, 8 * 65536));
    for (auto &ws : sets) {
        for (int waves : {1, cus * 8, cus * 32}) {
            const int steps = 2000;
            hipLaunchKernelGGL(k_lat, dim3(waves), dim3(64), 0, s, buf, ws.mask, steps, ticks, g_out);
            hipLaunchKernelGGL(k_lat, dim3(waves), dim3(64), 0, s, buf, ws.mask, steps, ticks, g_out);
            CHECK(hipStreamSynchronize(s));
            std::vector<unsigned long long> t(waves);
            CHECK(hipMemcpy(t.data(), ticks, 8 * waves, hipMemcpyDeviceToHost));
            double sum = 0; for (auto v : t) sum += (double)v;
            printf("latency %-5s %6d waves (one lane each, dependent 16-byte loads): %.0f ns per load\n", ws.name, waves, sum / waves / steps * 10.0);
        }
    }
    return 0;
}
