// How long the host waits for a kernel's result: hipStreamSynchronize against spinning on a word the kernel's last
// store writes into pinned host memory (system-scope release).  A short kernel (one workgroup) and one behind 50 us of work.
// build: hipcc -O2 --offload-arch=gfx950 -o sync_latency sync_latency.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_work(float *buf, int n, int rounds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = buf[i];
    for (int r = 0; r < rounds; ++r) v = v * 1.0001f + 0.5f;
    buf[i] = v;
}

__global__ void k_flag(unsigned *host_data, unsigned *host_flag, unsigned seq)
{
    if (threadIdx.x == 0) {
        host_data[0] = seq * 3u;
        __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *h = nullptr;
    CK(hipHostMalloc(&h, 256, hipHostMallocDefault));
    h[0] = h[16] = 0;
    float *d = nullptr;
    const int n = 1 << 22;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemset(d, 0, n * 4));
    for (int work = 0; work < 2; ++work) {
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<double> t;
            unsigned seq = 0;
            for (int it = 0; it < 300; ++it) {
                ++seq;
                CK(hipStreamSynchronize(st));
                const double t0 = now_us();
                if (work) k_work<<<n / 256, 256, 0, st>>>(d, n, 200);
                k_flag<<<1, 64, 0, st>>>(h, h + 16, seq + 1000u * (unsigned)(work * 2 + mode));
                if (mode == 0) {
                    CK(hipStreamSynchronize(st));
                } else {
                    const unsigned want = seq + 1000u * (unsigned)(work * 2 + mode);
                    while (__atomic_load_n(h + 16, __ATOMIC_ACQUIRE) != want) __builtin_ia32_pause();
                    if (h[0] != want * 3u) { std::fprintf(stderr, "data not visible\n"); return 1; }
                }
                t.push_back(now_us() - t0);
            }
            std::sort(t.begin(), t.end());
            std::printf("%s, %s: median %.1f us (p10 %.1f, p90 %.1f) from first launch to the host having the result\n",
                        work ? "behind a kernel" : "flag kernel alone", mode ? "spin on the pinned word" : "hipStreamSynchronize", t[t.size() / 2], t[t.size() / 10],
                        t[t.size() * 9 / 10]);
        }
    }
    // the kernel behind the work, timed alone, for reference
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    k_work<<<n / 256, 256, 0, st>>>(d, n, 200);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("k_work alone: %.1f us\n", ms * 1e3);
    return 0;
}
