// hip_floor — what ANY process pays before its first kernel has run on an MI355X: the HIP runtime's start (first call), one
// hipMalloc, one empty kernel (code object load + launch) and a synchronisation.  The floor the first registration() of a
// process is compared with (profiles/r06_cold_run.txt).  hipcc --offload-arch=gfx950 -O2 hip_floor.hip -o hip_floor
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void k_empty() {}

int main()
{
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { std::fprintf(stderr, "no device\n"); return 66; }
    const auto t1 = clk::now();
    (void)hipSetDevice(0);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const auto t2 = clk::now();
    void *p = nullptr;
    (void)hipMalloc(&p, 64 << 20);
    const auto t3 = clk::now();
    k_empty<<<1, 64, 0, st>>>();
    (void)hipStreamSynchronize(st);
    const auto t4 = clk::now();
    void *h = nullptr;
    (void)hipHostMalloc(&h, 32 << 20, hipHostMallocDefault);
    const auto t5 = clk::now();
    k_empty<<<1, 64, 0, st>>>();
    (void)hipStreamSynchronize(st);
    const auto t6 = clk::now();
    std::printf("hip_floor: first HIP call %.2f ms | set device + stream %.2f | hipMalloc 64 MB %.2f | first kernel + sync %.2f | hipHostMalloc 32 MB %.2f | "
                "second kernel + sync %.3f | main() to first kernel done %.2f ms\n",
                ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, t6), ms(t0, t4));
    return 0;
}
