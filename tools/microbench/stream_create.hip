// stream_create — what hipStreamCreateWithFlags costs in a fresh process on an MI355X: the first eight streams one after the
// other, or eight at once from eight threads; then pinned allocations by size.  (The first registration() of a process creates
// seven streams and half a dozen pinned buffers: profiles/r06_cold_run.txt.)
// hipcc --offload-arch=gfx950 -O2 -pthread stream_create.hip -o stream_create;  ./stream_create [seq|par]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

__global__ void k_empty() {}

int main(int argc, char **argv)
{
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const bool par = argc > 1 && !std::strcmp(argv[1], "par");
    const auto t0 = clk::now();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 66;
    (void)hipSetDevice(0);
    const auto t1 = clk::now();
    constexpr int kN = 8;
    hipStream_t st[kN];
    double each[kN];
    if (par) {
        std::vector<std::thread> th;
        for (int i = 0; i < kN; ++i)
            th.emplace_back([&, i] {
                (void)hipSetDevice(0);
                const auto a = clk::now();
                (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
                each[i] = ms(a, clk::now());
            });
        for (auto &t : th) t.join();
    } else {
        for (int i = 0; i < kN; ++i) {
            const auto a = clk::now();
            (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
            each[i] = ms(a, clk::now());
        }
    }
    const auto t2 = clk::now();
    std::printf("%s: runtime start %.1f ms | %d streams in %.1f ms:", par ? "eight threads" : "one after the other", ms(t0, t1), kN, ms(t1, t2));
    for (int i = 0; i < kN; ++i) std::printf(" %.1f", each[i]);
    // first use of every stream (a launch + sync)
    const auto t3 = clk::now();
    for (int i = 0; i < kN; ++i) {
        const auto a = clk::now();
        k_empty<<<1, 64, 0, st[i]>>>();
        (void)hipStreamSynchronize(st[i]);
        each[i] = ms(a, clk::now());
    }
    std::printf(" | first launch + sync on each:");
    for (int i = 0; i < kN; ++i) std::printf(" %.2f", each[i]);
    std::printf(" (%.1f ms)", ms(t3, clk::now()));
    // pinned buffers
    std::printf(" | hipHostMalloc");
    for (size_t mb : {1, 10, 10, 32, 32}) {
        void *p = nullptr;
        const auto a = clk::now();
        (void)hipHostMalloc(&p, mb << 20, hipHostMallocDefault);
        std::printf(" %zu MB %.2f", mb, ms(a, clk::now()));
    }
    std::printf(" ms\n");
    return 0;
}
