// dev: what the PCIe link gives a 9.8 MB frame -- pinned H2D alone, pinned D2H alone, both at once on two streams, and the host side
// of an upload (pageable -> pinned memcpy with 1, 2, 4 threads).   hipcc --offload-arch=gfx950 -O2 -pthread -o /tmp/link tools/microbench/link.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t B = 307200ull * 32;
    const int reps = 40;
    char *h_up, *h_down, *d_a, *d_b;
    hipHostMalloc(&h_up, B * 2); hipHostMalloc(&h_down, B * 2); hipMalloc(&d_a, B * 2); hipMalloc(&d_b, B * 2);
    memset(h_up, 1, B * 2); memset(h_down, 2, B * 2);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int w = 0; w < 3; ++w) { hipMemcpyAsync(d_a, h_up, B, hipMemcpyHostToDevice, s1); hipMemcpyAsync(h_down, d_b, B, hipMemcpyDeviceToHost, s2); }
    hipDeviceSynchronize();
    double t0 = now();
    for (int r = 0; r < reps; ++r) hipMemcpyAsync(d_a + (r & 1) * B, h_up + (r & 1) * B, B, hipMemcpyHostToDevice, s1);
    hipStreamSynchronize(s1);
    double t1 = now();
    printf("H2D alone : %.3f ms per 9.8 MB frame = %.1f GB/s\n", (t1 - t0) / reps, B * reps / (t1 - t0) / 1e6);
    t0 = now();
    for (int r = 0; r < reps; ++r) hipMemcpyAsync(h_down + (r & 1) * B, d_b + (r & 1) * B, B, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s2);
    t1 = now();
    printf("D2H alone : %.3f ms per frame = %.1f GB/s\n", (t1 - t0) / reps, B * reps / (t1 - t0) / 1e6);
    t0 = now();
    for (int r = 0; r < reps; ++r) {
        hipMemcpyAsync(d_a + (r & 1) * B, h_up + (r & 1) * B, B, hipMemcpyHostToDevice, s1);
        hipMemcpyAsync(h_down + (r & 1) * B, d_b + (r & 1) * B, B, hipMemcpyDeviceToHost, s2);
    }
    hipStreamSynchronize(s1); hipStreamSynchronize(s2);
    t1 = now();
    printf("both ways : %.3f ms per frame pair = %.1f GB/s each way\n", (t1 - t0) / reps, B * reps / (t1 - t0) / 1e6);
    // one 157 MB copy
    char *h_big, *d_big; hipHostMalloc(&h_big, B * 16); hipMalloc(&d_big, B * 16); memset(h_big, 3, B * 16);
    hipMemcpy(d_big, h_big, B * 16, hipMemcpyHostToDevice);
    t0 = now(); hipMemcpy(d_big, h_big, B * 16, hipMemcpyHostToDevice); t1 = now();
    printf("H2D 157 MB: %.3f ms = %.1f GB/s\n", t1 - t0, B * 16 / (t1 - t0) / 1e6);
    t0 = now(); hipMemcpy(h_big, d_big, B * 16, hipMemcpyDeviceToHost); t1 = now();
    printf("D2H 157 MB: %.3f ms = %.1f GB/s\n", t1 - t0, B * 16 / (t1 - t0) / 1e6);
    // host side: pageable -> pinned
    std::vector<char> page(B * 16, 5);
    for (int nt : {1, 2, 4, 8}) {
        t0 = now();
        for (int r = 0; r < 16; ++r) {
            std::vector<std::thread> th;
            const size_t step = B / nt;
            for (int t = 1; t < nt; ++t) th.emplace_back([&, t, r] { memcpy(h_up + t * step, page.data() + r * B + t * step, step); });
            memcpy(h_up, page.data() + r * B, step);
            for (auto &x : th) x.join();
        }
        t1 = now();
        printf("pageable -> pinned, %d thread(s): %.3f ms per frame = %.1f GB/s\n", nt, (t1 - t0) / 16, B * 16 / (t1 - t0) / 1e6);
    }
    // pinned -> fresh pageable (first touch)
    for (int nt : {1, 4}) {
        std::vector<char> *fresh = new std::vector<char>();
        fresh->reserve(B * 16);
        char *dst = fresh->data();
        t0 = now();
        for (int r = 0; r < 16; ++r) {
            std::vector<std::thread> th;
            const size_t step = B / nt;
            for (int t = 1; t < nt; ++t) th.emplace_back([&, t, r] { memcpy(dst + r * B + t * step, h_down + t * step, step); });
            memcpy(dst + r * B, h_down, step);
            for (auto &x : th) x.join();
        }
        t1 = now();
        printf("pinned -> fresh pageable, %d thread(s): %.3f ms per frame = %.1f GB/s\n", nt, (t1 - t0) / 16, B * 16 / (t1 - t0) / 1e6);
        delete fresh;
    }
    return 0;
}
