// prio_launch — does a small kernel on a HIGH-priority stream get its workgroups onto the CUs while a large kernel of another
// (normal-priority) stream is in its full phase?  (Pairs in flight: an alignment's 17-workgroup reduce + solve waits for wave slots
// behind the other alignment's 7 813-workgroup search launch -- profiles/r06_chain_overlap.txt.)
// A: `big` workgroups of 128 threads, each spinning ~20 us (8 192 wave slots -> several rounds, like the search launch).
// B: 17 workgroups of 256 threads, ~3 us of work, queued 15 us after A started, on a normal or a high-priority stream.
// Prints B's latency (queue -> done) for both.  hipcc --offload-arch=gfx950 -O2 prio_launch.hip -o prio_launch
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>

__global__ void k_spin(long long ticks, unsigned *sink)
{
    const long long t0 = wall_clock64();
    unsigned v = threadIdx.x;
    while (wall_clock64() - t0 < ticks) v = v * 1664525u + 1013904223u;
    if (v == 0xdeadbeefu) *sink = v;
}

int main()
{
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t a, b_norm, b_high;
    (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&b_norm, hipStreamNonBlocking);
    (void)hipStreamCreateWithPriority(&b_high, hipStreamNonBlocking, hi);
    unsigned *sink;
    (void)hipMalloc(&sink, 64);
    int rate_khz = 0;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const double ticks_per_us = rate_khz / 1000.0;
    std::printf("stream priority range: least %d .. greatest %d; wall clock %.1f ticks per us\n", lo, hi, ticks_per_us);
    for (int big : {8192, 16384, 32768}) {
        for (int rep = 0; rep < 3; ++rep)
            for (int which = 0; which < 2; ++which) {
                hipStream_t b = which ? b_high : b_norm;
                (void)hipDeviceSynchronize();
                const auto t0 = clk::now();
                k_spin<<<big, 128, 0, a>>>((long long)(20 * ticks_per_us), sink);
                while (us(t0, clk::now()) < 15.0) {}
                const auto t1 = clk::now();
                k_spin<<<17, 256, 0, b>>>((long long)(3 * ticks_per_us), sink);
                (void)hipStreamSynchronize(b);
                const auto t2 = clk::now();
                (void)hipStreamSynchronize(a);
                const auto t3 = clk::now();
                if (rep) std::printf("A: %5d workgroups of 20 us (done after %6.1f us) | B on a %s stream: done %6.1f us after it was queued\n", big, us(t0, t3),
                                     which ? "HIGH-priority  " : "normal-priority", us(t1, t2));
            }
    }
    return 0;
}
