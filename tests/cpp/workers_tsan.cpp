// workers_tsan.cpp — the helper threads of a context (csrc/workers.hpp: SourceWorker, TicketWorker, DownloadWorker) with stub
// jobs, posted / waited for / shut down from several threads at once.  Built and run by tests/test_workers_cpu.py under
// -fsanitize=thread and under -fsanitize=address,undefined on the CPU box: no GPU call in here.
// They serve rsreg_icp_set_source* (the source load beside the index build, incremental_icp.hpp:57-58), the frame uploads and
// side jobs of the scheme loops (types.hpp:30-43) and the streamed download of the merged cloud.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../realsense-pointcloud_amd/csrc/workers.hpp"

using namespace rsreg;

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

static void spin(int us)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(us)) {}
}

// one job at a time; every post is followed by a wait somewhere: the status of the LAST job is what wait returns
static void source_worker()
{
    SourceWorker w;
    std::atomic<int> ran{0};
    long long sum = 0;   // written by the jobs only (one at a time), read after the last wait
    for (int k = 0; k < 200; ++k) {
        w.post([&, k] { spin(k % 7); sum += k; ++ran; return k % 5 == 4 ? -3 : 0; });
        if (k % 3 == 0) REQUIRE(w.wait() == (k % 5 == 4 ? -3 : 0));
    }
    (void)w.wait();
    REQUIRE(ran == 200 && sum == 199 * 200 / 2);
    // waiters on other threads while the poster keeps posting
    std::atomic<bool> go{true};
    std::vector<std::thread> waiters;
    for (int t = 0; t < 3; ++t) waiters.emplace_back([&] { while (go) (void)w.wait(); });
    for (int k = 0; k < 200; ++k) w.post([&] { ++ran; return 0; });
    (void)w.wait();
    go = false;
    for (auto &t : waiters) t.join();
    REQUIRE(ran == 400);
    w.shutdown();
    w.shutdown();   // (rsreg_ctx_destroy may call it twice)
    SourceWorker never_used;
    never_used.shutdown();
}

// many posters, tickets waited for in any order by any thread; a failed job's status is reported once to a wait for THAT ticket, peek leaves it
static void ticket_worker()
{
    TicketWorker w;
    std::atomic<int> ran{0};
    std::vector<std::thread> posters;
    std::vector<std::vector<uint64_t>> tickets(4);
    for (int t = 0; t < 4; ++t)
        posters.emplace_back([&, t] {
            std::mt19937 rng(t);
            for (int k = 0; k < 150; ++k) {
                tickets[t].push_back(w.post([&, k] { spin(k % 5); ++ran; return 0; }));
                if (rng() % 4 == 0) REQUIRE(w.wait(tickets[t][rng() % tickets[t].size()]) == 0);
            }
        });
    for (auto &t : posters) t.join();
    for (auto &v : tickets)
        for (uint64_t k : v) REQUIRE(w.peek(k) == 0);
    REQUIRE(ran == 600);
    const uint64_t bad = w.post([] { return -7; });
    const uint64_t after = w.post([] { return 0; });
    REQUIRE(w.peek(after) == 0);     // a job's status is its own: the job behind a failed one has not failed
    REQUIRE(w.wait(after) == 0);
    REQUIRE(w.peek(bad) == -7);      // a third party sees it, again and again ...
    REQUIRE(w.peek(bad) == -7);
    REQUIRE(w.wait(bad) == -7);      // ... the poster's wait takes it
    REQUIRE(w.wait(bad) == 0);
    // shutdown with jobs still queued: they all run first
    for (int k = 0; k < 50; ++k) (void)w.post([&] { spin(20); ++ran; return 0; });
    w.shutdown();
    REQUIRE(ran == 650);
    w.shutdown();
}

// the streaming copy against memcpy: every size round the 64-byte blocks and the 4 096-byte threshold, every misalignment of source
// and destination, and the bytes on either side untouched
static void stream_copy_is_memcpy()
{
    std::vector<unsigned char> src(3u << 20), dst(3u << 20), want(3u << 20);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)(i * 2654435761u >> 24);
    const size_t sizes[] = {0, 1, 15, 16, 63, 64, 65, 4095, 4096, 4097, 4096 + 63, 8191, 100000, (1u << 20) + 13};
    for (size_t n : sizes)
        for (size_t so : {(size_t)0, (size_t)1, (size_t)7, (size_t)16, (size_t)33})
            for (size_t d_o : {(size_t)0, (size_t)3, (size_t)8, (size_t)16, (size_t)47}) {
                std::fill(dst.begin(), dst.end(), (unsigned char)0xA5);
                std::fill(want.begin(), want.end(), (unsigned char)0xA5);
                std::memcpy(want.data() + 64 + d_o, src.data() + so, n);
                stream_copy(dst.data() + 64 + d_o, src.data() + so, n);
                REQUIRE(std::memcmp(dst.data(), want.data(), n + 256) == 0);
            }
}

// four staging slots taken in turn, two jobs copied out side by side; the copy-out of a job runs when its "event" has fired
static void download_worker()
{
    DownloadWorker w;
    std::atomic<int> ready_calls{0};
    w.wait_ready = [&](const DownloadWorker::Job &j) -> int {
        ++ready_calls;
        spin(30);
        return j.device == 99 ? 5 : 0;   // (device 99: the stub's "event wait failed")
    };
    const size_t small = 4096, big = (size_t)5 << 20;   // (the big ones are copied out by four threads)
    std::vector<std::vector<char>> stage(DownloadWorker::kSlots, std::vector<char>(big)), dst(12, std::vector<char>(big)), sent(12);
    for (int k = 0; k < 12; ++k) {
        const int slot = k % DownloadWorker::kSlots;
        w.wait_slot(slot);   // (nobody copies out of this staging buffer any more)
        const size_t bytes = k % 3 == 0 ? big - 17 * (size_t)k : small;
        for (size_t i = 0; i < bytes; i += 997) stage[slot][i] = (char)(k + 1 + i / 997);
        sent[k].assign(stage[slot].begin(), stage[slot].begin() + (long)bytes);
        w.post(DownloadWorker::Job{nullptr, stage[slot].data(), dst[k].data(), bytes, slot, 0});
    }
    REQUIRE(w.wait_idle() == 0);
    for (int k = 0; k < 12; ++k) REQUIRE(std::memcmp(dst[k].data(), sent[k].data(), sent[k].size()) == 0);
    REQUIRE(ready_calls == 12);
    // a slot given back without a job (the error path of rsreg_cloud_download_async), from another thread
    w.wait_slot(0);
    std::thread giver([&] { spin(200); w.release_slot(0); });
    w.wait_slot(0);
    giver.join();
    w.release_slot(0);
    // a failed event wait: reported once by wait_idle, the slot is free again
    w.wait_slot(1);
    w.post(DownloadWorker::Job{nullptr, stage[1].data(), dst[0].data(), small, 1, 99});
    REQUIRE(w.wait_idle() == 5);
    REQUIRE(w.wait_idle() == 0);
    w.wait_slot(1);
    w.release_slot(1);
    // waiters on several threads
    std::vector<std::thread> th;
    for (int t = 0; t < 3; ++t) th.emplace_back([&] { for (int k = 0; k < 20; ++k) (void)w.wait_idle(); });
    for (int k = 0; k < 30; ++k) {
        const int slot = k % DownloadWorker::kSlots;
        w.wait_slot(slot);
        w.post(DownloadWorker::Job{nullptr, stage[slot].data(), dst[k % 12].data(), small, slot, 0});
    }
    for (auto &t : th) t.join();
    w.shutdown();
    w.shutdown();
}

// the pool of host threads behind host_parallel_for: loops from several caller threads at once, every element exactly once
static void host_pool_loops()
{
    HostPool pool(5);
    std::vector<std::thread> callers;
    std::atomic<long long> grand{0};
    for (int c = 0; c < 3; ++c)
        callers.emplace_back([&, c] {
            for (int k = 0; k < 60; ++k) {
                const size_t n = (size_t)(1000 + 7919 * k + c);
                std::vector<unsigned char> hit(n, 0);
                const std::function<void(size_t, size_t)> fn = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) ++hit[i]; };
                pool.run(n, 1 + (size_t)(k % 23), fn);
                long long sum = 0;
                for (unsigned char h : hit) { REQUIRE(h == 1); sum += h; }
                grand += sum;
            }
        });
    for (auto &t : callers) t.join();
    REQUIRE(grand > 0);
    HostPool none(0);   // no worker at all: the caller's thread does the loop
    int ran = 0;
    const std::function<void(size_t, size_t)> one = [&](size_t lo, size_t hi) { ran += (int)(hi - lo); };
    none.run(10, 4, one);
    REQUIRE(ran == 10);
    none.run(0, 4, one);
    REQUIRE(ran == 10);
}

int main()
{
    host_pool_loops();
    source_worker();
    ticket_worker();
    download_worker();
    stream_copy_is_memcpy();
    std::puts("workers ok");
    return 0;
}
