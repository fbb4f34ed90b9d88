// workers.hpp — the helper threads of a context (no GPU call in here: tests/cpp/workers_tsan.cpp runs them under
// -fsanitize=thread and -fsanitize=address on a machine without one).
#pragma once

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>
#if defined(__x86_64__) || defined(_M_X64)
#include <emmintrin.h>
#endif

namespace rsreg {

// The host side of a source load (a bounding-box round trip and ~25 launches: 0.13 ms of host time at any size) runs on a
// thread of the context's own, so that the caller's thread goes straight on to the target's index build
// (incremental_icp.hpp:57-58: setInputSource, then setInputTarget): the two queues are then FILLED side by side, not only
// drained side by side.  One job at a time; whoever needs the source (or hands one of its buffers on) waits for the
// job to have queued everything first (wait), then for the GPU as before (ev_src_done).
struct SourceWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, busy = false, stop = false;
    int rc = 0;

    void loop()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_job || stop; });
            if (stop) return;
            std::function<int()> f = std::move(job);
            has_job = false;
            busy = true;
            lk.unlock();
            const int r = f();
            lk.lock();
            rc = r;
            busy = false;
            cv.notify_all();
        }
    }
    void post(std::function<int()> f)
    {
        std::unique_lock<std::mutex> lk(m);
        if (!th.joinable()) th = std::thread([this] { loop(); });
        cv.wait(lk, [&] { return !has_job && !busy; });
        job = std::move(f);
        has_job = true;
        rc = 0;
        cv.notify_all();
    }
    int wait()   // until the posted job has run; its status
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !has_job && !busy; });
        return rc;
    }
    void shutdown()
    {
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return !has_job && !busy; });
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

// A copy of megabytes whose destination nobody reads soon -- a pinned staging buffer on its way to the DMA engine, a result the
// caller looks at after the registration: the stores go past the caches (x86-64: movntdq), so a line of the destination is
// not first READ from memory to be overwritten (a third of a plain copy's traffic) and the copy does not evict what the
// other threads work on.  Staging a 9.8 MB frame with the pool's threads: 0.25-0.31 -> see profiles/r06_link_pipeline.txt.
inline void stream_copy(void *dst_, const void *src_, size_t bytes)
{
#if defined(__x86_64__) || defined(_M_X64)
    char *dst = static_cast<char *>(dst_);
    const char *src = static_cast<const char *>(src_);
    if (bytes < 4096) { std::memcpy(dst, src, bytes); return; }
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;
    if (head) { std::memcpy(dst, src, head); dst += head; src += head; bytes -= head; }
    const size_t blocks = bytes / 64;
    for (size_t k = 0; k < blocks; ++k) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src) + 0), b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src) + 1);
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src) + 2), d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src) + 3);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 0, a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 1, b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 2, c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst) + 3, d);
        src += 64; dst += 64;
    }
    _mm_sfence();
    if (bytes % 64) std::memcpy(dst, src, bytes % 64);
#else
    std::memcpy(dst_, src_, bytes);
#endif
}

// Host-side record loops (32-byte records <-> packed xyz in pinned staging, staging <-> the caller's memory) are memory-bound
// copies of tens of MB that want a handful of cores for a fraction of a millisecond: the threads are kept (starting eight
// threads costs as much as the copy they are started for).  One loop at a time per pool (host_pool() below is the process-wide one,
// the download worker keeps small ones of its own); the caller's thread works too.
struct HostPool {
    std::vector<std::thread> th;
    std::mutex m, call_m;
    std::condition_variable cv, cv_done;
    const std::function<void(size_t, size_t)> *f = nullptr;
    size_t n = 0, parts = 0, next = 0, running = 0;
    bool stop = false;

    explicit HostPool(unsigned workers)
    {
        for (unsigned k = 0; k < workers; ++k) th.emplace_back([this] { loop(); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    // (m held) runs parts until none is left; returns with m held
    void work(std::unique_lock<std::mutex> &lk)
    {
        while (f && next < parts) {
            const size_t p = next++;
            const std::function<void(size_t, size_t)> *g = f;
            const size_t lo = n * p / parts, hi = n * (p + 1) / parts;
            ++running;
            lk.unlock();
            if (lo < hi) (*g)(lo, hi);
            lk.lock();
            if (--running == 0 && next >= parts) cv_done.notify_all();
        }
    }
    void loop()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || (f && next < parts); });
            if (stop) return;
            work(lk);
        }
    }
    // f(lo, hi) over [0, count) in `pieces` pieces (more pieces than threads: a slow thread does not hold the others up)
    void run(size_t count, size_t pieces, const std::function<void(size_t, size_t)> &fn)
    {
        if (count == 0) return;
        if (pieces <= 1 || th.empty()) { fn(0, count); return; }
        std::lock_guard<std::mutex> one(call_m);
        std::unique_lock<std::mutex> lk(m);
        f = &fn; n = count; parts = pieces; next = 0;
        cv.notify_all();
        work(lk);
        cv_done.wait(lk, [&] { return running == 0 && next >= parts; });
        f = nullptr;
    }
};

// Downloads that run beside the frame loop (rsreg_cloud_download_async): the copy lands in one of a few pinned staging
// buffers on a stream of its own; a thread waits for it and copies it out to the caller's (pageable) memory.  Two such threads
// (round 6), each with copiers of its own: the copy of a frame into pages nobody has touched yet is mostly the kernel zeroing
// them, one 2 MB page per faulting thread at a time, and took 0.30 ms a frame where the link needs 0.18 -- two frames side by
// side keep up with it.
struct DownloadWorker {
    static constexpr int kSlots = 4, kThreads = 2;
    struct Job {
        void *ev;           // the event behind the staging copy (a hipEvent_t in the library)
        const char *stage;
        char *dst;
        size_t bytes;
        int slot, device;
    };
    std::thread th[kThreads];
    std::mutex m;
    std::condition_variable cv;
    std::vector<Job> queue;
    int busy = 0;   // jobs being copied out
    bool stop = false;
    bool slot_busy[kSlots] = {false, false, false, false};
    int err = 0;
    std::function<int(const Job &)> wait_ready;   // blocks until the job's staging buffer is filled; 0 or an error code
    HostPool copiers[kThreads] = {HostPool(3), HostPool(3)};   // (with the job's own thread: four each)

    void loop(int me)
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return !queue.empty() || stop; });
            if (queue.empty() && stop) return;
            const Job j = queue.front();
            queue.erase(queue.begin());
            ++busy;
            lk.unlock();
            // (wait_ready: the creator's -- hipSetDevice + hipEventSynchronize in the library, a stub in tests/cpp/workers_tsan.cpp)
            const int e = wait_ready ? wait_ready(j) : 0;
            if (e == 0) {
                // a few threads of its own (kept: starting them per job cost a quarter of the copy): one core copies ~10 GB/s
                // into pages it touches first, and a frame of 10 MB would take three times as long as the link needs for it
                if (j.bytes >= (size_t)4 << 20) {
                    const std::function<void(size_t, size_t)> fn = [&j](size_t lo, size_t hi) { stream_copy(j.dst + lo * 4096, j.stage + lo * 4096, std::min(j.bytes, hi * 4096) - lo * 4096); };
                    copiers[me].run((j.bytes + 4095) / 4096, 8, fn);
                } else {
                    stream_copy(j.dst, j.stage, j.bytes);
                }
            }
            lk.lock();
            if (e != 0 && !err) err = e;
            slot_busy[j.slot] = false;
            --busy;
            cv.notify_all();
        }
    }
    void wait_slot(int slot)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !slot_busy[slot]; });
        slot_busy[slot] = true;
    }
    void release_slot(int slot)   // a slot taken by wait_slot whose job was never posted (an error in between)
    {
        std::unique_lock<std::mutex> lk(m);
        slot_busy[slot] = false;
        cv.notify_all();
    }
    void post(const Job &j)
    {
        std::unique_lock<std::mutex> lk(m);
        for (int k = 0; k < kThreads; ++k)
            if (!th[k].joinable() && (int)queue.size() + busy >= k) th[k] = std::thread([this, k] { loop(k); });   // (the second thread with the second job in flight)
        queue.push_back(j);
        cv.notify_all();
    }
    int wait_idle()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return queue.empty() && busy == 0; });
        const int e = err;
        err = 0;
        return e;
    }
    void shutdown()
    {
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return queue.empty() && busy == 0; });
            stop = true;
            cv.notify_all();
        }
        for (std::thread &t : th)
            if (t.joinable()) t.join();
    }
};

// Jobs run one after the other on a thread of their own, each with a ticket the poster can wait for.  The uploads that
// run beside the frame loop (rsreg_cloud_upload_deferred / _async) are staged here: copying a 9.8 MB frame into pinned
// memory takes 0.2 ms of a host thread, and on the caller's thread that is 0.2 ms per frame with nothing queued on the GPU.
struct TicketWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::function<int()>> queue;
    uint64_t posted = 0, done = 0;
    bool stop = false;
    std::vector<std::pair<uint64_t, int>> errs;   // (ticket, status) of the jobs that failed and whose poster has not asked yet

    void loop()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return !queue.empty() || stop; });
            if (queue.empty() && stop) return;
            std::function<int()> f = std::move(queue.front());
            queue.erase(queue.begin());
            lk.unlock();
            const int r = f();
            lk.lock();
            if (r) errs.emplace_back(done + 1, r);
            ++done;
            cv.notify_all();
        }
    }
    uint64_t post(std::function<int()> f)
    {
        std::unique_lock<std::mutex> lk(m);
        if (!th.joinable()) th = std::thread([this] { loop(); });
        queue.push_back(std::move(f));
        cv.notify_all();
        return ++posted;
    }
    int wait(uint64_t ticket)   // until job `ticket` has run; ITS status (the poster's own wait: the entry is taken out)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done >= ticket; });
        for (size_t k = 0; k < errs.size(); ++k) {
            if (errs[k].first == ticket) {
                const int e = errs[k].second;
                errs.erase(errs.begin() + (long)k);
                return e;
            }
        }
        return 0;
    }
    int peek(uint64_t ticket)   // the status of job `ticket` ALONE, for a third party: the error stays for the poster's own wait
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done >= ticket; });
        for (const auto &e : errs)
            if (e.first == ticket) return e.second;
        return 0;
    }
    void shutdown()
    {
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return done >= posted; });
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};


inline HostPool &host_pool()
{
    static HostPool pool(std::max(1u, std::min(16u, std::thread::hardware_concurrency())) - 1u);
    return pool;
}

}  // namespace rsreg
