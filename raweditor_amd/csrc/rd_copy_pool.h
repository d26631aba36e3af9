// rd_copy_pool.h -- part of librawdev.so's host side (included by rd_host_pipeline.inl).  Plain C++ with no HIP in it, so
// tests/cpp/test_copy_pool.cpp can run the locking under ThreadSanitizer on a machine without a GPU.
#pragma once
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------------
// host-side copy pool: staging buffer -> caller's pageable destination on several cores
// ------------------------------------------------------------------------------------------------
// A render into PAGEABLE host memory (a Rust Vec<u8>, a numpy array) cannot be the target of a DMA: the surface goes
// device -> pinned staging -> destination, and the second hop is a CPU memcpy.  One core moves ~10 GB/s (less while it
// takes the first-touch page faults of a fresh destination), PCIe delivers ~56 GB/s, so the hop is spread over a few
// helper threads.  Process-wide, started on first use, never joined (the object is leaked on purpose: no destructor
// runs against waiting threads at exit).  RD_COPY_THREADS = helpers (default 4; 0 = the calling thread alone).
namespace {
struct rd_copy_pool {
    struct job { char *d; const char *s; size_t n; };
    std::mutex run_mu;                         // one parallel copy at a time
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<job> jobs;
    size_t next = 0, pending = 0;
    unsigned helpers = 0;

    static rd_copy_pool &get()
    {
        static rd_copy_pool *pool = [] {
            rd_copy_pool *p = new rd_copy_pool;
            p->jobs.reserve(32);                  // helpers + 1 pieces per copy: push_back below never allocates
            const char *e = getenv("RD_COPY_THREADS");
            long want = e && *e ? strtol(e, nullptr, 10) : 4;
            const long hw = (long)std::thread::hardware_concurrency();
            if (hw > 0 && want > hw - 1) want = hw - 1;
            if (want < 0) want = 0;
            if (want > 16) want = 16;
            for (long i = 0; i < want; ++i) {
                try { std::thread([p] { p->work(); }).detach(); p->helpers += 1; } catch (...) { break; }
            }
            return p;
        }();
        return *pool;
    }
    bool take(job &j)                           // caller holds mu
    {
        if (next >= jobs.size()) return false;
        j = jobs[next++];
        return true;
    }
    void work()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            job j;
            if (!take(j)) { cv_work.wait(lk); continue; }
            lk.unlock();
            memcpy(j.d, j.s, j.n);
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
    // dst[0..n) = src[0..n), split into 2 MiB-aligned pieces over the helpers and the calling thread
    void copy(void *dst, const void *src, size_t n)
    {
        const size_t parts = helpers + 1u;
        if (parts == 1u || n < (4u << 20)) { memcpy(dst, src, n); return; }
        std::lock_guard<std::mutex> run(run_mu);
        std::unique_lock<std::mutex> lk(mu);
        jobs.clear(); next = 0;
        size_t piece = ((n + parts - 1) / parts + ((2u << 20) - 1)) & ~(size_t)((2u << 20) - 1);
        for (size_t off = 0; off < n; off += piece)
            jobs.push_back(job{ (char *)dst + off, (const char *)src + off, n - off < piece ? n - off : piece });
        pending = jobs.size();
        cv_work.notify_all();
        for (;;) {                              // the calling thread copies too
            job j;
            if (!take(j)) break;
            lk.unlock();
            memcpy(j.d, j.s, j.n);
            lk.lock();
            --pending;
        }
        cv_done.wait(lk, [this] { return pending == 0; });
    }
};
}  // namespace
