// rd_node_worker.h -- part of librawdev.so's host side (included by rd_host_batch.inl).  Plain C++ with no HIP in it, so
// tests/cpp/test_node_worker.cpp can run the hand-off under ThreadSanitizer on a machine without a GPU.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <thread>

// One host thread per device, for the life of the node batch (round 5; rounds 2-4 started and joined N threads inside every
// rd_node_batch_develop call -- 20 ms apart in the bench -- and a thread that failed to start there took the process down:
// the already started ones were destroyed joinable).  A worker sleeps on its condition variable, runs the job it is handed
// -- a plain function pointer + context: posting allocates nothing and cannot throw -- inside its own catch-all, and
// reports status + message.  Its device is made current once, when the thread starts (`on_start`), so the per-call device
// guard finds nothing to do.  The library passes its own hooks: on_start = hipSetDevice, caught = rd_caught (sorts the
// exception in flight into a status + rd_last_error()), last_error = rd_last_error (the worker thread's own).
struct rd_node_worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int (*call)(void *ctx, uint32_t d) = nullptr;
    void *ctx = nullptr;
    uint32_t index = 0;
    int device = 0;
    bool has_job = false, done = false, quit = false;
    int rc = 0;
    char msg[512] = "";
    void (*on_start)(int device) = nullptr;
    int (*caught)(const char *where) = nullptr;
    const char *(*last_error)() = nullptr;

    void loop()
    {
        if (on_start) on_start(device);
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return has_job || quit; });
            if (quit) return;
            has_job = false;
            int (*fn)(void *, uint32_t) = call;
            void *c = ctx;
            lk.unlock();
            int r;
            try { r = fn(c, index); }
            catch (...) { r = caught ? caught("rd_node_batch worker") : -6; }     // nothing leaves a thread either: that would be std::terminate
            lk.lock();
            rc = r;
            std::snprintf(msg, sizeof msg, "%s", r && last_error ? last_error() : "");
            done = true;
            cv.notify_all();
        }
    }
    void post(int (*fn)(void *, uint32_t), void *c)
    {
        { std::lock_guard<std::mutex> lk(mu); call = fn; ctx = c; done = false; has_job = true; }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
        return rc;
    }
    void stop()
    {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        th.join();
    }
};
