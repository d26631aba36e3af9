// The per-device worker threads of rd_node_batch (raweditor_amd/csrc/rd_node_worker.h) on their own: plain C++, no GPU.  Built
// and run by tests/test_host_cpu.py with -fsanitize=thread.  What the library does with them, in miniature: N workers started
// once; a caller posts one job to every worker and waits for all of them, call after call (two caller threads take turns
// under a mutex, like rd_node_batch's call_mu); jobs return statuses, some throw (the worker's catch-all must turn that into
// a status + message, never into std::terminate); a worker whose start "fails" is simply never started and stop() on it is a
// no-op; stop() joins the rest.  Exit code 0 = every result as expected (and no report from the sanitizer).
#include <atomic>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../../raweditor_amd/csrc/rd_node_worker.h"

static thread_local char t_err[128] = "";
static std::atomic<int> started{ 0 };

static int caught(const char *where)
{
    try { throw; }
    catch (const std::bad_alloc &) { std::snprintf(t_err, sizeof t_err, "%s: bad_alloc", where); return -4; }
    catch (const std::exception &e) { std::snprintf(t_err, sizeof t_err, "%s: %s", where, e.what()); return -6; }
    catch (...) { std::snprintf(t_err, sizeof t_err, "%s: unknown", where); return -6; }
}

struct job_ctx {
    int round;
    std::vector<long> *sums;
};

int main()
{
    const uint32_t N = 6;
    std::unique_ptr<rd_node_worker[]> w(new rd_node_worker[N]);
    for (uint32_t d = 0; d < N; ++d) {
        w[d].index = d; w[d].device = (int)d;
        w[d].on_start = [](int) { started.fetch_add(1); };
        w[d].caught = caught;
        w[d].last_error = []() -> const char * { return t_err; };
        if (d == N - 1) continue;                                // "failed to start": never started; stop() must cope
        w[d].th = std::thread([&w, d] { w[d].loop(); });
    }
    const uint32_t live = N - 1;
    std::vector<long> sums(live, 0);
    std::mutex call_mu;
    int bad = 0;
    auto caller = [&](int who) {
        for (int round = 0; round < 200; ++round) {
            std::lock_guard<std::mutex> call(call_mu);
            job_ctx ctx{ round * 2 + who, &sums };
            int (*fn)(void *, uint32_t) = [](void *c, uint32_t d) -> int {
                job_ctx *j = static_cast<job_ctx *>(c);
                if (j->round % 7 == 3 && d == 2) throw std::runtime_error("job failed on purpose");
                if (j->round % 11 == 5 && d == 4) throw 42;
                if (j->round % 13 == 6 && d == 1) { std::snprintf(t_err, sizeof t_err, "status from device %u", d); return -3; }
                (*j->sums)[d] += j->round;                        // each worker owns its own slot
                return 0;
            };
            for (uint32_t d = 0; d < live; ++d) w[d].post(fn, &ctx);
            for (uint32_t d = 0; d < live; ++d) {
                const int rc = w[d].wait();
                int want = 0;
                const char *text = "";
                if (ctx.round % 7 == 3 && d == 2) { want = -6; text = "job failed on purpose"; }
                else if (ctx.round % 11 == 5 && d == 4) { want = -6; text = "unknown"; }
                else if (ctx.round % 13 == 6 && d == 1) { want = -3; text = "status from device 1"; }
                if (rc != want || (want && !std::strstr(w[d].msg, text)) || (!want && w[d].msg[0])) {
                    std::fprintf(stderr, "caller %d round %d device %u: rc %d (want %d) msg '%s'\n", who, ctx.round, d, rc, want, w[d].msg);
                    ++bad;
                }
            }
        }
    };
    std::thread a(caller, 0), b(caller, 1);
    a.join(); b.join();
    for (uint32_t d = 0; d < N; ++d) w[d].stop();
    for (uint32_t d = 0; d < N; ++d) w[d].stop();                 // idempotent
    long expect[5] = { 0, 0, 0, 0, 0 };
    for (int r = 0; r < 400; ++r)
        for (uint32_t d = 0; d < live; ++d) {
            const bool failed = (r % 7 == 3 && d == 2) || (r % 11 == 5 && d == 4) || (r % 13 == 6 && d == 1);
            if (!failed) expect[d] += r;
        }
    for (uint32_t d = 0; d < live; ++d)
        if (sums[d] != expect[d]) { std::fprintf(stderr, "device %u: sum %ld, expected %ld\n", d, sums[d], expect[d]); ++bad; }
    if (started.load() != (int)live) { std::fprintf(stderr, "on_start ran %d times\n", started.load()); ++bad; }
    std::printf("%s\n", bad ? "FAILED" : "node workers ok");
    return bad ? 1 : 0;
}
