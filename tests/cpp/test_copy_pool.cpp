// The host-side copy pool of librawdev.so (raweditor_amd/csrc/rd_copy_pool.h) on its own: plain C++, no GPU.  Built and run
// by tests/test_host_cpu.py with -fsanitize=thread: several threads call copy() at once (as concurrent
// render_full_res_to_bytes calls do), sizes on both sides of the 4 MiB serial threshold and off every alignment, and every
// destination must equal its source.  Exit code 0 = all copies exact (and no report from the sanitizer).
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../raweditor_amd/csrc/rd_copy_pool.h"

static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

int main()
{
    rd_copy_pool &pool = rd_copy_pool::get();
    std::printf("helpers %u\n", pool.helpers);
    const size_t sizes[] = { 0, 1, 4095, (4u << 20) - 1, 4u << 20, (4u << 20) + 1, (6u << 20) + 12345 };
    int bad = 0;
    std::mutex bad_mu;
    auto worker = [&](uint32_t tid) {
        for (int round = 0; round < 2; ++round) {
            for (size_t n : sizes) {
                std::vector<unsigned char> src(n + 64), dst(n + 64, 0xee);
                for (size_t i = 0; i + 4 <= src.size(); i += 4) {      // a different word every 4 bytes
                    const uint32_t v = mix((uint32_t)i + tid * 977u + (uint32_t)round * 31u);
                    std::memcpy(&src[i], &v, 4);
                }
                const size_t off = (tid + round) % 7;              // unaligned on both sides
                pool.copy(dst.data() + off, src.data() + off, n);
                bool ok = true;
                ok = n == 0 || std::memcmp(dst.data() + off, src.data() + off, n) == 0;
                for (size_t i = 0; i < off && ok; ++i) ok = dst[i] == 0xee;                 // nothing outside [off, off + n)
                for (size_t i = off + n; i < dst.size() && ok; ++i) ok = dst[i] == 0xee;
                if (!ok) { std::lock_guard<std::mutex> lk(bad_mu); bad += 1; std::fprintf(stderr, "thread %u size %zu: mismatch\n", tid, n); }
            }
        }
    };
    std::vector<std::thread> ts;
    for (uint32_t t = 0; t < 4; ++t) ts.emplace_back(worker, t);
    for (auto &t : ts) t.join();
    std::printf("%s\n", bad ? "FAILED" : "copy pool ok");
    return bad ? 1 : 0;
}
