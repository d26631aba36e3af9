// tools/q8_monotone.hip -- is the pinned 8-bit code q(x) = trunc(255 * gamma(x) + 0.5) (rd_q8(rd_gamma_clamp(x)), rd_math.h /
// rd_kernels.h: what the RGBA8 surface stores) a MONOTONE step function of x?  A threshold table can only replace the
// evaluation if it is.  Walks every non-negative float encoding in order on the HOST (rd_math.h is host + device code, the
// same FMAs), checks that the code never decreases, lists the 255 steps and how many of them share a bucket of 2^16
// consecutive encodings (the table's granularity: top 16 bits of the float).
// Build: hipcc -O2 -ffp-contract=off -o tools/q8_monotone tools/q8_monotone.hip -pthread      (runs on the CPU, ~10 s on 8 cores)
#include <cstdio>
#include <thread>
#include <vector>
#include "../raweditor_amd/csrc/rd_math.h"

static inline uint32_t q8(float x) { return (uint32_t)__builtin_fmaf(rd_gamma_clamp(x), 255.0f, 0.5f); }

int main()
{
    const uint32_t hi = 0x7f800000u;                      // up to +inf inclusive
    const unsigned T = std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::vector<uint32_t>> steps(T);
    std::vector<unsigned long long> bad(T, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            const uint64_t a = (uint64_t)(hi + 1ull) * t / T, b = (uint64_t)(hi + 1ull) * (t + 1) / T;
            uint32_t prev = a ? q8(rd_u2f((uint32_t)a - 1u)) : 0u;
            for (uint64_t e = a; e < b; ++e) {
                const uint32_t q = q8(rd_u2f((uint32_t)e));
                if (q < prev) bad[t] += 1;
                else if (q > prev) { for (uint32_t k = prev; k < q; ++k) steps[t].push_back((uint32_t)e); }
                prev = q;
            }
        });
    for (auto &x : th) x.join();
    unsigned long long nbad = 0;
    std::vector<uint32_t> all;
    for (unsigned t = 0; t < T; ++t) { nbad += bad[t]; all.insert(all.end(), steps[t].begin(), steps[t].end()); }
    printf("encodings 0x00000000 .. 0x7f800000 walked in order: %llu decreases of the code (monotone: %s), %zu steps up\n", nbad,
           nbad ? "NO" : "yes", all.size());
    if (!all.empty())
        printf("first step (code 0 -> 1) at 0x%08x = %.9g, last (254 -> 255) at 0x%08x = %.9g; q(1.0) = %u, q(+inf) = %u, q(0x%08x) = %u\n",
               all.front(), rd_u2f(all.front()), all.back(), rd_u2f(all.back()), q8(1.0f), q8(__builtin_inff()), all.back() - 1u, q8(rd_u2f(all.back() - 1u)));
    for (int shift : { 16, 17, 18 }) {
        unsigned worst = 0, shared = 0;
        for (size_t i = 0; i < all.size();) {
            size_t j = i;
            while (j < all.size() && (all[j] >> shift) == (all[i] >> shift)) ++j;
            if (j - i > worst) worst = (unsigned)(j - i);
            if (j - i > 1) shared += 1;
            i = j;
        }
        printf("buckets of 2^%d encodings: at most %u step(s) in one bucket, %u bucket(s) hold more than one\n", shift, worst, shared);
    }
    // negative encodings, NaNs: code 0 (x < 0 -> pow = NaN -> clamp -> 0; -0 -> 0)
    unsigned long long nz = 0;
    for (uint64_t e = 0x80000000ull; e <= 0xffffffffull; e += 4099) nz += q8(rd_u2f((uint32_t)e)) != 0;   // a sample: 2^31 / 4099 encodings
    for (uint64_t e = 0x7f800001ull; e <= 0x7fffffffull; e += 127) nz += q8(rd_u2f((uint32_t)e)) != 0;    // positive NaNs
    printf("negative encodings / NaNs sampled with a non-zero code: %llu\n", nz);
    return nbad != 0;
}
