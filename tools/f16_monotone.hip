// tools/f16_monotone.hip -- is h(x) = binary16(rd_gamma_clamp(x)) (what the RGBA-f16 surface stores) a MONOTONE step function
// of x, and how densely do its steps sit in x's encodings?  (What the two-level threshold tables of rd_f16_lut_* rest on:
// they replace the transcendental shortcut + midpoint test of rd_f16_gamma_value as rd_q8_lut_bits did for the 8-bit code.)  Walks every encoding of
// [0, 1] in order on the HOST (rd_math.h is host + device code), counts decreases, lists how many steps share a bucket of
// 2^12 / 2^13 / 2^14 / 2^16 encodings and how many buckets contain a decrease.
// Build: hipcc -O2 -ffp-contract=off -o tools/f16_monotone tools/f16_monotone.hip -pthread      (CPU only)
#include <algorithm>
#include <cstdio>
#include <thread>
#include <vector>
#include "../raweditor_amd/csrc/rd_math.h"

static inline uint32_t h16(float x) { return __builtin_bit_cast(uint16_t, (_Float16)rd_gamma_clamp(x)); }

int main()
{
    const uint32_t hi = 0x3f800000u;                      // 1.0 inclusive
    const unsigned T = std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::vector<uint32_t>> steps(T), downs(T);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            const uint64_t a = (uint64_t)(hi + 1ull) * t / T, b = (uint64_t)(hi + 1ull) * (t + 1) / T;
            uint32_t prev = a ? h16(rd_u2f((uint32_t)a - 1u)) : 0u;
            for (uint64_t e = a; e < b; ++e) {
                const uint32_t q = h16(rd_u2f((uint32_t)e));
                if (q < prev) downs[t].push_back((uint32_t)e);
                else if (q > prev) { for (uint32_t k = prev; k < q; ++k) steps[t].push_back((uint32_t)e); }
                prev = q;
            }
        });
    for (auto &x : th) x.join();
    std::vector<uint32_t> all, dn;
    for (unsigned t = 0; t < T; ++t) { all.insert(all.end(), steps[t].begin(), steps[t].end()); dn.insert(dn.end(), downs[t].begin(), downs[t].end()); }
    printf("encodings 0x00000000 .. 0x3f800000 walked in order: %zu decreases of the half (monotone: %s), %zu steps up; h(1.0) = 0x%04x\n",
           dn.size(), dn.empty() ? "yes" : "NO", all.size(), h16(1.0f));
    if (!all.empty()) printf("first step at 0x%08x = %.9g, last at 0x%08x = %.9g\n", all.front(), rd_u2f(all.front()), all.back(), rd_u2f(all.back()));
    for (size_t i = 0; i < dn.size() && i < 12; ++i) printf("  decrease at 0x%08x = %.9g: 0x%04x -> 0x%04x\n", dn[i], rd_u2f(dn[i]), h16(rd_u2f(dn[i] - 1u)), h16(rd_u2f(dn[i])));
    for (int shift : { 12, 13, 14, 16 }) {
        unsigned worst = 0, shared = 0;
        size_t lo_domain = 0;                                     // steps below 2^-16 (outside a 16-binade table)
        for (size_t i = 0; i < all.size();) {
            size_t j = i;
            while (j < all.size() && (all[j] >> shift) == (all[i] >> shift)) ++j;
            if (all[i] >= 0x37800000u) { if (j - i > worst) worst = (unsigned)(j - i); if (j - i > 1) shared += 1; } else lo_domain += j - i;
            i = j;
        }
        std::vector<uint32_t> db;
        for (uint32_t e : dn) if (e >= 0x37800000u) db.push_back(e >> shift);
        db.erase(std::unique(db.begin(), db.end()), db.end());
        printf("x >= 2^-16, buckets of 2^%d encodings: at most %u step(s) in one bucket, %u bucket(s) hold more than one, %zu bucket(s) contain a decrease; %zu steps lie below 2^-16\n",
               shift, worst, shared, db.size(), lo_domain);
    }
    // smallest gap between consecutive steps (encodings), x >= 2^-16
    uint32_t gap = ~0u, at = 0;
    for (size_t i = 1; i < all.size(); ++i) if (all[i - 1] >= 0x37800000u && all[i] - all[i - 1] < gap && all[i] != all[i - 1]) { gap = all[i] - all[i - 1]; at = all[i]; }
    printf("smallest distance between two steps (x >= 2^-16): %u encodings, at 0x%08x = %.9g\n", gap, at, rd_u2f(at));
    return 0;
}
