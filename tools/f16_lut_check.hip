// tools/f16_lut_check.hip -- the f16 surface's two-level threshold tables (rd_kernels.h, rd_f16_lut_*) against the pinned
// function on the HOST, for every float encoding: the lookup is restated here in plain integer arithmetic exactly as the
// kernel does it (w = clamp01(x) * 2^-110; half = ((fine[w >> 13] + w) >> 13) + C[w >> 16]; s = E[w >> 16] + w; the pinned
// evaluation for 0 < x < 2^-16 and for the one non-monotone encoding) and compared with binary16(rd_gamma_clamp(x))
// and trunc(255 rd_gamma_clamp(x) + 0.5).  The device-side twin is rd_selftest_f16_lut (tests/test_gpu_q8.py).
// The tables come from the library itself (rd_f16_lut_tables: no device needed); the pinned function from rd_math.h.
// Build: hipcc -O2 -ffp-contract=off -o tools/f16_lut_check tools/f16_lut_check.hip -pthread -ldl     (CPU only, ~10 s on 8 cores)
// Run:   tools/f16_lut_check [path/to/librawdev.so]
#include <cstdio>
#include <dlfcn.h>
#include <thread>
#include <vector>
#include "../raweditor_amd/csrc/rd_math.h"

#define RD_F16_LUT_SCALE 0x1p-110f
#define RD_F16_LUT_NF 17409u
#define RD_F16_LUT_NC 2177u
#define RD_F16_LUT_DIP_W (0x3eefb555u - 0x37000000u)
static inline uint32_t rd_f16_bits_host(float g) { return __builtin_bit_cast(uint16_t, (_Float16)g); }   // the compiler's own conversion

int main(int argc, char **argv)
{
    std::vector<uint16_t> fine(RD_F16_LUT_NF + 1u);
    std::vector<uint32_t> coarse(RD_F16_LUT_NC * 2u);
    void *lib = dlopen(argc > 1 ? argv[1] : "raweditor_amd/librawdev.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    typedef int (*tables_fn)(uint16_t *, size_t, uint32_t *, size_t);
    tables_fn tables = (tables_fn)dlsym(lib, "rd_f16_lut_tables");
    if (!tables) { fprintf(stderr, "rd_f16_lut_tables not exported\n"); return 2; }
    const int rc = tables(fine.data(), fine.size(), coarse.data(), coarse.size());
    printf("rd_f16_lut_tables: %d\n", rc);
    if (rc) return 2;
    const unsigned T = std::max(1u, std::thread::hardware_concurrency());
    std::vector<unsigned long long> bad(T, 0), pinned(T, 0), highbits(T, 0);
    std::vector<uint32_t> first(T, 0xffffffffu);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            const uint64_t a = (1ull << 32) * t / T, b = (1ull << 32) * (t + 1) / T;
            for (uint64_t e = a; e < b; ++e) {
                const float x = rd_u2f((uint32_t)e);
                float xc = x != x ? 0.0f : x;                      // v_med3 / the clamp modifier: NaN -> 0
                xc = xc <= 0.0f ? 0.0f : xc > 1.0f ? 1.0f : xc;        // (-0 -> +0, as the device clamp gives: rd_selftest_f16_lut checks that side)
                const float w = xc * RD_F16_LUT_SCALE;
                const uint32_t wb = rd_f2u(w);
                uint32_t half = ((fine[wb >> 13] + wb) >> 13) + coarse[2u * (wb >> 16) + 1u];
                uint32_t s = coarse[2u * (wb >> 16)] + wb;
                const bool pin = (rd_f2u(xc) - 1u) < 0x377fffffu || wb == RD_F16_LUT_DIP_W;
                const float g = rd_gamma_clamp(x);
                const uint32_t he = rd_f16_bits_host(g), qe = (uint32_t)__builtin_fmaf(g, 255.0f, 0.5f);
                if (pin) { pinned[t] += 1; continue; }             // the kernel evaluates these with the pinned function itself
                if (half != he || (s >> 16) != qe) { bad[t] += 1; if ((uint32_t)e < first[t]) first[t] = (uint32_t)e; }
                if (half >> 16 || s >> 24) highbits[t] += 1;
            }
        });
    for (auto &x : th) x.join();
    unsigned long long nb = 0, np = 0, nh = 0;
    uint32_t f = 0xffffffffu;
    for (unsigned t = 0; t < T; ++t) { nb += bad[t]; np += pinned[t]; nh += highbits[t]; if (first[t] < f) f = first[t]; }
    printf("all 2^32 encodings: %llu mismatching (first 0x%08x), %llu sent to the pinned evaluation, %llu with stray high bits\n", nb, f, np, nh);
    return nb || nh ? 1 : 0;
}
