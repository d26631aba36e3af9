// tools/q8_exhaustive.hip -- the 8-bit surfaces' shortcut rd_q8_gamma (rd_kernels.h: hardware log2/exp2 + the pinned
// evaluation near a code boundary) against the pinned definition, for ALL 2^32 float encodings:
//   (1) on the device:  rd_q8_gamma(x) == rd_q8(rd_gamma_clamp(x))                      (the product's own exact path)
//   (2) on the host:    rd_q8_gamma(x) == (uint8)(clamp(ref_powf(x, 1/2.2)) * 255 + 0.5)   (the oracle's pow, clamp, pack)
// and, for the choice of RD_Q8_EPS, the largest distance between the shortcut's 255 * clamp(2^(log2(x)/2.2)) and the
// pinned 255 * gamma(x) over every encoding, and how many encodings take the pinned evaluation.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Ioracle -o tools/q8_exhaustive \
//         tools/q8_exhaustive.hip -Loracle -ldevelop_ref -Wl,-rpath,\$ORIGIN/../oracle -pthread
//   tools/q8_exhaustive [--device-only] [--lut]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../raweditor_amd/csrc/rd_kernels.h"
extern "C" {
#include "develop_ref.h"
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct q8_stats { unsigned long long mismatches, fallbacks; uint32_t first_bad; float max_dist; };

// out8[i] = rd_q8_gamma(base + i); stats accumulate over the launch
__global__ void __launch_bounds__(256) k(uint32_t base, uint8_t *out8, q8_stats *st)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = rd_u2f(base + i);
    const uint32_t fast = rd_q8_gamma(x);
    const uint32_t exact = rd_q8(rd_gamma_clamp(x));
    out8[i] = (uint8_t)fast;
    if (fast != exact || fast > 255u) {
        atomicAdd(&st->mismatches, 1ull);
        atomicMin(&st->first_bad, base + i);
    }
    if (x >= RD_FLT_MIN) {                                        // the shortcut's own intermediate, for the statistics
        float z;
        const float e = rd_hw_gamma01(x, z);                      // round 3: clamped hardware pow, add-magic code, distance to the tie
        const float t = __builtin_fmaf(e, 255.0f, RD_MAGIC23);
        const float dn = __builtin_fmaf(e, 255.0f, -(t - RD_MAGIC23));
        if (__builtin_fabsf(dn) > 0.5f - RD_Q8_EPS) atomicAdd(&st->fallbacks, 1ull);
        const float d = __builtin_fabsf(e * 255.0f - rd_gamma_clamp(x) * 255.0f);
        // atomicMax on the bits: d >= 0, so the integer order is the float order
        atomicMax(reinterpret_cast<uint32_t *>(&st->max_dist), rd_f2u(d));
    }
}

// round 4: the export kernel's LDS threshold table (rd_q8_lut_bits) instead of the transcendental shortcut: --lut
__global__ void __launch_bounds__(256) k_lut(uint32_t base, uint8_t *out8, q8_stats *st)
{
    __shared__ uint32_t lut[RD_Q8_LUT_WORDS];
    rd_q8_lut_load(lut);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = rd_u2f(base + i);
    const uint32_t fast = rd_q8_lut_bits(x, lut) >> 16;
    const uint32_t exact = rd_q8(rd_gamma_clamp(x));
    out8[i] = (uint8_t)fast;
    if (fast != exact) {
        atomicAdd(&st->mismatches, 1ull);
        atomicMin(&st->first_bad, base + i);
    }
}

static uint8_t oracle_q8(uint32_t bits)
{
    float x; memcpy(&x, &bits, 4);
    float c = ref_powf(x, 0.45454547f, 0);
    c = c > 0.0f ? c : 0.0f;                    // max(c, 0) with NaN -> 0 (shaders.rs:264; oracle's clamp convention)
    c = c < 1.0f ? c : 1.0f;
    uint8_t q;
    ref_pack_u8(&c, 1, &q);
    return q;
}

int main(int argc, char **argv)
{
    bool device_only = false, lut = false;
    for (int a = 1; a < argc; ++a) { device_only |= !strcmp(argv[a], "--device-only"); lut |= !strcmp(argv[a], "--lut"); }
    if (lut) {
        std::vector<uint32_t> table(RD_Q8_LUT_WORDS + 63u, 0u);
        rd_q8_lut_build(table.data());
        CK(hipMemcpyToSymbol(HIP_SYMBOL(rd_q8_lut_dev), table.data(), table.size() * sizeof(uint32_t)));
    }
    const uint32_t CH = 1u << 26;
    uint8_t *dev; CK(hipMalloc((void **)&dev, (size_t)CH));
    q8_stats *dst; CK(hipMalloc((void **)&dst, sizeof(q8_stats)));
    q8_stats st = { 0, 0, 0xffffffffu, 0.0f };
    CK(hipMemcpy(dst, &st, sizeof st, hipMemcpyHostToDevice));
    std::vector<uint8_t> host(CH);
    const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    std::atomic<unsigned long long> bad{ 0 };
    std::atomic<uint32_t> first_bad{ 0xffffffffu };
    for (uint32_t c = 0; c < 64; ++c) {
        const uint32_t base = c * CH;
        if (lut) hipLaunchKernelGGL(k_lut, dim3(CH / 256), dim3(256), 0, 0, base, dev, dst);
        else hipLaunchKernelGGL(k, dim3(CH / 256), dim3(256), 0, 0, base, dev, dst);
        CK(hipGetLastError());
        CK(hipMemcpy(host.data(), dev, (size_t)CH, hipMemcpyDeviceToHost));
        if (!device_only) {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([&, t]() {
                    unsigned long long b = 0;
                    for (uint64_t i = t; i < CH; i += nt)
                        if (host[i] != oracle_q8(base + (uint32_t)i)) {
                            ++b;
                            uint32_t cur = first_bad.load();
                            while (base + (uint32_t)i < cur && !first_bad.compare_exchange_weak(cur, base + (uint32_t)i)) {}
                        }
                    bad += b;
                });
            for (auto &x : th) x.join();
        }
        if (c % 16 == 15) { printf("checked %u / 64 chunks\n", c + 1); fflush(stdout); }
    }
    CK(hipMemcpy(&st, dst, sizeof st, hipMemcpyDeviceToHost));
    const char *what = lut ? "rd_q8_lut_bits (LDS threshold table)" : "rd_q8_gamma";
    if (!lut) printf("RD_Q8_EPS = %g codes\n", (double)RD_Q8_EPS);
    printf("%s vs rd_q8(rd_gamma_clamp) on the device, all 2^32 encodings: %llu mismatches", what, st.mismatches);
    if (st.mismatches) printf(" (first at 0x%08x)", st.first_bad);
    printf("\n");
    if (!device_only) {
        printf("%s vs the oracle's pow + clamp + pack on the host, all 2^32 encodings: %llu mismatches", what, (unsigned long long)bad);
        if (bad) printf(" (first at 0x%08x)", first_bad.load());
        printf("\n");
    }
    if (!lut) printf("largest |y' - y| over all x >= FLT_MIN: %.6g codes; encodings that take the pinned evaluation: %llu of 2^31 non-negative "
           "(%.4f %%)\n", (double)st.max_dist, st.fallbacks, 100.0 * (double)st.fallbacks / 2147483648.0);
    return (st.mismatches || bad) ? 1 : 0;
}
