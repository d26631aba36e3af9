// tools/gamma_exhaustive.hip -- rd_gamma_clamp (product, on the GPU) against the oracle's pow + clamp for ALL 2^32 float
// encodings, and rd_exp2f (host side of rd_math.h, used for the exposure uniform) against ref_exp2f on a dense sweep.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Ioracle -o tools/gamma_exhaustive \
//         tools/gamma_exhaustive.hip -Loracle -ldevelop_ref -Wl,-rpath,\$ORIGIN/../oracle -pthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include "../raweditor_amd/csrc/rd_math.h"
extern "C" {
#include "develop_ref.h"
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k(uint32_t base, uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = rd_f2u(rd_gamma_clamp(rd_u2f(base + i)));
}

static uint32_t oracle_gamma(uint32_t bits)
{
    float x; memcpy(&x, &bits, 4);
    float c = ref_powf(x, 0.45454547f, 0);
    c = c > 0.0f ? c : 0.0f;                    // max(c, 0) with NaN -> 0 (shaders.rs:264; oracle's clamp convention)
    c = c < 1.0f ? c : 1.0f;
    uint32_t r; memcpy(&r, &c, 4);
    return r;
}

int main()
{
    const uint32_t CH = 1u << 26;
    uint32_t *dev; CK(hipMalloc((void **)&dev, (size_t)CH * 4));
    std::vector<uint32_t> host(CH);
    const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    std::atomic<unsigned long long> bad{ 0 };
    uint32_t first_bad = 0; std::atomic<int> have_bad{ 0 };
    for (uint32_t c = 0; c < 64; ++c) {
        const uint32_t base = c * CH;
        hipLaunchKernelGGL(k, dim3(CH / 256), dim3(256), 0, 0, base, dev);
        CK(hipMemcpy(host.data(), dev, (size_t)CH * 4, hipMemcpyDeviceToHost));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                unsigned long long b = 0;
                for (uint64_t i = t; i < CH; i += nt)
                    if (host[i] != oracle_gamma(base + (uint32_t)i)) { ++b; if (!have_bad.exchange(1)) first_bad = base + (uint32_t)i; }
                bad += b;
            });
        for (auto &x : th) x.join();
        if (c % 8 == 7) { printf("checked %u / 64 chunks, %llu mismatches\n", c + 1, (unsigned long long)bad); fflush(stdout); }
    }
    printf("rd_gamma_clamp (GPU) vs oracle pow+clamp over all 2^32 encodings: %llu mismatches", (unsigned long long)bad);
    if (bad) printf(" (first at 0x%08x)", first_bad);
    printf("\n");
    // host rd_exp2f vs ref_exp2f: every float in [-130, 130] at stride 7 encodings plus the specials
    unsigned long long bad2 = 0, n2 = 0;
    for (uint32_t b = 0; b < 0xff800000u; b += 7u) {
        float z = rd_u2f(b);
        if (!(z >= -130.0f && z <= 130.0f)) { if ((b & 0x7fffffffu) < 0x7f800000u && (b & 0x7fffffffu) > 0x43020000u) continue; }
        ++n2;
        if (rd_f2u(rd_exp2f(z)) != rd_f2u(ref_exp2f(z)) && !(z != z)) ++bad2;
    }
    printf("rd_exp2f (host) vs ref_exp2f on %llu inputs: %llu mismatches\n", n2, bad2);
    return (bad || bad2) ? 1 : 0;
}
