// tools/f16_err_probe.hip -- how far the hardware pow (v_exp_f32(v_log_f32(x) / 2.2)) is from the pinned rd_gamma_clamp,
// relative, by |z| = |log2(x) / 2.2|, over every float in [FLT_MIN, 1): the data behind rd_f16_gamma's error bound.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -o tools/f16_err_probe tools/f16_err_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../raweditor_amd/csrc/rd_math.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k(uint32_t base, uint32_t *maxerr /* [64] in units of 2^-30 relative */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = rd_u2f(base + i);
    if (!(x >= RD_FLT_MIN) || !(x < 1.0f)) return;
    const float z = __builtin_amdgcn_logf(x) * RD_INV_GAMMA;
    const float e = __builtin_amdgcn_exp2f(z);
    const float g = rd_gamma_clamp(x);
    if (!(g > 0.0f)) return;
    const float rel = __builtin_fabsf(e - g) / g;
    uint32_t bin = (uint32_t)(-z);
    if (bin > 63u) bin = 63u;
    atomicMax(&maxerr[bin], (uint32_t)(rel * 1073741824.0f));
}

int main()
{
    uint32_t *d; CK(hipMalloc((void **)&d, 64 * 4)); CK(hipMemset(d, 0, 64 * 4));
    for (uint32_t c = 0; c < 0x3f8u; ++c)                      // encodings 0 .. 0x3f800000 in chunks of 2^20
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, c << 20, d);
    uint32_t h[64]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("|z| bin : max |e - g| / g in units of 2^-23 (f32 ulp at 1.0)\n");
    for (int b = 0; b < 64; ++b) if (h[b]) printf("  [%2d,%2d) : %.3f\n", b, b + 1, h[b] / 128.0);
    return 0;
}
