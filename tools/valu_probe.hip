// tools/valu_probe.hip -- does packed f32 (v_pk_fma_f32) pay on DEPENDENT chains, and with SGPR operands?
// Each lane runs ILP independent Horner-like chains of `len` dependent FMAs per iteration; scalar vs packed;
// multiplier/addend from kernel arguments (SGPRs) or from registers (VGPRs).  Full chip, 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -o tools/valu_probe tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int ILP, bool PACKED, bool VREG>
__global__ void __launch_bounds__(1024) chain(float *out, float a, float b, int iters)
{
    float av = a, bv = b;
    if (VREG) { asm volatile("" : "+v"(av), "+v"(bv)); }
    if (!PACKED) {
        float x[2 * ILP];
#pragma unroll
        for (int i = 0; i < 2 * ILP; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 2 * ILP; ++i) x[i] = __builtin_fmaf(x[i], av, bv);
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 2 * ILP; ++i) s += x[i];
        if (s == 1234.5678f) out[0] = s;
    } else {
        f2 x[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = f2{ (float)(threadIdx.x + i) * 1e-3f, (float)(threadIdx.x + i) * 2e-3f };
        const f2 a2 = { av, av }, b2 = { bv, bv };
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < ILP; ++i) x[i] = __builtin_elementwise_fma(x[i], a2, b2);
        }
        f2 s = { 0, 0 };
#pragma unroll
        for (int i = 0; i < ILP; ++i) s += x[i];
        if (s.x + s.y == 1234.5678f) out[0] = s.x;
    }
}

template <int ILP, bool PACKED, bool VREG>
static void run(const char *name, float *out, int blocks)
{
    const int iters = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((chain<ILP, PACKED, VREG>), dim3(blocks), dim3(1024), 0, 0, out, 0.999f, 0.001f, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((chain<ILP, PACKED, VREG>), dim3(blocks), dim3(1024), 0, 0, out, 0.999f, 0.001f, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4;
    const double values = (double)blocks * 1024 * 2 * ILP * 8 * iters;      // f32 FMAs (both forms process 2*ILP values per step)
    printf("%-44s blocks %4d: %8.1f us  %6.2f T fma/s\n", name, blocks, ms * 1e3, values / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out; CK(hipMalloc((void **)&out, 64));
    for (int blocks : { 512, 256 }) {
        run<1, false, false>("scalar  2 chains/lane  sgpr operands", out, blocks);
        run<1, true, false>("packed  1 pair chain   sgpr operands", out, blocks);
        run<1, false, true>("scalar  2 chains/lane  vgpr operands", out, blocks);
        run<1, true, true>("packed  1 pair chain   vgpr operands", out, blocks);
        run<3, false, false>("scalar  6 chains/lane  sgpr operands", out, blocks);
        run<3, true, false>("packed  3 pair chains  sgpr operands", out, blocks);
        run<3, true, true>("packed  3 pair chains  vgpr operands", out, blocks);
        run<8, false, false>("scalar 16 chains/lane  sgpr operands", out, blocks);
        run<8, true, false>("packed  8 pair chains  sgpr operands", out, blocks);
    }
    return 0;
}
