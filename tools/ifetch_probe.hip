// tools/ifetch_probe.hip -- does VALU throughput drop when the loop body is long (instruction fetch bound)?
// Loop bodies of N independent FMAs (8 accumulators round-robin), 4-byte (v_fmac_f32_e32) or 8-byte (v_fma_f32 VOP3 /
// v_fmaak_f32 with literal) encodings, full chip at 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/ifetch_probe tools/ifetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N, int ENC>
__global__ void __launch_bounds__(1024) k(float *out, float a, float b, int iters)
{
    float av = a, bv = b;
    asm volatile("" : "+v"(av), "+v"(bv));
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < N / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (ENC == 4) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
                if (ENC == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
                if (ENC == 9) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(x[i]) : "v"(av));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 1234.5678f) out[0] = s;
}

template <int N, int ENC>
static void run(float *out, int blocks)
{
    const int iters = (1 << 20) / N;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<N, ENC>), dim3(blocks), dim3(1024), 0, 0, out, 0.999f, 0.001f, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<N, ENC>), dim3(blocks), dim3(1024), 0, 0, out, 0.999f, 0.001f, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double instr = (double)blocks * 16 * N * iters;
    printf("body %5d instr x %d B = %6.1f KiB, blocks %d: %6.2f ns per wave-instr per SIMD  (%5.2f T lane-op/s)\n", N, ENC == 4 ? 4 : 8,
           N * (ENC == 4 ? 4 : 8) / 1024.0, blocks, ms * 1e6 / (instr / 1024), instr * 64 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out; CK(hipMalloc((void **)&out, 64));
    for (int blocks : { 512, 256 }) {
        run<64, 4>(out, blocks); run<512, 4>(out, blocks); run<2048, 4>(out, blocks); run<4096, 4>(out, blocks); run<8192, 4>(out, blocks);
        run<64, 8>(out, blocks); run<512, 8>(out, blocks); run<2048, 8>(out, blocks); run<4096, 8>(out, blocks); run<8192, 8>(out, blocks);
        run<2048, 9>(out, blocks); run<8192, 9>(out, blocks);
    }
    return 0;
}
