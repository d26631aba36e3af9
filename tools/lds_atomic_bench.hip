// LDS atomic throughput probe (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) k(uint32_t *out, int iters, uint32_t seed)
{
    __shared__ uint32_t lh[768 * 32];
    for (uint32_t i = threadIdx.x; i < 768 * 32; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, copy = lane & 31;
    uint32_t x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        uint32_t bin = (x >> 24);                 // random 0..255
        if (MODE == 0) atomicAdd(&lh[bin * 32 + copy], 1u);                    // product layout
        if (MODE == 1) atomicAdd(&lh[(it & 255) * 32 + copy], 1u);             // uniform bin: lanes l, l+32 same address
        if (MODE == 2) atomicAdd(&lh[(it & 127) * 64 + lane], 1u);             // 64 distinct addresses, distinct banks per half
        if (MODE == 3) acc += atomicAdd(&lh[bin * 32 + copy], 1u);             // returning
        if (MODE == 4) lh[(it & 127) * 64 + lane] = x;                         // plain ds_write_b32
        if (MODE == 5) atomicAdd(&lh[bin * 32 + (lane >> 1)], 1u);             // pairs share address
        if (MODE == 6) { if (lane < 32) atomicAdd(&lh[bin * 32 + copy], 1u); } // half wave active
        if (MODE == 7) atomicAdd((unsigned long long *)&lh[(bin * 32 + copy) & ~1u], 1ull); // u64
    }
    __syncthreads();
    uint32_t s = acc;
    for (uint32_t i = threadIdx.x; i < 768 * 32; i += blockDim.x) s += lh[i];
    if (s == 0x12345678u) out[0] = s;
}

int main()
{
    uint32_t *out; CK(hipMalloc((void **)&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[8] = { "ds_add_u32 random bin, copy=lane%32", "ds_add_u32 same bin, copy=lane%32", "ds_add_u32 64 distinct addrs",
                             "ds_add_rtn_u32 random bin", "ds_write_b32", "ds_add_u32 pairs share addr", "ds_add_u32 half wave", "ds_add_u64 random" };
    const int iters = 4096, blocks = 256;
    for (int m = 0; m < 8; ++m) {
        auto launch = [&]() {
            switch (m) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1u); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1u); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1u); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1u); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1u); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1u); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1u); break;
            default: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(1024), 0, 0, out, iters, 1u); break;
            }
        };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 2;
        double wave_instrs_per_cu = 16.0 * iters;   // one block per CU, 16 waves
        printf("%-40s %8.1f us  -> %6.1f ns per wave-instr per CU (%.1f cycles @2.1GHz)\n", names[m], ms * 1e3,
               ms * 1e6 / wave_instrs_per_cu, ms * 1e6 / wave_instrs_per_cu * 2.1);
    }
    return 0;
}
