// tools/atomic_probe.hip -- throughput of returning device-scope atomic adds from a full persistent grid
// (work-queue design question: how many tile grabs per microsecond can one / a few counters serve?).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/atomic_probe tools/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every wave's lane 0 performs `per_wave` dependent atomic adds on counter[(wave_global % ncounters) * 32]
__global__ void __launch_bounds__(1024) probe(unsigned *counters, unsigned ncounters, unsigned per_wave, unsigned *sink)
{
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned acc = 0;
    if ((threadIdx.x & 63u) == 0) {
        unsigned *c = counters + (size_t)(wave % ncounters) * 32u;
        for (unsigned i = 0; i < per_wave; ++i)
            acc += __hip_atomic_fetch_add(c, 1u + (acc & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (acc == 0xdeadbeefu) sink[0] = acc;
    }
}

int main()
{
    unsigned *counters, *sink;
    CK(hipMalloc((void **)&counters, 1 << 20)); CK(hipMemset(counters, 0, 1 << 20)); CK(hipMalloc((void **)&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (unsigned blocks : { 512u, 64u }) {
        for (unsigned nc : { 1u, 8u, 64u, 512u, 8192u }) {
            const unsigned per_wave = 64;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 0, 0, counters, nc, per_wave, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double n = (double)blocks * 16 * per_wave;
                if (rep == 2) printf("blocks %4u counters %5u: %9.1f us for %8.0f atomics = %7.1f M atomics/s, %6.1f ns per atomic (aggregate)\n", blocks, nc, ms * 1e3, n, n / ms / 1e3, ms * 1e6 / n);
            }
        }
    }
    return 0;
}
