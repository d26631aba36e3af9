// tools/satomic_probe.hip -- is s_atomic_add (scalar, returns through lgkmcnt) coherent across XCDs on plain hipMalloc memory?
// Every wave grabs `per_wave` tickets from ONE counter; the tickets must be a permutation of 0..N-1.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/satomic_probe tools/satomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) grab(unsigned *counter, unsigned ncounters, unsigned per_wave, unsigned *tickets)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    unsigned *c = counter + (size_t)(wave % ncounters) * 32u;
    for (unsigned i = 0; i < per_wave; ++i) {
        unsigned t = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(c) : "memory");
        if ((threadIdx.x & 63u) == 0) tickets[(size_t)wave * per_wave + i] = t;
    }
}

int main()
{
    const unsigned blocks = 512, per_wave = 16, n = blocks * 16 * per_wave;
    unsigned *counter, *tickets;
    CK(hipMalloc((void **)&counter, 1 << 20)); CK(hipMalloc((void **)&tickets, (size_t)n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (unsigned nc : { 1u, 8u, 64u }) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(counter, 0, 1 << 20));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(grab, dim3(blocks), dim3(1024), 0, 0, counter, nc, per_wave, tickets);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned> h(n), cnt(nc);
            CK(hipMemcpy(h.data(), tickets, (size_t)n * 4, hipMemcpyDeviceToHost));
            std::vector<unsigned> fin(1 << 18); CK(hipMemcpy(fin.data(), counter, 1 << 20, hipMemcpyDeviceToHost));
            // per counter: the tickets of the waves that use it must be 0..k-1 exactly once
            size_t bad = 0;
            for (unsigned c = 0; c < nc; ++c) {
                std::vector<unsigned> v;
                for (unsigned w = c; w < blocks * 16; w += nc) for (unsigned i = 0; i < per_wave; ++i) v.push_back(h[(size_t)w * per_wave + i]);
                std::sort(v.begin(), v.end());
                for (size_t i = 0; i < v.size(); ++i) bad += v[i] != i;
                bad += fin[(size_t)c * 32] != v.size();
            }
            if (rep == 2) printf("counters %3u: %8.1f us, %6.1f ns per atomic, %zu ticket errors\n", nc, ms * 1e3, ms * 1e6 / n, bad);
        }
    }
    return 0;
}
