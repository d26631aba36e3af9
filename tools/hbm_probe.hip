// tools/hbm_probe.hip -- which trivial streaming kernel reaches the box's ceiling?  (round 4: rd_measure_hbm's first form,
// 2048 x 256 threads with a one-access grid-stride loop, measured 5.0 TB/s copy / 4.2 TB/s fill on a box whose develop
// kernel moves 5.8 TB/s -- the probe, not the box, was the limit.)  Variants: accesses in flight per lane (UNROLL),
// workgroup size, grid size, nt or plain stores, interleaved (grid-stride) or chunked (a wave walks its own range).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/hbm_probe tools/hbm_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int U, bool NT, bool CHUNK>
__global__ void k_copy(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n)
{
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (CHUNK) {                       // wave w owns [w * per, (w + 1) * per): U x 1 KiB per step
        const size_t nw = T / 64, w = t / 64, lane = t % 64, per = n / nw;
        for (size_t i = w * per; i + 64 * U <= (w + 1) * per; i += 64 * U) {
            f4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = s[i + k * 64 + lane];
#pragma unroll
            for (int k = 0; k < U; ++k) { if (NT) __builtin_nontemporal_store(v[k], d + i + k * 64 + lane); else d[i + k * 64 + lane] = v[k]; }
        }
    } else {
        for (size_t i = t; i + (U - 1) * T < n; i += U * T) {
            f4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = s[i + k * T];
#pragma unroll
            for (int k = 0; k < U; ++k) { if (NT) __builtin_nontemporal_store(v[k], d + i + k * T); else d[i + k * T] = v[k]; }
        }
    }
}
template <int U, bool NT, bool CHUNK>
__global__ void k_fill(f4 *__restrict__ d, size_t n, float x)
{
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const f4 v = { x, x + 1, x + 2, 1.0f };
    if (CHUNK) {
        const size_t nw = T / 64, w = t / 64, lane = t % 64, per = n / nw;
        for (size_t i = w * per; i + 64 * U <= (w + 1) * per; i += 64 * U) {
#pragma unroll
            for (int k = 0; k < U; ++k) { if (NT) __builtin_nontemporal_store(v, d + i + k * 64 + lane); else d[i + k * 64 + lane] = v; }
        }
    } else {
        for (size_t i = t; i + (U - 1) * T < n; i += U * T) {
#pragma unroll
            for (int k = 0; k < U; ++k) { if (NT) __builtin_nontemporal_store(v, d + i + k * T); else d[i + k * T] = v; }
        }
    }
}
template <int U, bool CHUNK>
__global__ void k_read(const f4 *__restrict__ s, size_t n, float *sink)
{
    const size_t T = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0;
    if (CHUNK) {
        const size_t nw = T / 64, w = t / 64, lane = t % 64, per = n / nw;
        for (size_t i = w * per; i + 64 * U <= (w + 1) * per; i += 64 * U) {
            f4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = s[i + k * 64 + lane];
#pragma unroll
            for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        }
    } else {
        for (size_t i = t; i + (U - 1) * T < n; i += U * T) {
            f4 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = s[i + k * T];
#pragma unroll
            for (int k = 0; k < U; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

static hipStream_t st; static hipEvent_t e0, e1;
template <typename F> static double med_us(F launch, int reps = 7)
{
    std::vector<float> ms;
    for (int r = 0; r <= reps; ++r) {
        hipEventRecord(e0, st); launch(); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1); if (r) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2] * 1e3;
}

int main()
{
    const size_t bytes = 1ull << 30, n = bytes / 16;
    f4 *a, *b; float *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# 1 GiB buffers; GB/s: copy counts read + written bytes; median of 7 launches\n");
#define ROW(name, GB, ...) { double us = med_us([&] { __VA_ARGS__; }); printf("%-44s grid %5d x %4d : %8.1f us  %7.1f GB/s\n", name, g, blk, us, GB * bytes / us / 1e3); }
    for (int blk : { 256, 1024 })
        for (int g : { 1024, 2048, 4096, 8192, 16384 }) {
            if ((size_t)g * blk > 8192u * 1024u) continue;
            ROW("copy U1 nt interleaved", 2.0, hipLaunchKernelGGL((k_copy<1, true, false>), dim3(g), dim3(blk), 0, st, a, b, n));
            ROW("copy U4 nt interleaved", 2.0, hipLaunchKernelGGL((k_copy<4, true, false>), dim3(g), dim3(blk), 0, st, a, b, n));
            ROW("copy U8 nt interleaved", 2.0, hipLaunchKernelGGL((k_copy<8, true, false>), dim3(g), dim3(blk), 0, st, a, b, n));
            ROW("copy U4 plain interleaved", 2.0, hipLaunchKernelGGL((k_copy<4, false, false>), dim3(g), dim3(blk), 0, st, a, b, n));
            ROW("copy U4 nt chunked", 2.0, hipLaunchKernelGGL((k_copy<4, true, true>), dim3(g), dim3(blk), 0, st, a, b, n));
            ROW("copy U8 nt chunked", 2.0, hipLaunchKernelGGL((k_copy<8, true, true>), dim3(g), dim3(blk), 0, st, a, b, n));
            ROW("fill U1 nt interleaved", 1.0, hipLaunchKernelGGL((k_fill<1, true, false>), dim3(g), dim3(blk), 0, st, b, n, 1.0f));
            ROW("fill U4 nt interleaved", 1.0, hipLaunchKernelGGL((k_fill<4, true, false>), dim3(g), dim3(blk), 0, st, b, n, 1.0f));
            ROW("fill U8 nt interleaved", 1.0, hipLaunchKernelGGL((k_fill<8, true, false>), dim3(g), dim3(blk), 0, st, b, n, 1.0f));
            ROW("fill U4 plain interleaved", 1.0, hipLaunchKernelGGL((k_fill<4, false, false>), dim3(g), dim3(blk), 0, st, b, n, 1.0f));
            ROW("fill U4 nt chunked", 1.0, hipLaunchKernelGGL((k_fill<4, true, true>), dim3(g), dim3(blk), 0, st, b, n, 1.0f));
            ROW("fill U8 nt chunked", 1.0, hipLaunchKernelGGL((k_fill<8, true, true>), dim3(g), dim3(blk), 0, st, b, n, 1.0f));
            ROW("read U4 interleaved", 1.0, hipLaunchKernelGGL((k_read<4, false>), dim3(g), dim3(blk), 0, st, a, n, sink));
            ROW("read U8 interleaved", 1.0, hipLaunchKernelGGL((k_read<8, false>), dim3(g), dim3(blk), 0, st, a, n, sink));
            ROW("read U8 chunked", 1.0, hipLaunchKernelGGL((k_read<8, true>), dim3(g), dim3(blk), 0, st, a, n, sink));
        }
    { int g = 0, blk = 0; ROW("hipMemcpyAsync D2D", 2.0, hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st));
      ROW("hipMemsetAsync", 1.0, hipMemsetAsync(b, 0, bytes, st)); }
    return 0;
}
