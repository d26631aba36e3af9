// tools/div_exhaustive.hip -- is ONE residual correction enough for the levels division?
//
//     y = RN(1/d);  q0 = RN(a*y);  r = RN(a - d*q0) (fma);  q1 = RN(q0 + r*y) (fma)       (rd_colour_n, shaders.rs:239)
//
// The kernels run the correction twice (q2 from q1 the same way).  Whether q1 already equals the IEEE quotient RN(a/d)
// depends only on the two significands: scaling a or d by a power of two scales every intermediate exactly (as long as
// nothing leaves the normal range, which the host-side guard of rd_uniforms.h establishes before it selects this path).
// So this program checks ALL 2^23 x 2^23 significand pairs, a and d in [1, 2) -- 7.04e13 divisions -- on the GPU.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o tools/div_exhaustive tools/div_exhaustive.hip
//   tools/div_exhaustive [first_d_slice [n_slices]]        (128 slices of 65 536 divisors; prints progress per slice)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct stats { unsigned long long bad0, bad1, bad2; unsigned int first_a, first_d; };

__device__ __forceinline__ float u2f(unsigned int u) { return __builtin_bit_cast(float, u); }

// one block per divisor d = 1.mantissa, 256 threads sweep all 2^23 dividends a = 1.mantissa
__global__ void __launch_bounds__(256) k(unsigned int d_first, stats *st)
{
    const unsigned int dm = d_first + blockIdx.x;
    const float d = u2f(0x3f800000u | dm);
    const float y = 1.0f / d;                                // IEEE (no fast-math): correctly rounded reciprocal
    unsigned int bad0 = 0, bad1 = 0, bad2 = 0, fa = 0xffffffffu;
    for (unsigned int am = threadIdx.x; am < (1u << 23); am += 256u) {
        const float a = u2f(0x3f800000u | am);
        const float want = a / d;                            // IEEE quotient
        const float q0 = a * y;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, a), y, q0);
        const float q2 = __builtin_fmaf(__builtin_fmaf(-d, q1, a), y, q1);
        if (__builtin_bit_cast(unsigned int, q0) != __builtin_bit_cast(unsigned int, want)) ++bad0;     // liveness of the check
        if (__builtin_bit_cast(unsigned int, q1) != __builtin_bit_cast(unsigned int, want)) { ++bad1; fa = fa < am ? fa : am; }
        if (__builtin_bit_cast(unsigned int, q2) != __builtin_bit_cast(unsigned int, want)) ++bad2;
    }
    atomicAdd(&st->bad0, (unsigned long long)bad0);
    if (bad1 | bad2) {
        atomicAdd(&st->bad1, (unsigned long long)bad1);
        atomicAdd(&st->bad2, (unsigned long long)bad2);
        if (bad1 && atomicCAS(&st->first_d, 0xffffffffu, dm) == 0xffffffffu) st->first_a = fa;
    }
}

int main(int argc, char **argv)
{
    const unsigned int first = argc > 1 ? (unsigned int)atoi(argv[1]) : 0u;
    const unsigned int count = argc > 2 ? (unsigned int)atoi(argv[2]) : 128u;
    stats *dst, st = { 0, 0, 0, 0xffffffffu, 0xffffffffu };
    CK(hipMalloc((void **)&dst, sizeof st));
    CK(hipMemcpy(dst, &st, sizeof st, hipMemcpyHostToDevice));
    for (unsigned int s = first; s < first + count && s < 128u; ++s) {
        hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, s << 16, dst);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&st, dst, sizeof st, hipMemcpyDeviceToHost));
        printf("slice %3u / 128 done: mismatches so far: no correction %llu, one correction %llu, two corrections %llu\n", s + 1,
               st.bad0, st.bad1, st.bad2);
        fflush(stdout);
    }
    printf("all significand pairs a, d in [1,2) of slices %u..%u (%.3g divisions): one correction: %llu mismatches", first,
           first + count - 1, (double)count * 65536.0 * 8388608.0, st.bad1);
    if (st.bad1) printf(" (first: a = 0x%08x, d = 0x%08x)", 0x3f800000u | st.first_a, 0x3f800000u | st.first_d);
    printf("; two corrections: %llu mismatches; RN(a * RN(1/d)) alone: %llu mismatches\n", st.bad2, st.bad0);
    return 0;
}
