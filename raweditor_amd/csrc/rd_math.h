// rd_math.h -- the pinned transcendental pair of the develop path, host + device.
//
// WGSL pow() (reference src/gpu/shaders.rs:217, :261) has no bit-level definition: Vulkan
// drivers lower it to exp2(y*log2(x)) on v_log_f32/v_exp_f32.  librawdev pins that lowering on
// one polynomial pair (coefficients from tools/fit_pow.py) evaluated with explicit FMAs in a
// fixed order, so the surface is reproducible bit for bit on any IEEE-754 machine.
// DESIGN.md section 3 is the normative text; this file is its product implementation (the test
// oracle under oracle/ restates it separately and is never included from here).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RD_HD __host__ __device__ __forceinline__

#define RD_SQRT_HALF_BITS 0x3f3504f3u
#define RD_FLT_MIN 1.17549435e-38f
#define RD_INV_GAMMA 0.45454547f /* f32(1.0/2.2), shaders.rs:261 */

RD_HD uint32_t rd_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RD_HD float rd_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// log2(x), x >= FLT_MIN (finite or +inf): x = m*2^e with m in [sqrt(1/2), sqrt(2)), t = m-1,
// log2(x) = fma(t, P(t), e), P of degree 7 in Horner form.
RD_HD float rd_log2f(float x)
{
    uint32_t ix = rd_f2u(x) - RD_SQRT_HALF_BITS;
    int32_t e = (int32_t)ix >> 23;
    float m = rd_u2f((ix & 0x007fffffu) + RD_SQRT_HALF_BITS);
    float t = m - 1.0f;
    float p = -0x1.2a7c18p-3f;
    p = __builtin_fmaf(p, t, 0x1.e526cap-3f);
    p = __builtin_fmaf(p, t, -0x1.001218p-2f);
    p = __builtin_fmaf(p, t, 0x1.2596a4p-2f);
    p = __builtin_fmaf(p, t, -0x1.70bab4p-2f);
    p = __builtin_fmaf(p, t, 0x1.ec7b64p-2f);
    p = __builtin_fmaf(p, t, -0x1.7155bap-1f);
    p = __builtin_fmaf(p, t, 0x1.715472p+0f);
    return __builtin_fmaf(t, p, (float)e);
}

// 2^z for z in [-126, 128): n = rint(z) (ties to even), f = z-n, Q(f) of degree 6, exponent add.
// rint is taken with the add-magic idiom: for |z| < 2^22, t = RN(z + 1.5*2^23) carries the integer nearest to z (ties to
// even, the FPU's own rounding) in its low mantissa bits, t - 1.5*2^23 is that integer exactly, and because the low nine
// bits of the magic constant's encoding are zero, bits(t) << 23 == (uint32)n << 23.  Same n, same f, same result as
// rintf + float->int conversion, in four instructions, three of them full-rate adds (v_rndne_f32 and v_cvt_i32_f32 are
// half rate on gfx950).  No fast-math anywhere in this project, so (z + c) - c is not folded.
RD_HD float rd_exp2f_core(float z)
{
    const float magic = 12582912.0f;           // 1.5 * 2^23 = 0x4b400000
    float t = z + magic;
    float n = t - magic;
    float f = z - n;
    float p = 0x1.43e9d6p-13f;
    p = __builtin_fmaf(p, f, 0x1.5f4e2ep-10f);
    p = __builtin_fmaf(p, f, 0x1.3b2a72p-7f);
    p = __builtin_fmaf(p, f, 0x1.c6aec2p-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfbep-3f);
    p = __builtin_fmaf(p, f, 0x1.62e43p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return rd_u2f(rd_f2u(p) + (rd_f2u(t) << 23));
}

// The same value as rd_exp2f_core(z) for z in (-126, 127), as the PRODUCT p * 2^n instead of an add into p's exponent field:
// 2^n = bits 0x3f800000 + (n << 23) is a normal float there, and multiplying the normal p by a power of two that keeps the
// result normal is exact -- the same bits.  What it buys on the device: the multiply takes the [0, 1] clamp as an output
// modifier, so rd_gamma_clamp's min(., 1) costs nothing (v_lshl_add_u32 + v_mul_f32 clamp instead of v_lshl_add_u32 + v_min_u32).
RD_HD float rd_exp2f_core_clamp01(float z)
{
    const float magic = 12582912.0f;
    float t = z + magic;
    float n = t - magic;
    float f = z - n;
    float p = 0x1.43e9d6p-13f;
    p = __builtin_fmaf(p, f, 0x1.5f4e2ep-10f);
    p = __builtin_fmaf(p, f, 0x1.3b2a72p-7f);
    p = __builtin_fmaf(p, f, 0x1.c6aec2p-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfbep-3f);
    p = __builtin_fmaf(p, f, 0x1.62e43p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    const float r = p * rd_u2f((rd_f2u(t) << 23) + 0x3f800000u);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(r, 0.0f, 1.0f);            // folded into the multiply's clamp modifier (r is positive or NaN)
#else
    return r < 1.0f ? r : 1.0f;                                // (a NaN r belongs to an x the caller replaces by 0)
#endif
}

// Full-domain 2^z: NaN -> NaN, z >= 128 -> +inf, z < -126 -> 0 (sub-FLT_MIN results flush).
RD_HD float rd_exp2f(float z)
{
    if (z != z) return z;
    if (z >= 128.0f) return __builtin_inff();
    if (z < -126.0f) return 0.0f;
    return rd_exp2f_core(z);
}

// clamp(pow(x, 1/2.2), 0, 1) with the develop path's conventions (shaders.rs:261-264):
//   x < 0 or NaN -> pow = NaN -> clamp -> 0;  0 <= x < FLT_MIN -> 0;  large x / +inf -> 1.
// For x >= FLT_MIN, z = y*log2(x) lies in (-58, 59): inside rd_exp2f_core's domain, so the
// general guards of rd_exp2f are provably dead here and are left out.
RD_HD float rd_gamma_clamp(float x)
{
    float z = RD_INV_GAMMA * rd_log2f(x);
    const float v = rd_exp2f_core_clamp01(z);  // min(2^z, 1): v is a positive finite float whenever it is used
    return (x >= RD_FLT_MIN) ? v : 0.0f;
}
