// rd_uniforms.h -- the kernel-side uniform block and its derivation on the host.
//
// The reference uploads a 128-byte uniform block (src/gpu/pipeline.rs:17-46) and every fragment
// recomputes the uniform-only sub-expressions of fs_main (shaders.rs:200-205, :217, :233, :239,
// :245).  Here those sub-expressions are evaluated ONCE per frame on the host, in f32, with the
// exact operations the shader text spells (this TU is compiled with -ffp-contract=off), and reach
// the kernels as kernel arguments (SGPRs).  IEEE-754 makes host and device agree bit for bit.
#pragma once

#include "../../include/rawdev.h"
#include "rd_math.h"

struct rd_ku {
    float wb_r, wb_g, wb_b;      // shaders.rs:195 (wb_multipliers.rgb)
    float kr, kg, kb;            // 1+0.3*temperature, 1+0.3*tint, 1-0.3*temperature (:200-205)
    float m[9];                  // host row-major, consumed as columns (:209-214)
    float em;                    // pow(2, exposure) (:217)
    float highlights, shadows;   // :226, :230
    float cf;                    // 1 + contrast/100 (:233)
    float blacks, den;           // :239  den = (whites-blacks)+0.0001
    float rden;                  // RN(1/den), for the correctly rounded reciprocal division
    uint32_t fast_div;           // den and 1/den are normal finite numbers
    float s, oms;                // 1 + saturation/100 and 1-s (:245-247)
    float vibrance;              // :254
    float zoom, pan_x, pan_y;    // :46-51
    uint32_t black_level;        // extension (0 = reference)
    uint32_t elide;              // RD_EL_*: steps that are exact identities for THESE uniforms (export kernel)
};

// Steps of the colour stack the export kernel may skip because, for the uniforms of this frame, they return their input
// bit for bit: x*1 is x for every x; with a zero slider the factor 1 + (finite * 0) is exactly 1; an identity matrix gives
// (1*r + 0*g) + 0*b = r.  The "finite" is the catch (0 * inf is NaN), so everything except the x*1 cases is only flagged
// when rd_make_ku can bound every intermediate of the stack far below FLT_MAX from the uniforms and the input range
// [0, 16).  The sign of a zero may differ from the literal evaluation; it cannot reach the surface (the contrast step
// maps both zeros to -0.5, the gamma step maps both to +0).  The oracle never skips anything.
enum {
    RD_EL_K = 1u,        // temperature = tint = 0: the three factors are 1          (shaders.rs:200-205)
    RD_EL_MAT = 2u,      // identity colour matrix -- the only one the reference ever passes (color.rs:43-47)
    RD_EL_EM = 4u,       // exposure = 0: pow(2, 0) = 1                               (:217-218)
    RD_EL_HL = 8u,       // highlights = 0                                           (:226)
    RD_EL_SH = 16u,      // shadows = 0                                              (:230)
    RD_EL_SAT = 32u,     // saturation = 0: mix(Y, c, 1)                             (:245-247)
    RD_EL_VIB = 64u,     // vibrance = 0: mix(Y2, c, 1)                              (:251-257)
    RD_EL_FIX = 128u,    // levels divide: the numerator is zero or finite and far from BOTH exponent limits and the
                         // denominator is within 2^+-40, so no intermediate of the one-correction quotient of rd_colour_n
                         // leaves the normal range (the exhaustive significand proof applies) and none of the IEEE special
                         // cases (NaN, inf, zero denominators, exponent overflow) can occur
    RD_EL_BLK = 256u,    // blacks = 0: c - 0 = c                                    (:239)
    // every step that mixes the channels of a pixel is an identity: the stack acts on r, g and b separately
    RD_EL_SEPARABLE = RD_EL_MAT | RD_EL_HL | RD_EL_SH | RD_EL_SAT | RD_EL_VIB,
};

static inline rd_ku rd_make_ku(const rd_edit_params &p, const float wb[4], const float cm[9],
                               float zoom, float pan_x, float pan_y, uint32_t black_level,
                               uint32_t math_mode = RD_MATH_STRICT)
{
    rd_ku u;
    u.wb_r = wb[0]; u.wb_g = wb[1]; u.wb_b = wb[2];
    if (math_mode == RD_MATH_CONTRACTED) {           // a*b+c of the shader text -> one fma
        u.kr = __builtin_fmaf(p.temperature, 0.3f, 1.0f);
        u.kb = __builtin_fmaf(-p.temperature, 0.3f, 1.0f);
        u.kg = __builtin_fmaf(p.tint, 0.3f, 1.0f);
    } else {
        u.kr = 1.0f + p.temperature * 0.3f;
        u.kb = 1.0f - p.temperature * 0.3f;
        u.kg = 1.0f + p.tint * 0.3f;
    }
    for (int i = 0; i < 9; ++i) u.m[i] = cm[i];
    // pow(2.0, e) = exp2(e * log2(2.0)), log2(2.0) == 1 exactly in the pinned pair.
    u.em = rd_exp2f(p.exposure * rd_log2f(2.0f));
    u.highlights = p.highlights;
    u.shadows = p.shadows;
    u.cf = 1.0f + (p.contrast / 100.0f);
    u.blacks = p.blacks;
    u.den = (p.whites - p.blacks) + 0.0001f;
    u.rden = 1.0f / u.den;
    {
        const float ad = __builtin_fabsf(u.den), ay = __builtin_fabsf(u.rden);
        u.fast_div = (ad >= RD_FLT_MIN && ad <= 3.0e38f && ay >= RD_FLT_MIN && ay <= 3.0e38f) ? 1u : 0u;
    }
    u.s = 1.0f + (p.saturation / 100.0f);
    u.oms = 1.0f - u.s;
    u.vibrance = p.vibrance;
    u.zoom = zoom; u.pan_x = pan_x; u.pan_y = pan_y;
    u.black_level = black_level;
    u.elide = 0u;
    if (u.kr == 1.0f && u.kg == 1.0f && u.kb == 1.0f) u.elide |= RD_EL_K;
    if (u.em == 1.0f) u.elide |= RD_EL_EM;
    {   // magnitude bounds through the stack, in double; any NaN makes a comparison false and leaves the flags clear
        const double in = 16.0;                                    // 65535 / 4096 < 16
        const double ar = in * __builtin_fabs((double)u.wb_r) * __builtin_fabs((double)u.kr);
        const double ag = in * __builtin_fabs((double)u.wb_g) * __builtin_fabs((double)u.kg);
        const double ab = in * __builtin_fabs((double)u.wb_b) * __builtin_fabs((double)u.kb);
        double a = 0.0;
        for (int row = 0; row < 3; ++row) {
            const double v = __builtin_fabs((double)u.m[row]) * ar + __builtin_fabs((double)u.m[3 + row]) * ag +
                             __builtin_fabs((double)u.m[6 + row]) * ab;
            a = v > a ? v : a;
        }
        a *= __builtin_fabs((double)u.em);                         // after :217-218; |L| <= 1.0001 a
        const double hl = 1.0 + 1.0001 * a * __builtin_fabs((double)u.highlights);
        const double sh = 1.0 + (1.0 + 1.0001 * a) * __builtin_fabs((double)u.shadows);
        const double a5 = a * hl * sh;                             // after :230
        const double a6 = (a5 + 0.5) * __builtin_fabs((double)u.cf) + 0.5;
        const double a7 = (a6 + __builtin_fabs((double)u.blacks)) * __builtin_fabs((double)u.rden) * 1.0001;
        const double a8 = a7 * 1.0001 * __builtin_fabs((double)u.oms) + a7 * __builtin_fabs((double)u.s);
        const double lim = 1.0e30;
        const bool front = ar < lim && ag < lim && ab < lim && a < lim;
        const bool identity = u.m[0] == 1.0f && u.m[4] == 1.0f && u.m[8] == 1.0f && u.m[1] == 0.0f && u.m[2] == 0.0f &&
                              u.m[3] == 0.0f && u.m[5] == 0.0f && u.m[6] == 0.0f && u.m[7] == 0.0f;
        if (front && identity) u.elide |= RD_EL_MAT;
        if (front && u.highlights == 0.0f) u.elide |= RD_EL_HL;
        if (front && u.shadows == 0.0f) u.elide |= RD_EL_SH;
        if (front && a5 < lim && a6 < lim && a7 < lim && u.fast_div && u.s == 1.0f && u.oms == 0.0f) u.elide |= RD_EL_SAT;
        if (front && a5 < lim && a6 < lim && a7 < lim && a8 < lim && u.fast_div && u.vibrance == 0.0f) u.elide |= RD_EL_VIB;
        // Lower limit of the numerator a = v - blacks, v = (c - 0.5) * cf + 0.5 (shaders.rs:233-239, never skipped in strict
        // mode): t + 0.5 is exact and a multiple of 2^-25 for t in [-1, -0.25] and at least 0.25 in magnitude otherwise,
        // so v is 0 or |v| >= 2^-25, a multiple of 2^-48.  With blacks = 0 or |blacks| >= 2^-50 (a multiple of 2^-73) a is
        // 0 or |a| >= 2^-73; with |den| <= 2^40 the quotient stays above 2^-114 and a non-zero residual a - den*q, a multiple
        // of 2^(E_a - 47), above 2^-120: all normal.
        const double ad = __builtin_fabs((double)u.den), abk = __builtin_fabs((double)u.blacks);
        const bool num_ok = u.blacks == 0.0f || (abk >= 0x1p-50 && abk < lim);
        if (front && a5 < lim && a6 < lim && a7 < lim && u.fast_div && ad >= 0x1p-40 && ad <= 0x1p40 && num_ok) u.elide |= RD_EL_FIX;
        if (u.blacks == 0.0f) u.elide |= RD_EL_BLK;
    }
    return u;
}
