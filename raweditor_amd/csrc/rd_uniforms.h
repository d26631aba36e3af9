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
    return u;
}
