// rd_kernels.h -- gfx950 kernels of the develop path (compiled only with --offload-arch=gfx950).
//
//   rd_develop_quads : the export hot path.  Full-resolution identity map (shaders.rs:184-187 with
//                      zoom 1 / pan 0), nearest-neighbour demosaic (shaders.rs:104-169), colour
//                      stack + gamma + clamp (shaders.rs:192-266), surface store and the 3x256
//                      histogram (pipeline.rs:720-736) in ONE launch per frame.
//   rd_develop_map   : any target size / zoom / pan (preview 1280 px, histogram 128 px; the
//                      point-sampling map of shaders.rs:23-60), same arithmetic, one pixel per lane.
//   rd_hist_u8       : calculate_histogram on an RGBA8 buffer (pipeline.rs:720-736).
//   rd_reduce_slab*  : fold the per-workgroup histogram slabs.
//
// Why there is no LDS stencil tile: the reference's demosaic only ever reads
// the 2x2 block {rows 2k-1, 2k} x {cols 2q, 2q+1} for the four output pixels of that same block
// (selection table of shaders.rs:127-155 with the py+1 parity shift).  One lane owns one such block:
// every CFA sample is loaded from HBM exactly once, straight into a register, and the four output
// pixels need only THREE colour evaluations (the odd-row pair is the same triple; the even-row
// pair shares r and g).  LDS is spent on the histogram instead.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rd_math.h"
#include "rd_uniforms.h"

#ifndef RD_BLOCK
#define RD_BLOCK 1024       // 16 waves; two workgroups per CU = 8 waves per SIMD (tools/microbench.hip may override)
#endif
#define RD_MAX_BLOCKS 1024  // slab capacity (workgroups per launch)
#ifndef RD_HK
#define RD_HK 8             // private histogram copies per bin: lane l adds into copy l % RD_HK
#endif

// Third value of the kernels' MATH parameter beside rd_math_mode's two (rawdev.h): NO arithmetic.  The export kernel's
// f32 instance with every memory instruction, ticket, sweep and LDS stage of the real one and the colour stack, the gamma
// and the histogram removed -- the surfaces receive the raw samples as floats.  Reachable only through
// rd_batch_probe_pattern (a measurement aid: the memory pattern's own ceiling on this box, in these buffers); no render
// entry point accepts it.
#define RD_MATH_PROBE 2

typedef float rd_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t rd_u2 __attribute__((ext_vector_type(2)));
typedef _Float16 rd_h2 __attribute__((ext_vector_type(2)));
// Frame pointers that a multi-frame launch reads from its descriptor array carry no address space of their own (hipcc
// would emit flat_load / flat_store for them, which count in vmcnt AND lgkmcnt and retire out of order: every wait
// becomes vmcnt(0)).  All pixel traffic therefore goes through explicitly global pointers.
#define RD_GLOBAL __attribute__((address_space(1)))

struct rd_rgb { float r, g, b; };

// The elision flags of a frame (rd_uniforms.h).  tools/isa_budget.py compiles the kernels with -DRD_BUDGET_ELIDE=<flags> to
// pin them at compile time: the assembly then holds exactly ONE path through the colour stack (the one a frame with those
// flags takes), so its static instruction count is that frame's dynamic count.  Never defined in the product build.
__device__ __forceinline__ uint32_t rd_elide_of(const rd_ku &u)
{
#ifdef RD_BUDGET_ELIDE
    (void)u;
    return RD_BUDGET_ELIDE;
#else
    return u.elide;
#endif
}

// ---------------------------------------------------------------------------------------------
// The colour stack (shaders.rs:192-266): rd_colour_n below; rd_dot709 is its Rec.709 luma (:222, :243, :256).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rd_dot709(float r, float g, float b)
{
    return ((r * 0.2126f) + (g * 0.7152f)) + (b * 0.0722f);
}

// a / u.den, correctly rounded (shaders.rs:239).  The denominator is uniform, so its correctly rounded reciprocal u.rden
// comes from the host, and the quotient is y = RN(1/d), q0 = RN(a*y), then residual corrections q' = RN(q + RN(a - d*q)*y)
// instead of the ~14-instruction generic v_div_scale/v_rcp/v_div_fmas expansion.
//   * RD_EL_FIX set (rd_uniforms.h proves a is zero or far from both exponent limits and d is within 2^+-40): ONE correction,
//     3 FMAs.  Whether q1 == RN(a/d) depends on the two significands only (powers of two scale every intermediate exactly
//     while nothing leaves the normal range), and tools/div_exhaustive.hip checks ALL 2^23 x 2^23 significand pairs on the
//     GPU: 0 mismatches in 7.04e13 divisions (profiles/r02_div_exhaustive.txt).
//   * otherwise, u.fast_div (d and 1/d normal finite numbers): two corrections (Markstein) and v_div_fixup_f32 for the IEEE
//     special cases (a = 0/inf/NaN); tools/div_check.c: 10^9 samples equal.
//   * otherwise the generic divide.
// (rd_levels_divide below: all values of a tile behind one branch.)

// v / u.den for M values (shaders.rs:239), the variants described above; MATH = RD_MATH_CONTRACTED: v * RN(1/den).
template <int M, int MATH>
__device__ __forceinline__ void rd_levels_divide(const rd_ku &u, float (&v)[M])
{
    if (MATH == RD_MATH_CONTRACTED) {
#pragma unroll
        for (int k = 0; k < M; ++k) v[k] = v[k] * u.rden;
    } else if (rd_elide_of(u) & RD_EL_FIX) {                     // one residual correction (proof: above)
#pragma unroll
        for (int k = 0; k < M; ++k) {
            const float t = v[k] * u.rden;
            v[k] = __builtin_fmaf(__builtin_fmaf(-u.den, t, v[k]), u.rden, t);
        }
    } else if (u.fast_div) {                                     // two corrections + the IEEE special cases
#pragma unroll
        for (int k = 0; k < M; ++k) {
            const float a = v[k];
            float t = a * u.rden;
            float e = __builtin_fmaf(-u.den, t, a);
            t = __builtin_fmaf(e, u.rden, t);
            e = __builtin_fmaf(-u.den, t, a);
            v[k] = __builtin_amdgcn_div_fixupf(__builtin_fmaf(e, u.rden, t), u.den, a);
        }
    } else {
#pragma unroll
        for (int k = 0; k < M; ++k) v[k] = v[k] / u.den;
    }
}

__device__ __forceinline__ float rd_dot709_c(float r, float g, float b)      // the same dot product, contracted
{
    return __builtin_fmaf(b, 0.0722f, __builtin_fmaf(g, 0.7152f, r * 0.2126f));
}

// shaders.rs:192-266 on N demosaiced triples, advanced together stage by stage.  MATH = RD_MATH_STRICT: the literal
// operation order of the WGSL text (this TU is compiled with -ffp-contract=off, so nothing fuses; the divide is the
// IEEE-correct one).  MATH = RD_MATH_CONTRACTED: the same text with every a*b+c contracted to one fma and the levels
// division done as x*RN(1/d) -- the lowering an AMD shader compiler applies to the reference's WGSL (DESIGN.md section
// 3b; oracle: colour_stack_contracted); ~62 VALU + 3 pow per triple instead of ~105 + 3 pow.  Every stage that u.elide
// (rd_uniforms.h) marks as an exact identity for this frame's uniforms is skipped behind ONE wave-uniform branch for all N.  With all sliders away
// from their defaults nothing is skipped; a typical edit (identity matrix -- the reference never passes another one --
// and a few untouched sliders) sheds a quarter of the linear part.  Results are written back into r, g, b.
// GAMMA = false stops before the gamma / clamp step (:261-264) and returns the linear values: the 8-bit surfaces finish
// with rd_q8_gamma below.
// The front of the stack: white balance, temperature / tint, colour matrix (:195-214) -- the part in which a 2x2 block's three
// triples share products (the compiler's common-subexpression pass finds them: 30 instead of 45 matrix instructions).
template <int N, int MATH>
__device__ __forceinline__ void rd_colour_front(const rd_ku &u, float (&r)[N], float (&g)[N], float (&b)[N])
{
    constexpr bool C = MATH == RD_MATH_CONTRACTED;
    const uint32_t el = rd_elide_of(u);
#pragma unroll
    for (int i = 0; i < N; ++i) { r[i] = r[i] * u.wb_r; g[i] = g[i] * u.wb_g; b[i] = b[i] * u.wb_b; }          // :195
    if (!(el & RD_EL_K)) {
#pragma unroll
        for (int i = 0; i < N; ++i) { r[i] = r[i] * u.kr; b[i] = b[i] * u.kb; g[i] = g[i] * u.kg; }            // :200-205
    }
    if (!(el & RD_EL_MAT)) {
#pragma unroll
        for (int i = 0; i < N; ++i) {                                                                          // :209-214
            float x, y, z;
            if (C) {
                x = __builtin_fmaf(u.m[6], b[i], __builtin_fmaf(u.m[3], g[i], u.m[0] * r[i]));
                y = __builtin_fmaf(u.m[7], b[i], __builtin_fmaf(u.m[4], g[i], u.m[1] * r[i]));
                z = __builtin_fmaf(u.m[8], b[i], __builtin_fmaf(u.m[5], g[i], u.m[2] * r[i]));
            } else {
                x = ((u.m[0] * r[i]) + (u.m[3] * g[i])) + (u.m[6] * b[i]);
                y = ((u.m[1] * r[i]) + (u.m[4] * g[i])) + (u.m[7] * b[i]);
                z = ((u.m[2] * r[i]) + (u.m[5] * g[i])) + (u.m[8] * b[i]);
            }
            r[i] = x; g[i] = y; b[i] = z;
        }
    }
}

// The rest: exposure ... vibrance (:217-257) and, with GAMMA, the gamma / clamp step (:261-264).
template <int N, int MATH, bool GAMMA = true>
__device__ __forceinline__ void rd_colour_tail(const rd_ku &u, float (&r)[N], float (&g)[N], float (&b)[N])
{
    constexpr bool C = MATH == RD_MATH_CONTRACTED;
    const uint32_t el = rd_elide_of(u);
    if (!(el & RD_EL_EM)) {
#pragma unroll
        for (int i = 0; i < N; ++i) { r[i] = r[i] * u.em; g[i] = g[i] * u.em; b[i] = b[i] * u.em; }            // :217-218
    }
    if ((el & (RD_EL_HL | RD_EL_SH)) != (RD_EL_HL | RD_EL_SH)) {
        float L[N];
#pragma unroll
        for (int i = 0; i < N; ++i) L[i] = C ? rd_dot709_c(r[i], g[i], b[i]) : rd_dot709(r[i], g[i], b[i]);    // :222
        if (!(el & RD_EL_HL)) {
#pragma unroll
            for (int i = 0; i < N; ++i) {                                                                      // :226
                const float hl = C ? __builtin_fmaf(L[i], u.highlights, 1.0f) : 1.0f + (L[i] * u.highlights);
                r[i] = r[i] * hl; g[i] = g[i] * hl; b[i] = b[i] * hl;
            }
        }
        if (!(el & RD_EL_SH)) {
#pragma unroll
            for (int i = 0; i < N; ++i) {                                                                      // :230
                const float sh = C ? __builtin_fmaf(1.0f - L[i], u.shadows, 1.0f) : 1.0f + ((1.0f - L[i]) * u.shadows);
                r[i] = r[i] * sh; g[i] = g[i] * sh; b[i] = b[i] * sh;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                                                              // :233-234
        if (C) {
            r[i] = __builtin_fmaf(r[i] - 0.5f, u.cf, 0.5f);
            g[i] = __builtin_fmaf(g[i] - 0.5f, u.cf, 0.5f);
            b[i] = __builtin_fmaf(b[i] - 0.5f, u.cf, 0.5f);
        } else {
            r[i] = (r[i] - 0.5f) * u.cf + 0.5f;
            g[i] = (g[i] - 0.5f) * u.cf + 0.5f;
            b[i] = (b[i] - 0.5f) * u.cf + 0.5f;
        }
    }
    if (!(el & RD_EL_BLK)) {
#pragma unroll
        for (int i = 0; i < N; ++i) { r[i] = r[i] - u.blacks; g[i] = g[i] - u.blacks; b[i] = b[i] - u.blacks; }   // :239
    }
    {
        float q[3 * N];
#pragma unroll
        for (int i = 0; i < N; ++i) { q[3 * i] = r[i]; q[3 * i + 1] = g[i]; q[3 * i + 2] = b[i]; }
        rd_levels_divide<3 * N, MATH>(u, q);
#pragma unroll
        for (int i = 0; i < N; ++i) { r[i] = q[3 * i]; g[i] = q[3 * i + 1]; b[i] = q[3 * i + 2]; }
    }
    if (!(el & RD_EL_SAT)) {
#pragma unroll
        for (int i = 0; i < N; ++i) {                                                                          // :243-247
            const float Y = C ? rd_dot709_c(r[i], g[i], b[i]) : rd_dot709(r[i], g[i], b[i]);
            const float ys = Y * u.oms;
            if (C) { r[i] = __builtin_fmaf(r[i], u.s, ys); g[i] = __builtin_fmaf(g[i], u.s, ys); b[i] = __builtin_fmaf(b[i], u.s, ys); }
            else { r[i] = ys + r[i] * u.s; g[i] = ys + g[i] * u.s; b[i] = ys + b[i] * u.s; }
        }
    }
    if (!(el & RD_EL_VIB)) {
#pragma unroll
        for (int i = 0; i < N; ++i) {                                                                          // :251-257
            const float sat = __builtin_fmaxf(r[i], __builtin_fmaxf(g[i], b[i])) - __builtin_fminf(r[i], __builtin_fminf(g[i], b[i]));
            if (C) {
                const float a2 = __builtin_fmaf(u.vibrance, 1.0f - sat, 1.0f);
                const float Y2 = rd_dot709_c(r[i], g[i], b[i]);
                const float yv = Y2 * (1.0f - a2);
                r[i] = __builtin_fmaf(r[i], a2, yv); g[i] = __builtin_fmaf(g[i], a2, yv); b[i] = __builtin_fmaf(b[i], a2, yv);
            } else {
                const float va = u.vibrance * (1.0f - sat);
                const float Y2 = rd_dot709(r[i], g[i], b[i]);
                const float a2 = 1.0f + va;
                const float yv = Y2 * (1.0f - a2);
                r[i] = yv + r[i] * a2; g[i] = yv + g[i] * a2; b[i] = yv + b[i] * a2;
            }
        }
    }
    if constexpr (GAMMA) {
#pragma unroll
        for (int i = 0; i < N; ++i) { r[i] = rd_gamma_clamp(r[i]); g[i] = rd_gamma_clamp(g[i]); b[i] = rd_gamma_clamp(b[i]); }   // :261-264
    }
}

template <int N, int MATH, bool GAMMA = true>
__device__ __forceinline__ void rd_colour_n(const rd_ku &u, float (&r)[N], float (&g)[N], float (&b)[N])
{
    rd_colour_front<N, MATH>(u, r, g, b);
    rd_colour_tail<N, MATH, GAMMA>(u, r, g, b);
}

// The stack for a frame whose channel-mixing steps are all exact identities (RD_EL_SEPARABLE: identity matrix,
// highlights = shadows = vibrance = 0, saturation = 0): what is left of rd_colour_n acts on each channel alone -- white
// balance, temperature / tint, exposure, contrast, levels; the same operations in the same order, so the same bits -- and a
// 2x2 block's three triples (C,A,B), (C,D,A), (C,D,B) hold only five distinct values: v = { r of C; g of A, D; b of B, A }.
template <int MATH>
__device__ __forceinline__ void rd_colour_separable(const rd_ku &u, float (&v)[5])
{
    constexpr bool C = MATH == RD_MATH_CONTRACTED;
    const uint32_t el = rd_elide_of(u);
    v[0] = v[0] * u.wb_r; v[1] = v[1] * u.wb_g; v[2] = v[2] * u.wb_g; v[3] = v[3] * u.wb_b; v[4] = v[4] * u.wb_b;       // :195
    if (!(el & RD_EL_K)) {                                                                                         // :200-205
        v[0] = v[0] * u.kr; v[3] = v[3] * u.kb; v[4] = v[4] * u.kb; v[1] = v[1] * u.kg; v[2] = v[2] * u.kg;
    }
    if (!(el & RD_EL_EM)) {
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = v[k] * u.em;                                                            // :217-218
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)                                                                                    // :233-234
        v[k] = C ? __builtin_fmaf(v[k] - 0.5f, u.cf, 0.5f) : (v[k] - 0.5f) * u.cf + 0.5f;
    if (!(el & RD_EL_BLK)) {
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = v[k] - u.blacks;                                                        // :239
    }
    rd_levels_divide<5, MATH>(u, v);
}

template <int MATH>
__device__ __forceinline__ rd_rgb rd_colour_m(const rd_ku &u, float r, float g, float b)      // one triple
{
    float tr[1] = { r }, tg[1] = { g }, tb[1] = { b };
    rd_colour_n<1, MATH>(u, tr, tg, tb);
    return rd_rgb{ tr[0], tg[0], tb[0] };
}

// f32(raw)/4096 (shaders.rs:106-110, :167-168) with the optional integer black level.
__device__ __forceinline__ float rd_norm(uint32_t raw, uint32_t bl)
{
    raw = raw > bl ? raw - bl : 0u;
    return (float)raw * (1.0f / 4096.0f);
}

// The same for black level 0 (the reference has none) without a conversion: 0x45000000 is 2^11, whose ulp is
// 2^-12, so the encoding 0x45000000 | raw IS the float 2^11 + raw * 2^-12 (raw < 2^23), and subtracting 2^11 leaves
// raw / 4096 exactly -- one or + one full-rate subtract instead of v_cvt_f32_u32 (half rate) + multiply.
__device__ __forceinline__ float rd_norm0(uint32_t raw16) { return rd_u2f(raw16 | 0x45000000u) - 2048.0f; }

// Rgba8Unorm quantisation (pipeline.rs:322), pinned as trunc(RN(x*255) + 0.5).  x is always a clamped gamma value in
// [0, 1], and for every float in [0, 1] one fma gives the same integer as the multiply-then-add
// (tools/q8_fma_check.c: all 1 065 353 217 encodings; tests/test_host_cpu.py runs it), so the code costs one
// instruction less.
__device__ __forceinline__ uint32_t rd_q8(float x) { return (uint32_t)__builtin_fmaf(x, 255.0f, 0.5f); }

// rd_q8(rd_gamma_clamp(x)) for the surfaces that keep only the 8-bit code (RGBA8 -- the reference's own target format,
// pipeline.rs:322 -- and RGB8), in a third of the instructions.  The code is a step function of x with 255 steps; between
// two steps ANY evaluation of pow(x, 1/2.2) that is accurate to a small fraction of a code gives the same answer.  So:
// the hardware's v_log_f32 / v_exp_f32 (about 1 ULP each; the lowering the reference's Vulkan drivers use, rd_math.h)
// give e = 2^(log2(x) / 2.2), whose 255 e lies within 3.05e-5 codes of the pinned 255 g for every float
// (tools/q8_exhaustive.hip measures the largest distance over all 2^32 encodings, profiles/r02_q8_exhaustive.txt), and
// only a lane whose 255 e lies within RD_Q8_EPS (8x that) of a half-integer -- where the two could round to different
// codes -- takes the pinned evaluation (about one lane in 2000; the branch is skipped when no lane of the wave needs it).
//
// Round 3: the same decision in instructions that issue at the full rate (tools/isa_budget.py prices them; the round-2
// form spent 46 issue cycles per value, this one 28 -- profiles/r03_isa_budget_*.txt):
//   * e is clamped to [0, 1] by the OUTPUT MODIFIER of v_exp_f32 (no v_min; with the HSA default DX10_CLAMP a NaN -- a
//     negative x -- becomes 0, so every x below FLT_MIN, zero, negative or NaN gives e = 0 and code 0 without a compare
//     and select of its own);
//   * the code is taken by the add-magic idiom: t = RN(255 e + 2^23) is ONE fma and carries round-to-nearest(255 e) in the
//     low byte of its encoding (0x4b0000qq) -- no v_cvt_u32_f32, no v_fract; nearest-even instead of the pin's
//     trunc(. + 0.5) differs only on exact ties, which lie inside the zone the pinned evaluation decides;
//   * the distance to the nearest half-integer is d = fma(e, 255, -(t - 2^23)) (exact difference, rounded once).
// Every float encoding is checked against rd_q8(rd_gamma_clamp(x)) on the device (rd_selftest_q8, tests/test_gpu_q8.py
// in the -m gpu suite) and against the oracle's pow + clamp + pack on the host by tools/q8_exhaustive.hip.
#ifndef RD_Q8_EPS
#define RD_Q8_EPS 0.00025f
#endif
#define RD_MAGIC23 8388608.0f               /* 2^23 = 0x4b000000: RN(v + 2^23) carries RN(v) in its low mantissa bits */
#define RD_MAGIC23_BITS 0x4b000000u

// Lane constants of the shortcuts.  A VOP3 instruction takes no literal on gfx9 and an SGPR source halves the issue rate of
// v_fma_f32 (tools/valu_probe2.hip), so the export kernel parks these in VGPRs once per wave (rd_kc_parked); the map
// kernel and the self-tests let the compiler place them.
struct rd_kc {
    float k255;                                                  // 255.0
    float f16_ka, f16_kb;                                        // RD_F16_KA / 64, RD_F16_KB / 64 (rd_f16_gamma)
};

// e = clamp(2^(log2(x) / 2.2), 0, 1) on the hardware pair; z = log2(x) / 2.2.  fmed3(., 0, 1) is the clamp idiom hipcc
// folds into the producing instruction's output modifier (v_exp_f32_e64 ... clamp).
__device__ __forceinline__ float rd_hw_gamma01(float x, float &z)
{
    z = __builtin_amdgcn_logf(x) * RD_INV_GAMMA;
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(z), 0.0f, 1.0f);
}

// The 8-bit code of x as the ENCODING 0x4b0000qq (qq = the code): the histogram address and the pixel pack take their
// byte from it without a conversion.
__device__ __forceinline__ uint32_t rd_q8_gamma_bits(float x, const rd_kc &kc)
{
    float z;
    const float e = rd_hw_gamma01(x, z);
    const float t = __builtin_fmaf(e, kc.k255, RD_MAGIC23);      // RN(255 e + 2^23): in [2^23, 2^23 + 255]
    const float r = t - RD_MAGIC23;                              // the rounded code as a float (exact)
    const float d = __builtin_fmaf(e, kc.k255, -r);              // 255 e - code, |d| <= 0.5
    uint32_t tb = rd_f2u(t);
    if (__builtin_fabsf(d) > 0.5f - RD_Q8_EPS) tb = RD_MAGIC23_BITS | rd_q8(rd_gamma_clamp(x));
    return tb;
}

__device__ __forceinline__ uint32_t rd_q8_gamma(float x)
{
    const rd_kc kc = { 255.0f, 0.0f, 0.0f };
    return rd_q8_gamma_bits(x, kc) & 0xffu;
}

// ---------------------------------------------------------------------------------------------
// Round 4: the 8-bit code from a THRESHOLD TABLE in LDS (the export kernel's RGBA8 / RGB8 surfaces; RD_Q8_LUT=0 builds the
// round-3 transcendental shortcut above instead, for A/B).
//
// q(x) = rd_q8(rd_gamma_clamp(x)) is a monotone step function of x with 255 steps (tools/q8_monotone.hip walks all 2^31
// non-negative encodings in order: 0 decreases; the first step is at x = 1.105e-6, the last at 0.99569), and no 2^16
// consecutive float encodings hold more than ONE step.  So the code is  base[bucket] + (x >= threshold[bucket])  with
// bucket = the top 16 bits of the encoding -- and with three tricks the whole thing is three full-rate VALU instructions,
// one shift-and-mask and one ds_read_b32 instead of v_log + v_mul + v_exp + 2 fma + sub + compare (28 -> ~10 issue cycles):
//   * xc = clamp(x, 0, 1) (negative, NaN -> 0; the clamp is an output modifier of the instruction that produced x where
//     the compiler can fold it, one v_med3_f32 otherwise), then w = xc * 2^-96: an exact power-of-two scaling that moves
//     [2^-30, 1] onto the exponent fields 1 .. 31, so the top 16 bits of w ARE the table index 0 .. 0xF80 -- no lower clamp,
//     no subtraction (everything below 2^-30 lands in the denormal buckets 0 .. 127, which all hold code 0);
//   * an entry is E = (base << 16) + (0x10000 - t) - (bucket << 16), t = the threshold's offset inside the bucket (0x10000:
//     no step): then s = E + bits(w) = (base << 16) + (0x10000 - t) + (low 16 bits of w) carries "low bits >= t" into bit 16
//     by itself -- one v_add_u32, no compare, no select -- and bits 16..23 of s are the code (bits 24..31 are zero);
//   * the pixel pack (v_perm_b32) and the histogram address take the code from byte 2 of s; it is never extracted.
// The table (3969 words = 15.5 KiB) is built on the host from rd_gamma_clamp itself (rd_q8_lut_build: 255 bisections),
// lives in a __device__ array and is copied into LDS when a workgroup starts: 24 KiB histogram + 15.5 KiB table leave two
// workgroups per CU.  rd_selftest_q8_lut runs it against the pinned function for ALL 2^32 encodings on the device.
// ---------------------------------------------------------------------------------------------
#ifndef RD_Q8_LUT
#define RD_Q8_LUT 1
#endif
#define RD_Q8_LUT_WORDS 3969u                /* buckets 0 .. 0xF80 (w = 2^-96: x = 1.0) */
#define RD_Q8_LUT_LDS_WORDS 3972u            /* the LDS copy, padded to whole 16-byte loads */
#define RD_Q8_LUT_SCALE 0x1p-96f             /* 0x0f800000 */
#define RD_Q8_LUT_REBIAS 0x30000000u         /* bits(x) - bits(x * 2^-96) for normal results: 96 << 23 */

__device__ __attribute__((aligned(16))) uint32_t rd_q8_lut_dev[RD_Q8_LUT_WORDS + 63u];       // filled by the host before the first launch (rawdev.hip)

// Host: the table from the pinned function.  thr[k-1] = the smallest encoding whose code is >= k (bisection: q is monotone).
static inline void rd_q8_lut_build(uint32_t *lut /* RD_Q8_LUT_WORDS */)
{
    uint32_t thr[255];
    for (uint32_t k = 1; k <= 255u; ++k) {
        uint32_t lo = 0u, hi = 0x3f800000u;                  // q(lo) < k <= q(hi)
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            const uint32_t q = (uint32_t)__builtin_fmaf(rd_gamma_clamp(rd_u2f(mid)), 255.0f, 0.5f);
            if (q >= k) hi = mid; else lo = mid;
        }
        thr[k - 1u] = hi - RD_Q8_LUT_REBIAS;                 // as an encoding of w = x * 2^-96 (every threshold is far above 2^-30)
    }
    uint32_t k = 0;                                          // thresholds at or below the bucket start = the code there
    for (uint32_t b = 0; b < RD_Q8_LUT_WORDS; ++b) {
        const uint32_t start = b << 16, end = start + 0x10000u;
        while (k < 255u && thr[k] <= start) ++k;
        uint32_t t = 0x10000u;
        if (k < 255u && thr[k] < end) t = thr[k] - start;    // at most one step per bucket (tools/q8_monotone.hip)
        lut[b] = (k << 16) + (0x10000u - t) - start;
    }
}

// s: the code of x in bits 16..23 (bits 24..31 zero, bits 0..15 a by-product).  `lut` is the LDS copy of the table.
__device__ __forceinline__ uint32_t rd_q8_lut_bits(float x, const uint32_t *lut)
{
    const float xc = __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f);     // negative, NaN -> 0 (DX10_CLAMP / min3); > 1 -> 1
    const uint32_t wb = rd_f2u(xc * RD_Q8_LUT_SCALE);
    const uint32_t a = (uint32_t)((int32_t)wb >> 14) & 0x3ffcu;  // bucket * 4 (w >= 0: the arithmetic shift is the one measured at full rate)
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lut) + a) + wb;
}

// The export kernel copies the table into LDS once per workgroup.
__device__ __forceinline__ void rd_q8_lut_load(uint32_t *lut)
{
    typedef uint32_t rd_u4v __attribute__((ext_vector_type(4)));       // 16 bytes per lane: a quarter of the instructions (the load is
    const rd_u4v *src = reinterpret_cast<const rd_u4v *>(rd_q8_lut_dev);   // paid by every launch, however small)
    rd_u4v *dst = reinterpret_cast<rd_u4v *>(lut);
    for (uint32_t i = threadIdx.x; i < RD_Q8_LUT_LDS_WORDS / 4u; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// The same shortcut for the RGBA-f16 surface (BASELINE config 5): what leaves the kernel is the binary16 rounding of the
// pinned gamma (and, with the fused histogram, its 8-bit code).  The hardware pow e = 2^z, z = log2(x) / 2.2, is within
// (1.2 |z| + 2.25) * 2^-23 * e of the pinned value -- its error is that of z, which grows with |log2 x|
// (tools/f16_err_probe.hip: largest relative distance 1.41 * 2^-23 for |z| < 1, 3.9 below 7, 12.1 below 29, 23.2 beyond;
// the bound keeps 1.5x that) -- i.e. within K = 2.4 |z| + 4.5 of its own f32 ulps.  binary16 keeps 10 of the 23 fraction
// bits and rounds to nearest, so the pinned value and e round to the same half unless the 13 discarded bits of e lie
// within K of the midpoint 0x1000; only then (about one lane in 800 for bright pixels), and in binary16's subnormal
// range, the pinned evaluation decides.  The function returns a FLOAT whose binary16 rounding is the pinned half (e, or
// the pinned g for the deciding lanes), so the caller converts pairs with one v_cvt_pk_f16_f32.
// Round 3, same decisions at the full issue rate (68 -> 48 cycles per value): the clamp is v_exp_f32's output modifier
// (e = 0 for every x below FLT_MIN, zero, negative or NaN: half 0 and code 0 with no compare of their own); the 13 low
// bits are compared as (bits - 4096) / 64 against K / 64, K / 64 = clamp(|z| * KA/64 + KB/64) by the fma's output
// modifier, so an infinite z (x = 0) no longer sends a wave of black pixels through the pinned evaluation (K > 64 only
// where binary16 is subnormal anyway); "0 < e < 2^-14" is one integer compare on bits(e) - 1.
// Checked for all 2^32 encodings against binary16(rd_gamma_clamp(x)) by rd_selftest_f16 / tests/test_gpu_q8.py.
#define RD_F16_KA 2.4f
#define RD_F16_KB 4.5f
template <bool WANT_Q>
__device__ __forceinline__ float rd_f16_gamma_value(float x, const rd_kc &kc, uint32_t &tb)
{
    float z;
    float e = rd_hw_gamma01(x, z);
    const uint32_t eb = rd_f2u(e);
    // (the 13 discarded fraction bits - 4096) / 64, exact: 2^17 + bits * 2^-6 is representable, minus (2^17 + 64)
    const float lowm = rd_u2f((eb & 0x1fffu) | 0x48000000u) - 131136.0f;
    const float k = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_fabsf(z), kc.f16_ka, kc.f16_kb), 0.0f, 1.0f);
    bool near = __builtin_fabsf(lowm) <= k;                      // within K of a rounding midpoint
    near = near || (eb - 1u) < 0x387fffffu;                     // 0 < e < 2^-14: a subnormal half
    tb = 0u;
    if (WANT_Q) {
        const float t = __builtin_fmaf(e, kc.k255, RD_MAGIC23);
        const float r = t - RD_MAGIC23;
        const float d = __builtin_fmaf(e, kc.k255, -r);
        tb = rd_f2u(t);
        near = near || __builtin_fabsf(d) > 0.5f - RD_Q8_EPS;
    }
    if (near) {
        e = rd_gamma_clamp(x);
        if (WANT_Q) tb = RD_MAGIC23_BITS | rd_q8(e);
    }
    return e;
}

// binary16 pair (lo, hi) of two such values: one v_cvt_pk_f16_f32 (gfx950; round to nearest even)
__device__ __forceinline__ uint32_t rd_pack_h2(float lo, float hi)
{
    typedef float rd_f2v __attribute__((ext_vector_type(2)));
    const rd_f2v v = { lo, hi };
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rd_h2));
}

template <bool WANT_Q>
__device__ __forceinline__ void rd_f16_gamma(float x, uint32_t &h, uint32_t &q)      // self-tests: half and code of one value
{
    const rd_kc kc = { 255.0f, RD_F16_KA / 64.0f, RD_F16_KB / 64.0f };
    uint32_t tb;
    const float e = rd_f16_gamma_value<WANT_Q>(x, kc, tb);
    h = __builtin_bit_cast(uint16_t, (_Float16)e);
    q = tb & 0xffu;
}

// ---------------------------------------------------------------------------------------------
// Round 4, second table: binary16(gamma(x)) AND the 8-bit code from two-level THRESHOLD TABLES in LDS (the export kernel's
// RGBA-f16 surface; RD_F16_LUT=0 builds the transcendental shortcut above instead, for A/B).
//
// h(x) = binary16(rd_gamma_clamp(x)) is a step function of x with 15 360 steps on [0, 1]; tools/f16_monotone.hip walks every
// encoding: ONE decrease in the whole range (0x3eefb555: the half dips 0x39ab -> 0x39aa for that single encoding), and for
// x >= 2^-16 no 2^13 consecutive encodings hold more than one step (except that dip's bucket).  With w = clamp01(x) * 2^-110
// -- exact; [2^-16, 1] lands on exponent fields 1 .. 17, everything smaller is a denormal w -- the tables are
//   * FINE, one u16 per 2^13 encodings (index bits(w) >> 13 = 0 .. 17408, 34 KiB): ((d - j + 6) << 13) + (0x2000 - t), d = steps
//     between the start of the bucket's group of eight and the bucket, j = the bucket's place in the group, t = the step's
//     offset in the bucket (none: 0x2000) -- d - j is in [-6, 0] because a bucket holds at most one step and seven at least one;
//   * COARSE, one {E, C} pair per 2^16 encodings (index bits(w) >> 16, 17 KiB): E = the 8-bit code's entry exactly as in
//     rd_q8_lut_build (for this scaling), C = (the half at the group's start) - 6 - 8 * group.
// Then  half = ((fine + bits(w)) >> 13) + C  -- the add carries "low 13 bits >= t" into bit 13 by itself and the un-masked
// high bits of w are the bucket index, which C takes out again -- and  s = E + bits(w)  carries the code in bits 16..23:
// multiply, two shift-and-mask pairs, three adds and a shift per value (16 issue cycles + two LDS reads) for what the
// shortcut spends 38 + 10 on (v_log, v_exp, midpoint test, subnormal test, code, tie test).  Two kinds of lane take the pinned
// evaluation instead: 0 < x < 2^-16 (7839 of the steps live down there, and w is a denormal -- or 0 below 2^-40 -- that no
// longer tells them apart: bits(x) - 1 < bits(2^-16) - 1, one subtraction and one unsigned compare) and the dip's one encoding
// (one v_cmp_eq).  x = 0 (and every negative or NaN x: the clamp) is w = 0: bucket 0, half 0, code 0.
// rd_f16_lut_build derives everything from rd_gamma_clamp itself and REFUSES (the library then fails loudly) if a bucket
// holds two steps or the dips are not exactly the one the kernel tests for; rd_selftest_f16_lut runs the device code against
// the pinned function for all 2^32 encodings.
// ---------------------------------------------------------------------------------------------
#ifndef RD_F16_LUT
#define RD_F16_LUT 1
#endif
#define RD_F16_LUT_SCALE 0x1p-110f           /* 0x08800000 */
#define RD_F16_LUT_REBIAS 0x37000000u        /* bits(x) - bits(x * 2^-110) for normal results: 110 << 23 */
#define RD_F16_LUT_NF 17409u                 /* fine buckets 0 .. 0x4400 (w = 2^-110: x = 1.0) */
#define RD_F16_LUT_NC 2177u                  /* coarse buckets 0 .. 0x880 */
#define RD_F16_LUT_DIP_X 0x3eefb555u         /* the one encoding where binary16(gamma(x)) is below its predecessor's */
#define RD_F16_LUT_FINE_LDS 17416u            /* u16 entries of the LDS copy: RD_F16_LUT_NF padded to whole 16-byte loads */
#define RD_F16_LUT_COARSE_LDS 4356u          /* words of the LDS copy: 2 * RD_F16_LUT_NC padded likewise */

__device__ __attribute__((aligned(16))) uint32_t rd_f16_fine_dev[(RD_F16_LUT_NF + 1u) / 2u + 64u];    // u16 pairs; filled by the host before the first launch
__device__ __attribute__((aligned(16))) uint32_t rd_f16_coarse_dev[RD_F16_LUT_NC * 2u + 64u];         // {E, C} pairs

// binary16 of a float in [0, 1], round to nearest even (host side of the builder; the device converts with v_cvt_f16_f32)
static inline uint32_t rd_f16_bits_host(float g)
{
    const uint32_t b = rd_f2u(g);
    const int32_t e = (int32_t)(b >> 23) - 127;
    uint32_t m = (b & 0x7fffffu) | 0x800000u;
    if (b == 0u) return 0u;
    if (e < -25) return 0u;                                       // below half of the smallest subnormal
    uint32_t shift, base;
    if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); base = 0u; }          // subnormal half: value = m * 2^(e-23), unit 2^-24
    else { shift = 13u; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1u);
    if (rem > halfway || (rem == halfway && (q & 1u))) q += 1u;   // a carry out of the fraction moves into the exponent: still right
    return base + q;
}

// Host: both tables from the pinned function.  fine: RD_F16_LUT_NF + 1 entries, coarse: 2 * RD_F16_LUT_NC words.  Returns 0,
// or a negative code when the function's shape is not what the kernel's lookup assumes (-1: two steps in one fine bucket,
// -2: a field out of range, -3: the set of dips is not { RD_F16_LUT_DIP_X }, -4: out of memory).
static inline int rd_f16_lut_build(uint16_t *fine, uint32_t *coarse)
{
    auto H = [](uint32_t xb) { return rd_f16_bits_host(rd_gamma_clamp(rd_u2f(xb))); };
    auto Q = [](uint32_t xb) { return (uint32_t)__builtin_fmaf(rd_gamma_clamp(rd_u2f(xb)), 255.0f, 0.5f); };
    const uint32_t top = 0x3f800000u, hmax = H(top);              // 0x3c00
    // thrH[h - 1] = the smallest encoding from which the half is >= h (first up-crossing), dips listed apart
    uint32_t *thrH = new (std::nothrow) uint32_t[hmax];
    if (!thrH) return -4;
    uint32_t dips[8], ndips = 0;
    for (uint32_t h = 1; h <= hmax; ++h) {
        uint32_t lo = 0u, hi = top;                               // H(lo) < h <= H(hi)
        while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (H(mid) >= h) hi = mid; else lo = mid; }
        for (uint32_t k = 1; k <= 8u && hi > k;) {              // a dip just below: take the FIRST crossing
            if (H(hi - k) >= h) { hi -= k; k = 1; } else ++k;
        }
        thrH[h - 1u] = hi;
        for (uint32_t k = 1; k <= 16u && hi + k <= top; ++k)
            if (H(hi + k) < h) { bool seen = false; for (uint32_t d = 0; d < ndips; ++d) seen |= dips[d] == hi + k; if (!seen && ndips < 8u) dips[ndips++] = hi + k; }
    }
    uint32_t thrQ[255];
    for (uint32_t k = 1; k <= 255u; ++k) {
        uint32_t lo = 0u, hi = top;
        while (hi - lo > 1u) { const uint32_t mid = lo + (hi - lo) / 2u; if (Q(mid) >= k) hi = mid; else lo = mid; }
        thrQ[k - 1u] = hi;
    }
    int rc = (ndips == 1u && dips[0] == RD_F16_LUT_DIP_X) ? 0 : -3;
    const uint32_t first_normal = 1u << 10;                       // fine bucket of w = 2^-126 (x = 2^-16)
    for (uint32_t i = 0; i <= RD_F16_LUT_NF; ++i) fine[i] = 0u;
    for (uint32_t c = 0; c < RD_F16_LUT_NC; ++c) { coarse[2u * c] = 0u; coarse[2u * c + 1u] = 0u; }
    uint32_t kh = 0, kq = 0;
    for (uint32_t c = first_normal >> 3; c < RD_F16_LUT_NC; ++c) {
        const uint32_t xs_c = (c << 16) + RD_F16_LUT_REBIAS;
        while (kh < hmax && thrH[kh] <= xs_c) ++kh;               // kh = the half at the group's start
        while (kq < 255u && thrQ[kq] <= xs_c) ++kq;
        uint32_t tq = 0x10000u;
        if (kq < 255u && thrQ[kq] - xs_c < 0x10000u) tq = thrQ[kq] - xs_c;
        coarse[2u * c] = (kq << 16) + (0x10000u - tq) - (c << 16);
        coarse[2u * c + 1u] = kh - 6u - 8u * c;
        uint32_t k = kh;
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t i = c * 8u + j;
            if (i >= RD_F16_LUT_NF) break;
            const uint32_t xs = (i << 13) + RD_F16_LUT_REBIAS;
            while (k < hmax && thrH[k] <= xs) ++k;
            uint32_t t = 0x2000u;
            if (k < hmax && thrH[k] - xs < 0x2000u) {
                t = thrH[k] - xs;
                if (k + 1u < hmax && thrH[k + 1u] - xs < 0x2000u && rc == 0) rc = -1;
            }
            const int32_t field = (int32_t)(k - kh) - (int32_t)j + 6;
            if ((field < 0 || field > 6) && rc == 0) rc = -2;
            fine[i] = (uint16_t)(((uint32_t)field << 13) + (0x2000u - t));
        }
    }
    delete[] thrH;
    return rc;
}

#define RD_F16_LUT_DIP_W (RD_F16_LUT_DIP_X - RD_F16_LUT_REBIAS)

// half: binary16(gamma(x)) as an integer (bits 16..31 zero); s: the 8-bit code in bits 16..23 (rd_q8_lut_bits' format);
// pinned: this lane needs the pinned evaluation instead (0 < x < 2^-16, or the dip).  fine / coarse: the LDS copies.
// xc: clamp01(x), what the pinned evaluation of such a lane starts from (the same result as from x, and x's only use is then
// the clamp, which the compiler folds into the instruction that produced x).
__device__ __forceinline__ void rd_f16_lut_lookup(float x, const uint16_t *fine, const uint32_t *coarse, uint32_t &half, uint32_t &s, bool &pinned, float &xc)
{
    typedef uint32_t rd_u2v __attribute__((ext_vector_type(2)));
    xc = __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f);                 // negative, NaN -> 0; > 1 -> 1
    const float w = xc * RD_F16_LUT_SCALE;                       // exact
    const uint32_t wb = rd_f2u(w);
    uint32_t t1 = wb >> 12, t2 = wb >> 13;
    asm("" : "+v"(t1), "+v"(t2));                                 // keep them shifts (v_bfe_u32 issues at half rate)
    const uint32_t fe = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(fine) + (t1 & 0xfffeu));
    const rd_u2v ce = *reinterpret_cast<const rd_u2v *>(reinterpret_cast<const char *>(coarse) + (t2 & 0xfff8u));
    uint32_t sum = fe + wb;
    asm("" : "+v"(sum));
    half = (sum >> 13) + ce.y;
    s = ce.x + wb;
    pinned = (rd_f2u(xc) - 1u) < 0x377fffffu || wb == RD_F16_LUT_DIP_W;           // 0 < xc < 2^-16 (one unsigned compare), or the dip
}

// The export kernel copies both tables into LDS once per workgroup.
__device__ __forceinline__ void rd_f16_lut_load(uint16_t *fine, uint32_t *coarse)
{
    typedef uint32_t rd_u4v __attribute__((ext_vector_type(4)));       // 16 bytes per lane (see rd_q8_lut_load)
    rd_u4v *f = reinterpret_cast<rd_u4v *>(fine), *c = reinterpret_cast<rd_u4v *>(coarse);
    const rd_u4v *fs = reinterpret_cast<const rd_u4v *>(rd_f16_fine_dev), *cs = reinterpret_cast<const rd_u4v *>(rd_f16_coarse_dev);
    for (uint32_t i = threadIdx.x; i < RD_F16_LUT_FINE_LDS / 8u; i += blockDim.x) f[i] = fs[i];
    for (uint32_t i = threadIdx.x; i < RD_F16_LUT_COARSE_LDS / 4u; i += blockDim.x) c[i] = cs[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Histogram: RD_HK private copies of every bin in LDS, copy = lane % RD_HK, so a flat frame (all 64
// lanes in one bin) serialises 64/RD_HK-deep on an address instead of 64-deep.  RD_HK = 8 keeps the
// table at 24 KiB, which together with the 48 KiB store-transpose stage lets TWO 1024-thread
// workgroups share a CU (32 waves); measured on a constant frame it costs nothing (LDS atomics are
// ~5 cycles per wave-instruction and far from the bottleneck), RD_HK = 32 (96 KiB, one workgroup per
// CU) was 10 % slower overall.  Flushed once per workgroup, WITHOUT global atomics, into that
// workgroup's slab row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rd_hist_zero(uint32_t *lh)
{
    for (uint32_t i = threadIdx.x; i < 768u * RD_HK; i += blockDim.x) lh[i] = 0u;
    __syncthreads();
}

__device__ __forceinline__ void rd_hist_add(uint32_t *lh, uint32_t copy, uint32_t qr, uint32_t qg,
                                            uint32_t qb, uint32_t inc)
{
    atomicAdd(&lh[(qr)*RD_HK + copy], inc);
    atomicAdd(&lh[(256u + qg) * RD_HK + copy], inc);
    atomicAdd(&lh[(512u + qb) * RD_HK + copy], inc);
}

// The same three adds for codes that arrive as add-magic encodings 0x4b0000qq (rd_q8_gamma_bits): the byte address of
// bin q, copy c is q * 4 RD_HK + 4 c, and (bits << s) + base with base = 4 c - (0x4b000000 << s) (mod 2^32) is exactly that
// -- one v_lshl_add_u32 per bin, no extraction of the code; the channel offset rides in the ds instruction's offset field.
// `hbase` = rd_hist_base(copy), once per wave.
#define RD_HIST_SHIFT (2 + (RD_HK == 8 ? 3 : RD_HK == 4 ? 2 : RD_HK == 16 ? 4 : RD_HK == 32 ? 5 : 0))
static_assert(RD_HK == 4 || RD_HK == 8 || RD_HK == 16 || RD_HK == 32, "RD_HK must be 4, 8, 16 or 32");
__device__ __forceinline__ uint32_t rd_hist_base(uint32_t copy) { return copy * 4u - (RD_MAGIC23_BITS << RD_HIST_SHIFT); }

__device__ __forceinline__ void rd_hist_add_bits(uint32_t *lh, uint32_t hbase, uint32_t tr, uint32_t tg, uint32_t tb, uint32_t inc)
{
    char *base = reinterpret_cast<char *>(lh);
    atomicAdd(reinterpret_cast<uint32_t *>(base + ((tr << RD_HIST_SHIFT) + hbase)), inc);
    atomicAdd(reinterpret_cast<uint32_t *>(base + ((tg << RD_HIST_SHIFT) + hbase)) + 256u * RD_HK, inc);
    atomicAdd(reinterpret_cast<uint32_t *>(base + ((tb << RD_HIST_SHIFT) + hbase)) + 512u * RD_HK, inc);
}

// The same for codes that sit in bits 16..23 of s (rd_q8_lut_bits; bits 24..31 zero, bits 0..15 arbitrary): the byte address
// of bin q, copy c is ((s >> 16) << RD_HIST_SHIFT) + 4 c -- a full-rate shift and one v_lshl_add_u32 (6 issue cycles; the
// add-magic encodings above need 4).  The empty asm keeps the shift a shift: left to itself hipcc fuses it into
// v_bfe_u32 + v_lshl_or_b32, two VOP3 integer instructions that issue at half rate on this part (8 cycles;
// tools/valu_probe2.hip, profiles/r04_valu_probe_lut.txt: v_lshrrev_b32 1.14 ns, v_bfe_u32 1.79 ns per wave-instruction).
__device__ __forceinline__ uint32_t rd_sar16(uint32_t s)
{
    uint32_t c = s >> 16;
    asm("" : "+v"(c));
    return c;
}
__device__ __forceinline__ void rd_hist_add_b2(uint32_t *lh, uint32_t copy4, uint32_t sr, uint32_t sg, uint32_t sb, uint32_t inc)
{
    char *base = reinterpret_cast<char *>(lh);
    atomicAdd(reinterpret_cast<uint32_t *>(base + ((rd_sar16(sr) << RD_HIST_SHIFT) + copy4)), inc);
    atomicAdd(reinterpret_cast<uint32_t *>(base + ((rd_sar16(sg) << RD_HIST_SHIFT) + copy4)) + 256u * RD_HK, inc);
    atomicAdd(reinterpret_cast<uint32_t *>(base + ((rd_sar16(sb) << RD_HIST_SHIFT) + copy4)) + 512u * RD_HK, inc);
}

template <bool ALPHA>
__device__ __forceinline__ uint32_t rd_pack_rgba_b2(uint32_t sr, uint32_t sg, uint32_t sb)       // codes in byte 2 of each
{
    const uint32_t rg = __builtin_amdgcn_perm(sg, sr, 0x0c0c0602u);              // [r, g, 0, 0]
    return __builtin_amdgcn_perm(sb, rg, ALPHA ? 0xff060100u : 0x0c060100u);     // [r, g, b, 0xff / 0]
}

// RGBA8 / RGB8 pixel from three such encodings: two byte permutes (v_perm_b32: selector bytes 0-3 pick the second operand's
// bytes, 4-7 the first operand's, 0x0c gives 0x00 and 0xff gives 0xff) instead of shift-or, shift-or, or.
template <bool ALPHA>
__device__ __forceinline__ uint32_t rd_pack_rgba_bits(uint32_t tr, uint32_t tg, uint32_t tb)
{
    const uint32_t rg = __builtin_amdgcn_perm(tg, tr, 0x0c0c0400u);              // [r, g, 0, 0]
    return __builtin_amdgcn_perm(tb, rg, ALPHA ? 0xff040100u : 0x0c040100u);     // [r, g, b, 0xff / 0]
}

__device__ __forceinline__ void rd_hist_flush(const uint32_t *lh, uint32_t *slab32,
                                              unsigned long long *slab64)
{
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < 768u; t += blockDim.x) {
        uint32_t sum = 0;
#pragma unroll 8
        for (uint32_t i = 0; i < RD_HK; ++i) sum += lh[t * RD_HK + ((t + i) & (RD_HK - 1))];
        if (slab64) slab64[(size_t)blockIdx.x * 768u + t] += sum;   // batch: u64, private row
        else slab32[(size_t)blockIdx.x * 768u + t] = sum;            // single frame: overwrite
    }
}

// ---------------------------------------------------------------------------------------------
// Surface stores.  FMT is an rd_format.  `px` is a pixel index into the tightly packed surface.
// ---------------------------------------------------------------------------------------------
template <int FMT>
__device__ __forceinline__ void rd_store_px(void *out, size_t px, const rd_rgb &c, uint32_t qr,
                                            uint32_t qg, uint32_t qb)
{
    if (FMT == RD_FMT_RGBA_F32) {
        rd_f4 v = { c.r, c.g, c.b, 1.0f };
        __builtin_nontemporal_store(v, reinterpret_cast<rd_f4 *>(out) + px);
    } else if (FMT == RD_FMT_RGBA_F16) {
        rd_h2 lo = { (_Float16)c.r, (_Float16)c.g };
        rd_h2 hi = { (_Float16)c.b, (_Float16)1.0f };
        rd_u2 v = { __builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi) };
        __builtin_nontemporal_store(v, reinterpret_cast<rd_u2 *>(out) + px);
    } else if (FMT == RD_FMT_RGB_U8) {
        uint8_t *o = reinterpret_cast<uint8_t *>(out) + px * 3u;    // small targets only: three byte stores
        o[0] = (uint8_t)qr; o[1] = (uint8_t)qg; o[2] = (uint8_t)qb;
    } else {
        uint32_t v = qr | (qg << 8) | (qb << 16) | 0xff000000u;
        __builtin_nontemporal_store(v, reinterpret_cast<uint32_t *>(out) + px);
    }
}

// ---------------------------------------------------------------------------------------------
// rd_develop_quads -- the export hot path.
//
// Work unit u (0 <= u <= H/2) = CFA/output rows a = 2u-1 and b = 2u (a = -1 and b = H do not
// exist and are skipped); a lane owns columns 2q, 2q+1 of one unit.  With A,B = cfa[a][2q..2q+1]
// and C,D = cfa[b][2q..2q+1] (row indices clamped to the image, which reproduces get_neighbor's
// edge clamp for the first and last row) the selection table of shaders.rs:127-155 gives
//     row a (py odd),  both columns : (r,g,b) = (C, A, B)
//     row b (py even), column 2q    : (r,g,b) = (C, D, A)     <- blue taken from a green site
//     row b (py even), column 2q+1  : (r,g,b) = (C, D, B)
// Requires cfa 4-byte aligned (the host falls back to rd_develop_map otherwise).  An odd W leaves column W - 1 to
// rd_develop_lastcol (below): the lanes here own the W / 2 whole quads of a row pair.
//
// Scheduling: a WAVE owns a tile = 64 consecutive quads of one unit (128 px x 2 rows).  Every wave
// starts on tile `slot` (static) and then draws further tiles from a ticket counter with a SCALAR
// atomic (s_atomic_add: returns through lgkmcnt into an SGPR, so it neither occupies a lane nor a
// place in the in-order vmcnt queue the loads and stores live in).  Why not static round-robin:
// measured per-workgroup timelines (tools/microbench.hip, MB_TIMELINE) show that of the two
// workgroups sharing a CU the first-dispatched one runs ~1.6x faster (older waves win arbitration)
// and that the XCDs differ by 5-8 % -- with a static deal the CU then runs half-empty for the last
// third of the launch.  One address serves only ~88 M atomics/s (11.4 ns each, tools/atomic_probe.hip),
// so the tickets are spread over K = gridDim/4 counters of 64 client waves each; counter k deals the
// tiles nwaves + t*K + k (interleaved, so all counters advance together and the launch keeps ONE
// moving front through the frame), and its 64 clients are picked across all XCDs, both workgroups
// of a CU and four wave slots, so every group drains at the chip's average speed and no stealing
// is needed.  Every client ends with exactly one failed draw, so a counter's final value is known
// (tmax + clients) and the wave that draws it resets the counter for the next launch.
// Tile bookkeeping (unit, tile-in-unit, row bases) is wave-uniform and lives in SGPRs/SALU; per-lane
// integer work is one lane offset.  The next tile's two loads (256 B per row per wave) are issued
// before the current tile's arithmetic and stores, so a load never waits behind a younger store in
// the in-order vmcnt queue.
//
// Stores: a lane's two f32 pixels are 32 B, so storing them directly gives every store
// instruction a 32-B lane stride (half-used 64-B requests; measured 2.4 TB/s when store-bound).
// The f32 surface therefore goes through a wave-private LDS transpose (3 ds_write_b96 +
// 4 ds_read_b128 per lane; the alpha words of the stage are written once per kernel) so that every
// global_store_dwordx4 writes 1 KiB contiguous; the tile also WAITS there for its stores, one
// iteration long, instead of in registers.  f16 (16 B per lane per row) and u8 (8 B) are contiguous
// as they are, skip the transpose and carry their packed tile in registers.
// ---------------------------------------------------------------------------------------------
#define RD_WAVES (RD_BLOCK / 64)
#define RD_TQ_STRIDE 32u        // dwords between ticket counters (128 B: one counter per cache line)
#define RD_TQ_CLIENTS (RD_WAVES * 4u)   // waves per counter when the launch uses more than one (needs RD_WAVES % 4 == 0)
// Two 1024-thread workgroups fit a CU only while a wave needs <= 80 SGPRs (800 per SIMD, 16-register granules and a
// 16-register trap-handler reserve per wave: 96 + 16 allows 7 waves per SIMD, not 8).  hipcc reports "occupancy 8" either
// way; with 83 SGPRs the second workgroup of every CU was measured to start only when the first had finished (+12 %).
#ifndef RD_NUM_SGPR
#define RD_NUM_SGPR 80
#endif
// Round 5: the f32 instances also tell the compiler that 8 waves per SIMD is the target (amdgpu_waves_per_eu): __launch_bounds__(1024)
// alone promises only 4, so its scheduler feels free to use up to 128 VGPRs and whether an instance stays under 64 is luck
// (with the slider uniforms parked, RD_F32_PARK, four instances came out at 65-67).  With the attribute the scheduler orders the
// code for <= 64 (the build still checks: no scratch, <= 64 VGPRs); measured neutral on the f32 kernel (75.36 vs 75.33 us per
// frame, alternating).  The narrow surfaces keep the default (4, 8): they fit anyway and the attribute's instruction order
// costs them 4 issue cycles per tile (0.15-0.2 % measured, profiles/r05_f32_park_ab.txt).
#ifndef RD_WAVES_PER_EU
#define RD_WAVES_PER_EU __attribute__((amdgpu_waves_per_eu(FMT == RD_FMT_RGBA_F32 ? 8 : 4, 8)))
#endif
#ifndef RD_SWEEP_POLICY
#define RD_SWEEP_POLICY ""      // cache-policy bits of the sweeps' LDS-DMA loads (probe builds: " nt", " sc1", ...)
#endif

#ifdef RD_COLOUR_HOOK_HEADER  // tools/microbench.hip only: swaps in reduced-VALU stand-ins to find the memory floor
#include RD_COLOUR_HOOK_HEADER
#endif
#ifndef RD_COLOUR
#define RD_COLOUR rd_colour_m<MATH>
#endif

// TILES says how a unit's 64-quad tiles cover its W/2 quads:
//   RD_TILES_WHOLE    W % 128 == 0 (6016, 11648): the tiles abut;
//   RD_TILES_OVERLAP  any other even W >= 128 (round 5; the frames cameras make are 6000, 8256, 5472, 7360 ... wide and
//                     shaders.rs:181-187 renders any size): the LAST tile of a unit starts at quad W/2 - 64 instead of
//                     64 * (tiles - 1), i.e. it overlaps its predecessor by 64 - (W/2) % 64 quads, recomputes them and stores
//                     the same bytes a second time; only the histogram must not count those lanes twice, and an LDS atomic
//                     may sit behind a lane mask (it is not vector memory).  Costs 8-22 issue cycles per tile
//                     (tools/isa_budget.py), which is why the abutting case keeps its own instance;
//   RD_TILES_MASKED   W < 128: one partial tile per unit, lane masks everywhere.
// FULL (= not MASKED) = every tile is a whole 64 quads: no lane masks, so every vector-memory instruction of the loop body
// is issued unconditionally.  Why that matters: vmcnt
// retires in issue order, and hipcc can only leave the four stores of tile i in flight while it
// waits for the prefetched loads of tile i+1 (s_waitcnt vmcnt(4)) if it can COUNT them; one store
// behind a branch makes it fall back to vmcnt(0), which drains the store queue every iteration
// (measured: +20 us per frame).  For the same reason the first/last unit (one of the two rows does
// not exist) is handled by redirecting that row's store onto the other row with the other row's
// value -- a duplicate store of correct data -- instead of branching around it.
// One tile's results, packed the way its surface stores want them, carried in registers from the
// iteration that computes them to the next one, which stores them (f16 / u8 / rgb8; the f32 tile waits
// in the LDS stage).
//   RD_TILES_SHIFT    the f32 surface at a width that is not a multiple of 4, W >= 128 (round 6).  Its rows are 16 W bytes, so
//                     they start 16, 32 or 48 bytes into a 64-byte block, and with tiles cut at pixel 128 k every 1-KiB store
//                     begins and ends inside a block that another wave completes at another time.  Measured, time per pixel
//                     against 6016: 6008 px (rows on 128-byte lines) x1.03, 6004 (64-byte blocks) x1.04, 6002 (32 bytes) x1.16,
//                     6001 x1.21-1.27 -- the memory system wants whole 64-byte blocks (aligning to 32-byte sectors alone, tried
//                     first, changed nothing).  Here a tile OWNS 62 quads and every wave still develops 64: tile k's window is the
//                     quads [62 k, 62 k + 64), and in row r it stores the pixels [124 k + s_r, 124 (k + 1) + s_r), where
//                     s_r = (-r W) mod 4 is the shift that puts row r's stores on 64-byte boundaries -- different for the two rows
//                     of the pair, at most 3, which is what the two spare quads are for.  Tile 0 starts every row at pixel 0, the
//                     last tile is pulled back to end at the row's last quad (like OVERLAP) and stores up to it.  A quad is
//                     counted in the histogram by the tile that owns it; an odd width's last column stays rd_develop_lastcol's.
//                     3 % more tiles, on a kernel that waits for memory.
#define RD_TILES_MASKED 0
#define RD_TILES_WHOLE 1
#define RD_TILES_OVERLAP 2
#define RD_TILES_SHIFT 3
template <int FMT> struct rd_tile_out;
template <> struct rd_tile_out<RD_FMT_RGBA_F32> { };      // nothing: the f32 tile waits in the wave's LDS stage, not in registers
template <> struct rd_tile_out<RD_FMT_RGBA_F16> { uint32_t a0, a1, b0, b1, c0, c1; };   // half2 pairs: rg, b1
template <> struct rd_tile_out<RD_FMT_RGBA_U8> { uint32_t v1, v2, v3; };
template <> struct rd_tile_out<RD_FMT_RGB_U8> { uint32_t v1, v2, v3; };                   // 0x00bbggrr

#ifdef RD_PROBE   // tools/microbench.hip only: per-workgroup timeline (100 MHz ticks, shader-clock ticks, HW_ID, XCC_ID, tiles)
__device__ uint32_t rd_probe_buf[RD_MAX_BLOCKS * 8];
#endif

// One frame of a multi-frame launch (rd_develop_batch below): what a single-frame launch gets as kernel arguments.
// The host (rawdev.hip, rd_batch_develop) fills an array of these in HBM; a wave reads an entry with scalar loads
// when its tile stream crosses into that frame.
struct alignas(64) rd_frame_desc {
    const uint16_t *cfa;
    void *out;
    rd_ku u;
    uint32_t pad_[48 - 4 - sizeof(rd_ku) / 4];
};
static_assert(sizeof(rd_frame_desc) == 192, "rd_frame_desc is three 64-byte lines");

// MULTI = false: one frame (or one row band: units [unit0, unit1)) per launch, uniforms in kernel arguments.
// MULTI = true : the launch covers `nframes` whole frames of one size (descs[0 .. nframes-1]); the tile index runs
//                through all of them (frame-major), so the ticket front moves from one frame into the next without
//                a launch boundary -- no drain tail, no inter-launch gap, and the next frame's first tiles are
//                computed while the previous frame's last ones are still being stored.  A wave re-reads the
//                uniforms (scalar loads, ~190 B) only when its compute stage changes frame: about once per
//                (tiles per frame / resident waves) = 11 tiles at 24 MP.
// STAMP = true (one diagnostic instance, rd_batch_measure_clock; in every other instance no stamp executes): thread 0 of each
//                workgroup reads the shader-cycle counter and the 100 MHz real-time counter after the prologue and at the end
//                and stores the two differences into stamps[2 * blockIdx.x ..] -- a buffer nothing else reads: the clock the
//                part holds UNDER THIS KERNEL (MI355X_MICROARCH.md, "DVFS give-back" item 6).
template <int FMT, bool HIST, int TILES, int MATH, bool BURST, bool MULTI, bool STAMP = false>
__device__ __forceinline__ void
rd_quads_body(const uint16_t *__restrict__ cfa_arg, void *__restrict__ out_arg, uint32_t W, uint32_t H, uint32_t unit0,
              uint32_t unit1, uint32_t tpu, uint32_t tpu_magic, uint32_t tq_k, uint32_t tq_tmax, uint32_t *tq,
              const rd_ku &u_arg, uint32_t *slab32, unsigned long long *slab64,
              const rd_frame_desc *__restrict__ descs, uint32_t nframes, uint32_t tpf, uint32_t tpf_magic,
              uint32_t *stamps = nullptr)
{
    // Register budget: two workgroups per CU need <= 80 SGPRs AND <= 64 VGPRs per wave (RD_NUM_SGPR above).  The uniforms of
    // the front of the stack (white balance, temperature/tint, matrix) and of the levels divide (17 values) are therefore parked in VGPRs -- the asm
    // keeps the compiler from folding them back into scalar operands -- which leaves every instance at <= 78 SGPRs / <= 63 VGPRs
    // (raweditor_amd/kernel_resources.json; the build fails beyond the budget).  (An SGPR source also halves the issue rate of v_mul/v_add/v_fma_f32, tools/valu_probe2.hip, but
    // that is not what limits this kernel: DESIGN.md section 6.)
    constexpr bool FULL = TILES != RD_TILES_MASKED;
    constexpr bool SHIFT = TILES == RD_TILES_SHIFT;
    static_assert(!SHIFT || FMT == RD_FMT_RGBA_F32, "RD_TILES_SHIFT is the f32 surface's instance");
    rd_ku u;
    auto adopt = [&](const rd_ku &src_mem) {                     // take over one frame's uniforms (MULTI: on every frame change)
        rd_ku src = src_mem;
        if constexpr (MULTI) {                                   // descriptor fields: have every scalar load issued and landed
            asm volatile("" : "+s"(src.wb_r), "+s"(src.wb_g), "+s"(src.wb_b), "+s"(src.kr), "+s"(src.kg), "+s"(src.kb),   // before the
                              "+s"(src.m[0]), "+s"(src.m[1]), "+s"(src.m[2]), "+s"(src.m[3]), "+s"(src.m[4]), "+s"(src.m[5]),  // first park
                              "+s"(src.m[6]), "+s"(src.m[7]), "+s"(src.m[8]), "+s"(src.den), "+s"(src.rden));                // (one round trip, not 17)
        }
        u = src;
#define RD_PARK(f) asm volatile("v_mov_b32 %0, %1" : "=v"(u.f) : "s"(src.f))
        RD_PARK(wb_r); RD_PARK(wb_g); RD_PARK(wb_b); RD_PARK(kr); RD_PARK(kg); RD_PARK(kb);
        RD_PARK(m[0]); RD_PARK(m[1]); RD_PARK(m[2]); RD_PARK(m[3]); RD_PARK(m[4]); RD_PARK(m[5]); RD_PARK(m[6]); RD_PARK(m[7]); RD_PARK(m[8]);
        RD_PARK(den); RD_PARK(rden);                             // 36 uses per tile in the divide's FMA chains
#ifndef RD_F32_PARK
#define RD_F32_PARK 0          // 1 = the f32 multi-frame kernel parks its slider uniforms too (tools/build_ab_libs.sh: librawdev_r5park.so).
#endif                         // Measured (profiles/r05_f32_park_ab.txt): 84 fewer issue cycles per tile, 0.0 % on boxes at their memory
                               // floor and -0.25 % on fast boxes, where it also never reached the faster of the two placement levels: off.
        // six to eight more: an SGPR source halves the issue rate of v_mul / v_add / v_fma.  (f32: only with RD_F32_PARK, and then in
        // the multi-frame kernel only; the single-frame instances sit at 63-64 VGPRs without them and one of them spilled with them)
        if constexpr (FMT != RD_FMT_RGBA_F32 || (RD_F32_PARK && MULTI)) {
            RD_PARK(em); RD_PARK(cf); RD_PARK(blacks); RD_PARK(s); RD_PARK(oms); RD_PARK(vibrance);
            if constexpr (FMT != RD_FMT_RGB_U8 && FMT != RD_FMT_RGBA_F32) { RD_PARK(highlights); RD_PARK(shadows); }     // (RGB8's LDS repack and the f32 surface's pinned gamma need the registers)
        }
#undef RD_PARK
    };
#if defined(RD_COLOUR_HOOK_HEADER) || defined(RD_NO_Q8_SHORTCUT)
    constexpr bool Q8LUT = false, F16LUT = false;
#else
    constexpr bool Q8LUT = (RD_Q8_LUT != 0) && (FMT == RD_FMT_RGBA_U8 || FMT == RD_FMT_RGB_U8);   // codes from the LDS threshold table
    constexpr bool F16LUT = (RD_F16_LUT != 0) && FMT == RD_FMT_RGBA_F16;                         // halves and codes from the two-level tables
#endif
    // lane constants of the gamma shortcuts (rd_kc), in VGPRs for the same reason; the asm keeps them there
    rd_kc kc = { 255.0f, RD_F16_KA / 64.0f, RD_F16_KB / 64.0f };
    if constexpr (FMT != RD_FMT_RGBA_F32 && !Q8LUT && !F16LUT) {
        asm volatile("" : "+v"(kc.k255));
        if constexpr (FMT == RD_FMT_RGBA_F16) asm volatile("" : "+v"(kc.f16_ka), "+v"(kc.f16_kb));
    }
    typedef uint32_t rd_u4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) uint32_t lh[HIST ? 768 * RD_HK : 4];
    __shared__ __attribute__((aligned(16))) uint32_t qlut[Q8LUT ? RD_Q8_LUT_LDS_WORDS : 4];
    __shared__ __attribute__((aligned(16))) uint16_t hfine[F16LUT ? RD_F16_LUT_FINE_LDS : 8];
    __shared__ __attribute__((aligned(16))) uint32_t hcoarse[F16LUT ? RD_F16_LUT_COARSE_LDS : 4];
    __shared__ rd_f4 stage[FMT == RD_FMT_RGBA_F32 ? RD_BLOCK * 3 : 1];
    __shared__ uint16_t rgb16[FMT == RD_FMT_RGB_U8 ? RD_WAVES * 384 : 1];   // per wave: 2 rows x 384 B
    __shared__ rd_f4 pf_dump[BURST ? 64 : 1];                    // where the LDS-DMA sweeps land (never read)
    // Workgroup prologue (round 5: one phase instead of three).  Every global load of the code / half tables is issued first
    // -- up to five 16-byte loads per lane in flight -- the histogram is zeroed (16 bytes per store) while they travel, then
    // the tables are written to LDS and ONE barrier ends it.  It was: table loop (a dependent load -> LDS store per trip,
    // three trips for the f16 tables), barrier, second table, barrier, 4-byte zeroing loop, barrier -- several microseconds at
    // the head of EVERY launch, which a launch per row band (BASELINE config 5 as worded: 128 launches per step) pays 8
    // times per frame.
    {
        typedef uint32_t rd_u4v __attribute__((ext_vector_type(4)));
        constexpr uint32_t NQ = Q8LUT ? RD_Q8_LUT_LDS_WORDS / 4u : 0u, NF = F16LUT ? RD_F16_LUT_FINE_LDS / 8u : 0u,
                           NC = F16LUT ? RD_F16_LUT_COARSE_LDS / 4u : 0u;
        constexpr uint32_t KQ = (NQ + RD_BLOCK - 1u) / RD_BLOCK, KF = (NF + RD_BLOCK - 1u) / RD_BLOCK, KC = (NC + RD_BLOCK - 1u) / RD_BLOCK;
        const rd_u4v *qs = reinterpret_cast<const rd_u4v *>(rd_q8_lut_dev), *fs = reinterpret_cast<const rd_u4v *>(rd_f16_fine_dev),
                     *cs = reinterpret_cast<const rd_u4v *>(rd_f16_coarse_dev);
        rd_u4v vq[KQ ? KQ : 1], vf[KF ? KF : 1], vc[KC ? KC : 1];
#pragma unroll
        for (uint32_t k = 0; k < KQ; ++k) { const uint32_t i = threadIdx.x + k * RD_BLOCK; if (i < NQ) vq[k] = qs[i]; }
#pragma unroll
        for (uint32_t k = 0; k < KF; ++k) { const uint32_t i = threadIdx.x + k * RD_BLOCK; if (i < NF) vf[k] = fs[i]; }
#pragma unroll
        for (uint32_t k = 0; k < KC; ++k) { const uint32_t i = threadIdx.x + k * RD_BLOCK; if (i < NC) vc[k] = cs[i]; }
        if (HIST) {
            static_assert((768u * RD_HK) % 4u == 0, "histogram table in 16-byte pieces");
            const rd_u4v z = { 0u, 0u, 0u, 0u };
            for (uint32_t i = threadIdx.x; i < 768u * RD_HK / 4u; i += RD_BLOCK) reinterpret_cast<rd_u4v *>(lh)[i] = z;
        }
#pragma unroll
        for (uint32_t k = 0; k < KQ; ++k) { const uint32_t i = threadIdx.x + k * RD_BLOCK; if (i < NQ) reinterpret_cast<rd_u4v *>(qlut)[i] = vq[k]; }
#pragma unroll
        for (uint32_t k = 0; k < KF; ++k) { const uint32_t i = threadIdx.x + k * RD_BLOCK; if (i < NF) reinterpret_cast<rd_u4v *>(hfine)[i] = vf[k]; }
#pragma unroll
        for (uint32_t k = 0; k < KC; ++k) { const uint32_t i = threadIdx.x + k * RD_BLOCK; if (i < NC) reinterpret_cast<rd_u4v *>(hcoarse)[i] = vc[k]; }
        if (Q8LUT || F16LUT || HIST) __syncthreads();
    }
    if constexpr (FMT == RD_FMT_RGBA_F32) {
        // The store stage holds [lane][c1, c2, c3] as RGBA; alpha is 1.0 for every pixel of every tile, so it is written
        // here once and the tiles only ever write r, g, b (12 bytes) next to it: no per-tile assembly of {r, g, b, 1}
        // register quadruples (15 v_mov per tile), a quarter fewer LDS bytes written.
        float *sa = reinterpret_cast<float *>(stage + (size_t)(threadIdx.x >> 6) * 192u + (threadIdx.x & 63u) * 3u);
        sa[3] = 1.0f; sa[7] = 1.0f; sa[11] = 1.0f;
        __builtin_amdgcn_wave_barrier();
    }
    uint64_t stamp_t0 = 0, stamp_r0 = 0;
    if constexpr (STAMP) { stamp_t0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
#ifdef RD_PROBE
    const uint64_t probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t probe_tiles = 0;
    if (threadIdx.x == 0) rd_probe_buf[blockIdx.x * 8u + 6u] = 0u;
#endif

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qpr = W >> 1;                                 // quads per unit
    // first quad of tile pq of a unit (wave-uniform, SALU): tiles abut, except that RD_TILES_OVERLAP pulls the last one back
    // to end at the row's end
    auto qbase = [&](uint32_t pq) -> uint32_t {
        if constexpr (SHIFT) {                                   // the 64-quad window of tile pq starts at its own 62 quads
            const uint32_t b = pq * 62u, last = qpr - 64u;
            return b < last ? b : last;
        }
        const uint32_t b = pq * 64u;
        if constexpr (TILES == RD_TILES_OVERLAP) { const uint32_t last = qpr - 64u; return b < last ? b : last; }
        else return b;
    };
    const uint32_t ntiles = MULTI ? nframes * tpf : (unit1 - unit0) * tpu;   // < 2^32: checked on the host
    const uint32_t nwaves = gridDim.x * RD_WAVES;
    const uint32_t copy = lane & (RD_HK - 1);
    // First tile: static, slot = wave_in_block * gridDim + block (block-cyclic, so a launch smaller than the grid is
    // spread over all CUs).
    uint32_t tile = wave * gridDim.x + blockIdx.x;

    // tile -> (unit, tile in unit) without a divide: q = mulhi(t, floor(2^32 / tpu)) is floor(t / tpu) or one less.
    auto split = [&](uint32_t t, uint32_t &un, uint32_t &q) {
        uint32_t d = __umulhi(t, tpu_magic);
        uint32_t r = t - d * tpu;
        if (r >= tpu) { d += 1u; r -= tpu; }
        un = unit0 + d; q = r;
    };
    // Ticket counter of this wave's group (see the header comment).  tq_k == 1: one counter for the whole launch.
    const uint32_t tq_clients = tq_k > 1u ? RD_TQ_CLIENTS : nwaves;
    const uint32_t tq_idx = tq_k > 1u ? (wave & 3u) * (gridDim.x >> 4) + ((blockIdx.x >= (gridDim.x >> 1) ? blockIdx.x - (gridDim.x >> 1) : blockIdx.x) >> 3) : 0u;
    uint32_t *const tq_cnt = tq + (size_t)tq_idx * RD_TQ_STRIDE;
    // Next tile of this wave, or ~0u when its counter is exhausted (every wave gets that answer exactly once).
    uint32_t dealt = tile;                                       // tq_k == 0 (A/B switch RD_STATIC_DEAL=1): static round-robin deal
    auto draw = [&]() -> uint32_t {
        if (tq_k == 0u) { dealt += nwaves; return dealt < ntiles ? dealt : ~0u; }
        for (;;) {
            uint32_t t = 1u;
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(tq_cnt) : "memory");
            if (t >= tq_tmax) {
                if (t == tq_tmax + tq_clients - 1u) {            // last draw on this counter in this launch: reset it
                    uint32_t z = 0u;
                    asm volatile("s_atomic_and %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(z) : "s"(tq_cnt) : "memory");
                }
                return ~0u;
            }
            t = nwaves + t * tq_k + tq_idx;
            if (t < ntiles) return t;                            // only ticket tmax-1 can point past the end
        }
    };
    // MULTI: global tile -> (frame, tile in frame) the same way, then split().
    auto locate = [&](uint32_t t, uint32_t &fr, uint32_t &tin, uint32_t &un, uint32_t &q) {
        if (MULTI) {
            uint32_t d = __umulhi(t, tpf_magic);
            uint32_t r = t - d * tpf;
            if (r >= tpf) { d += 1u; r -= tpf; }
            fr = d; t = r;
        } else {
            fr = 0u;
        }
        tin = t;
        split(t, un, q);
    };
    uint32_t unit, qt, fr_c, tin0;
    locate(tile < ntiles ? tile : 0u, fr_c, tin0, unit, qt);
    // The three pipeline stages (load next / compute current / store previous) may sit in three different frames:
    // cfa_n, (u, out_c), out_p.
    const RD_GLOBAL uint16_t *cfa = (const RD_GLOBAL uint16_t *)(MULTI ? descs[fr_c].cfa : cfa_arg);     // load stage
    RD_GLOBAL void *out_c = (RD_GLOBAL void *)(MULTI ? descs[fr_c].out : out_arg);   // compute stage's surface; becomes the store stage's
    adopt(MULTI ? descs[fr_c].u : u_arg);
    uint32_t fr_n = fr_c;
    (void)tin0;

    // cfa[ra][2q..2q+1] and cfa[rb][2q..2q+1] of tile (pu, pq) for this lane; rows clamped to the image.
    auto load_tile = [&](uint32_t pu, uint32_t pq, uint32_t &top, uint32_t &bot) {
        const uint32_t ra = pu ? 2u * pu - 1u : 0u;
        const uint32_t rb = 2u * pu < H ? 2u * pu : H - 1u;
#ifdef RD_PROBE_NO_TILE_LOADS                                    // probe builds only (tools/): what the main loop's loads cost the memory system
        (void)ra; (void)rb; top = lane * 0x00010001u + pu; bot = top ^ 0x01230123u; return;
#endif
        // (a quad's two samples are ONE dword load.  With an odd W every other row starts on an odd 16-bit boundary, so the
        //  dword is declared 2-byte aligned: gfx950 under HSA runs with unaligned global access, hipcc keeps global_load_dword
        //  and the even widths' code does not change -- tools/isa_budget.py before / after)
        typedef uint32_t rd_u32_a2 __attribute__((aligned(2)));
        if constexpr (FULL) {
            // wave-uniform base (SALU, 64-bit) + the lane's own dword: global_load_dword v_lane4, s[base] -- no per-tile
            // 64-bit VALU address arithmetic (it was 14 issue cycles per tile, tools/isa_budget.py)
            const size_t x0 = (size_t)qbase(pq) * 2u;
            top = reinterpret_cast<const RD_GLOBAL rd_u32_a2 *>(cfa + ((size_t)ra * W + x0))[lane];
            bot = reinterpret_cast<const RD_GLOBAL rd_u32_a2 *>(cfa + ((size_t)rb * W + x0))[lane];
        } else {
            uint32_t q = pq * 64u + lane;
            q = q < qpr ? q : qpr - 1u;                          // clamp: loaded but never used
            top = reinterpret_cast<const RD_GLOBAL rd_u32_a2 *>(cfa + (size_t)ra * W)[q];
            bot = reinterpret_cast<const RD_GLOBAL rd_u32_a2 *>(cfa + (size_t)rb * W)[q];
        }
    };

    // demosaic + colour stack + histogram of one tile -> packed results
    auto compute_tile = [&](uint32_t tu, uint32_t tq, uint32_t top, uint32_t bot) {
        const bool has_a = tu != 0u, has_b = 2u * tu < H;        // wave-uniform
        // lanes that count in the histogram: all (WHOLE); all but the quads a pulled-back last tile shares with its
        // predecessor (OVERLAP: tq * 64 - qbase(tq) of them, 0 for every other tile); the lanes inside the row (MASKED)
        // (SHIFT: the quads [62 tq, 62 tq + 62) are this tile's own -- the last tile's reach the row's end; the rest of the window is a neighbour's)
        const bool valid = TILES == RD_TILES_WHOLE ||
                           (SHIFT ? (qbase(tq) + lane >= tq * 62u && qbase(tq) + lane < tq * 62u + 62u)
                                  : TILES == RD_TILES_OVERLAP ? lane >= tq * 64u - qbase(tq) : (tq * 64u + lane) < qpr);
        float A, B, C, D;
#ifdef RD_BUDGET_ELIDE                                           // tools/isa_budget.py: one path in the assembly
        constexpr bool black0 = true;
#else
        const bool black0 = u.black_level == 0u;
#endif
        if (black0) {                                            // wave-uniform; the reference's case
            A = rd_norm0(top & 0xffffu); B = rd_norm0(top >> 16);
            C = rd_norm0(bot & 0xffffu); D = rd_norm0(bot >> 16);
        } else {
            A = rd_norm(top & 0xffffu, u.black_level); B = rd_norm(top >> 16, u.black_level);
            C = rd_norm(bot & 0xffffu, u.black_level); D = rd_norm(bot >> 16, u.black_level);
        }
#if defined(RD_COLOUR_HOOK_HEADER) || defined(RD_NO_Q8_SHORTCUT)   // microbench stand-ins / A/B builds (tools/): the narrow
        constexpr bool Q8ONLY = false, H16 = false;                    // surfaces go through the pinned gamma like the f32 one
#else
        constexpr bool Q8ONLY = FMT == RD_FMT_RGBA_U8 || FMT == RD_FMT_RGB_U8;    // only the 8-bit codes leave the kernel
        constexpr bool H16 = FMT == RD_FMT_RGBA_F16;                              // only binary16 values (+ codes) leave it
#endif
#ifdef RD_COLOUR_HOOK_HEADER
        constexpr bool F32T = false;
#else
        constexpr bool F32T = FMT == RD_FMT_RGBA_F32;      // f32 surface: gamma, histogram and stage write triple by triple (below)
#endif
        // 8-bit codes of the three triples.  BITS (the narrow surfaces' shortcut paths): as add-magic ENCODINGS 0x4b0000qq
        // (rd_q8_gamma_bits) -- the histogram address and the pixel pack take the byte straight from them; otherwise plain codes.
        constexpr bool BITS = Q8ONLY || H16;
        uint32_t q1r = 0, q1g = 0, q1b = 0, q2r = 0, q2g = 0, q2b = 0, q3r = 0, q3g = 0, q3b = 0;
        uint32_t ha0 = 0, ha1 = 0, hb0 = 0, hb1 = 0, hc0 = 0, hc1 = 0;       // binary16 pairs (r, g), (b, 1.0) of c1, c2, c3 (H16)
        rd_rgb c1 = { 0.0f, 0.0f, 0.0f }, c2 = c1, c3 = c1;
        const uint32_t hbase = rd_hist_base(copy);               // loop-invariant: hoisted
        auto count = [&](uint32_t qr, uint32_t qg, uint32_t qb, uint32_t inc) {      // one pixel's codes into the histogram
            if constexpr (Q8LUT || F16LUT) rd_hist_add_b2(lh, copy * 4u, qr, qg, qb, inc);
            else if constexpr (BITS) rd_hist_add_bits(lh, hbase, qr, qg, qb, inc);
            else rd_hist_add(lh, copy, qr, qg, qb, inc);
        };
        auto code = [&](float v) -> uint32_t {                   // the 8-bit code of a linear value, as the surface's encoding of it
            if constexpr (Q8LUT) return rd_q8_lut_bits(v, qlut);
            else return rd_q8_gamma_bits(v, kc);
        };
        // F16LUT: binary16 of one value as an integer + its code (bits 16..23 of q), both from the tables; the rare lane
        // below the tables' domain (or on the dip) takes the pinned evaluation
        auto half_of = [&](float v, uint32_t &q) -> uint32_t {
            uint32_t h = 0u;
            if constexpr (F16LUT) {
                bool pinned;
                float vc;
                rd_f16_lut_lookup(v, hfine, hcoarse, h, q, pinned, vc);
                if (pinned) {
                    const float e = rd_gamma_clamp(vc);
                    h = __builtin_bit_cast(uint16_t, (_Float16)e);
                    q = rd_q8(e) << 16;
                }
            }
            return h;
        };
        // three values at once: six LDS reads in flight behind one wait, and ONE branch for the rare lanes instead of three
        auto halves_of = [&](float v0, float v1, float v2, uint32_t &h0, uint32_t &h1, uint32_t &h2, uint32_t &q0, uint32_t &q1, uint32_t &q2) {
            if constexpr (F16LUT) {
                bool p0, p1, p2;
                float c0, c1, c2;
                rd_f16_lut_lookup(v0, hfine, hcoarse, h0, q0, p0, c0);
                rd_f16_lut_lookup(v1, hfine, hcoarse, h1, q1, p1, c1);
                rd_f16_lut_lookup(v2, hfine, hcoarse, h2, q2, p2, c2);
                if (p0 || p1 || p2) {
                    if (p0) { const float e = rd_gamma_clamp(c0); h0 = __builtin_bit_cast(uint16_t, (_Float16)e); q0 = rd_q8(e) << 16; }
                    if (p1) { const float e = rd_gamma_clamp(c1); h1 = __builtin_bit_cast(uint16_t, (_Float16)e); q1 = rd_q8(e) << 16; }
                    if (p2) { const float e = rd_gamma_clamp(c2); h2 = __builtin_bit_cast(uint16_t, (_Float16)e); q2 = rd_q8(e) << 16; }
                }
            } else { h0 = h1 = h2 = 0u; (void)v0; (void)v1; (void)v2; (void)q0; (void)q1; (void)q2; }
        };
        bool separable = false;                                  // wave-uniform
#ifndef RD_COLOUR_HOOK_HEADER
        // (not for the f32 surface: it sits on its memory floor with or without the shortcut -- 76.5 us per frame either way --
        // and the extra path costs its general case 0.3 %)
        if constexpr (Q8ONLY || H16) separable = (rd_elide_of(u) & RD_EL_SEPARABLE) == RD_EL_SEPARABLE;
        if (separable) {
            // No step of THIS frame's stack mixes channels (the usual edit): five distinct values instead of nine
            // (rd_colour_separable); triple 1 = (v0, v1, v3), triple 2 = (v0, v2, v4), triple 3 = (v0, v2, v3).
            float v[5] = { C, A, D, B, A };
            rd_colour_separable<MATH>(u, v);
            if constexpr (Q8ONLY) {
                q1r = code(v[0]); q1g = code(v[1]); q1b = code(v[3]);
                q2g = code(v[2]); q2b = code(v[4]);
                q2r = q1r; q3r = q1r; q3g = q2g; q3b = q1b;
                if (HIST && valid) {
                    if (has_a) count(q1r, q1g, q1b, 2u);
                    if (has_b) { count(q2r, q2g, q2b, 1u); count(q3r, q3g, q3b, 1u); }
                }
            } else if constexpr (H16 && F16LUT) {
                uint32_t hr, hg, hb;
                halves_of(v[0], v[1], v[3], hr, hg, hb, q1r, q1g, q1b);
                ha0 = hr | (hg << 16); ha1 = hb | 0x3c000000u;                 // (r, g), (b, 1.0)
                if (HIST && valid && has_a) count(q1r, q1g, q1b, 2u);
                const uint32_t hg2 = half_of(v[2], q2g), hb2 = half_of(v[4], q2b);
                q2r = q1r; q3r = q1r; q3g = q2g; q3b = q1b;
                hb0 = hr | (hg2 << 16); hb1 = hb2 | 0x3c000000u;
                hc0 = hb0; hc1 = ha1;
                if (HIST && valid && has_b) { count(q2r, q2g, q2b, 1u); count(q3r, q3g, q3b, 1u); }
            } else if constexpr (H16) {
                const float er = rd_f16_gamma_value<HIST>(v[0], kc, q1r), eg = rd_f16_gamma_value<HIST>(v[1], kc, q1g);
                const float eb = rd_f16_gamma_value<HIST>(v[3], kc, q1b);
                ha0 = rd_pack_h2(er, eg); ha1 = rd_pack_h2(eb, 1.0f);
                if (HIST && valid && has_a) count(q1r, q1g, q1b, 2u);
                const float eg2 = rd_f16_gamma_value<HIST>(v[2], kc, q2g), eb2 = rd_f16_gamma_value<HIST>(v[4], kc, q2b);
                q2r = q1r; q3r = q1r; q3g = q2g; q3b = q1b;
                hb0 = rd_pack_h2(er, eg2); hb1 = rd_pack_h2(eb2, 1.0f);
                hc0 = hb0; hc1 = ha1;
                if (HIST && valid && has_b) { count(q2r, q2g, q2b, 1u); count(q3r, q3g, q3b, 1u); }
            }
        }
#endif
        if (!separable) {
#ifdef RD_COLOUR_HOOK_HEADER
        c1 = RD_COLOUR(u, C, A, B);
        c2 = RD_COLOUR(u, C, D, A);
        c3 = RD_COLOUR(u, C, D, B);
#else
        float tr[3] = { C, C, C }, tg[3] = { A, D, D }, tb[3] = { B, A, B };       // row a: (C,A,B); row b: (C,D,A), (C,D,B)
        if constexpr (MATH != RD_MATH_PROBE) rd_colour_n<3, MATH, !(Q8ONLY || H16 || F32T)>(u, tr, tg, tb);
        c1 = rd_rgb{ tr[0], tg[0], tb[0] }; c2 = rd_rgb{ tr[1], tg[1], tb[1] }; c3 = rd_rgb{ tr[2], tg[2], tb[2] };
#endif
        if constexpr (Q8ONLY) {
            q1r = code(c1.r); q1g = code(c1.g); q1b = code(c1.b);
            q2r = code(c2.r); q2g = code(c2.g); q2b = code(c2.b);
            q3r = code(c3.r); q3g = code(c3.g); q3b = code(c3.b);
        } else if constexpr (H16 && F16LUT) {                     // triple by triple, as below
            uint32_t hr, hg, hb;                                  // (the codes are counted after the third triple, below)
            halves_of(c1.r, c1.g, c1.b, hr, hg, hb, q1r, q1g, q1b);
            ha0 = hr | (hg << 16); ha1 = hb | 0x3c000000u;
            halves_of(c2.r, c2.g, c2.b, hr, hg, hb, q2r, q2g, q2b);
            hb0 = hr | (hg << 16); hb1 = hb | 0x3c000000u;
            halves_of(c3.r, c3.g, c3.b, hr, hg, hb, q3r, q3g, q3b);
            hc0 = hr | (hg << 16); hc1 = hb | 0x3c000000u;
        } else if constexpr (H16) {                               // triple by triple: halves packed and codes counted at once,
            float er, eg, eb;                                     // so that at most one triple's values are live
            er = rd_f16_gamma_value<HIST>(c1.r, kc, q1r); eg = rd_f16_gamma_value<HIST>(c1.g, kc, q1g); eb = rd_f16_gamma_value<HIST>(c1.b, kc, q1b);
            ha0 = rd_pack_h2(er, eg); ha1 = rd_pack_h2(eb, 1.0f);
            if (HIST && valid && has_a) count(q1r, q1g, q1b, 2u);
            er = rd_f16_gamma_value<HIST>(c2.r, kc, q2r); eg = rd_f16_gamma_value<HIST>(c2.g, kc, q2g); eb = rd_f16_gamma_value<HIST>(c2.b, kc, q2b);
            hb0 = rd_pack_h2(er, eg); hb1 = rd_pack_h2(eb, 1.0f);
            if (HIST && valid && has_b) count(q2r, q2g, q2b, 1u);
            er = rd_f16_gamma_value<HIST>(c3.r, kc, q3r); eg = rd_f16_gamma_value<HIST>(c3.g, kc, q3g); eb = rd_f16_gamma_value<HIST>(c3.b, kc, q3b);
            hc0 = rd_pack_h2(er, eg); hc1 = rd_pack_h2(eb, 1.0f);
            if (HIST && valid && has_b) count(q3r, q3g, q3b, 1u);
        } else if constexpr (F32T) {
            // One triple at a time: pinned gamma, codes into the histogram, r g b into the wave-private store stage
            // ([lane][c1, c2, c3]; the alphas are already there), so a finished triple holds no registers.  The stage still
            // holds the PREVIOUS tile until store_tile has read it -- which the loop does before it computes this one, and
            // a wave's LDS operations execute in order.
            rd_f4 *st = stage + (size_t)wave * 192u;
            if constexpr (MATH == RD_MATH_PROBE) {                // the pattern probe: the samples as they are, nothing counted
                *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 0u]) = c1;
                *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 1u]) = c2;
                *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 2u]) = c3;
            } else {
            const rd_rgb g1 = { rd_gamma_clamp(c1.r), rd_gamma_clamp(c1.g), rd_gamma_clamp(c1.b) };
            if (HIST && valid && has_a) count(rd_q8(g1.r), rd_q8(g1.g), rd_q8(g1.b), 2u);
            *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 0u]) = g1;
            const rd_rgb g2 = { rd_gamma_clamp(c2.r), rd_gamma_clamp(c2.g), rd_gamma_clamp(c2.b) };
            if (HIST && valid && has_b) count(rd_q8(g2.r), rd_q8(g2.g), rd_q8(g2.b), 1u);
            *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 1u]) = g2;
            const rd_rgb g3 = { rd_gamma_clamp(c3.r), rd_gamma_clamp(c3.g), rd_gamma_clamp(c3.b) };
            if (HIST && valid && has_b) count(rd_q8(g3.r), rd_q8(g3.g), rd_q8(g3.b), 1u);
            *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 2u]) = g3;
            }
        } else if (HIST || FMT == RD_FMT_RGBA_U8 || FMT == RD_FMT_RGB_U8) {
            q1r = rd_q8(c1.r); q1g = rd_q8(c1.g); q1b = rd_q8(c1.b);
            q2r = rd_q8(c2.r); q2g = rd_q8(c2.g); q2b = rd_q8(c2.b);
            q3r = rd_q8(c3.r); q3g = rd_q8(c3.g); q3b = rd_q8(c3.b);
        }
        if (HIST && valid && (!H16 || F16LUT) && !F32T) {
            if (has_a) count(q1r, q1g, q1b, 2u);
            if (has_b) { count(q2r, q2g, q2b, 1u); count(q3r, q3g, q3b, 1u); }
        }
        }   // !separable
        rd_tile_out<FMT> r;
        if constexpr (FMT == RD_FMT_RGBA_F32) {
            if constexpr (!F32T) {                               // microbench stand-ins only: all three at the end
                rd_f4 *st = stage + (size_t)wave * 192u;
                *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 0u]) = c1;
                *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 1u]) = c2;
                *reinterpret_cast<rd_rgb *>(&st[lane * 3u + 2u]) = c3;
            }
        } else if constexpr (FMT == RD_FMT_RGBA_F16 && H16) {
            r.a0 = ha0; r.a1 = ha1; r.b0 = hb0; r.b1 = hb1; r.c0 = hc0; r.c1 = hc1;
        } else if constexpr (FMT == RD_FMT_RGBA_F16) {
            r.a0 = rd_pack_h2(c1.r, c1.g); r.a1 = rd_pack_h2(c1.b, 1.0f);
            r.b0 = rd_pack_h2(c2.r, c2.g); r.b1 = rd_pack_h2(c2.b, 1.0f);
            r.c0 = rd_pack_h2(c3.r, c3.g); r.c1 = rd_pack_h2(c3.b, 1.0f);
        } else if constexpr (Q8LUT) {                            // RGBA8 / RGB8 from byte 2 of the table sums: two byte permutes per pixel
            r.v1 = rd_pack_rgba_b2<FMT == RD_FMT_RGBA_U8>(q1r, q1g, q1b);
            r.v2 = rd_pack_rgba_b2<FMT == RD_FMT_RGBA_U8>(q2r, q2g, q2b);
            r.v3 = rd_pack_rgba_b2<FMT == RD_FMT_RGBA_U8>(q3r, q3g, q3b);
        } else if constexpr (BITS) {                             // RGBA8 / RGB8 from the encodings: two byte permutes per pixel
            r.v1 = rd_pack_rgba_bits<FMT == RD_FMT_RGBA_U8>(q1r, q1g, q1b);
            r.v2 = rd_pack_rgba_bits<FMT == RD_FMT_RGBA_U8>(q2r, q2g, q2b);
            r.v3 = rd_pack_rgba_bits<FMT == RD_FMT_RGBA_U8>(q3r, q3g, q3b);
        } else if constexpr (FMT == RD_FMT_RGB_U8) {
            r.v1 = q1r | (q1g << 8) | (q1b << 16);
            r.v2 = q2r | (q2g << 8) | (q2b << 16);
            r.v3 = q3r | (q3g << 8) | (q3b << 16);
        } else {
            r.v1 = q1r | (q1g << 8) | (q1b << 16) | 0xff000000u;
            r.v2 = q2r | (q2g << 8) | (q2b << 16) | 0xff000000u;
            r.v3 = q3r | (q3g << 8) | (q3b << 16) | 0xff000000u;
        }
        return r;
    };

    // surface stores of one tile.  A missing row (first / last unit) is redirected onto the existing
    // one with that row's value, so the number of store instructions never depends on the tile.
    auto store_tile = [&](RD_GLOBAL void *out, uint32_t tu, uint32_t tq, const rd_tile_out<FMT> &r) {
        const bool has_a = tu != 0u, has_b = 2u * tu < H;
        const size_t row_b_px = (size_t)(has_b ? 2u * tu : 2u * tu - 1u) * W;
        const size_t row_a_px = has_a ? (size_t)(2u * tu - 1u) * W : row_b_px;
        const uint32_t q0 = qbase(tq);                           // the tile's first quad
        const bool valid = FULL || q0 + lane < qpr;
        if constexpr (SHIFT) {
            // 64-byte-aligned store windows (see RD_TILES_SHIFT above): in window-relative pixels row a stores [ra0, ra0 + na) and
            // row b [rb0, rb0 + nb); a lane past the end of its window repeats the window's last pixel (same address, same
            // data), so the four stores stay unconditional.
            (void)r; (void)valid;
            const rd_f4 *st = stage + (size_t)wave * 192u;
            __builtin_amdgcn_wave_barrier();
            RD_GLOBAL rd_f4 *o = reinterpret_cast<RD_GLOBAL rd_f4 *>(out);
            const bool last = tq + 1u == tpu;
            const uint32_t row_a = has_a ? 2u * tu - 1u : 2u * tu, row_b = has_b ? 2u * tu : 2u * tu - 1u;    // absolute rows, as row_*_px
            const uint32_t sa = (0u - row_a * W) & 3u, sb = (0u - row_b * W) & 3u;                             // the rows' shifts
            const uint32_t nat = 124u * tq - 2u * q0;            // where pixel 124 tq lies in the window (0, or more in the pulled-back last tile)
            // (a window never leaves the tile's 128 pixels: next to a last tile that owns a single quad the window before it is
            //  pulled back too, and a shift of 3 would push its end -- or the last tile's start -- past them; a window that comes out
            //  empty re-stores its tile's last pixel)
            const uint32_t ea = last ? 128u : (nat + 124u + sa < 128u ? nat + 124u + sa : 128u);
            const uint32_t eb = last ? 128u : (nat + 124u + sb < 128u ? nat + 124u + sb : 128u);
            const uint32_t ra0 = tq ? (nat + sa < ea ? nat + sa : ea - 1u) : 0u, rb0 = tq ? (nat + sb < eb ? nat + sb : eb - 1u) : 0u;
            const uint32_t na = ea - ra0, nb = eb - rb0;
            const uint32_t ma = na - 1u, mb = nb - 1u;
            RD_GLOBAL rd_f4 *oa = o + (row_a_px + (size_t)q0 * 2u), *ob = o + (row_b_px + (size_t)q0 * 2u);   // wave-uniform bases (SALU)
#pragma unroll
            for (uint32_t half = 0; half < 2u; ++half) {
                const uint32_t p = half * 64u + lane;
                const uint32_t xa = ra0 + (p < ma ? p : ma), xb = rb0 + (p < mb ? p : mb);   // window-relative pixels (one v_min each)
                __builtin_assume(xa < 128u && xb < 128u);        // (inside the tile: the byte offset fits the 32-bit lane offset of a store)
                // row a is the pair's odd row: both pixels of a quad are c1; row b: c2 / c3 by the pixel's parity
                uint32_t ia = (xa >> 1) * 3u, ib = (xb >> 1) * 3u + 1u + (xb & 1u);
                if (__builtin_expect(!(has_a && has_b), 0)) {    // first / last unit only (wave-uniform): a missing row is the other one
                    // again -- row_a_px == row_b_px then, so is the shift, so xa == xb: only the stage INDEX has to follow (a branch, as
                    // in the f16 path: as selects it would cost every tile)
                    if (!has_a) ia = ib; else ib = ia;
                    asm volatile("" : "+v"(ia), "+v"(ib));
                }
                const rd_f4 va = st[ia], vb = st[ib];
                // (wave-uniform base + a 32-bit byte offset per lane: global_store_dwordx4 v_off, v[data], s[base], as the other tilings)
                __builtin_nontemporal_store(va, reinterpret_cast<RD_GLOBAL rd_f4 *>(reinterpret_cast<RD_GLOBAL char *>(oa) + (uint32_t)(xa * 16u)));
                __builtin_nontemporal_store(vb, reinterpret_cast<RD_GLOBAL rd_f4 *>(reinterpret_cast<RD_GLOBAL char *>(ob) + (uint32_t)(xb * 16u)));
            }
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (FMT == RD_FMT_RGBA_F32) {
            (void)r;
            const rd_f4 *st = stage + (size_t)wave * 192u;       // written by compute_tile
            __builtin_amdgcn_wave_barrier();
            RD_GLOBAL rd_f4 *o = reinterpret_cast<RD_GLOBAL rd_f4 *>(out);
#pragma unroll
            for (uint32_t half = 0; half < 2u; ++half) {
                const uint32_t p = half * 64u + lane;                  // pixel within the tile
                const uint32_t px = q0 * 2u + p;                       // column
                const uint32_t ja = (p >> 1) * 3u;                     // row a: c1 of quad p/2
                const uint32_t jb = ja + 1u + (p & 1u);                // row b: c2 (even col) / c3 (odd col)
                const rd_f4 va = st[has_a ? ja : jb];                  // a missing row re-stores the other one: the select is
                const rd_f4 vb = st[has_b ? jb : ja];                  // made on the LDS index (2 per half), not on 8 floats
                if (FULL || px < 2u * qpr) {                           // (an odd W: column W - 1 is rd_develop_lastcol's)
                    // wave-uniform base (row + the tile's first column: SALU) + the lane's own pixel, like the loads
#ifdef RD_ST_PLAIN
                    (o + (row_a_px + (size_t)q0 * 2u))[p] = va;
                    (o + (row_b_px + (size_t)q0 * 2u))[p] = vb;
#else
                    __builtin_nontemporal_store(va, (o + (row_a_px + (size_t)q0 * 2u)) + p);
                    __builtin_nontemporal_store(vb, (o + (row_b_px + (size_t)q0 * 2u)) + p);
#endif
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (FMT == RD_FMT_RGBA_F16) {
            // 2 px = 16 B per lane per row.  A row starts at pixel row * W: 8-byte aligned for every W, 16-byte aligned only when
            // row * W is even -- so the address is formed in pixels and the 16-byte store is declared 8-byte aligned (one
            // global_store_dwordx4 either way)
            typedef rd_u4 rd_u4_a8 __attribute__((aligned(8)));
            RD_GLOBAL rd_u2 *o = reinterpret_cast<RD_GLOBAL rd_u2 *>(out);
            rd_u4 va = { r.a0, r.a1, r.a0, r.a1 }, vb = { r.b0, r.b1, r.c0, r.c1 };
            if (__builtin_expect(!(has_a && has_b), 0)) {        // first / last unit only (wave-uniform): a BRANCH -- as selects
                if (!has_a) va = vb; else vb = va;               // it cost every tile eight v_cndmask (the asm keeps it a branch)
                asm volatile("" : "+v"(va.x), "+v"(va.y), "+v"(va.z), "+v"(va.w), "+v"(vb.x), "+v"(vb.y), "+v"(vb.z), "+v"(vb.w));
            }
            if (valid) {                                         // wave-uniform base + the lane's own 16 bytes (see load_tile)
                __builtin_nontemporal_store(va, reinterpret_cast<RD_GLOBAL rd_u4_a8 *>(o + (row_a_px + (size_t)q0 * 2u)) + lane);
                __builtin_nontemporal_store(vb, reinterpret_cast<RD_GLOBAL rd_u4_a8 *>(o + (row_b_px + (size_t)q0 * 2u)) + lane);
            }
        } else if constexpr (FMT == RD_FMT_RGB_U8) {
            // 2 px = 6 B per lane per row: repack the wave's 384-B rows through LDS (three 16-bit writes per
            // lane and row) and store whole dwords: lane l stores dword l, lanes 0..31 also dword 64+l.
            // Full tiles only (the host routes W < 128 to rd_develop_map).  A row (3 W bytes) and a tile (6 q0 bytes) start on
            // an EVEN byte, a multiple of 4 only when W % 4 == 0 and q0 is even: the dword stores are declared 2-byte
            // aligned (rd_u32_a2) -- gfx950 under HSA runs with unaligned global access enabled, hipcc keeps them
            // global_store_dword, and the oracle tests at W = 6002 / 130 / 202 exercise the odd case.
            uint16_t *s16 = rgb16 + (size_t)wave * 384u;
            uint32_t pa = r.v1, pb = r.v1, pc = r.v2, pd = r.v3;       // row a: (c1,c1)  row b: (c2,c3)
            if (!has_a) { pa = pc; pb = pd; }
            if (!has_b) { pc = pa; pd = pb; }
            s16[lane * 3u + 0u] = (uint16_t)(pa & 0xffffu);                            // r0 g0
            s16[lane * 3u + 1u] = (uint16_t)((pa >> 16) | ((pb & 0xffu) << 8));        // b0 r1
            s16[lane * 3u + 2u] = (uint16_t)(pb >> 8);                                 // g1 b1
            s16[192u + lane * 3u + 0u] = (uint16_t)(pc & 0xffffu);
            s16[192u + lane * 3u + 1u] = (uint16_t)((pc >> 16) | ((pd & 0xffu) << 8));
            s16[192u + lane * 3u + 2u] = (uint16_t)(pd >> 8);
            __builtin_amdgcn_wave_barrier();
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s16);
            // (an odd W puts every other row on an ODD byte: the dwords are declared 1-byte aligned)
            typedef uint32_t rd_u32_a1 __attribute__((aligned(1)));
            RD_GLOBAL rd_u32_a1 *oa = reinterpret_cast<RD_GLOBAL rd_u32_a1 *>(reinterpret_cast<RD_GLOBAL uint8_t *>(out) + (row_a_px + (size_t)q0 * 2u) * 3u);
            RD_GLOBAL rd_u32_a1 *ob = reinterpret_cast<RD_GLOBAL rd_u32_a1 *>(reinterpret_cast<RD_GLOBAL uint8_t *>(out) + (row_b_px + (size_t)q0 * 2u) * 3u);
            const uint32_t a0 = s32[lane], b0 = s32[96u + lane];
            const uint32_t l2 = lane & 31u;
            const uint32_t a1 = s32[64u + l2], b1 = s32[160u + l2];
#ifdef RD_RGB8_ST_PLAIN       // A/B build (tools/build_ab_libs.sh r6): write-back stores, so that the L2 may merge the two halves of a
            oa[lane] = a0;         // 32-byte sector that a ragged row splits between neighbouring tiles (profiles/r06_rgb8_ragged_ab.txt)
            ob[lane] = b0;
            if (lane < 32u) { oa[64u + lane] = a1; ob[64u + lane] = b1; }
#else
            __builtin_nontemporal_store(a0, oa + lane);
            __builtin_nontemporal_store(b0, ob + lane);
            if (lane < 32u) {
                __builtin_nontemporal_store(a1, oa + 64u + lane);
                __builtin_nontemporal_store(b1, ob + 64u + lane);
            }
#endif
            __builtin_amdgcn_wave_barrier();
        } else {
            // 2 px = 8 B per lane per row; formed in pixels and declared 4-byte aligned for the same reason
            typedef rd_u2 rd_u2_a4 __attribute__((aligned(4)));
            RD_GLOBAL uint32_t *o = reinterpret_cast<RD_GLOBAL uint32_t *>(out);
            rd_u2 va = { r.v1, r.v1 }, vb = { r.v2, r.v3 };
            if (__builtin_expect(!(has_a && has_b), 0)) {        // first / last unit only (wave-uniform): a branch, not selects
                if (!has_a) va = vb; else vb = va;
                asm volatile("" : "+v"(va.x), "+v"(va.y), "+v"(vb.x), "+v"(vb.y));
            }
            if (valid) {
                __builtin_nontemporal_store(va, reinterpret_cast<RD_GLOBAL rd_u2_a4 *>(o + (row_a_px + (size_t)q0 * 2u)) + lane);
                __builtin_nontemporal_store(vb, reinterpret_cast<RD_GLOBAL rd_u2_a4 *>(o + (row_b_px + (size_t)q0 * 2u)) + lane);
            }
        }
    };

    if (tile < ntiles) {
        uint32_t top, bot;
        if (BURST) {
            // ---- read-only burst (f32 surface only: the other surfaces are VALU-bound and gain nothing).
            // HBM3E serves this kernel's 1:8 read:write mix badly when 256-B reads are sprinkled between the
            // write streams: every isolated read costs the DRAM channel a write->read->write turnaround
            // (measured: the same bytes take 85-88 us mixed but 65 us when the CFA plane is already in the
            // Infinity Cache).  So before anything is stored the grid sweeps this launch's CFA rows once,
            // 16 B per lane, fire-and-forget: LDS-DMA loads (global_load_lds_dwordx4) into a 1-KiB dump area
            // nobody reads, so no VGPR is tied up and nothing waits for the data.  The whole
            // chip is in this phase together (persistent grid, nothing stored yet); the 48 MB land in the
            // 256 MiB Infinity Cache in one pure-read burst while every wave computes its first tile, and
            // the main loop's loads are served on-die.  nt stores do not displace the lines (measured).
            // Any wave may pull any line (the cache is shared), so the sweep ignores tile ownership.  The
            // count is static (8 x 1 KiB per wave covers 64 MB per launch) so the compiler can still count
            // vmcnt for the tile-0 loads issued above; larger launches finish with a waited loop.
            // MULTI: the sweep covers frame 0 of the launch; the later frames are swept by prefetch_frame below.
            typedef uint32_t rd_u4 __attribute__((ext_vector_type(4)));
            const uint32_t row_lo = MULTI ? 0u : unit0 ? 2u * unit0 - 1u : 0u;
            const uint32_t row_hi = MULTI ? H : 2u * (unit1 - 1u) < H ? 2u * (unit1 - 1u) + 1u : H;    // exclusive
            const RD_GLOBAL uint16_t *cfa_b = (const RD_GLOBAL uint16_t *)(MULTI ? descs[0].cfa : cfa_arg);
            // (rows of a width that is not a multiple of 8 start off a 16-byte boundary: the sweep starts at the boundary below
            // the band's first row -- still inside the plane -- and covers whole 16-B chunks up to the band's end)
            const size_t b_lo = ((size_t)row_lo * W * sizeof(uint16_t)) & ~(size_t)15u;
            const RD_GLOBAL rd_u4 *src = reinterpret_cast<const RD_GLOBAL rd_u4 *>(reinterpret_cast<const RD_GLOBAL char *>(cfa_b) + b_lo);
            const size_t n16 = ((size_t)row_hi * W * sizeof(uint16_t) - b_lo) / 16u;
            const size_t g0 = (size_t)(blockIdx.x * RD_WAVES + wave) * 64u, gstride = (size_t)nwaves * 64u;
            // (host guarantees for BURST launches: cfa 16-byte aligned, W >= 128, at least 1 MB of CFA rows, so src is aligned,
            //  n16 >= 64 and every instruction below is unconditional; W may be odd -- the sweep walks bytes)
            {                                                    // launches > 64 MB only: the part beyond 8 x 1 KiB per wave
                uint32_t sink = 0;
                for (size_t i0 = g0 + 8u * gstride; i0 + 64u <= n16; i0 += gstride) {
                    const rd_u4 v = src[i0 + lane];
                    sink ^= v.x ^ v.y ^ v.z ^ v.w;
                }
                asm volatile("" ::"v"(sink));
            }
            // Tile-0 loads + the eight LDS-DMA sweeps + "wait for the two loads only" as ONE asm statement:
            // hipcc does not count LDS-DMA instructions in its vmcnt bookkeeping and would wait vmcnt(0) for
            // the tile-0 data, i.e. for the whole burst.  Inside the statement the order is ours: the two
            // loads are the oldest of ten outstanding operations, so vmcnt(8) is exactly "they have landed".
            // Nothing else is in flight here (first memory instructions of the wave after the waited loop).
            {
                uint32_t q = qbase(qt) + lane;
                const uint32_t ra = unit ? 2u * unit - 1u : 0u;
                const uint32_t rb = 2u * unit < H ? 2u * unit : H - 1u;
                const RD_GLOBAL uint32_t *pt = reinterpret_cast<const RD_GLOBAL uint32_t *>(cfa + (size_t)ra * W) + q;
                const RD_GLOBAL uint32_t *pb = reinterpret_cast<const RD_GLOBAL uint32_t *>(cfa + (size_t)rb * W) + q;
                const uint32_t lds_base = __builtin_amdgcn_readfirstlane(          // every wave dumps into the same 1 KiB
                    (uint32_t)(size_t)(__attribute__((address_space(3))) void *)(&pf_dump[0]));
                const RD_GLOBAL rd_u4 *a[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    size_t i0 = g0 + (size_t)k * gstride;
                    i0 = i0 + 64u <= n16 ? i0 : 0;               // wave-uniform clamp: re-touch the first chunk
                    a[k] = src + i0 + lane;
                }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
                asm volatile("global_load_dword %0, %2, off\n\t"
                             "global_load_dword %1, %3, off\n\t"
                             "s_mov_b32 m0, %4\n\t"
                             "s_nop 0\n\t"
                             "global_load_lds_dwordx4 %5, off" RD_SWEEP_POLICY "\n\t"
                             "global_load_lds_dwordx4 %6, off" RD_SWEEP_POLICY "\n\t"
                             "global_load_lds_dwordx4 %7, off" RD_SWEEP_POLICY "\n\t"
                             "global_load_lds_dwordx4 %8, off" RD_SWEEP_POLICY "\n\t"
                             "global_load_lds_dwordx4 %9, off" RD_SWEEP_POLICY "\n\t"
                             "global_load_lds_dwordx4 %10, off" RD_SWEEP_POLICY "\n\t"
                             "global_load_lds_dwordx4 %11, off" RD_SWEEP_POLICY "\n\t"
                             "global_load_lds_dwordx4 %12, off" RD_SWEEP_POLICY "\n\t"
                             "s_waitcnt vmcnt(8)"
                             : "=&v"(top), "=&v"(bot)
                             : "v"(pt), "v"(pb), "s"(lds_base), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]),
                               "v"(a[5]), "v"(a[6]), "v"(a[7])
                             : "memory", "m0");
#pragma clang diagnostic pop
            }
        } else {
            load_tile(unit, qt, top, bot);
        }
        // Land the first tile's loads BEFORE the loop.  hipcc places one static s_waitcnt per use and
        // merges the loop-entry and back-edge states; with loads still pending at the loop header it
        // emits vmcnt(1)/vmcnt(2) there, which on the back edge means "drain the stores in flight".
        asm volatile("" : "+v"(top), "+v"(bot));

        // MULTI + BURST: the read burst of the frames after the first.  There is no chip-wide quiet phase inside a
        // multi-frame launch, but there is a chip-wide CLOCK: all ticket counters advance together, so every wave
        // enters frame f within about one tile time of every other.  The wave whose load stage draws its first tile
        // of frame f issues its share of frame f's sweep -- the same fire-and-forget LDS-DMA as above, 1 KiB per
        // instruction, into a dump area nobody reads -- so the whole chip requests the plane in one short window and
        // the frame's remaining loads are served by the Infinity Cache.  hipcc does not count these in its vmcnt
        // bookkeeping; they retire in order with the tile loads issued just after them, long before the compute
        // stage that follows has finished.  Measured (tools/bench_batch_ab.py, 8 frames per launch): 79.4 us per
        // frame with the sweep at frame entry, 84.1 us without any sweep of the later frames; starting the sweep
        // earlier (when the load stage stands at 90 ... 99 % of the PREVIOUS frame's tiles) was no better
        // (79.3 ... 80.1), much earlier (50 ... 75 %) worse: the sweep then holds back the stores of the tiles in
        // flight.  The threshold variant is gone; "at frame entry" is what remains.  Round 3 tried the two other extremes --
        // every frame of the launch swept at launch start, and the next frame's plane swept piece by piece in address
        // order as the ticket front moves through the current one -- both slower (profiles/r03_memory_floor.txt).
        auto prefetch_frame = [&](uint32_t fr) {
            const char *base = reinterpret_cast<const char *>(descs[fr].cfa);
            const uint32_t n1k = (uint32_t)(((size_t)H * W * sizeof(uint16_t)) >> 10);   // whole 1-KiB pieces of a plane
            const uint32_t dump = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void *)(&pf_dump[0]));
            const uint32_t voff = lane * 16u;
            uint32_t piece = blockIdx.x * RD_WAVES + wave;       // pieces g, g + nwaves, ...: eight cover 64 MB at 8192 waves
#pragma unroll 1
            for (int k = 0; k < 8 && piece < n1k; ++k, piece += nwaves) {            // wave-uniform
                const char *sb = base + ((size_t)piece << 10);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" RD_SWEEP_POLICY ::"s"(dump), "v"(voff), "s"(sb) : "memory", "m0");
#pragma clang diagnostic pop
            }
        };

        // Software pipeline, one tile deep on each side:
        //   iteration i:  issue loads(i+1) | store tile i-1 (registers -> LDS transpose -> HBM) | compute tile i
        // vmcnt retires in order, so waiting for loads(i+1) only requires the stores of tile i-2 to
        // have completed; the stores of tile i-1 overlap this wave's own arithmetic.
        uint32_t nunit = unit, nqt = qt;
        bool more;
        auto next_tile = [&]() {                                 // draw, locate, move the load stage into the tile's frame
            const uint32_t ntile = draw();
            more = ntile != ~0u;
            if (more) {
                uint32_t f, tin;
                locate(ntile, f, tin, nunit, nqt);
                (void)tin;
                if (MULTI && f != fr_n) {                    // the load stage enters another frame
                    fr_n = f;
                    cfa = (const RD_GLOBAL uint16_t *)descs[f].cfa;
                    if (BURST) prefetch_frame(f);            // this wave's share of the frame's sweep
                }
            }
        };
        next_tile();
        uint32_t ntop, nbot;
        load_tile(nunit, nqt, ntop, nbot);
        rd_tile_out<FMT> pend = compute_tile(unit, qt, top, bot);
        uint32_t punit = unit, pqt = qt;
        RD_GLOBAL void *out_p = out_c;
        asm volatile("" : "+v"(ntop), "+v"(nbot));               // same reason: nothing pending at the loop header
        while (more) {
#ifdef RD_PROBE
            ++probe_tiles;
#endif
            unit = nunit; qt = nqt; top = ntop; bot = nbot;
            if (MULTI && fr_n != fr_c) {                         // the compute stage enters another frame: its uniforms and surface
                fr_c = fr_n;
                out_c = (RD_GLOBAL void *)descs[fr_c].out;
                adopt(descs[fr_c].u);
            }
            next_tile();
            load_tile(nunit, nqt, ntop, nbot);                   // unconditional (the last pass re-reads its own)
            store_tile(out_p, punit, pqt, pend);
            pend = compute_tile(unit, qt, top, bot);
            punit = unit; pqt = qt; out_p = out_c;
        }
        store_tile(out_p, punit, pqt, pend);
    } else {
        (void)draw();                                            // launch smaller than the grid: still one (failed) draw per wave
    }
    if (HIST) rd_hist_flush(lh, slab32, slab64);
    if constexpr (STAMP) {
        if (threadIdx.x == 0 && stamps) {
            stamps[blockIdx.x * 2u + 0u] = (uint32_t)(__builtin_amdgcn_s_memtime() - stamp_t0);
            stamps[blockIdx.x * 2u + 1u] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - stamp_r0);
        }
    }
#ifdef RD_PROBE
    if (lane == 0) atomicAdd(&rd_probe_buf[blockIdx.x * 8u + 6u], probe_tiles + (tile < ntiles ? 1u : 0u));
    if (threadIdx.x == 0) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        rd_probe_buf[blockIdx.x * 8u + 0u] = (uint32_t)(t1 - probe_t0);
        rd_probe_buf[blockIdx.x * 8u + 1u] = (uint32_t)(r1 - probe_r0);
        rd_probe_buf[blockIdx.x * 8u + 2u] = (uint32_t)probe_r0;
        rd_probe_buf[blockIdx.x * 8u + 3u] = (uint32_t)r1;
        rd_probe_buf[blockIdx.x * 8u + 4u] = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
        rd_probe_buf[blockIdx.x * 8u + 5u] = __builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
    }
#endif
}

// One frame, or one row band of a frame, per launch: rd_render*, the export ring, and rd_batch_develop when the
// multi-frame launch is switched off (RD_BATCH_PERSISTENT=0).
template <int FMT, bool HIST, int TILES, int MATH = RD_MATH_STRICT, bool BURST = (FMT == RD_FMT_RGBA_F32)>
__global__ void __launch_bounds__(RD_BLOCK) __attribute__((amdgpu_num_sgpr(RD_NUM_SGPR))) RD_WAVES_PER_EU
rd_develop_quads(const uint16_t *__restrict__ cfa, void *__restrict__ out, uint32_t W, uint32_t H,
                 uint32_t unit0, uint32_t unit1, uint32_t tpu, uint32_t tpu_magic, uint32_t tq_k,
                 uint32_t tq_tmax, uint32_t *tq, rd_ku u_arg, uint32_t *slab32, unsigned long long *slab64)
{
    rd_quads_body<FMT, HIST, TILES, MATH, BURST, false>(cfa, out, W, H, unit0, unit1, tpu, tpu_magic, tq_k, tq_tmax, tq, u_arg,
                                                       slab32, slab64, nullptr, 1u, 0u, 0u);
}

// `nframes` whole frames of one size per launch (descs[0 .. nframes-1], frame-major tile index; tpf = tiles per
// frame): the batch path.  Histogram counts of all frames meet in the workgroup's LDS table (u32: the host keeps
// nframes * W * H below 2^32) and are added to its u64 slab row at the end of the launch.
template <int FMT, bool HIST, int TILES, int MATH = RD_MATH_STRICT, bool BURST = (FMT == RD_FMT_RGBA_F32), bool STAMP = false>
__global__ void __launch_bounds__(RD_BLOCK) __attribute__((amdgpu_num_sgpr(RD_NUM_SGPR))) RD_WAVES_PER_EU
rd_develop_batch(const rd_frame_desc *__restrict__ descs, uint32_t nframes, uint32_t W, uint32_t H, uint32_t tpu,
                 uint32_t tpu_magic, uint32_t tpf, uint32_t tpf_magic, uint32_t tq_k, uint32_t tq_tmax, uint32_t *tq,
                 unsigned long long *slab64, uint32_t *stamps)
{
    rd_quads_body<FMT, HIST, TILES, MATH, BURST, true, STAMP>(nullptr, nullptr, W, H, 0u, H / 2u + 1u, tpu, tpu_magic, tq_k, tq_tmax, tq,
                                                             descs[0].u, nullptr, slab64, descs, nframes, tpf, tpf_magic, stamps);
}

// ---------------------------------------------------------------------------------------------
// rd_develop_map -- general target (tw x th), zoom and pan: one output pixel per lane.
// vs_main evaluated at the pixel centre (shaders.rs:31-57), bounds test (:174-178), trunc to
// pixel_coords (:184-187), then the per-pixel demosaic with get_neighbor's clamps.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rd_tap(const uint16_t *cfa, int32_t W, int32_t H, int32_t x,
                                        int32_t y, uint32_t bl)
{
    x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
    y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
    return rd_norm(cfa[(size_t)y * (size_t)W + (size_t)x], bl);
}

// One pixel (px, py) of the frame, inside it: the demosaic selection of shaders.rs:127-155 with get_neighbor's clamps and the
// colour stack.  LINEAR = true stops before the gamma / clamp step (the 8-bit surfaces finish with rd_q8_gamma).  Shared by
// rd_develop_map (after its pixel map) and rd_develop_lastcol (the last column of an odd-width frame).
template <int MATH, bool LINEAR>
__device__ __forceinline__ rd_rgb rd_develop_px(const uint16_t *cfa, int32_t W, int32_t H, int32_t px, int32_t py, const rd_ku &u)
{
    const float n = rd_tap(cfa, W, H, px, py, u.black_level);
    const bool even_row = ((py + 1) & 1) == 0;   // shaders.rs:115-116
    const bool even_col = (px & 1) == 0;
    float r, g, b;
    if (even_row) {
        if (even_col) { g = n; b = rd_tap(cfa, W, H, px + 1, py, u.black_level); r = rd_tap(cfa, W, H, px, py + 1, u.black_level); }
        else          { b = n; g = rd_tap(cfa, W, H, px - 1, py, u.black_level); r = rd_tap(cfa, W, H, px - 1, py + 1, u.black_level); }
    } else {
        if (even_col) { r = n; g = rd_tap(cfa, W, H, px + 1, py, u.black_level); b = rd_tap(cfa, W, H, px, py - 1, u.black_level); }
        else          { g = n; r = rd_tap(cfa, W, H, px - 1, py, u.black_level); b = rd_tap(cfa, W, H, px, py - 1, u.black_level); }
    }
    if constexpr (LINEAR) {
        float tr[1] = { r }, tg[1] = { g }, tb[1] = { b };
        rd_colour_n<1, MATH, false>(u, tr, tg, tb);
        return rd_rgb{ tr[0], tg[0], tb[0] };
    } else {
        return rd_colour_m<MATH>(u, r, g, b);
    }
}

template <int FMT, bool HIST, int MATH = RD_MATH_STRICT>
__global__ void __launch_bounds__(RD_BLOCK)
rd_develop_map(const uint16_t *__restrict__ cfa, void *__restrict__ out, uint32_t W, uint32_t H,
               uint32_t tw, uint32_t th, rd_ku u, uint32_t *slab32, unsigned long long *slab64)
{
    constexpr bool Q8ONLY = FMT == RD_FMT_RGBA_U8 || FMT == RD_FMT_RGB_U8;
    __shared__ uint32_t lh[HIST ? 768 * RD_HK : 1];
    if (HIST) rd_hist_zero(lh);
    const uint32_t copy = threadIdx.x & (RD_HK - 1);
    const uint32_t total = tw * th;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const uint32_t j = idx / tw, i = idx - j * tw;
        const float sx = ((float)i + 0.5f) / (float)tw;
        const float sy = ((float)j + 0.5f) / (float)th;
        const float tx = ((sx - 0.5f) / u.zoom - u.pan_x) + 0.5f;
        const float ty = ((sy - 0.5f) / u.zoom - u.pan_y) + 0.5f;
        rd_rgb c = { 0.0f, 0.0f, 0.0f };
        if (!(tx < 0.0f || tx > 1.0f || ty < 0.0f || ty > 1.0f)) {   // the shader's own test (:174-175): a NaN coordinate (zoom = 0) passes it
            int32_t px = tx != tx ? 0 : (int32_t)(tx * (float)W);    // ... and i32(NaN) is 0 (the pinned lowering; what v_cvt_i32_f32 returns)
            int32_t py = ty != ty ? 0 : (int32_t)(ty * (float)H);
            // tx == 1.0 exactly: px == W, one past the frame.  The shader carries that coordinate on -- the Bayer parity is taken on
            // it -- and only its loads see the border (rd_tap clamps every one, the centre's included).
            c = rd_develop_px<MATH, Q8ONLY>(cfa, W, H, px, py, u);   // Q8ONLY: linear values; rd_q8_gamma finishes (out-of-bounds pixels stay 0 -> code 0)
        }
        uint32_t qr = 0, qg = 0, qb = 0;
        if constexpr (Q8ONLY) { qr = rd_q8_gamma(c.r); qg = rd_q8_gamma(c.g); qb = rd_q8_gamma(c.b); }
        else if (HIST) { qr = rd_q8(c.r); qg = rd_q8(c.g); qb = rd_q8(c.b); }
        rd_store_px<FMT>(out, idx, c, qr, qg, qb);
        if (HIST) rd_hist_add(lh, copy, qr, qg, qb, 1u);
    }
    if (HIST) rd_hist_flush(lh, slab32, slab64);
}

// ---------------------------------------------------------------------------------------------
// rd_develop_lastcol -- the last column of frames whose width is ODD (round 6: the reference renders any texture size,
// shaders.rs:181-187, loader.rs:57-58; a cropped plane can have one).
// The export kernel's lanes own 2 x 2 blocks, so it covers columns [0, W - 1) of such a frame (its row stride is W all the
// same: rows then start on odd 16-bit boundaries, see load_tile).  Column W - 1 -- an even index, whose right-hand neighbour
// clamps onto itself (shaders.rs:163-166) -- is one pixel per row: one lane per (frame, row) evaluates it with the map
// kernel's per-pixel code (rd_develop_px: the same bits as every other path), stores it, and ADDS its histogram counts to the
// slab rows the export kernel has just written.  Launched behind that kernel on its stream; rows [row0, row1) of `nframes`
// frames (descs != nullptr: a multi-frame launch's descriptors; otherwise one frame from the arguments).
// ---------------------------------------------------------------------------------------------
template <int FMT, bool HIST, int MATH = RD_MATH_STRICT>
__global__ void __launch_bounds__(256)
rd_develop_lastcol(const uint16_t *__restrict__ cfa_arg, void *__restrict__ out_arg, const rd_frame_desc *__restrict__ descs,
                   uint32_t nframes, uint32_t W, uint32_t H, uint32_t row0, uint32_t row1, rd_ku u_arg, uint32_t *slab32,
                   unsigned long long *slab64)
{
    constexpr bool Q8ONLY = FMT == RD_FMT_RGBA_U8 || FMT == RD_FMT_RGB_U8;
    __shared__ uint32_t lh[HIST ? 768 * RD_HK : 1];
    if (HIST) rd_hist_zero(lh);
    const uint32_t copy = threadIdx.x & (RD_HK - 1);
    const uint32_t rows = row1 - row0, total = nframes * rows;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const uint32_t f = idx / rows, y = row0 + (idx - f * rows);
        const uint16_t *cfa = descs ? descs[f].cfa : cfa_arg;
        void *out = descs ? descs[f].out : out_arg;
        const rd_ku u = descs ? descs[f].u : u_arg;
        const rd_rgb c = rd_develop_px<MATH, Q8ONLY>(cfa, (int32_t)W, (int32_t)H, (int32_t)W - 1, (int32_t)y, u);
        uint32_t qr = 0, qg = 0, qb = 0;
        if constexpr (Q8ONLY) { qr = rd_q8_gamma(c.r); qg = rd_q8_gamma(c.g); qb = rd_q8_gamma(c.b); }
        else if (HIST) { qr = rd_q8(c.r); qg = rd_q8(c.g); qb = rd_q8(c.b); }
        rd_store_px<FMT>(out, (size_t)y * W + (W - 1u), c, qr, qg, qb);
        if (HIST) rd_hist_add(lh, copy, qr, qg, qb, 1u);
    }
    if (HIST) {                                                   // ADD to this workgroup's slab row (the export kernel wrote it)
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < 768u; t += blockDim.x) {
            uint32_t sum = 0;
#pragma unroll 8
            for (uint32_t i = 0; i < RD_HK; ++i) sum += lh[t * RD_HK + ((t + i) & (RD_HK - 1))];
            if (slab64) slab64[(size_t)blockIdx.x * 768u + t] += sum;
            else slab32[(size_t)blockIdx.x * 768u + t] += sum;
        }
    }
}

// calculate_histogram (pipeline.rs:720-736) on an RGBA8 buffer of npx pixels.
__global__ void __launch_bounds__(RD_BLOCK)
rd_hist_u8(const uint32_t *__restrict__ rgba, uint32_t npx, uint32_t *slab32)
{
    __shared__ uint32_t lh[768 * RD_HK];
    rd_hist_zero(lh);
    const uint32_t copy = threadIdx.x & (RD_HK - 1);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < npx; idx += stride) {
        const uint32_t v = rgba[idx];
        rd_hist_add(lh, copy, v & 0xffu, (v >> 8) & 0xffu, (v >> 16) & 0xffu, 1u);
    }
    rd_hist_flush(lh, slab32, nullptr);
}

// Fold the per-workgroup slab rows: block b owns bins [32b, 32b+32), thread (g, j) sums rows g, g+8, ...
// of bin 32b+j; the partial sums per bin meet in LDS.  Launch: 24 blocks x RD_FOLD_THREADS threads (round 3: 1024
// instead of 256 -- the fold sits behind every single-frame render with a histogram, and a thread's chain of dependent row
// loads was most of its 11.5 us).
#define RD_FOLD_THREADS 1024     // 32 bins x 32 row groups: a thread sums at most RD_MAX_BLOCKS / 32 rows (16 of a 512-row slab)
template <typename T>
__device__ __forceinline__ T rd_fold_bins(T *__restrict__ slab, uint32_t nblocks, bool clear)
{
    constexpr uint32_t NG = RD_FOLD_THREADS / 32u;
    __shared__ T part[NG][32];
    const uint32_t j = threadIdx.x & 31u, g = threadIdx.x >> 5;
    const uint32_t bin = blockIdx.x * 32u + j;
    T sum = 0;
    for (uint32_t w = g; w < nblocks; w += NG) {
        sum += slab[(size_t)w * 768u + bin];
        if (clear) slab[(size_t)w * 768u + bin] = 0;
    }
    part[g][j] = sum;
    __syncthreads();
    T tot = 0;
    if (g == 0) {
#pragma unroll
        for (uint32_t k = 0; k < NG; ++k) tot += part[k][j];
    }
    return tot;
}

// out32[bin] = sum over workgroups of slab32[wg][bin]
__global__ void __launch_bounds__(RD_FOLD_THREADS) rd_reduce_slab32(uint32_t *__restrict__ slab32, uint32_t nblocks,
                                                        uint32_t *__restrict__ out32)
{
    const uint32_t tot = rd_fold_bins<uint32_t>(slab32, nblocks, false);
    if (threadIdx.x < 32u) out32[blockIdx.x * 32u + threadIdx.x] = tot;
}

// acc[0..768) += src[0..768); dst = acc: a device's folded histogram interval added to its running sum (everything enqueued
// since the last fetch), the sum stored into page-locked host memory (rd_node_batch_histogram_enqueue).
__global__ void __launch_bounds__(768) rd_acc_hist64(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ acc,
                                                     unsigned long long *__restrict__ dst)
{
    const unsigned long long v = acc[threadIdx.x] + src[threadIdx.x];
    acc[threadIdx.x] = v;
    dst[threadIdx.x] = v;
    __threadfence_system();
}

// out64[bin] = sum over workgroups of slab64[wg][bin]; the slab is zeroed for the next batch.
__global__ void __launch_bounds__(RD_FOLD_THREADS) rd_reduce_slab64(unsigned long long *__restrict__ slab64, uint32_t nblocks,
                                                        unsigned long long *__restrict__ out64)
{
    const unsigned long long tot = rd_fold_bins<unsigned long long>(slab64, nblocks, true);
    if (threadIdx.x < 32u) out64[blockIdx.x * 32u + threadIdx.x] = tot;
}
