/*
 * oracle/develop_ref.h -- CPU restatement of the RawEditor develop path (TEST INFRASTRUCTURE).
 *
 * This is the parity ORACLE, not product code.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (raweditor_amd/, librawdev.so) never
 * includes, links or calls anything in oracle/.
 *
 * It restates, in plain scalar C with a pinned operation order (no FMA contraction, true IEEE
 * divide), what the reference's WGSL fragment shader and its wgpu host compute:
 *     /root/reference/src/gpu/shaders.rs:23-60    output pixel -> CFA pixel map  (vs_main)
 *     /root/reference/src/gpu/shaders.rs:104-169  nearest-neighbour demosaic     (debayer, get_neighbor)
 *     /root/reference/src/gpu/shaders.rs:171-267  10-slider colour stack, gamma, clamp (fs_main)
 *     /root/reference/src/gpu/pipeline.rs:125-133 preview / histogram target sizes
 *     /root/reference/src/gpu/pipeline.rs:322     Rgba8Unorm target (8-bit pack)
 *     /root/reference/src/gpu/pipeline.rs:720-736 3x256 histogram of the RGBA8 bytes
 *
 * PARITY PINNING: the reference holds no golden vectors, fixtures or tests for this path
 * (its 7 unit tests never touch src/gpu/), it is Rust+WGSL (no toolchain in the image), and
 * no NEF sample ships with it.  What pins the oracle:
 *   (i)   the reference's own shader TEXT, executed: tools/make_wgsl_golden.py reads the WGSL string
 *         of shaders.rs:14-267 where it lies and runs vs_main / fs_main through the WGSL evaluator
 *         of oracle/wgsl_eval.py (written from the WGSL specification; it holds no reference text);
 *         the vectors are tests/golden/wgsl_golden.npz and this file reproduces them bit for bit in
 *         both pow modes (tests/test_wgsl_pin_cpu.py) -- operation order, constants, selection table,
 *         abstract-float folding and conversions are therefore the text's, not a reading of it;
 *   (ii)  the analytic known-answer vectors K1..K10 of SURVEY.md section 8c (tests/golden/);
 *   (iii) agreement with an independently written numpy twin (oracle/develop_np.py).
 * What stays "parity unpinned": the behaviour WGSL leaves to the implementation (pow accuracy,
 * fma contraction, dot order, mix form, min/max of NaN, out-of-bounds textureLoad, the rasteriser's
 * interpolation) -- fixed here as named choices that only a run of the
 * reference on wgpu could confirm -- and the fixed-function UNORM8 / binary16 conversions.
 */
#ifndef DEVELOP_REF_H
#define DEVELOP_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* state/edit.rs:15-77 -- field order is the serde / uniform-block order. */
typedef struct {
    float exposure, contrast, highlights, shadows, whites, blacks;
    float vibrance, saturation, temperature, tint;
} ref_edit_params;

/* gpu/pipeline.rs:17-46 minus padding: everything fs_main/vs_main read from the uniform block. */
typedef struct {
    ref_edit_params p;
    float wb[4];  /* [R, G, B, G2]; only .rgb is used (shaders.rs:195) */
    float cm[9];  /* host row-major; the shader consumes the rows as COLUMNS (shaders.rs:209-214) */
    float zoom, pan_x, pan_y;
    uint32_t black_level; /* extension, 0 = reference behaviour (the reference subtracts nothing) */
    uint32_t math_mode;   /* REF_MATH_STRICT (literal WGSL order, default) or REF_MATH_CONTRACTED */
} ref_uniforms;

/* REF_MATH_CONTRACTED restates the same shader the way an AMD shader compiler is allowed to (and does)
 * lower it: every a*b+c of the WGSL expression tree becomes one fma, and x/d by the uniform d becomes
 * x*RN(1/d) (WGSL permits both: contraction is unspecified, division is 2.5 ULP).  The exact sequence
 * is spelled out in colour_stack_contracted(); DESIGN.md section 3b. */
enum { REF_MATH_STRICT = 0, REF_MATH_CONTRACTED = 1 };

enum { REF_POW_PINNED = 0, REF_POW_LIBM = 1 };

/* The pinned transcendental pair. */
float ref_log2f(float x);
float ref_exp2f(float z);
float ref_powf(float x, float y, int pow_mode);

void ref_default_params(ref_edit_params *p);                      /* state/edit.rs:81-95 */
void ref_derived_dims(uint32_t w, uint32_t h, uint32_t *pw, uint32_t *ph,
                      uint32_t *hw, uint32_t *hh);                  /* pipeline.rs:125-133 */

/* One output pixel (i,j) of a tw x th target -> rgba[4] (f32, post-clamp, alpha 1). */
void ref_pixel(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
               uint32_t tw, uint32_t th, uint32_t i, uint32_t j, int pow_mode, float rgba[4]);

/* Whole target, rows [row0,row1). out is tw*th*4 floats (full surface; only the band is written). */
void ref_render_f32_rows(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
                         uint32_t tw, uint32_t th, uint32_t row0, uint32_t row1, int pow_mode,
                         float *out);
void ref_render_f32_band(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
                         uint32_t tw, uint32_t th, uint32_t row0, uint32_t row1, int pow_mode,
                         float *out_band);   /* rows [row0,row1) only, band-relative storage */
void ref_render_f32(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
                    uint32_t tw, uint32_t th, int pow_mode, float *out);
/* Row-parallel over nthreads pthreads (cpu_baseline leg of bench.py). */
void ref_render_f32_mt(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
                       uint32_t tw, uint32_t th, int pow_mode, float *out, int nthreads);

/* Timing harness (bench.py cpu_baseline): persistent threads, first-touch band buffers, whole frames until budget_s. */
void ref_bench_mt(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u, int nthreads, double budget_s,
                  int max_frames, int *frames_done, double *seconds);

void ref_pack_u8(const float *rgba, size_t nfloats, uint8_t *out);   /* Rgba8Unorm store */
void ref_pack_f16(const float *rgba, size_t nfloats, uint16_t *out); /* IEEE binary16, RNE */
void ref_histogram(const uint8_t *rgba, size_t npx, uint32_t hist[768]); /* pipeline.rs:720-736 */

#ifdef __cplusplus
}
#endif
#endif
