/*
 * oracle/develop_ref.c -- CPU restatement of the RawEditor develop path.
 * TEST INFRASTRUCTURE ONLY (see develop_ref.h): never linked into or called by the product.
 *
 * Build (oracle/Makefile):  gcc -O2 -std=c11 -mfma -ffp-contract=off -fno-fast-math ...
 *   -ffp-contract=off : the operation order written here is the law; the only fused
 *                       operations are the explicit fmaf() calls of ref_log2f / ref_exp2f.
 *   -mfma             : fmaf() inlines to vfmadd (exact, single rounding either way).
 *
 * Every function names the reference lines it restates (paths relative to /root/reference).
 */
#include "develop_ref.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Pinned transcendental pair.  WGSL pow() (src/gpu/shaders.rs:217, :261) is lowered by
 * Vulkan drivers to exp2(y*log2(x)) on hardware approximations; we pin that formulation
 * on one polynomial pair (tools/fit_pow.py) so CPU and GPU agree bit for bit.
 * ------------------------------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

#define REF_SQRT_HALF_BITS 0x3f3504f3u
#define REF_FLT_MIN 1.17549435e-38f

/* log2(x) for x >= FLT_MIN (finite or +inf).  m in [sqrt(1/2), sqrt(2)), t = m-1,
 * log2(x) = e + t*P(t), P degree 7, Horner with fmaf. */
float ref_log2f(float x)
{
    static const float c[8] = {
        0x1.715472p+0f, -0x1.7155bap-1f, 0x1.ec7b64p-2f, -0x1.70bab4p-2f,
        0x1.2596a4p-2f, -0x1.001218p-2f, 0x1.e526cap-3f, -0x1.2a7c18p-3f };
    uint32_t ix = f2u(x) - REF_SQRT_HALF_BITS;
    int32_t e = (int32_t)ix >> 23;
    float m = u2f((ix & 0x007fffffu) + REF_SQRT_HALF_BITS);
    float t = m - 1.0f;
    float p = c[7];
    p = fmaf(p, t, c[6]);
    p = fmaf(p, t, c[5]);
    p = fmaf(p, t, c[4]);
    p = fmaf(p, t, c[3]);
    p = fmaf(p, t, c[2]);
    p = fmaf(p, t, c[1]);
    p = fmaf(p, t, c[0]);
    return fmaf(t, p, (float)e);
}

/* 2^z.  NaN -> NaN; z >= 128 -> +inf; z < -126 -> 0 (results below FLT_MIN flush to zero);
 * else n = rint(z) (ties to even), f = z-n in [-1/2,1/2], Q(f) degree 6, exponent add. */
float ref_exp2f(float z)
{
    static const float q[7] = {
        1.0f, 0x1.62e43p-1f, 0x1.ebfbep-3f, 0x1.c6aec2p-5f,
        0x1.3b2a72p-7f, 0x1.5f4e2ep-10f, 0x1.43e9d6p-13f };
    if (z != z) return z;
    if (z >= 128.0f) return INFINITY;
    if (z < -126.0f) return 0.0f;
    float n = rintf(z);
    float f = z - n;
    float p = q[6];
    p = fmaf(p, f, q[5]);
    p = fmaf(p, f, q[4]);
    p = fmaf(p, f, q[3]);
    p = fmaf(p, f, q[2]);
    p = fmaf(p, f, q[1]);
    p = fmaf(p, f, q[0]);
    return u2f(f2u(p) + ((uint32_t)(int32_t)n << 23));
}

/* pow(x,y) = exp2(y*log2(x)); x<0 -> NaN, 0 <= x < FLT_MIN -> 0 (y>0).  REF_POW_LIBM swaps in
 * a double-precision pow rounded once, used only to bound the pinned pair's deviation. */
float ref_powf(float x, float y, int pow_mode)
{
    if (x != x || y != y) return NAN;
    if (x < 0.0f) return NAN;
    if (x < REF_FLT_MIN) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : INFINITY);
    if (pow_mode == REF_POW_LIBM) return (float)pow((double)x, (double)y);
    return ref_exp2f(y * ref_log2f(x));
}

/* state/edit.rs:81-95 */
void ref_default_params(ref_edit_params *p)
{
    memset(p, 0, sizeof *p);
    p->whites = 1.0f;
}

/* gpu/pipeline.rs:125-133 -- truncating f32 arithmetic exactly as written there. */
void ref_derived_dims(uint32_t w, uint32_t h, uint32_t *pw, uint32_t *ph, uint32_t *hw, uint32_t *hh)
{
    float aspect = (float)w / (float)h;
    uint32_t preview_w = w < 1280u ? w : 1280u;
    *pw = preview_w;
    *ph = (uint32_t)((float)preview_w / aspect);
    *hw = 128u;
    *hh = (uint32_t)((float)128u / aspect);
}

/* shaders.rs:161-169 get_neighbor: clamp coords, load, /4096. */
static inline float tap(const uint16_t *cfa, int32_t w, int32_t h, int32_t x, int32_t y, uint32_t bl)
{
    if (x < 0) x = 0;
    if (x > w - 1) x = w - 1;
    if (y < 0) y = 0;
    if (y > h - 1) y = h - 1;
    uint32_t raw = cfa[(size_t)y * (size_t)w + (size_t)x];
    raw = raw > bl ? raw - bl : 0u;   /* integer black level (extension; bl = 0 is the reference) */
    return (float)raw * (1.0f / 4096.0f);
}

/* shaders.rs:104-158 debayer: nearest-neighbour selection, row parity on (py+1). */
static void debayer(const uint16_t *cfa, int32_t w, int32_t h, int32_t px, int32_t py, uint32_t bl, float rgb[3])
{
    float n = tap(cfa, w, h, px, py, bl);
    int even_row = ((py + 1) % 2) == 0;
    int even_col = (px % 2) == 0;
    float r, g, b;
    if (even_row) {
        if (even_col) { g = n; b = tap(cfa, w, h, px + 1, py, bl); r = tap(cfa, w, h, px, py + 1, bl); }
        else          { b = n; g = tap(cfa, w, h, px - 1, py, bl); r = tap(cfa, w, h, px - 1, py + 1, bl); }
    } else {
        if (even_col) { r = n; g = tap(cfa, w, h, px + 1, py, bl); b = tap(cfa, w, h, px, py - 1, bl); }
        else          { g = n; r = tap(cfa, w, h, px - 1, py, bl); b = tap(cfa, w, h, px, py - 1, bl); }
    }
    rgb[0] = r; rgb[1] = g; rgb[2] = b;
}

static inline float dot709(float r, float g, float b)
{
    return ((r * 0.2126f) + (g * 0.7152f)) + (b * 0.0722f);
}

/* shaders.rs:192-266 colour stack on one debayered pixel. */
static void colour_stack(const ref_uniforms *u, int pow_mode, float c[3])
{
    const ref_edit_params *p = &u->p;
    float r = c[0], g = c[1], b = c[2];
    /* 2. white balance (:195) */
    r = r * u->wb[0]; g = g * u->wb[1]; b = b * u->wb[2];
    /* 2.5 temperature / tint (:200-205) */
    r = r * (1.0f + p->temperature * 0.3f);
    b = b * (1.0f - p->temperature * 0.3f);
    g = g * (1.0f + p->tint * 0.3f);
    /* 3. matrix (:209-214): mat3x3(row0,row1,row2) takes COLUMNS => out = M^T c */
    const float *m = u->cm;
    float x = ((m[0] * r) + (m[3] * g)) + (m[6] * b);
    float y = ((m[1] * r) + (m[4] * g)) + (m[7] * b);
    float z = ((m[2] * r) + (m[5] * g)) + (m[8] * b);
    r = x; g = y; b = z;
    /* 4. exposure (:217-218) */
    float em = ref_powf(2.0f, p->exposure, pow_mode);
    r = r * em; g = g * em; b = b * em;
    /* 5. highlights / shadows (:222-230), same L for both */
    float L = dot709(r, g, b);
    float hl = 1.0f + (L * p->highlights);
    r = r * hl; g = g * hl; b = b * hl;
    float sh = 1.0f + ((1.0f - L) * p->shadows);
    r = r * sh; g = g * sh; b = b * sh;
    /* 6. contrast (:233-234) */
    float cf = 1.0f + (p->contrast / 100.0f);
    r = (r - 0.5f) * cf + 0.5f; g = (g - 0.5f) * cf + 0.5f; b = (b - 0.5f) * cf + 0.5f;
    /* 7. levels (:239) */
    float den = (p->whites - p->blacks) + 0.0001f;
    r = (r - p->blacks) / den; g = (g - p->blacks) / den; b = (b - p->blacks) / den;
    /* 8. saturation (:243-247): mix(x,y,a) = x*(1-a) + y*a */
    float Y = dot709(r, g, b);
    float s = 1.0f + (p->saturation / 100.0f);
    float ys = Y * (1.0f - s);
    r = ys + r * s; g = ys + g * s; b = ys + b * s;
    /* 9. vibrance (:251-257) */
    float sat = fmaxf(r, fmaxf(g, b)) - fminf(r, fminf(g, b));
    float va = p->vibrance * (1.0f - sat);
    float Y2 = dot709(r, g, b);
    float a2 = 1.0f + va;
    float yv = Y2 * (1.0f - a2);
    r = yv + r * a2; g = yv + g * a2; b = yv + b * a2;
    /* 10. gamma (:261) */
    const float inv_gamma = (float)(1.0 / 2.2);
    r = ref_powf(r, inv_gamma, pow_mode);
    g = ref_powf(g, inv_gamma, pow_mode);
    b = ref_powf(b, inv_gamma, pow_mode);
    /* 11. clamp (:264), NaN -> 0 (maxNum semantics) */
    c[0] = fminf(fmaxf(r, 0.0f), 1.0f);
    c[1] = fminf(fmaxf(g, 0.0f), 1.0f);
    c[2] = fminf(fmaxf(b, 0.0f), 1.0f);
}

/* shaders.rs:192-266 with every mul+add of the expression tree contracted to one fma and the levels
 * division done as a multiplication by the correctly rounded reciprocal (REF_MATH_CONTRACTED). */
static inline float dot709_fma(float r, float g, float b)
{
    return fmaf(b, 0.0722f, fmaf(g, 0.7152f, r * 0.2126f));
}

static void colour_stack_contracted(const ref_uniforms *u, int pow_mode, float c[3])
{
    const ref_edit_params *p = &u->p;
    float r = c[0], g = c[1], b = c[2];
    r = r * u->wb[0]; g = g * u->wb[1]; b = b * u->wb[2];                          /* :195 */
    r = r * fmaf(p->temperature, 0.3f, 1.0f);                                      /* :200 */
    b = b * fmaf(-p->temperature, 0.3f, 1.0f);                                     /* :201  1 - t*0.3 */
    g = g * fmaf(p->tint, 0.3f, 1.0f);                                             /* :205 */
    const float *m = u->cm;                                                        /* :209-214 (columns) */
    float x = fmaf(m[6], b, fmaf(m[3], g, m[0] * r));
    float y = fmaf(m[7], b, fmaf(m[4], g, m[1] * r));
    float z = fmaf(m[8], b, fmaf(m[5], g, m[2] * r));
    float em = ref_powf(2.0f, p->exposure, pow_mode);                              /* :217-218 */
    r = x * em; g = y * em; b = z * em;
    float L = dot709_fma(r, g, b);                                                 /* :222 */
    float hl = fmaf(L, p->highlights, 1.0f);                                       /* :226 */
    r = r * hl; g = g * hl; b = b * hl;
    float sh = fmaf(1.0f - L, p->shadows, 1.0f);                                   /* :230 */
    r = r * sh; g = g * sh; b = b * sh;
    float cf = 1.0f + (p->contrast / 100.0f);                                      /* :233 (uniform) */
    r = fmaf(r - 0.5f, cf, 0.5f); g = fmaf(g - 0.5f, cf, 0.5f); b = fmaf(b - 0.5f, cf, 0.5f);
    float den = (p->whites - p->blacks) + 0.0001f;                                 /* :239 (uniform) */
    float rden = 1.0f / den;
    r = (r - p->blacks) * rden; g = (g - p->blacks) * rden; b = (b - p->blacks) * rden;
    float Y = dot709_fma(r, g, b);                                                 /* :243-247 */
    float s = 1.0f + (p->saturation / 100.0f);
    float ys = Y * (1.0f - s);
    r = fmaf(r, s, ys); g = fmaf(g, s, ys); b = fmaf(b, s, ys);
    float sat = fmaxf(r, fmaxf(g, b)) - fminf(r, fminf(g, b));                     /* :251 */
    float a2 = fmaf(p->vibrance, 1.0f - sat, 1.0f);                                /* :254, :257  1 + vibrance*(1-sat) */
    float Y2 = dot709_fma(r, g, b);                                                /* :256 */
    float yv = Y2 * (1.0f - a2);
    r = fmaf(r, a2, yv); g = fmaf(g, a2, yv); b = fmaf(b, a2, yv);
    const float inv_gamma = (float)(1.0 / 2.2);                                    /* :261 */
    r = ref_powf(r, inv_gamma, pow_mode);
    g = ref_powf(g, inv_gamma, pow_mode);
    b = ref_powf(b, inv_gamma, pow_mode);
    c[0] = fminf(fmaxf(r, 0.0f), 1.0f);                                            /* :264 */
    c[1] = fminf(fmaxf(g, 0.0f), 1.0f);
    c[2] = fminf(fmaxf(b, 0.0f), 1.0f);
}

/* shaders.rs:23-60 (vs_main, evaluated at the pixel centre) + :174-187 (bounds, pixel_coords). */
void ref_pixel(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
               uint32_t tw, uint32_t th, uint32_t i, uint32_t j, int pow_mode, float rgba[4])
{
    float sx = ((float)i + 0.5f) / (float)tw;
    float sy = ((float)j + 0.5f) / (float)th;
    float tx = ((sx - 0.5f) / u->zoom - u->pan_x) + 0.5f;
    float ty = ((sy - 0.5f) / u->zoom - u->pan_y) + 0.5f;
    rgba[3] = 1.0f;
    if (tx < 0.0f || tx > 1.0f || ty < 0.0f || ty > 1.0f) {
        rgba[0] = rgba[1] = rgba[2] = 0.0f;  /* :174-178 -- as written there: a NaN coordinate (zoom = 0) passes all four tests */
        return;
    }
    /* :184-187 i32(f32): truncation; of a NaN it is implementation-defined -- 0 here, what GPUs' conversions return */
    int32_t px = tx != tx ? 0 : (int32_t)(tx * (float)w);
    int32_t py = ty != ty ? 0 : (int32_t)(ty * (float)h);
    /* tx == 1.0 exactly gives px == w, one past the texture, and the shader carries that coordinate on (:184-192): the Bayer
     * parity is taken on it as it is; only the loads see a border -- get_neighbor clamps (:164-167) and the centre load of
     * :106 is out of bounds, which this repository lowers as a clamp too (tap()).  Pinned by the evaluated shader text:
     * tests/golden/wgsl_golden.npz case tex_coord_one. */
    float c[3];
    debayer(cfa, (int32_t)w, (int32_t)h, px, py, u->black_level, c);
    if (u->math_mode == REF_MATH_CONTRACTED) colour_stack_contracted(u, pow_mode, c);
    else colour_stack(u, pow_mode, c);
    rgba[0] = c[0]; rgba[1] = c[1]; rgba[2] = c[2];
}

void ref_render_f32_rows(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
                         uint32_t tw, uint32_t th, uint32_t row0, uint32_t row1, int pow_mode,
                         float *out)
{
    for (uint32_t j = row0; j < row1; ++j)
        for (uint32_t i = 0; i < tw; ++i)
            ref_pixel(cfa, w, h, u, tw, th, i, j, pow_mode, out + ((size_t)j * tw + i) * 4);
}

/* Rows [row0,row1) of the target only; out_band holds (row1-row0)*tw*4 floats. */
void ref_render_f32_band(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
                         uint32_t tw, uint32_t th, uint32_t row0, uint32_t row1, int pow_mode,
                         float *out_band)
{
    for (uint32_t j = row0; j < row1; ++j)
        for (uint32_t i = 0; i < tw; ++i)
            ref_pixel(cfa, w, h, u, tw, th, i, j, pow_mode, out_band + ((size_t)(j - row0) * tw + i) * 4);
}

void ref_render_f32(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
                    uint32_t tw, uint32_t th, int pow_mode, float *out)
{
    ref_render_f32_rows(cfa, w, h, u, tw, th, 0, th, pow_mode, out);
}

typedef struct {
    const uint16_t *cfa; uint32_t w, h; const ref_uniforms *u; uint32_t tw, th, row0, row1;
    int pow_mode; float *out;
} band_job;

static void *band_main(void *arg)
{
    band_job *b = (band_job *)arg;
    ref_render_f32_rows(b->cfa, b->w, b->h, b->u, b->tw, b->th, b->row0, b->row1, b->pow_mode, b->out);
    return NULL;
}

void ref_render_f32_mt(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u,
                       uint32_t tw, uint32_t th, int pow_mode, float *out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > th) nthreads = (int)th;
    if (nthreads <= 1) { ref_render_f32(cfa, w, h, u, tw, th, pow_mode, out); return; }
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    band_job *jobs = (band_job *)malloc(sizeof(band_job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; ++t) {
        uint32_t r0 = (uint32_t)(((uint64_t)th * (uint64_t)t) / (uint64_t)nthreads);
        uint32_t r1 = (uint32_t)(((uint64_t)th * (uint64_t)(t + 1)) / (uint64_t)nthreads);
        jobs[t] = (band_job){ cfa, w, h, u, tw, th, r0, r1, pow_mode, out };
        pthread_create(&tid[t], NULL, band_main, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) pthread_join(tid[t], NULL);
    free(jobs);
    free(tid);
}

/* Timing harness for bench.py's cpu_baseline leg: `nthreads` persistent threads, each owning one row band of the frame
 * and a band-sized output buffer that it allocates and first-touches itself (so the pages sit on the thread's own NUMA
 * node), all rendering the same frame again and again between barriers until `budget_s` seconds have passed.
 * *frames_done / *seconds give whole-frame throughput.  The pixels are the same ref_pixel() arithmetic as everywhere
 * else; they are not returned (parity is the tests' business, this only measures). */
typedef struct {
    const uint16_t *cfa; uint32_t w, h; const ref_uniforms *u; uint32_t row0, row1; int pow_mode;
    pthread_barrier_t *bar; volatile int *stop; double sink;
} bench_job;

static void *bench_main(void *arg)
{
    bench_job *b = (bench_job *)arg;
    const size_t n = (size_t)(b->row1 - b->row0) * b->w * 4;
    float *band = (float *)malloc(n * sizeof(float));
    if (band) memset(band, 0, n * sizeof(float));            /* first touch by the owning thread */
    for (;;) {
        pthread_barrier_wait(b->bar);                        /* frame start (thread 0 decides about stopping in between) */
        if (*b->stop) break;
        if (band) {
            ref_render_f32_band(b->cfa, b->w, b->h, b->u, b->w, b->h, b->row0, b->row1, b->pow_mode, band);
            b->sink += band[n - 4];
        }
        pthread_barrier_wait(b->bar);                        /* frame end */
    }
    free(band);
    return NULL;
}

#include <time.h>
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void ref_bench_mt(const uint16_t *cfa, uint32_t w, uint32_t h, const ref_uniforms *u, int nthreads, double budget_s,
                  int max_frames, int *frames_done, double *seconds)
{
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > h) nthreads = (int)h;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    bench_job *jobs = (bench_job *)malloc(sizeof(bench_job) * (size_t)nthreads);
    pthread_barrier_t bar;
    volatile int stop = 0;
    pthread_barrier_init(&bar, NULL, (unsigned)nthreads + 1u);       /* workers + this thread */
    for (int t = 0; t < nthreads; ++t) {
        uint32_t r0 = (uint32_t)(((uint64_t)h * (uint64_t)t) / (uint64_t)nthreads);
        uint32_t r1 = (uint32_t)(((uint64_t)h * (uint64_t)(t + 1)) / (uint64_t)nthreads);
        jobs[t] = (bench_job){ cfa, w, h, u, r0, r1, REF_POW_PINNED, &bar, &stop, 0.0 };
        pthread_create(&tid[t], NULL, bench_main, &jobs[t]);
    }
    /* one untimed frame (page faults, clocks), then the timed ones */
    pthread_barrier_wait(&bar); pthread_barrier_wait(&bar);
    int frames = 0;
    const double t0 = now_s();
    double el = 0.0;
    do {
        pthread_barrier_wait(&bar);
        pthread_barrier_wait(&bar);
        ++frames;
        el = now_s() - t0;
    } while (el < budget_s && frames < max_frames);
    stop = 1;
    pthread_barrier_wait(&bar);
    for (int t = 0; t < nthreads; ++t) pthread_join(tid[t], NULL);
    pthread_barrier_destroy(&bar);
    free(jobs);
    free(tid);
    if (frames_done) *frames_done = frames;
    if (seconds) *seconds = el;
}

/* pipeline.rs:322 Rgba8Unorm store: round-to-nearest of x*255, pinned as trunc(x*255 + 0.5). */
void ref_pack_u8(const float *rgba, size_t nfloats, uint8_t *out)
{
    for (size_t k = 0; k < nfloats; ++k) out[k] = (uint8_t)(rgba[k] * 255.0f + 0.5f);
}

/* IEEE binary16, round-to-nearest-even; inputs are in [0,1] but the conversion is general. */
static uint16_t f32_to_f16(float f)
{
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t a = x & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((a > 0x7f800000u) ? 0x0200u : 0u));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);      /* rounds to >= 65520 -> inf */
    if (a < 0x33000001u) return (uint16_t)sign;                   /* < 2^-25 (or == with tie to even 0) */
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t man = (a & 0x007fffffu) | 0x00800000u;
    uint32_t shift, half;
    if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); half = 0; }
    else { shift = 13; half = (uint32_t)(e + 15) << 10; }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q += 1;
    if (e < -14) return (uint16_t)(sign | q);                     /* subnormal (carry into normal is fine) */
    return (uint16_t)(sign | (half + q - 0x400u));                /* q carries the implicit bit */
}

void ref_pack_f16(const float *rgba, size_t nfloats, uint16_t *out)
{
    for (size_t k = 0; k < nfloats; ++k) out[k] = f32_to_f16(rgba[k]);
}

/* pipeline.rs:720-736: channel-major [R[256], G[256], B[256]], alpha ignored. */
void ref_histogram(const uint8_t *rgba, size_t npx, uint32_t hist[768])
{
    memset(hist, 0, 768 * sizeof(uint32_t));
    for (size_t k = 0; k < npx; ++k) {
        hist[0 * 256 + rgba[4 * k + 0]] += 1;
        hist[1 * 256 + rgba[4 * k + 1]] += 1;
        hist[2 * 256 + rgba[4 * k + 2]] += 1;
    }
}
