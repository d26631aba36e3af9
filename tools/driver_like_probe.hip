// tools/driver_like_probe.hip -- how far is a DRIVER-LIKE lowering of the reference's shader from this repository's pinned
// evaluation, on the RGBA8 surface the reference renders to (pipeline.rs:322)?  Measured on the GPU, not assumed.
//
// The reference runs `fs_main` (src/gpu/shaders.rs:171-267) through wgpu/naga and a Vulkan driver's shader compiler.  WGSL
// leaves that compiler free to (a) contract a*b+c into one fma, (b) lower pow(x, y) to exp2(y * log2(x)) on the hardware's
// v_log_f32 / v_exp_f32, (c) divide by reciprocal-and-multiply (2.5 ULP), and leaves the UNORM tie rule to the ROP.  This
// probe evaluates, for every pixel of random frames with random slider stacks, BOTH
//   pinned : rd_colour_m<RD_MATH_STRICT> + rd_q8                      (the product's strict path == oracle/develop_ref.c)
//   driver : the same shader text with (a) + (b) + (c) and the UNORM pack rounded to nearest-even -- what an AMD shader
//            compiler plausibly emits for the reference's WGSL (this GPU's own v_log_f32 / v_exp_f32 / v_rcp_f32)
// and counts the RGBA8 bytes that differ and by how much.  It bounds what tests/golden/reference_kit (INTEGRATION.md
// section 6) should show on real AMD hardware; it is NOT a parity proof (nobody has run the reference here).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -o tools/driver_like_probe tools/driver_like_probe.hip
//   tools/driver_like_probe [frames=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../raweditor_amd/csrc/rd_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct probe_stats { unsigned long long bytes, diff1, diff2plus, px_any; uint32_t worst; float in[3], lin_pinned[3], lin_driver[3]; rd_edit_params p; };

__device__ __forceinline__ float hw_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

// fs_main with every a*b+c contracted, pow on the hardware pair, division by v_rcp_f32 (shaders.rs:192-266; uniforms as the
// shader computes them per fragment, from the raw sliders)
__device__ __forceinline__ rd_rgb driver_like(const rd_edit_params &p, const float wb[4], const float m[9], float r, float g, float b,
                                              rd_rgb *linear)
{
    r *= wb[0]; g *= wb[1]; b *= wb[2];                                                     // :195
    r *= __builtin_fmaf(p.temperature, 0.3f, 1.0f); b *= __builtin_fmaf(-p.temperature, 0.3f, 1.0f);      // :200-201
    g *= __builtin_fmaf(p.tint, 0.3f, 1.0f);                                                // :205
    const float x = __builtin_fmaf(m[6], b, __builtin_fmaf(m[3], g, m[0] * r));            // :209-214, rows consumed as columns
    const float y = __builtin_fmaf(m[7], b, __builtin_fmaf(m[4], g, m[1] * r));
    const float z = __builtin_fmaf(m[8], b, __builtin_fmaf(m[5], g, m[2] * r));
    const float em = hw_pow(2.0f, p.exposure);                                              // :217
    r = x * em; g = y * em; b = z * em;
    const float L = __builtin_fmaf(b, 0.0722f, __builtin_fmaf(g, 0.7152f, r * 0.2126f));   // :222
    const float hl = __builtin_fmaf(L, p.highlights, 1.0f), sh = __builtin_fmaf(1.0f - L, p.shadows, 1.0f);
    r = r * hl * sh; g = g * hl * sh; b = b * hl * sh;                                      // :226, :230
    const float cf = 1.0f + p.contrast * __builtin_amdgcn_rcpf(100.0f);                    // :233
    r = __builtin_fmaf(r - 0.5f, cf, 0.5f); g = __builtin_fmaf(g - 0.5f, cf, 0.5f); b = __builtin_fmaf(b - 0.5f, cf, 0.5f);
    const float rden = __builtin_amdgcn_rcpf((p.whites - p.blacks) + 0.0001f);             // :239
    r = (r - p.blacks) * rden; g = (g - p.blacks) * rden; b = (b - p.blacks) * rden;
    const float s = 1.0f + p.saturation * __builtin_amdgcn_rcpf(100.0f);                   // :245
    float Y = __builtin_fmaf(b, 0.0722f, __builtin_fmaf(g, 0.7152f, r * 0.2126f));
    float ys = Y * (1.0f - s);
    r = __builtin_fmaf(r, s, ys); g = __builtin_fmaf(g, s, ys); b = __builtin_fmaf(b, s, ys);                // mix(Y, c, s)
    const float sat = __builtin_fmaxf(r, __builtin_fmaxf(g, b)) - __builtin_fminf(r, __builtin_fminf(g, b));  // :251-257
    const float a2 = __builtin_fmaf(p.vibrance, 1.0f - sat, 1.0f);
    Y = __builtin_fmaf(b, 0.0722f, __builtin_fmaf(g, 0.7152f, r * 0.2126f));
    ys = Y * (1.0f - a2);
    r = __builtin_fmaf(r, a2, ys); g = __builtin_fmaf(g, a2, ys); b = __builtin_fmaf(b, a2, ys);
    *linear = rd_rgb{ r, g, b };
    rd_rgb o = { hw_pow(r, RD_INV_GAMMA), hw_pow(g, RD_INV_GAMMA), hw_pow(b, RD_INV_GAMMA) };                 // :261
    o.r = __builtin_fminf(__builtin_fmaxf(o.r, 0.0f), 1.0f);                                // :264 (maxNum: NaN -> 0)
    o.g = __builtin_fminf(__builtin_fmaxf(o.g, 0.0f), 1.0f);
    o.b = __builtin_fminf(__builtin_fmaxf(o.b, 0.0f), 1.0f);
    return o;
}

__global__ void __launch_bounds__(256) probe(const uint16_t *cfa, uint32_t W, uint32_t H, rd_ku u, rd_edit_params p, probe_stats *st)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W * H) return;
    const int32_t py = idx / W, px = idx - py * W;
    const float n = rd_tap(cfa, W, H, px, py, 0);
    const bool even_row = ((py + 1) & 1) == 0, even_col = (px & 1) == 0;                    // shaders.rs:115-116
    float r, g, b;
    if (even_row) {
        if (even_col) { g = n; b = rd_tap(cfa, W, H, px + 1, py, 0); r = rd_tap(cfa, W, H, px, py + 1, 0); }
        else          { b = n; g = rd_tap(cfa, W, H, px - 1, py, 0); r = rd_tap(cfa, W, H, px - 1, py + 1, 0); }
    } else {
        if (even_col) { r = n; g = rd_tap(cfa, W, H, px + 1, py, 0); b = rd_tap(cfa, W, H, px, py - 1, 0); }
        else          { g = n; r = rd_tap(cfa, W, H, px - 1, py, 0); b = rd_tap(cfa, W, H, px, py - 1, 0); }
    }
    const rd_rgb a = rd_colour_m<RD_MATH_STRICT>(u, r, g, b);
    float lr[1] = { r }, lg[1] = { g }, lb[1] = { b };
    rd_colour_n<1, RD_MATH_STRICT, false>(u, lr, lg, lb);                                   // the pinned LINEAR values (before gamma)
    const float wb[4] = { u.wb_r, u.wb_g, u.wb_b, 1.0f };
    rd_rgb dl;
    const rd_rgb d = driver_like(p, wb, u.m, r, g, b, &dl);
    const uint32_t qa[3] = { rd_q8(a.r), rd_q8(a.g), rd_q8(a.b) };
    const uint32_t qd[3] = { (uint32_t)__builtin_rintf(d.r * 255.0f), (uint32_t)__builtin_rintf(d.g * 255.0f), (uint32_t)__builtin_rintf(d.b * 255.0f) };
    uint32_t worst = 0, n1 = 0, n2 = 0;
    for (int c = 0; c < 3; ++c) {
        const uint32_t df = qa[c] > qd[c] ? qa[c] - qd[c] : qd[c] - qa[c];
        worst = df > worst ? df : worst;
        n1 += df == 1u; n2 += df > 1u;
    }
    if (n1) atomicAdd(&st->diff1, (unsigned long long)n1);
    if (n2) atomicAdd(&st->diff2plus, (unsigned long long)n2);
    if (worst) {
        atomicAdd(&st->px_any, 1ull);
        if (atomicMax(&st->worst, worst) < worst) {                                          // (racy on purpose: any one worst pixel will do)
            st->in[0] = r; st->in[1] = g; st->in[2] = b; st->p = p;
            st->lin_pinned[0] = lr[0]; st->lin_pinned[1] = lg[0]; st->lin_pinned[2] = lb[0];
            st->lin_driver[0] = dl.r; st->lin_driver[1] = dl.g; st->lin_driver[2] = dl.b;
        }
    }
}

static uint64_t g_seed = 0x52415745ull;
static double urand() { g_seed = g_seed * 6364136223846793005ull + 1442695040888963407ull; return (double)(g_seed >> 11) / 9007199254740992.0; }
static float uni(double lo, double hi) { return (float)(lo + (hi - lo) * urand()); }

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 64;
    const uint32_t W = 1536, H = 1024;
    std::vector<uint16_t> cfa((size_t)W * H);
    uint16_t *d_cfa; probe_stats *d_st;
    CK(hipMalloc((void **)&d_cfa, cfa.size() * 2)); CK(hipMalloc((void **)&d_st, sizeof(probe_stats)));
    const float wb[4] = { 2.0f, 1.0f, 1.5f, 1.0f };
    const float cam[9] = { 1.6f, -0.4f, -0.2f, -0.3f, 1.5f, -0.2f, 0.0f, -0.5f, 1.5f }, ident[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    struct { const char *name; int kind; } sets[] = { { "all sliders default, identity matrix (what the app shows on load)", 0 },
                                                      { "all ten sliders uniform in the UI ranges, identity matrix", 1 },
                                                      { "all ten sliders uniform in the UI ranges, camera matrix (bench.py's stacks)", 2 },
                                                      { "gentle edits (a third of each UI range), identity matrix", 3 } };
    printf("driver-like lowering (fma contraction + v_log_f32 / v_exp_f32 pow + v_rcp_f32 divides + nearest-even UNORM) vs the pinned\n"
           "evaluation, RGBA8 codes of r, g, b, %d random %ux%u frames (uniform 12-bit CFA) per slider set:\n", frames, W, H);
    for (auto &set : sets) {
        probe_stats st;
        memset(&st, 0, sizeof st);
        CK(hipMemcpy(d_st, &st, sizeof st, hipMemcpyHostToDevice));
        for (int f = 0; f < frames; ++f) {
            for (auto &v : cfa) v = (uint16_t)(urand() * 4096.0);
            CK(hipMemcpy(d_cfa, cfa.data(), cfa.size() * 2, hipMemcpyHostToDevice));
            rd_edit_params p;
            memset(&p, 0, sizeof p);
            p.whites = 1.0f;
            if (set.kind) {
                const double k = set.kind == 3 ? 1.0 / 3.0 : 1.0;
                p.exposure = uni(-5 * k, 5 * k); p.contrast = uni(-10 * k, 10 * k); p.highlights = uni(-k, k); p.shadows = uni(-k, k);
                p.whites = uni(1.0 - 0.2 * k, 1.0 + 0.2 * k); p.blacks = uni(0.0, 0.2 * k); p.vibrance = uni(-k, k);
                p.saturation = uni(-100 * k, 100 * k); p.temperature = uni(-k, k); p.tint = uni(-k, k);
            }
            const rd_ku u = rd_make_ku(p, wb, set.kind == 2 ? cam : ident, 1.0f, 0.0f, 0.0f, 0, RD_MATH_STRICT);
            hipLaunchKernelGGL(probe, dim3((W * H + 255) / 256), dim3(256), 0, 0, d_cfa, W, H, u, p, d_st);
            CK(hipGetLastError());
        }
        CK(hipMemcpy(&st, d_st, sizeof st, hipMemcpyDeviceToHost));
        const double nb = 3.0 * W * H * frames;
        printf("  %-78s: %.4f %% of the bytes differ by 1 code, %.6f %% by more (largest difference %u); %.4f %% of the pixels touched\n",
               set.name, 100.0 * st.diff1 / nb, 100.0 * st.diff2plus / nb, st.worst, 100.0 * st.px_any / ((double)W * H * frames));
        if (st.worst > 1)
            printf("      the worst pixel: demosaiced (%.6g, %.6g, %.6g); linear values before the gamma, pinned (%.6g, %.6g, %.6g) vs driver-like (%.6g, %.6g, %.6g);\n"
                   "      sliders exposure %.3f contrast %.3f highlights %.3f shadows %.3f whites %.3f blacks %.3f vibrance %.3f saturation %.2f temperature %.3f tint %.3f\n",
                   st.in[0], st.in[1], st.in[2], st.lin_pinned[0], st.lin_pinned[1], st.lin_pinned[2], st.lin_driver[0], st.lin_driver[1], st.lin_driver[2],
                   st.p.exposure, st.p.contrast, st.p.highlights, st.p.shadows, st.p.whites, st.p.blacks, st.p.vibrance, st.p.saturation, st.p.temperature, st.p.tint);
    }
    return 0;
}
