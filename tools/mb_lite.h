// tools/mb_lite.h -- microbench only: reduced-VALU stand-ins for rd_colour (WRONG results on purpose),
// used to find the memory-side floor of the export kernel.  Included through RD_COLOUR_HOOK_HEADER.
#pragma once
// reduced-VALU stand-ins for rd_colour (wrong results; only to see how the kernel scales with VALU load)
__device__ __forceinline__ rd_rgb mb_colour_lite(const rd_ku &u, float r, float g, float b, int level)
{
    r = r * u.wb_r; g = g * u.wb_g; b = b * u.wb_b;
    r = r * u.kr; b = b * u.kb; g = g * u.kg;
    float x = ((u.m[0] * r) + (u.m[3] * g)) + (u.m[6] * b);
    float y = ((u.m[1] * r) + (u.m[4] * g)) + (u.m[7] * b);
    float z = ((u.m[2] * r) + (u.m[5] * g)) + (u.m[8] * b);
    r = x * u.em; g = y * u.em; b = z * u.em;
    float L = rd_dot709(r, g, b);
    float hl = 1.0f + (L * u.highlights);
    r = r * hl; g = g * hl; b = b * hl;
    r = (r - 0.5f) * u.cf + 0.5f; g = (g - 0.5f) * u.cf + 0.5f; b = (b - 0.5f) * u.cf + 0.5f;
    rd_rgb o;
    if (level == 1) { o.r = rd_gamma_clamp(r); o.g = __builtin_fminf(__builtin_fmaxf(g, 0.f), 1.f); o.b = __builtin_fminf(__builtin_fmaxf(b, 0.f), 1.f); }
    else { o.r = __builtin_fminf(__builtin_fmaxf(r, 0.f), 1.f); o.g = __builtin_fminf(__builtin_fmaxf(g, 0.f), 1.f); o.b = __builtin_fminf(__builtin_fmaxf(b, 0.f), 1.f); }
    return o;
}


#ifdef MB_LITE
#define RD_COLOUR(u, r, g, b) mb_colour_lite(u, r, g, b, MB_LITE - 1)
#endif
