/* tools/q8_fma_check.c -- the Rgba8Unorm pack is pinned as trunc(RN(x * 255) + 0.5) (two roundings: the oracle's
 * (uint8_t)(x * 255.0f + 0.5f) without contraction).  The kernels compute it as trunc(fma(x, 255, 0.5)) (one rounding, one
 * instruction less per code); this checks that the two agree for EVERY float in [0, 1] -- the only values that reach the
 * pack (the gamma step clamps).  gcc -O2 -mfma -ffp-contract=off tools/q8_fma_check.c -lm -o /tmp/q8_fma_check */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(void){
    unsigned long bad=0; uint32_t first=0;
    for(uint32_t b=0;b<=0x3f800000u;++b){ float x; memcpy(&x,&b,4);
        volatile float m = x*255.0f; float y1 = m+0.5f; float y2 = fmaf(x,255.0f,0.5f);
        if((uint32_t)y1!=(uint32_t)y2){ if(!bad) first=b; ++bad; } }
    printf("mismatches %lu first 0x%08x\n",bad,first); return 0; }
