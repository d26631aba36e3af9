/* tools/q8_magic_check.c -- the f32 surface's histogram codes (round 5).  The pin is q = trunc(RN(x * 255) + 0.5) (the oracle's
 * (uint8_t)(x * 255.0f + 0.5f)); the f32 kernel took it as trunc(fma(x, 255, 0.5)) + v_cvt_u32_f32 (tools/q8_fma_check.c).
 * The add-magic form t = fma(x, 255, 2^23) needs no conversion: 2^23's ulp is 1, so RN(255 x + 2^23) carries
 * round-to-nearest-EVEN(255 x) in the low byte of its encoding 0x4b0000qq.  Nearest-even and trunc(. + 0.5) differ only where
 * 255 x is exactly k + 0.5, and the only float in [0, 1] with that property is 0.5 (255 x = 127.5 needs 255 | 2k + 1):
 * there both give 128.  Arguments aside, this checks EVERY float in [0, 1] -- the only values that reach the pack (the
 * gamma step clamps).  gcc -O2 -mfma -ffp-contract=off tools/q8_magic_check.c -lm -o /tmp/q8_magic_check */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
int main(void){
    unsigned long bad=0; uint32_t first=0;
    for(uint32_t b=0;b<=0x3f800000u;++b){ float x; memcpy(&x,&b,4);
        volatile float m = x*255.0f; const float y1 = m+0.5f;                 /* the pin: two roundings */
        const float t = fmaf(x,255.0f,8388608.0f); uint32_t tb; memcpy(&tb,&t,4);
        if((uint32_t)y1 != (tb & 0xffu) + ((tb >> 8 & 1u) << 8) || (tb & 0xfffffe00u) != 0x4b000000u){ if(!bad) first=b; ++bad; } }
    printf("mismatches %lu first 0x%08x\n",bad,first); return bad != 0; }
