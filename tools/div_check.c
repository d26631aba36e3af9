/* Check that the reciprocal-correction sequences equal IEEE division a/d bit for bit.
 *   y = RN(1/d);  q0 = a*y;  q1 = fma(fma(-d,q0,a), y, q0);  q2 = fma(fma(-d,q1,a), y, q1)
 * gcc -O2 -mfma -ffp-contract=off tools/div_check.c -lm -o /tmp/div_check && /tmp/div_check */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static inline uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static uint64_t s=88172645463325252ull;
static inline uint64_t rnd(void){s^=s<<13;s^=s>>7;s^=s<<17;return s;}
int main(void){
    unsigned long bad1=0,bad2=0,n=0;
    /* d: denominators the levels step can produce ((whites-blacks)+1e-4 over a wide range) plus random normals */
    for(int di=0;di<40000;++di){
        float d;
        if(di<20000){ float w=0.0f+ (float)(rnd()%2000001)/1000000.0f; float b=(float)(rnd()%1000001)/1000000.0f-0.3f; d=(w-b)+0.0001f; }
        else { uint32_t e=(uint32_t)(rnd()%60)+97; d=u2f((e<<23)|(uint32_t)(rnd()&0x7fffff)| ((rnd()&1)?0x80000000u:0)); }
        if(d==0.0f||!isfinite(d)) continue;
        float y=1.0f/d;
        for(int k=0;k<25000;++k){
            uint32_t e=(uint32_t)(rnd()%80)+87;           /* |a| in [2^-40, 2^40) */
            float a=u2f((e<<23)|(uint32_t)(rnd()&0x7fffff)|((rnd()&1)?0x80000000u:0));
            if(k<64) a=(float)k*0.015625f-0.5f;           /* small exact values incl. 0 */
            float want=a/d;
            float q0=a*y;
            float q1=fmaf(fmaf(-d,q0,a),y,q0);
            float q2=fmaf(fmaf(-d,q1,a),y,q1);
            if(f2u(q1)!=f2u(want)) ++bad1;
            if(f2u(q2)!=f2u(want)) { if(bad2<5) printf("q2 mismatch a=%a d=%a want=%a got=%a\n",a,d,want,q2); ++bad2; }
            ++n;
        }
    }
    printf("n=%lu  one-correction mismatches=%lu  two-correction mismatches=%lu\n",n,bad1,bad2);
    return 0;
}
