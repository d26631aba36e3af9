/* Exhaustive accuracy scan of the pinned pow(x, 1/2.2) (oracle/develop_ref.c) against a
 * double-precision pow, over every float in [lo, hi].  Build & run:
 *   gcc -O2 -mfma -ffp-contract=off tools/pow_accuracy.c oracle/develop_ref.c -Ioracle -lm -pthread -o /tmp/pow_acc && /tmp/pow_acc
 */
#include "develop_ref.h"
#include <math.h>
#include <stdio.h>
#include <string.h>

static double ulp_err(float got, double want)
{
    float w = (float)want;
    int e; frexpf(w, &e);
    double ulp = ldexp(1.0, e - 24);
    return fabs((double)got - want) / ulp;
}

int main(void)
{
    const float y = (float)(1.0 / 2.2);
    struct { float lo, hi; } ranges[] = { {1.17549435e-38f, 1e-6f}, {1e-6f, 1e-3f}, {1e-3f, 0.0625f},
                                          {0.0625f, 0.5f}, {0.5f, 1.0f}, {1.0f, 16.0f} };
    for (unsigned r = 0; r < sizeof ranges / sizeof ranges[0]; ++r) {
        unsigned lo, hi; memcpy(&lo, &ranges[r].lo, 4); memcpy(&hi, &ranges[r].hi, 4);
        double worst = 0, sum = 0; float worst_x = 0; unsigned long n = 0, nonmono = 0;
        float prev = 0;
        for (unsigned u = lo; u <= hi; ++u) {
            float x; memcpy(&x, &u, 4);
            float got = ref_powf(x, y, REF_POW_PINNED);
            double e = ulp_err(got, pow((double)x, (double)y));
            if (e > worst) { worst = e; worst_x = x; }
            sum += e; ++n;
            if (u > lo && got < prev) ++nonmono;
            prev = got;
        }
        printf("[%g, %g]: n=%lu max %.3f ulp at x=%.9g, mean %.3f ulp, non-monotone steps %lu\n",
               ranges[r].lo, ranges[r].hi, n, worst, worst_x, sum / n, nonmono);
    }
    printf("pow(1)=%.9g pow(2,1)=%.9g exp2(0)=%.9g exp2(1)=%.9g exp2(-1)=%.9g log2(2)=%.9g log2(0.5)=%.9g\n",
           ref_powf(1.0f, y, 0), ref_powf(2.0f, 1.0f, 0), ref_exp2f(0), ref_exp2f(1), ref_exp2f(-1), ref_log2f(2), ref_log2f(0.5f));
    return 0;
}
