// Exhaustive check: is  q = x*r; e = fma(-b,q,x); q' = fma(e,r,q)  (r = RN(1/b)) equal to the IEEE
// quotient x/b for EVERY finite float x?  Prints the number of mismatches per magnitude range.
// gcc -O2 -fopenmp -ffp-contract=off tools/check_constdiv.c -lm -o /tmp/check_constdiv && /tmp/check_constdiv 9 3 639 191
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv)
{
    for (int a = 1; a < argc; ++a) {
        const float b = (float)atof(argv[a]);
        const float r = 1.0f / b;
        uint64_t bad = 0, bad_normal = 0;
        uint32_t first_bad = 0, lo_ok = 0, hi_ok = 0;
        uint32_t min_bad_exp = 255, max_bad_exp = 0;
#pragma omp parallel for reduction(+ : bad, bad_normal) reduction(min : min_bad_exp) reduction(max : max_bad_exp)
        for (uint32_t u = 0; u < 0x7f800000u; ++u) {
            float x;
            memcpy(&x, &u, 4);
            float q = x * r;
            float e = fmaf(-b, q, x);
            float q2 = fmaf(e, r, q);
            float ref = x / b;
            if (memcmp(&q2, &ref, 4)) {
                ++bad;
                uint32_t ex = u >> 23;
                if (ex < min_bad_exp) min_bad_exp = ex;
                if (ex > max_bad_exp) max_bad_exp = ex;
                if (ex >= 30 && ex <= 250) ++bad_normal;
            }
        }
        printf("b=%g  mismatches=%llu  (biased exponent of bad x: min %u max %u)  mismatches with exp in [30,250]: %llu\n",
               b, (unsigned long long)bad, min_bad_exp, max_bad_exp, (unsigned long long)bad_normal);
    }
    return 0;
}
