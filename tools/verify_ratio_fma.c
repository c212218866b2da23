/* Exhaustive check of the table-free ratio of compare_short_multi_kernel (k_sliding.hip, round 6):
 *     pf = (float)possible;  rh = 1.0f / pf;  rl = fma(-pf, rh, 1.0f) * rh;
 *     ratio = fma(hf, rh, hf * rl)                                   with hf = (float)hits
 * against the correctly rounded (float)hits / (float)possible of LBAudioDetectiveFingerprint.m:175, for EVERY
 * 0 <= hits <= possible <= 100 (the 5151 entries of the quotient table the other scans read from LDS), bit for bit.
 * Every operation above is a single correctly rounded IEEE float32 operation on the device as well
 * (-ffp-contract=off, -fhip-fp32-correctly-rounded-divide-sqrt, explicit __fmaf_rn / __fmul_rn / __fdiv_rn).
 * Prints the number of disagreements (0) and exits non-zero when there is one.
 * build: gcc -O2 -mfma -ffp-contract=off tools/verify_ratio_fma.c -o /tmp/verify_ratio_fma -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#ifndef LIMIT
#define LIMIT 100          /* pairs of a sub-fingerprint: 100 at 200 Booleans; -DLIMIT=128 for the uniform corpus' 256 */
#endif

static inline uint32_t to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(void) {
    int bad = 0, n = 0;
    for (int p = 0; p <= LIMIT; ++p) {
        const float pf = (float)p;
        volatile float rh = p ? 1.0f / pf : 0.0f;
        volatile float e = fmaf(-pf, rh, p ? 1.0f : 0.0f);
        volatile float rl = e * rh;
        for (int h = 0; h <= p; ++h, ++n) {
            const float hf = (float)h;
            volatile float t = hf * rl;
            const float got = fmaf(hf, rh, t);
            const float want = p ? hf / pf : 0.0f;       /* Fp.m:170-175: no possible pair -> 0 */
            if (to_bits(got) != to_bits(want)) {
                ++bad;
                printf("possible %d hits %d: %a != %a\n", p, h, got, want);
            }
        }
    }
    printf("%d pairs checked, %d disagree\n", n, bad);
    return bad != 0;
}
