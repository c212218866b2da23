/* Exhaustive check of the constant-division shortcut used by k_haar_select32.hip:
 *     q0 = x * r;  e = fma(-d, q0, x);  q = fma(e, r, q0)        with r = RN(1 / d)
 * against the correctly rounded x / d for EVERY float32 bit pattern x, for the three divisors the Haar
 * uses (sqrtf(2), sqrtf(32), sqrtf(128)).  Prints, per divisor, how many inputs disagree and the
 * magnitude range that contains every disagreement.
 * build: gcc -O2 -mfma -fopenmp -ffp-contract=off tools/verify_const_div.c -o /tmp/verify_const_div -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(void) {
    const float ds[3] = {sqrtf(2.0f), sqrtf(32.0f), sqrtf(128.0f)};
    for (int t = 0; t < 3; ++t) {
        const float d = ds[t], r = 1.0f / d;
        unsigned long long bad = 0;
        uint32_t lo_bad = 0xFFFFFFFFu, hi_bad = 0;
#pragma omp parallel for reduction(+ : bad) reduction(min : lo_bad) reduction(max : hi_bad) schedule(static)
        for (long long i = 0; i < (1LL << 32); ++i) {
            const uint32_t u = (uint32_t)i;
            const float x = from_bits(u);
            const float want = x / d;
            const float q0 = x * r;
            const float e = fmaf(-d, q0, x);
            const float q = fmaf(e, r, q0);
            int same = to_bits(q) == to_bits(want);
            if (!same && want != want && q != q) same = 1; /* both NaN */
            if (!same) {
                ++bad;
                const uint32_t mag = u & 0x7fffffffu;
                if (mag < lo_bad) lo_bad = mag;
                if (mag > hi_bad) hi_bad = mag;
            }
        }
        printf("d = %.9g (bits %08x), r = %.9g: %llu mismatches", d, to_bits(d), r, bad);
        if (bad) printf(", |x| bits in [%08x, %08x] = [%g, %g]", lo_bad, hi_bad, from_bits(lo_bad), from_bits(hi_bad));
        printf("\n");
    }
    return 0;
}
