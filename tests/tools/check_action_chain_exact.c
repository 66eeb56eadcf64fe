/*
 * check_action_chain_exact.c -- exhaustive proof that the short float32 sequences the HIP kernel uses for the
 * action chain (A1-A3: PBDroneEnv.rescale_action / _preprocessAction / cmd2pwm / pwm2rpm) are BIT-IDENTICAL to the
 * IEEE-754 correctly rounded operations numpy performs in the reference, over every float32 input that can reach
 * them.  TEST TOOL (tests/test_action_chain_exact.py builds and runs it); not part of the product.
 *
 *   division by a constant c:   q0 = a * RN(1/c);  r = fma(-q0, c, a);  q = fma(r, RN(1/c), q0)      vs   a / c
 *   square root:                s = approx;  one ulp below / above tested with fma residuals          vs   sqrtf
 *
 * Usage: check_action_chain_exact [stride]   (stride 1 = exhaustive, default; larger = subsample for quick runs)
 * Build: gcc -O2 -mfma -fopenmp -ffp-contract=off check_action_chain_exact.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float div_const(float a, float c, float y)
{
    float q0 = a * y;
    float r = fmaf(-q0, c, a);
    return fmaf(r, y, q0);
}

/* sqrt: the hardware seed is within 1 ulp; emulate the worst case by starting from sqrtf(x) +- 1 ulp */
static inline float sqrt_fix(float x, float s)
{
    float sm = u2f(f2u(s) - 1), sp = u2f(f2u(s) + 1);
    float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);
    float out = s;
    if (rm <= 0.0f) out = sm;
    if (rp > 0.0f) out = sp;
    return out;
}

int main(int argc, char **argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 1u;
    const float KF = (float)3.16e-10, SCALE = (float)0.2685, CONST_ = (float)4070.3;
    const float A_LOW = (float)(3.16e-10 * ((0.2685 * 20000.0 + 4070.3) * (0.2685 * 20000.0 + 4070.3)));
    const float A_HIGH = (float)(3.16e-10 * ((0.2685 * 65535.0 + 4070.3) * (0.2685 * 65535.0 + 4070.3)));
    const float DEN = A_HIGH - A_LOW;
    const float yDEN = 1.0f / DEN, yKF = 1.0f / KF, ySCALE = 1.0f / SCALE;
    long long bad_den = 0, bad_kf = 0, bad_scale = 0, bad_sqrt = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0;

    /* (a - A_LOW) / DEN for every float32 action a in [-1.5, 1.5] (the action space is [-1, 1]; noise is clipped) */
#pragma omp parallel for reduction(+ : bad_den, n1) schedule(static)
    for (long long k = 0; k <= (long long)f2u(1.5f); k += stride) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            float a = u2f((uint32_t)k | ((uint32_t)sgn << 31));
            float num = a - A_LOW;
            if (f2u(div_const(num, DEN, yDEN)) != f2u(num / DEN)) ++bad_den;
            ++n1;
        }
    }
    /* thrust / KF for every float32 thrust in [A_LOW, A_HIGH] */
#pragma omp parallel for reduction(+ : bad_kf, bad_sqrt, n2, n4) schedule(static)
    for (long long k = f2u(A_LOW); k <= (long long)f2u(A_HIGH); k += stride) {
        float t = u2f((uint32_t)k);
        float q = t / KF;
        if (f2u(div_const(t, KF, yKF)) != f2u(q)) ++bad_kf;
        ++n2;
        /* sqrt of the quotient, from a seed one ulp low, exact, and one ulp high */
        float s = sqrtf(q);
        for (int d = -1; d <= 1; ++d) {
            float seed = u2f(f2u(s) + d);
            if (f2u(sqrt_fix(q, seed)) != f2u(s)) ++bad_sqrt;
            ++n4;
        }
    }
    /* (s - CONST) / SCALE for every float32 s in [sqrt(A_LOW/KF) - 1, sqrt(A_HIGH/KF) + 1] */
    float slo = sqrtf(A_LOW / KF) - 1.0f, shi = sqrtf(A_HIGH / KF) + 1.0f;
#pragma omp parallel for reduction(+ : bad_scale, n3) schedule(static)
    for (long long k = f2u(slo); k <= (long long)f2u(shi); k += stride) {
        float num = u2f((uint32_t)k) - CONST_;
        if (f2u(div_const(num, SCALE, ySCALE)) != f2u(num / SCALE)) ++bad_scale;
        ++n3;
    }
    printf("{\"stride\": %u, \"den\": [%lld, %lld], \"kf\": [%lld, %lld], \"scale\": [%lld, %lld], \"sqrt\": [%lld, %lld]}\n",
           stride, bad_den, n1, bad_kf, n2, bad_scale, n3, bad_sqrt, n4);
    return (bad_den || bad_kf || bad_scale || bad_sqrt) ? 1 : 0;
}
