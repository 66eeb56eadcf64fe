/* Host check of the series behind box_muller_pair (drl-dronenavigation_amd/csrc/dn_kernels.hip): the same operation sequence in
 * C (frexp + 2 atanh series, Taylor sine / cosine after the quadrant reduction; the IEEE division and sqrt stand in for
 * the device's v_rcp_f64 / v_rsq_f64 + Newton steps) against the libm form log / sqrt / cos / sin, over 2*10^7 random
 * word pairs plus the edge words.   gcc -O2 -o /tmp/bm profiles/box_muller_host_check.c -lm && /tmp/bm
 * Last run: worst abs 9.415e-14, float32 flips 5 of 40000016.  (Measurement aid, not part of the product or the tests;
 * the device code is checked by tests/test_gpu_parity.py::test_gaussian_draws_match_oracle_to_float32_rounding.) */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static void bm(uint32_t ra, uint32_t rb, double *o0, double *o1)
{
    const double u1 = ((double)ra + 0.5) * (1.0 / 4294967296.0);
    const double u2 = ((double)rb + 0.5) * (1.0 / 4294967296.0);
    int e; double m = frexp(u1, &e);
    if (m < 0.70710678118654752440) { m = m + m; e = e - 1; }
    const double num = m - 1.0, den = m + 1.0;
    double r = 1.0 / den;
    const double sv = num * r, s2 = sv * sv;
    double pl = 1.0 / 15.0;
    pl = fma(pl, s2, 1.0 / 13.0); pl = fma(pl, s2, 1.0 / 11.0); pl = fma(pl, s2, 1.0 / 9.0); pl = fma(pl, s2, 1.0 / 7.0);
    pl = fma(pl, s2, 1.0 / 5.0); pl = fma(pl, s2, 1.0 / 3.0); pl = fma(pl, s2, 1.0);
    const double ln_u1 = fma((double)e, 0.69314718055994530942, (sv + sv) * pl);
    const double t = -2.0 * ln_u1;
    const double rad = sqrt(t);
    const double k4 = rint(u2 * 4.0);
    const double th = (2.0 * 3.14159265358979323846) * fma(k4, -0.25, u2);
    const double t2 = th * th;
    double ps = 1.0 / 6227020800.0;
    ps = fma(ps, t2, -1.0 / 39916800.0); ps = fma(ps, t2, 1.0 / 362880.0); ps = fma(ps, t2, -1.0 / 5040.0);
    ps = fma(ps, t2, 1.0 / 120.0); ps = fma(ps, t2, -1.0 / 6.0);
    const double sn = fma(th * t2, ps, th);
    double pc = -1.0 / 87178291200.0;
    pc = fma(pc, t2, 1.0 / 479001600.0); pc = fma(pc, t2, -1.0 / 3628800.0); pc = fma(pc, t2, 1.0 / 40320.0);
    pc = fma(pc, t2, -1.0 / 720.0); pc = fma(pc, t2, 1.0 / 24.0); pc = fma(pc, t2, -0.5);
    const double cs = fma(pc, t2, 1.0);
    const int k = (int)k4 & 3;
    const double c = (k & 1) ? sn : cs, d = (k & 1) ? cs : sn;
    *o0 = rad * ((k == 1 || k == 2) ? -c : c);
    *o1 = rad * ((k >= 2) ? -d : d);
}
int main(void)
{
    double worst = 0, worst_rel = 0; long flips = 0, N = 20000000;
    uint64_t x = 88172645463325252ull;
    for (long i = 0; i < N + 8; ++i) {
        uint32_t ra, rb;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17; ra = (uint32_t)(x >> 32); rb = (uint32_t)x;
        if (i == N) { ra = 0; rb = 0; } if (i == N + 1) { ra = 0xFFFFFFFFu; rb = 0xFFFFFFFFu; }
        if (i == N + 2) { ra = 0xFFFFFFFFu; rb = 0x80000000u; } if (i == N + 3) { ra = 1; rb = 0x3FFFFFFFu; }
        if (i == N + 4) { ra = 0x80000000u; rb = 0x40000000u; } if (i == N + 5) { ra = 0xB504F333u; rb = 0xC0000000u; }
        if (i == N + 6) { ra = 0xB504F334u; rb = 0x1FFFFFFFu; } if (i == N + 7) { ra = 0x7FFFFFFFu; rb = 0xE0000000u; }
        double a0, a1; bm(ra, rb, &a0, &a1);
        const double u1 = ((double)ra + 0.5) * (1.0 / 4294967296.0), u2 = ((double)rb + 0.5) * (1.0 / 4294967296.0);
        const double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * 3.14159265358979323846 * u2;
        const double b0 = rad * cos(ang), b1 = rad * sin(ang);
        const double e0 = fabs(a0 - b0), e1 = fabs(a1 - b1);
        if (e0 > worst) worst = e0; if (e1 > worst) worst = e1;
        if ((float)a0 != (float)b0) ++flips; if ((float)a1 != (float)b1) ++flips;
    }
    printf("worst abs %.3e  float32 flips %ld of %ld\n", worst, flips, 2 * (N + 8));
    return 0;
}
