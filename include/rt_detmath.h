/*
 * rt_detmath.h — deterministic sin/cos/log for the sampling routines.
 *
 * The reference calls Rust's f64::sin / cos / ln (src/vec4.rs:50-61, src/object/sphere.rs:131-145,
 * rand_distr's normal sampler), i.e. whatever libm the platform has: results differ in the last
 * ulp between platforms, and between glibc and the GPU's OCML.  On mirror-like curved geometry
 * such one-ulp differences are amplified ~10x per bounce, so a CPU oracle and a GPU kernel that
 * use different libms drift apart after ~8 bounces although both are "right".
 *
 * These functions use only IEEE +, -, *, /, floor and integer bit operations, written as separate
 * operations (build with -ffp-contract=off): the same input gives the same bits with g++ on x86
 * and hipcc on gfx950.  Accuracy is ~1 ulp (tests/test_detmath.py), i.e. as good as any libm the
 * reference may have run on.  Used by oracle/oracle.cpp and by the f64 kernels.
 *
 * Domain: det_sincos for |x| <= 64 (callers pass [0, 2 pi)); det_log for normal x > 0.
 */
#ifndef RT_DETMATH_H
#define RT_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_DM_HD __host__ __device__ inline
#else
#define RT_DM_HD inline
#endif

RT_DM_HD double rt_dm_floor(double x) { return __builtin_floor(x); }

/* sin(x), cos(x): Cody-Waite reduction by pi/2 (two-constant split, exact for |k| < 2^20),
 * Taylor polynomials on [-pi/4, pi/4] in Horner form. */
RT_DM_HD void det_sincos(double x, double* s_out, double* c_out) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11; /* pi/2 - pio2_hi */
    double kf = rt_dm_floor(x * two_over_pi + 0.5);
    double r = (x - kf * pio2_hi) - kf * pio2_lo;
    double z = r * r;
    /* sin r = r + r z (S1 + z (S2 + ...)), S_k = (-1)^k / (2k+1)! */
    double ps = -7.64716373181981647590e-13;            /* -1/15! */
    ps = 1.60590438368216145994e-10 + z * ps;           /*  1/13! */
    ps = -2.50521083854417187751e-08 + z * ps;          /* -1/11! */
    ps = 2.75573192239858906526e-06 + z * ps;           /*  1/9!  */
    ps = -1.98412698412698412698e-04 + z * ps;          /* -1/7!  */
    ps = 8.33333333333333333333e-03 + z * ps;           /*  1/5!  */
    ps = -1.66666666666666666667e-01 + z * ps;          /* -1/3!  */
    double sr = r + (r * z) * ps;
    /* cos r = 1 - z/2 + z^2 (C2 + z (C3 + ...)), C_k = (-1)^k / (2k)! */
    double pc = 4.77947733238738529744e-14;             /*  1/16! */
    pc = -1.14707455977297247139e-11 + z * pc;          /* -1/14! */
    pc = 2.08767569878680989792e-09 + z * pc;           /*  1/12! */
    pc = -2.75573192239858906526e-07 + z * pc;          /* -1/10! */
    pc = 2.48015873015873015873e-05 + z * pc;           /*  1/8!  */
    pc = -1.38888888888888888889e-03 + z * pc;          /* -1/6!  */
    pc = 4.16666666666666666667e-02 + z * pc;           /*  1/4!  */
    double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    int k = (int)kf & 3;
    double s, c;
    if (k == 0) { s = sr; c = cr; }
    else if (k == 1) { s = cr; c = -sr; }
    else if (k == 2) { s = -sr; c = -cr; }
    else { s = -cr; c = sr; }
    *s_out = s;
    *c_out = c;
}

RT_DM_HD double det_sin(double x) { double s, c; det_sincos(x, &s, &c); return s; }
RT_DM_HD double det_cos(double x) { double s, c; det_sincos(x, &s, &c); return c; }

/* ln(x) for normal x > 0: x = m 2^e with m in (sqrt(1/2), sqrt(2)], ln m = 2 atanh((m-1)/(m+1)). */
RT_DM_HD double det_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    uint64_t bits;
    memcpy(&bits, &x, sizeof bits);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    uint64_t mb = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &mb, sizeof m);
    if (m > 1.41421356237309504880) {
        m = m * 0.5;
        e = e + 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 4.76190476190476190476e-02;              /* 1/21 */
    p = 5.26315789473684210526e-02 + z * p;             /* 1/19 */
    p = 5.88235294117647058824e-02 + z * p;             /* 1/17 */
    p = 6.66666666666666666667e-02 + z * p;             /* 1/15 */
    p = 7.69230769230769230769e-02 + z * p;             /* 1/13 */
    p = 9.09090909090909090909e-02 + z * p;             /* 1/11 */
    p = 1.11111111111111111111e-01 + z * p;             /* 1/9  */
    p = 1.42857142857142857143e-01 + z * p;             /* 1/7  */
    p = 2.00000000000000000000e-01 + z * p;             /* 1/5  */
    p = 3.33333333333333333333e-01 + z * p;             /* 1/3  */
    double lm = 2.0 * (s + (s * z) * p);
    double ef = (double)e;
    return ef * ln2_hi + (ef * ln2_lo + lm);
}

#endif /* RT_DETMATH_H */
