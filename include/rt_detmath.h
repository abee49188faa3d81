/*
 * rt_detmath.h — deterministic sin/cos/log for the sampling routines, acos/atan2 for the UV maps.
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
 * Domain: det_sincos for |x| <= 64 (callers pass [0, 2 pi)); det_log for normal x > 0;
 * det_atan / det_atan2 / det_acos: every input (the case analysis of the classic libm algorithms:
 * zeros, infinities and NaN come out as C's atan2 / acos define them).  They additionally use the
 * IEEE square root, which is correctly rounded on both sides.  The sphere and sky UV maps
 * (src/object/sphere.rs:69-75, src/object/sky.rs:40-46) feed texel lookups and checker parities:
 * a one-ulp difference in u can flip a whole texel, so both sides must share these bits too.
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

/* atan(x): reduction to |t| < 7/16 around the break points 0.5, 1, 1.5, inf (atan x = atan c + atan((x - c) / (1 + c x)),
 * the four atan c as hi + lo pairs), odd minimax polynomial of degree 23 in two interleaved Horner chains. */
RT_DM_HD double det_atan(double x) {
    const double hi0 = 4.63647609000806093515e-01, lo0 = 2.26987774529616870924e-17; /* atan 0.5 */
    const double hi1 = 7.85398163397448278999e-01, lo1 = 3.06161699786838301793e-17; /* atan 1   */
    const double hi2 = 9.82793723247329054082e-01, lo2 = 1.39033110312309984516e-17; /* atan 1.5 */
    const double hi3 = 1.57079632679489655800e+00, lo3 = 6.12323399573676603587e-17; /* atan inf */
    if (x != x) return x + x;
    uint64_t bits;
    memcpy(&bits, &x, sizeof bits);
    const int neg = (int)(bits >> 63);
    double a = neg ? -x : x;
    double hi = 0.0, lo = 0.0, t;
    int reduced = 1;
    if (a >= 7.378697629483820646e19) { /* 2^66: atan = +-pi/2 to the last bit */
        double r = hi3 + lo3;
        return neg ? -r : r;
    }
    if (a < 0.4375) {
        if (a < 1.862645149230957e-09) return x; /* 2^-29 */
        t = x;
        reduced = 0;
    } else if (a < 0.6875) {
        t = (2.0 * a - 1.0) / (2.0 + a); hi = hi0; lo = lo0;
    } else if (a < 1.1875) {
        t = (a - 1.0) / (a + 1.0); hi = hi1; lo = lo1;
    } else if (a < 2.4375) {
        t = (a - 1.5) / (1.0 + 1.5 * a); hi = hi2; lo = lo2;
    } else {
        t = -1.0 / a; hi = hi3; lo = lo3;
    }
    double z = t * t;
    double w = z * z;
    double s1 = 1.62858201153657823623e-02;
    s1 = 4.97687799461593236017e-02 + w * s1;
    s1 = 6.66107313738753120669e-02 + w * s1;
    s1 = 9.09088713343650656196e-02 + w * s1;
    s1 = 1.42857142725034663711e-01 + w * s1;
    s1 = 3.33333333333329318027e-01 + w * s1;
    s1 = z * s1;
    double s2 = -3.65315727442169155270e-02;
    s2 = -5.83357013379057348645e-02 + w * s2;
    s2 = -7.69187620504482999495e-02 + w * s2;
    s2 = -1.11111104054623557880e-01 + w * s2;
    s2 = -1.99999999998764832476e-01 + w * s2;
    s2 = w * s2;
    if (!reduced) return t - t * (s1 + s2);
    double r = hi - ((t * (s1 + s2) - lo) - t);
    return neg ? -r : r;
}

/* atan2(y, x) with C's special cases; the angle of the quotient, moved to the quadrant of (x, y). */
RT_DM_HD double det_atan2(double y, double x) {
    const double pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16;
    const double pio2 = 1.5707963267948965580e+00, pio4 = 7.8539816339744827900e-01;
    if (x != x || y != y) return x + y;
    uint64_t bx, by;
    memcpy(&bx, &x, sizeof bx);
    memcpy(&by, &y, sizeof by);
    const int m = (int)(by >> 63) | ((int)(bx >> 63) << 1); /* bit 0: y negative, bit 1: x negative */
    const uint64_t ax = bx & 0x7FFFFFFFFFFFFFFFull, ay = by & 0x7FFFFFFFFFFFFFFFull;
    const uint64_t inf = 0x7FF0000000000000ull;
    if (ay == 0) return m == 0 ? 0.0 : (m == 1 ? -0.0 : (m == 2 ? pi : -pi));
    if (ax == 0) return (m & 1) ? -pio2 : pio2;
    if (ax == inf) {
        if (ay == inf) return m == 0 ? pio4 : (m == 1 ? -pio4 : (m == 2 ? 3.0 * pio4 : -3.0 * pio4));
        return m == 0 ? 0.0 : (m == 1 ? -0.0 : (m == 2 ? pi : -pi));
    }
    if (ay == inf) return (m & 1) ? -pio2 : pio2;
    const int k = (int)(ay >> 52) - (int)(ax >> 52);
    double z;
    int q = m;
    if (k > 60) { z = pio2 + 0.5 * pi_lo; q = m & 1; }  /* |y / x| > 2^60 */
    else if ((m & 2) && k < -60) z = 0.0;               /* |y / x| < 2^-60, x < 0 */
    else {
        double quo = y / x;
        z = det_atan(quo < 0.0 ? -quo : quo);
    }
    if (q == 0) return z;
    if (q == 1) return -z;
    if (q == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

/* acos(x): rational approximation R(z) = P(z) / Q(z) of (asin(t) - t) / t^3 on z = t^2 <= 1/4; for |x| > 1/2 through
 * acos x = 2 asin(sqrt((1 - x) / 2)), with the square root's rounding error carried for x > 1/2. */
RT_DM_HD double det_acos_r(double z) {
    double p = 3.47933107596021167570e-05;
    p = 7.91534994289814532176e-04 + z * p;
    p = -4.00555345006794114027e-02 + z * p;
    p = 2.01212532134862925881e-01 + z * p;
    p = -3.25565818622400915405e-01 + z * p;
    p = 1.66666666666666657415e-01 + z * p;
    p = z * p;
    double q = 7.70381505559019352791e-02;
    q = -6.88283971605453293030e-01 + z * q;
    q = 2.02094576023350569471e+00 + z * q;
    q = -2.40339491173441421878e+00 + z * q;
    q = 1.0 + z * q;
    return p / q;
}
RT_DM_HD double det_acos(double x) {
    const double pi = 3.14159265358979311600e+00;
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    if (x != x) return x + x;
    const double a = x < 0.0 ? -x : x;
    if (a >= 1.0) {
        if (a == 1.0) return x > 0.0 ? 0.0 : pi + 2.0 * pio2_lo;
        return (x - x) / (x - x); /* NaN */
    }
    if (a < 0.5) {
        if (a <= 6.938893903907228e-18) return pio2_hi + pio2_lo; /* 2^-57 */
        double r = det_acos_r(x * x);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (x < 0.0) {
        double z = (1.0 + x) * 0.5;
        double s = __builtin_sqrt(z);
        double w = det_acos_r(z) * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    double z = (1.0 - x) * 0.5;
    double s = __builtin_sqrt(z);
    uint64_t sb;
    memcpy(&sb, &s, sizeof sb);
    sb &= 0xFFFFFFFF00000000ull;
    double df;
    memcpy(&df, &sb, sizeof df);
    double c = (z - df * df) / (s + df);
    double w = det_acos_r(z) * s + c;
    return 2.0 * (df + w);
}

#endif /* RT_DETMATH_H */
