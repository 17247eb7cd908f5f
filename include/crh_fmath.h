/* crh_fmath.h — deterministic f32 elementary functions shared by host and gfx950 device code.
 *
 * Why this exists: the reference (Rust) calls the platform libm for the few transcendentals on the
 * tessellation path — `acos` (stroke.rs:111), complex `arg`/`powf` = atan2 / cos / sin / pow
 * (curve.rs:230-238 through geometric_algebra 0.3.0) — so its float payloads already differ between
 * platforms in the last ulp. To make "GPU bytes == CPU bytes" a testable property, the boundary's
 * numerical contract replaces libm by the functions below: every one is evaluated in IEEE binary64
 * with only + - * / sqrt and integer bit operations (correctly rounded on x86-64 and on gfx950 when
 * compiled with -ffp-contract=off), then rounded once to binary32. The result is within 1 ulp(f32)
 * of the exact value (checked against libm in tests/test_fmath.py) and bit-identical on both targets
 * (checked on the GPU in tests/test_gpu_fmath.py).
 *
 * No magic minimax tables: plain range reduction + Taylor/atanh series, so the file can be audited
 * by eye. Speed is irrelevant — these run once per curve interval / join, not per sample.
 */
#ifndef CRH_FMATH_H
#define CRH_FMATH_H

#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define CRH_HD __host__ __device__ static inline
#else
#define CRH_HD static inline
#endif

#define CRH_PI 3.14159265358979323846
#define CRH_PI_2 1.57079632679489661923
#define CRH_PI_4 0.78539816339744830962
#define CRH_PI_8 0.39269908169872415481

CRH_HD uint64_t crh_d2u(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
CRH_HD double crh_u2d(uint64_t u) {
    double x;
    memcpy(&x, &u, 8);
    return x;
}
CRH_HD uint32_t crh_f2u(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return u;
}
CRH_HD float crh_u2f(uint32_t u) {
    float x;
    memcpy(&x, &u, 4);
    return x;
}
CRH_HD float crh_nanf(void) { return crh_u2f(0x7fc00000u); }
CRH_HD int crh_isnan_d(double x) { return x != x; }
CRH_HD double crh_abs_d(double x) { return crh_u2d(crh_d2u(x) & 0x7fffffffffffffffull); }
CRH_HD double crh_floor_d(double x) {
    /* |x| < 2^51 on every call site */
    double t = (double)(int64_t)x;
    return (t > x) ? t - 1.0 : t;
}

/* atan(z) for |z| <= tan(pi/16) ~ 0.19892: alternating Taylor series, 13 terms (next term < 2e-19). */
CRH_HD double crh_d_atan_small(double z) {
    const double z2 = z * z;
    double s = 1.0 / 25.0;
    s = 1.0 / 23.0 - z2 * s;
    s = 1.0 / 21.0 - z2 * s;
    s = 1.0 / 19.0 - z2 * s;
    s = 1.0 / 17.0 - z2 * s;
    s = 1.0 / 15.0 - z2 * s;
    s = 1.0 / 13.0 - z2 * s;
    s = 1.0 / 11.0 - z2 * s;
    s = 1.0 / 9.0 - z2 * s;
    s = 1.0 / 7.0 - z2 * s;
    s = 1.0 / 5.0 - z2 * s;
    s = 1.0 / 3.0 - z2 * s;
    s = 1.0 - z2 * s;
    return z * s;
}

/* atan2 in binary64 (finite inputs; NaN in -> NaN out; atan2(+-0, +-0) follows C99). */
CRH_HD double crh_d_atan2(double y, double x) {
    if (crh_isnan_d(x) || crh_isnan_d(y)) return x + y;
    const int neg_x = (int)(crh_d2u(x) >> 63);
    const int neg_y = (int)(crh_d2u(y) >> 63);
    const double ax = crh_abs_d(x), ay = crh_abs_d(y);
    double res;
    if (ax == 0.0 && ay == 0.0) {
        res = 0.0;
    } else {
        const int swap = ay > ax;
        const double r = swap ? ax / ay : ay / ax; /* in [0,1] */
        double base, z;
        if (r > 0.66817863791929898) { /* tan(3 pi/16) */
            z = (r - 1.0) / (r + 1.0);
            base = CRH_PI_4;
        } else if (r > 0.19891236737965800) { /* tan(pi/16) */
            const double c = 0.41421356237309503; /* tan(pi/8) = sqrt(2) - 1 */
            z = (r - c) / (1.0 + r * c);
            base = CRH_PI_8;
        } else {
            z = r;
            base = 0.0;
        }
        res = base + crh_d_atan_small(z);
        if (swap) res = CRH_PI_2 - res;
    }
    if (neg_x) res = CRH_PI - res;
    return neg_y ? -res : res;
}

/* sin and cos of a (|a| < 2^20) in binary64: Cody-Waite reduction by pi/2, Taylor on [-pi/4, pi/4]. */
CRH_HD void crh_d_sincos(double a, double* s_out, double* c_out) {
    if (crh_isnan_d(a) || crh_abs_d(a) > 1048576.0) {
        *s_out = *c_out = crh_u2d(0x7ff8000000000000ull);
        return;
    }
    const double k = crh_floor_d(a * 0.63661977236758134308 + 0.5); /* a * 2/pi */
    const double pio2_hi = 1.57079632673412561417e+00;              /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11;              /* pi/2 - pio2_hi */
    const double r = (a - k * pio2_hi) - k * pio2_lo;
    const double r2 = r * r;
    /* sin r = r (1 - r2/(2*3) (1 - r2/(4*5) (1 - ...))) up to r^21 */
    double s = 1.0 - r2 * (1.0 / 420.0);                            /* 20*21 */
    s = 1.0 - r2 * (1.0 / 342.0) * s;                               /* 18*19 */
    s = 1.0 - r2 * (1.0 / 272.0) * s;                               /* 16*17 */
    s = 1.0 - r2 * (1.0 / 210.0) * s;                               /* 14*15 */
    s = 1.0 - r2 * (1.0 / 156.0) * s;                               /* 12*13 */
    s = 1.0 - r2 * (1.0 / 110.0) * s;                               /* 10*11 */
    s = 1.0 - r2 * (1.0 / 72.0) * s;                                /* 8*9 */
    s = 1.0 - r2 * (1.0 / 42.0) * s;                                /* 6*7 */
    s = 1.0 - r2 * (1.0 / 20.0) * s;                                /* 4*5 */
    s = 1.0 - r2 * (1.0 / 6.0) * s;                                 /* 2*3 */
    s = r * s;
    /* cos r = 1 - r2/(1*2) (1 - r2/(3*4) (1 - ...)) up to r^20 */
    double c = 1.0 - r2 * (1.0 / 380.0);                            /* 19*20 */
    c = 1.0 - r2 * (1.0 / 306.0) * c;                               /* 17*18 */
    c = 1.0 - r2 * (1.0 / 240.0) * c;                               /* 15*16 */
    c = 1.0 - r2 * (1.0 / 182.0) * c;                               /* 13*14 */
    c = 1.0 - r2 * (1.0 / 132.0) * c;                               /* 11*12 */
    c = 1.0 - r2 * (1.0 / 90.0) * c;                                /* 9*10 */
    c = 1.0 - r2 * (1.0 / 56.0) * c;                                /* 7*8 */
    c = 1.0 - r2 * (1.0 / 30.0) * c;                                /* 5*6 */
    c = 1.0 - r2 * (1.0 / 12.0) * c;                                /* 3*4 */
    c = 1.0 - r2 * (1.0 / 2.0) * c;                                 /* 1*2 */
    const int q = (int)((int64_t)k & 3);
    switch (q) {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}

/* natural log of a positive finite normal binary64. */
CRH_HD double crh_d_log(double b) {
    uint64_t u = crh_d2u(b);
    int64_t e = (int64_t)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = crh_u2d(u); /* [1,2) */
    if (m > 1.41421356237309505) {
        m *= 0.5;
        e += 1;
    }
    /* ln m = 2 atanh(z), z = (m-1)/(m+1), |z| <= 0.1716 ; 16 terms (next < 1e-26) */
    const double z = (m - 1.0) / (m + 1.0);
    const double z2 = z * z;
    double s = 1.0 / 31.0;
    s = 1.0 / 29.0 + z2 * s;
    s = 1.0 / 27.0 + z2 * s;
    s = 1.0 / 25.0 + z2 * s;
    s = 1.0 / 23.0 + z2 * s;
    s = 1.0 / 21.0 + z2 * s;
    s = 1.0 / 19.0 + z2 * s;
    s = 1.0 / 17.0 + z2 * s;
    s = 1.0 / 15.0 + z2 * s;
    s = 1.0 / 13.0 + z2 * s;
    s = 1.0 / 11.0 + z2 * s;
    s = 1.0 / 9.0 + z2 * s;
    s = 1.0 / 7.0 + z2 * s;
    s = 1.0 / 5.0 + z2 * s;
    s = 1.0 / 3.0 + z2 * s;
    s = 1.0 + z2 * s;
    const double ln2_hi = 6.93147180369123816490e-01; /* fdlibm split of ln 2 */
    const double ln2_lo = 1.90821492927058770002e-10;
    const double de = (double)e;
    return (de * ln2_hi) + (2.0 * z * s + de * ln2_lo);
}

/* exp(y) for |y| < 700 in binary64. */
CRH_HD double crh_d_exp(double y) {
    const double n = crh_floor_d(y * 1.44269504088896338700 + 0.5); /* y / ln 2 */
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double r = (y - n * ln2_hi) - n * ln2_lo; /* |r| <= 0.3466 */
    /* Taylor to r^18 (next term < 1e-25), Horner with 1/k factors */
    double s = 1.0 + r * (1.0 / 18.0);
    s = 1.0 + r * (1.0 / 17.0) * s;
    s = 1.0 + r * (1.0 / 16.0) * s;
    s = 1.0 + r * (1.0 / 15.0) * s;
    s = 1.0 + r * (1.0 / 14.0) * s;
    s = 1.0 + r * (1.0 / 13.0) * s;
    s = 1.0 + r * (1.0 / 12.0) * s;
    s = 1.0 + r * (1.0 / 11.0) * s;
    s = 1.0 + r * (1.0 / 10.0) * s;
    s = 1.0 + r * (1.0 / 9.0) * s;
    s = 1.0 + r * (1.0 / 8.0) * s;
    s = 1.0 + r * (1.0 / 7.0) * s;
    s = 1.0 + r * (1.0 / 6.0) * s;
    s = 1.0 + r * (1.0 / 5.0) * s;
    s = 1.0 + r * (1.0 / 4.0) * s;
    s = 1.0 + r * (1.0 / 3.0) * s;
    s = 1.0 + r * (1.0 / 2.0) * s;
    s = 1.0 + r * s;
    const int64_t ni = (int64_t)n;
    if (ni < -1000) return 0.0;
    if (ni > 1000) return crh_u2d(0x7ff0000000000000ull);
    return s * crh_u2d((uint64_t)(ni + 1023) << 52);
}

/* b^e for b >= 0 in binary64 (b = 0 -> 0 for e > 0, 1 for e == 0, inf for e < 0). */
CRH_HD double crh_d_pow(double b, double e) {
    if (crh_isnan_d(b) || crh_isnan_d(e)) return b + e;
    if (e == 0.0) return 1.0;
    if (b == 0.0) return e > 0.0 ? 0.0 : crh_u2d(0x7ff0000000000000ull);
    if (b == 1.0) return 1.0;
    const double inf = crh_u2d(0x7ff0000000000000ull);
    if (b == inf) return e > 0.0 ? inf : 0.0;
    if (crh_abs_d(e) == inf) {
        const int grow = (b > 1.0) == (e > 0.0);
        return grow ? inf : 0.0;
    }
    const double y = e * crh_d_log(b);
    if (y > 700.0) return inf;
    if (y < -745.0) return 0.0;
    return crh_d_exp(y);
}

/* cube root of any finite binary64. */
CRH_HD double crh_d_cbrt(double x) {
    if (x == 0.0 || crh_isnan_d(x)) return x;
    const double a = crh_abs_d(x);
    double r = crh_d_pow(a, 1.0 / 3.0);
    /* one Newton step removes the pow() rounding */
    r = r - (r * r * r - a) / (3.0 * r * r);
    return x < 0.0 ? -r : r;
}

/* ---- binary32 front ends (the functions the tessellation path calls) -------------------------- */

/* f32::atan2 / ComplexNumber::arg */
CRH_HD float crh_atan2f(float y, float x) { return (float)crh_d_atan2((double)y, (double)x); }

/* f32::acos (NaN outside [-1, 1], as libm) */
CRH_HD float crh_acosf(float x) {
    const double d = (double)x;
    if (!(d >= -1.0 && d <= 1.0)) return crh_nanf();
    return (float)crh_d_atan2(sqrt((1.0 - d) * (1.0 + d)), d);
}

/* f32::sin_cos */
CRH_HD void crh_sincosf(float a, float* s, float* c) {
    double ds, dc;
    crh_d_sincos((double)a, &ds, &dc);
    *s = (float)ds;
    *c = (float)dc;
}

/* f32::powf for a non-negative base */
CRH_HD float crh_powf(float b, float e) { return (float)crh_d_pow((double)b, (double)e); }

/* WGSL `%` on f32: e1 - e2 * trunc(e1 / e2) (shaders.wgsl:211) — NOT libm fmod. */
CRH_HD float crh_wgsl_mod(float a, float b) {
    const float q = a / b;
    const float t = truncf(q); /* exact in f32 */
    return a - b * t;
}

#endif /* CRH_FMATH_H */
