// csrc/ga.hpp — device-side projective geometric algebra + polynomial solvers for the tessellation kernels.
//
// The reference takes this arithmetic from the un-vendored crate geometric_algebra 0.3.0 (Cargo.toml:19). The
// operation contract (element layouts, signs, association) is the one SURVEY.md Appendix A derives from the
// reference's call sites; every kernel is compiled with -ffp-contract=off because Rust never fuses a*b+c.
// Transcendentals come from include/crh_fmath.h (the boundary's deterministic libm replacement).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/crh_fmath.h"

namespace crh {

constexpr float kErrorMargin = 0.0001f; // error.rs:19
constexpr float kEpsilon = 1.1920929e-7f; // f32::EPSILON

// Host+device so that tools/ can exercise the very same arithmetic on a CPU when bisecting a parity failure;
// the product only ever calls these from kernels.
#define CRH_D __host__ __device__ __forceinline__

// ppga2d::Point = (w, x*w, y*w) (utils.rs:111-118); ppga2d::Plane = (e0, nx, ny) (utils.rs:101-103)
struct Pt {
    float w, x, y;
};
struct Pl {
    float c, x, y;
};

CRH_D Pt operator+(Pt a, Pt b) { return {a.w + b.w, a.x + b.x, a.y + b.y}; }
CRH_D Pt operator*(Pt a, float s) { return {a.w * s, a.x * s, a.y * s}; }
CRH_D Pl operator+(Pl a, Pl b) { return {a.c + b.c, a.x + b.x, a.y + b.y}; }
CRH_D Pl operator-(Pl a, Pl b) { return {a.c - b.c, a.x - b.x, a.y - b.y}; }
CRH_D Pl operator*(Pl a, float s) { return {a.c * s, a.x * s, a.y * s}; }
CRH_D Pl neg(Pl a) { return {-a.c, -a.x, -a.y}; }

CRH_D Pt vec_to_point(float x, float y) { return {1.0f, x, y}; }                       // utils.rs:111
CRH_D Pt weighted_vec_to_point(float w, float x, float y) { return {w, x * w, y * w}; } // utils.rs:116
CRH_D float2 point_to_vec(Pt p) { return make_float2(p.x / p.w, p.y / p.w); }            // utils.rs:106
CRH_D Pl rotate_cw(Pl v) { return {0.0f, v.y, -v.x}; }                                  // utils.rs:101

// Point v Point -> the line through both (normal = direction rotated 90 degrees clockwise)
CRH_D Pl join(Pt a, Pt b) { return {a.y * b.x - a.x * b.y, a.w * b.y - a.y * b.w, a.x * b.w - a.w * b.x}; }
// Plane v Point -> scalar, three-term sum left to right
CRH_D float join(Pl l, Pt p) { return l.c * p.w + l.x * p.x + l.y * p.y; }
CRH_D float join(Pt p, Pl l) { return p.w * l.c + p.x * l.x + p.y * l.y; }
CRH_D float triple(Pt a, Pt b, Pt c) { return join(join(a, b), c); }
// Plane ^ Plane -> Point (meet)
CRH_D Pt meet(Pl a, Pl b) { return {a.y * b.x - a.x * b.y, a.c * b.y - a.y * b.c, a.x * b.c - a.c * b.x}; }
CRH_D float dot(Pl a, Pl b) { return a.x * b.x + a.y * b.y; }
// Plane . Point -> the line through P perpendicular to l; applied twice == grade-1 part of (l . P) P (stroke.rs:71-75)
CRH_D Pl contract(Pl l, Pt p) { return {l.x * p.y - l.y * p.x, l.y * p.w, -l.x * p.w}; }
CRH_D float sqmag(Pl l) { return l.x * l.x + l.y * l.y; }
CRH_D float mag(Pl l) { return sqrtf(sqmag(l)); }
CRH_D Pl signum(Pl l) { return l * (1.0f / mag(l)); }
CRH_D Pt line_line_intersection(Pl a, Pl b) { // utils.rs:67-70
    const Pt p = meet(a, b);
    return p * (1.0f / p.w);
}
CRH_D float f32_signum(float x) { return (x != x) ? x : ((crh_f2u(x) >> 31) ? -1.0f : 1.0f); }
CRH_D bool is_nan(float x) { return x != x; }
CRH_D bool is_finite(float x) { return (crh_f2u(x) & 0x7f800000u) != 0x7f800000u; }

// mat_vec_transform! (curve.rs:12-23): right-nested sums
CRH_D Pt mvt2(const Pt* p, float a0, float a1) { return p[0] * a0 + p[1] * a1; }
CRH_D Pt mvt3(const Pt* p, float a0, float a1, float a2) { return p[0] * a0 + (p[1] * a1 + p[2] * a2); }
CRH_D Pt mvt4(const Pt* p, float a0, float a1, float a2, float a3) { return p[0] * a0 + (p[1] * a1 + (p[2] * a2 + p[3] * a3)); }

CRH_D void quadratic_power_basis(const Pt cp[3], Pt pb[3]) { // curve.rs:26-32
    pb[0] = cp[0] * 1.0f;
    pb[1] = mvt2(cp, -2.0f, 2.0f);
    pb[2] = mvt3(cp, 1.0f, -2.0f, 1.0f);
}
CRH_D void cubic_power_basis(const Pt cp[4], Pt pb[4]) { // curve.rs:35-42
    pb[0] = cp[0] * 1.0f;
    pb[1] = mvt2(cp, -3.0f, 3.0f);
    pb[2] = mvt3(cp, 3.0f, -6.0f, 3.0f);
    pb[3] = mvt4(cp, -1.0f, 3.0f, -3.0f, 1.0f);
}
CRH_D Pt quadratic_point(const Pt pb[3], float t) { return mvt3(pb, 1.0f, t, t * t); } // curve.rs:86
CRH_D Pl quadratic_tangent(const Pt pb[3], float t) {                                  // curve.rs:91-95
    return join(mvt3(pb, 1.0f, t, t * t), mvt3(pb, 0.0f, 1.0f, 2.0f * t));
}
CRH_D Pt cubic_point(const Pt pb[4], float t) { return mvt4(pb, 1.0f, t, t * t, t * t * t); } // curve.rs:105
CRH_D Pl cubic_tangent(const Pt pb[4], float t) {                                             // curve.rs:110-114
    return join(mvt4(pb, 1.0f, t, t * t, t * t * t), mvt4(pb, 0.0f, 1.0f, 2.0f * t, 3.0f * (t * t)));
}
CRH_D void reparametrize_cubic(const Pt pb[4], float a, float b, Pt out[4]) { // curve.rs:58-83
    const float a2 = a * a, a3 = a * a * a, b2 = b * b, b3 = b * b * b;
    out[0] = mvt4(pb, 1.0f, a, a2, a3);
    out[1] = mvt4(pb, 0.0f, b - a, -2.0f * a2 + 2.0f * a * b, 3.0f * a2 * b - 3.0f * a3);
    out[2] = mvt4(pb, 0.0f, 0.0f, (a - b) * (a - b), -6.0f * a2 * b + 3.0f * a * b2 + 3.0f * a3);
    out[3] = mvt4(pb, 0.0f, 0.0f, 0.0f, 3.0f * a2 * b - 3.0f * a * b2 - a3 + b3);
}

// curve.rs:133-144
CRH_D void inflection_coefficients(const Pt pb[4], bool integral, float d[4]) {
    d[0] = integral ? 0.0f : triple(pb[1], pb[2], pb[3]) * -1.0f;
    d[1] = triple(pb[0], pb[2], pb[3]) * 1.0f;
    d[2] = triple(pb[0], pb[1], pb[3]) * -1.0f;
    d[3] = triple(pb[0], pb[1], pb[2]) * 1.0f;
    const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
    d[0] *= inv;
    d[1] *= inv;
    d[2] *= inv;
    d[3] *= inv;
}

// polynomial::Root: numerator (re, im), denominator; "no root" = ([1, 0], 0) (curve.rs:157)
struct Root {
    float re, im, den;
};
CRH_D Root no_root() { return {1.0f, 0.0f, 0.0f}; }

// Solvers: same closed forms, same evaluation order and precision as the boundary's numerical contract
// (f32 for degree <= 2, binary64 Cardano / trigonometric / Ferrari above). `n` = number of roots written.
CRH_D float solve_linear(float c0, float c1, Root* r, int& n) {
    n = 0;
    if (fabsf(c1) <= kErrorMargin) return 0.0f;
    r[n++] = {-c0, 0.0f, c1};
    return 1.0f;
}
CRH_D float solve_quadratic(float c0, float c1, float c2, Root* r, int& n) {
    if (fabsf(c2) <= kErrorMargin) return solve_linear(c0, c1, r, n);
    const float D = c1 * c1 - 4.0f * c2 * c0;
    const float den = 2.0f * c2;
    n = 2;
    if (D < 0.0f) {
        const float q = sqrtf(-D);
        r[0] = {-c1, q, den};
        r[1] = {-c1, -q, den};
    } else {
        const float q = sqrtf(D);
        r[0] = {-c1 + q, 0.0f, den};
        r[1] = {-c1 - q, 0.0f, den};
    }
    return D;
}
// (The binary64 body is one out-of-line function per kernel. It hands its roots back BY VALUE: through a pointer the caller's array had to
// live in scratch memory — 48 B per lane of every kernel that classifies a rational cubic.)
struct CubicRoots {
    Root r0, r1, r2;
    float discriminant;
};
inline __host__ __device__ __noinline__ CubicRoots solve_cubic_roots(float f0, float f1, float f2, float f3) {
    Root r[3];
    const double a = f3, b = f2, c = f1, d = f0;
    const double d0 = b * b - 3.0 * a * c;
    const double d1 = 2.0 * b * b * b - 9.0 * a * b * c + 27.0 * a * a * d;
    const double inner = d1 * d1 - 4.0 * d0 * d0 * d0;
    const double disc = 18.0 * a * b * c * d - 4.0 * b * b * b * d + b * b * c * c - 4.0 * a * c * c * c - 27.0 * a * a * d * d;
    const float den = (float)(3.0 * a);
    if (inner >= 0.0) {
        const double s = sqrt(inner);
        const double C = crh_d_cbrt((d1 + (d1 < 0.0 ? -s : s)) * 0.5);
        if (C == 0.0) {
            r[0] = r[1] = r[2] = {(float)(-b), 0.0f, den};
        } else {
            const double q = d0 / C;
            const double re = -0.5 * (C + q);
            const double im = 0.86602540378443864676 * (C - q);
            r[0] = {(float)(-(b + (C + q))), 0.0f, den};
            r[1] = {(float)(-(b + re)), (float)(-im), den};
            r[2] = {(float)(-(b + re)), (float)(im), den};
        }
    } else {
        const double m = 2.0 * sqrt(d0);
        const double theta = crh_d_atan2(sqrt(-inner), d1) * (1.0 / 3.0);
        for (int k = 0; k < 3; ++k) {
            double sn, cs;
            crh_d_sincos(theta + 2.09439510239319549231 * (double)k, &sn, &cs);
            r[k] = {(float)(-(b + m * cs)), 0.0f, den};
        }
    }
    return CubicRoots{r[0], r[1], r[2], (float)disc};
}
CRH_D float solve_cubic(float f0, float f1, float f2, float f3, Root* r, int& n) {
    if (fabsf(f3) <= kErrorMargin) return solve_quadratic(f0, f1, f2, r, n);
    const CubicRoots s = solve_cubic_roots(f0, f1, f2, f3);
    n = 3;
    r[0] = s.r0, r[1] = s.r1, r[2] = s.r2;
    return s.discriminant;
}
CRH_D void push_monic_quadratic(Root* r, double s1, double s0, double shift) {
    const double D = s1 * s1 - 4.0 * s0;
    if (D < 0.0) {
        const double im = 0.5 * sqrt(-D);
        r[0] = {(float)(-0.5 * s1 + shift), (float)im, 1.0f};
        r[1] = {(float)(-0.5 * s1 + shift), (float)(-im), 1.0f};
    } else {
        const double sq = sqrt(D);
        r[0] = {(float)(0.5 * (-s1 + sq) + shift), 0.0f, 1.0f};
        r[1] = {(float)(0.5 * (-s1 - sq) + shift), 0.0f, 1.0f};
    }
}
struct QuarticRoots {
    Root r0, r1, r2, r3;
    float discriminant;
};
inline __host__ __device__ __noinline__ QuarticRoots solve_quartic_roots(float cf0, float cf1, float cf2, float cf3, float cf4) {
    Root r[4];
    const double a4 = cf4;
    const double b = cf3 / a4, c = cf2 / a4, d = cf1 / a4, e = cf0 / a4;
    const double p = c - 0.375 * b * b;
    const double q = 0.125 * b * b * b - 0.5 * b * c + d;
    const double rr = -0.01171875 * b * b * b * b + 0.0625 * b * b * c - 0.25 * b * d + e;
    const double shift = -0.25 * b;
    if (fabs(q) <= 1e-12 * (1.0 + fabs(p) + fabs(rr))) {
        const double D = p * p - 4.0 * rr;
        if (D < 0.0) {
            const double zr = -0.5 * p, zi = 0.5 * sqrt(-D);
            const double m = sqrt(sqrt(zr * zr + zi * zi));
            double sn, cs;
            crh_d_sincos(0.5 * crh_d_atan2(zi, zr), &sn, &cs);
            r[0] = {(float)(m * cs + shift), (float)(m * sn), 1.0f};
            r[1] = {(float)(-m * cs + shift), (float)(-m * sn), 1.0f};
            r[2] = {(float)(m * cs + shift), (float)(-m * sn), 1.0f};
            r[3] = {(float)(-m * cs + shift), (float)(m * sn), 1.0f};
        } else {
            const double sq = sqrt(D);
            const double z[2] = {0.5 * (-p + sq), 0.5 * (-p - sq)};
            for (int k = 0; k < 2; ++k) {
                if (z[k] >= 0.0) {
                    const double y = sqrt(z[k]);
                    r[2 * k] = {(float)(y + shift), 0.0f, 1.0f};
                    r[2 * k + 1] = {(float)(-y + shift), 0.0f, 1.0f};
                } else {
                    const double y = sqrt(-z[k]);
                    r[2 * k] = {(float)shift, (float)y, 1.0f};
                    r[2 * k + 1] = {(float)shift, (float)(-y), 1.0f};
                }
            }
        }
        return QuarticRoots{r[0], r[1], r[2], r[3], (float)D};
    }
    const double rb = p, rc = 0.25 * p * p - rr, rd = -0.125 * q * q;
    const double d0 = rb * rb - 3.0 * rc;
    const double d1 = 2.0 * rb * rb * rb - 9.0 * rb * rc + 27.0 * rd;
    const double inner = d1 * d1 - 4.0 * d0 * d0 * d0;
    double m;
    if (inner >= 0.0) {
        const double s = sqrt(inner);
        const double C = crh_d_cbrt((d1 + (d1 < 0.0 ? -s : s)) * 0.5);
        m = (C == 0.0) ? -rb * (1.0 / 3.0) : -(rb + C + d0 / C) * (1.0 / 3.0);
    } else {
        const double mm = 2.0 * sqrt(d0);
        const double theta = crh_d_atan2(sqrt(-inner), d1) * (1.0 / 3.0);
        m = -1e300;
        for (int k = 0; k < 3; ++k) {
            double sn, cs;
            crh_d_sincos(theta + 2.09439510239319549231 * (double)k, &sn, &cs);
            const double cand = -(rb + mm * cs) * (1.0 / 3.0);
            if (cand > m) m = cand;
        }
    }
    if (!(m > 0.0)) m = 0.0;
    const double s = sqrt(2.0 * m);
    if (s == 0.0) {
        push_monic_quadratic(r, 0.0, 0.5 * p, shift);
        push_monic_quadratic(r + 2, 0.0, 0.5 * p, shift);
    } else {
        push_monic_quadratic(r, s, 0.5 * p + m - q / (2.0 * s), shift);
        push_monic_quadratic(r + 2, -s, 0.5 * p + m + q / (2.0 * s), shift);
    }
    return QuarticRoots{r[0], r[1], r[2], r[3], (float)inner};
}
CRH_D float solve_quartic(const float cf[5], Root* r, int& n) {
    if (fabsf(cf[4]) <= kErrorMargin) return solve_cubic(cf[0], cf[1], cf[2], cf[3], r, n);
    const QuarticRoots s = solve_quartic_roots(cf[0], cf[1], cf[2], cf[3], cf[4]);
    n = 4;
    r[0] = s.r0, r[1] = s.r1, r[2] = s.r2, r[3] = s.r3;
    return s.discriminant;
}

// curve.rs:151-190
CRH_D float integral_inflection_points(const float d[4], bool loop_self_intersection, Root roots[3]) {
    const float discriminant = 3.0f * (d[2] * d[2]) - 4.0f * d[1] * d[3];
    roots[1] = no_root();
    roots[2] = no_root();
    if (fabsf(d[1]) <= kErrorMargin) {
        if (fabsf(d[2]) <= kErrorMargin) {
            roots[0] = {-1.0f, 0.0f, 1.0f};
            return -1.0f;
        }
        roots[0] = {d[3], 0.0f, 3.0f * d[2]};
        return 1.0f;
    }
    const float factor = discriminant < 0.0f ? (loop_self_intersection ? -1.0f : 0.0f) : 1.0f / 3.0f;
    const float s = sqrtf(discriminant * factor);
    roots[0] = {d[2] + s, 0.0f, 2.0f * d[1]};
    roots[1] = {d[2] - s, 0.0f, 2.0f * d[1]};
    return discriminant;
}
// curve.rs:197-226
CRH_D float rational_inflection_points(const float d[4], bool loop_self_intersection, Root roots[3]) {
    if (fabsf(d[0]) <= kErrorMargin) return integral_inflection_points(d, loop_self_intersection, roots);
    Root solved[3];
    int n = 0;
    float discriminant = solve_cubic(d[3] * -1.0f, d[2] * 3.0f, d[1] * -3.0f, d[0], solved, n);
    for (int k = 0; k < 3; ++k) roots[k] = k < n ? solved[k] : no_root();
    if (!loop_self_intersection) return discriminant;
    Root h[2];
    int hn = 0;
    discriminant = solve_quadratic(d[1] * d[3] - d[2] * d[2], d[1] * d[2] - d[0] * d[3], d[0] * d[2] - d[1] * d[1], h, hn);
    if (discriminant > 0.0f) {
        roots[2] = roots[0]; // real_root == 0 for every branch of solve_cubic
        if (hn == 2) {
            roots[0] = h[0];
            roots[1] = h[1];
        } else if (hn == 1) {
            roots[0] = h[0];
            roots[1] = no_root();
        }
    }
    return -discriminant;
}

// epga1d::ComplexNumber (curve.rs:230-238)
struct Cx {
    float re, im;
};
CRH_D Cx cmul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
CRH_D Cx cdiv(Cx a, Cx b) {
    const float s = 1.0f / (b.re * b.re + b.im * b.im);
    return cmul(a, Cx{b.re * s, -b.im * s});
}
CRH_D Cx cpowf(Cx a, float e) {
    const float m = crh_powf(sqrtf(a.re * a.re + a.im * a.im), e);
    float s, c;
    crh_sincosf(crh_atan2f(a.im, a.re) * e, &s, &c);
    return {m * c, m * s};
}
CRH_D Cx cpowi(Cx a, uint32_t n) {
    Cx r = {1.0f, 0.0f};
    for (;;) {
        if (n & 1u) r = cmul(r, a);
        n >>= 1;
        if (n == 0) break;
        a = cmul(a, a);
    }
    return r;
}

} // namespace crh
