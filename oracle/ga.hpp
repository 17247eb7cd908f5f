// oracle/ga.hpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the product; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Restatement of the arithmetic the reference takes from the un-vendored crate
// `geometric_algebra 0.3.0` (Cargo.toml:19, Cargo.lock:520-523, checksum dea41acf…20959):
// ppga2d::{Point, Plane}, ppga3d::{Point, Plane}, epga1d::ComplexNumber and polynomial::{solve_*, Root}.
// The crate source is not in /root/reference and cannot be fetched, so the element layouts and sign
// conventions below are the ones pinned by the reference's own call sites (SURVEY.md Appendix A.1:
// start caps must extend backwards stroke.rs:274-282, the SAT helper documents clockwise input
// utils.rs:83-98, StrokeOptions::offset documents left/right path.rs:179). PARITY UNPINNED: there are
// no golden vectors in the reference; solver internals and summation association are this file's
// choice, stated next to each function.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/crh_fmath.h"

namespace oracle {

constexpr float ERROR_MARGIN = 0.0001f; // error.rs:19
constexpr float F32_EPSILON = 1.1920929e-7f;

inline float f32_signum(float x) { // Rust f32::signum: NaN -> NaN, +-0 -> +-1
    if (x != x) return x;
    return std::signbit(x) ? -1.0f : 1.0f;
}

// ---- ppga2d ---------------------------------------------------------------------------------------
// Point::new(w, x*w, y*w) (utils.rs:111-118); Plane::new(c, nx, ny) with [0] the e0 part (utils.rs:101-103).
struct Point {
    float g[3];
    float& operator[](int i) { return g[i]; }
    float operator[](int i) const { return g[i]; }
};
struct Plane {
    float g[3];
    float& operator[](int i) { return g[i]; }
    float operator[](int i) const { return g[i]; }
};
inline Point operator+(Point a, Point b) { return {{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
inline Point operator*(Point a, float s) { return {{a[0] * s, a[1] * s, a[2] * s}}; }
inline Plane operator+(Plane a, Plane b) { return {{a[0] + b[0], a[1] + b[1], a[2] + b[2]}}; }
inline Plane operator-(Plane a, Plane b) { return {{a[0] - b[0], a[1] - b[1], a[2] - b[2]}}; }
inline Plane operator*(Plane a, float s) { return {{a[0] * s, a[1] * s, a[2] * s}}; }
inline Plane operator-(Plane a) { return {{-a[0], -a[1], -a[2]}}; }

// Dual: index-preserving copy (stroke.rs:19-20 zeroes [0] afterwards; curve.rs:312,336-338)
inline Point dual(Plane p) { return {{p[0], p[1], p[2]}}; }
inline Plane dual(Point p) { return {{p[0], p[1], p[2]}}; }

// Point v Point -> Plane: the line through both; normal = (b - a) rotated 90 deg clockwise (y up).
inline Plane regressive(Point a, Point b) {
    return {{a[2] * b[1] - a[1] * b[2], a[0] * b[2] - a[2] * b[0], a[1] * b[0] - a[0] * b[1]}};
}
// Plane v Point and Point v Plane -> scalar (3-term sum, left to right).
inline float regressive(Plane l, Point p) { return l[0] * p[0] + l[1] * p[1] + l[2] * p[2]; }
inline float regressive(Point p, Plane l) { return p[0] * l[0] + p[1] * l[1] + p[2] * l[2]; }
// Plane ^ Plane -> Point (meet)
inline Point outer(Plane a, Plane b) {
    return {{a[2] * b[1] - a[1] * b[2], a[0] * b[2] - a[2] * b[0], a[1] * b[0] - a[0] * b[1]}};
}
// Plane . Plane -> scalar (e0 is null)
inline float inner(Plane a, Plane b) { return a[1] * b[1] + a[2] * b[2]; }
// Plane . Point -> Plane: the line through P perpendicular to l
inline Plane inner(Plane l, Point p) { return {{l[1] * p[2] - l[2] * p[1], l[2] * p[0], -l[1] * p[0]}}; }
// grade-1 part of Plane * Point: the same contraction (stroke.rs:71-75,86)
inline Plane geometric_product_grade1(Plane l, Point p) { return inner(l, p); }
inline float squared_magnitude(Plane l) { return l[1] * l[1] + l[2] * l[2]; }
inline float magnitude(Plane l) { return std::sqrt(squared_magnitude(l)); }
// Signum: multiply by the reciprocal of the magnitude; a zero line gives NaN in [0] (0 * inf).
inline Plane signum(Plane l) { return l * (1.0f / magnitude(l)); }
inline Plane geometric_quotient(Plane l, float s) { return l * (1.0f / s); }

// ---- utils.rs hot helpers ------------------------------------------------------------------------
inline Plane rotate_90_degree_clockwise(Plane v) { return {{0.0f, v[2], -v[1]}}; } // utils.rs:101-103
inline void point_to_vec(Point p, float out[2]) {                                   // utils.rs:106-108
    out[0] = p[1] / p[0];
    out[1] = p[2] / p[0];
}
inline Point vec_to_point(const float v[2]) { return {{1.0f, v[0], v[1]}}; }                            // utils.rs:111-113
inline Point weighted_vec_to_point(float w, const float v[2]) { return {{w, v[0] * w, v[1] * w}}; }      // utils.rs:116-118
inline Point line_line_intersection(Plane a, Plane b) {                                                  // utils.rs:67-70
    Point p = outer(a, b);
    return p * (1.0f / p[0]);
}

// ---- ppga3d (fill.rs:70-85 only) -----------------------------------------------------------------
struct Point3 {
    float g[4];
    float& operator[](int i) { return g[i]; }
    float operator[](int i) const { return g[i]; }
};
struct Plane3 {
    float g[4];
    float& operator[](int i) { return g[i]; }
    float operator[](int i) const { return g[i]; }
};
inline Point3 operator+(Point3 a, Point3 b) { return {{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}}; }
inline Point3 operator*(Point3 a, float s) { return {{a[0] * s, a[1] * s, a[2] * s, a[3] * s}}; }
inline Plane3 operator*(Plane3 a, float s) { return {{a[0] * s, a[1] * s, a[2] * s, a[3] * s}}; }
// (A v B) v C: Pluecker minors of the line A v B, then a 3-term sum per plane coefficient. The overall
// sign is immaterial: fill.rs:81 rescales by 1 / -plane[3].
inline Plane3 regressive3(Point3 A, Point3 B, Point3 C) {
    const float m01 = A[0] * B[1] - A[1] * B[0];
    const float m02 = A[0] * B[2] - A[2] * B[0];
    const float m03 = A[0] * B[3] - A[3] * B[0];
    const float m12 = A[1] * B[2] - A[2] * B[1];
    const float m13 = A[1] * B[3] - A[3] * B[1];
    const float m23 = A[2] * B[3] - A[3] * B[2];
    return {{m12 * C[3] - m13 * C[2] + m23 * C[1], -(m02 * C[3] - m03 * C[2] + m23 * C[0]), m01 * C[3] - m03 * C[1] + m13 * C[0],
             -(m01 * C[2] - m02 * C[1] + m12 * C[0])}};
}
inline float squared_magnitude(Plane3 p) { return p[1] * p[1] + p[2] * p[2] + p[3] * p[3]; }

// ---- epga1d::ComplexNumber (curve.rs:230-238) -----------------------------------------------------
struct Complex {
    float re, im;
};
inline Complex cmul(Complex a, Complex b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// a / b = a * conj(b) * (1 / |b|^2)
inline Complex cdiv(Complex a, Complex b) {
    const float s = 1.0f / (b.re * b.re + b.im * b.im);
    const Complex inv = {b.re * s, -b.im * s};
    return cmul(a, inv);
}
inline float carg(Complex a) { return crh_atan2f(a.im, a.re); }
// powf(e) = from_polar(|z|^e, arg * e)
inline Complex cpowf(Complex a, float e) {
    const float mag = crh_powf(std::sqrt(a.re * a.re + a.im * a.im), e);
    float s, c;
    crh_sincosf(carg(a) * e, &s, &c);
    return {mag * c, mag * s};
}
// powi(n), n >= 0: square-and-multiply starting from the identity
inline Complex cpowi(Complex a, int64_t n) {
    Complex r = {1.0f, 0.0f};
    while (true) {
        if (n & 1) r = cmul(r, a);
        n >>= 1;
        if (n == 0) break;
        a = cmul(a, a);
    }
    return r;
}

// ---- polynomial::{Root, solve_*} ------------------------------------------------------------------
// Interface pinned by the call sites: ascending coefficients; Root{numerator: complex,
// denominator}; parameter = numerator.real() / denominator, skipped when denominator == 0. Internals
// are closed-form textbook solvers evaluated in binary64 and rounded to f32 (stated per function).
struct Root {
    float num_re, num_im, den;
};
struct Roots {
    int n;
    Root r[4];
};
inline Root no_root() { return {1.0f, 0.0f, 0.0f}; } // Root::new([1.0, 0.0], 0.0), curve.rs:157

// c0 + c1 t = 0. First tuple element: 1 when a root exists, else 0.
inline float solve_linear(const float c[2], float eps, Roots& out) {
    out.n = 0;
    if (std::fabs(c[1]) <= eps) return 0.0f;
    out.r[out.n++] = {-c[0], 0.0f, c[1]};
    return 1.0f;
}

// c0 + c1 t + c2 t^2 = 0: quadratic formula, roots (-c1 + q)/(2 c2) then (-c1 - q)/(2 c2), q = sqrt(D)
// (complex when D < 0). Returns D = c1^2 - 4 c2 c0, all in f32 like the crate's in-tree sibling curve.rs:152.
inline float solve_quadratic(const float c[3], float eps, Roots& out) {
    if (std::fabs(c[2]) <= eps) return solve_linear(c, eps, out);
    const float D = c[1] * c[1] - 4.0f * c[2] * c[0];
    const float den = 2.0f * c[2];
    out.n = 2;
    if (D < 0.0f) {
        const float q = std::sqrt(-D);
        out.r[0] = {-c[1], q, den};
        out.r[1] = {-c[1], -q, den};
    } else {
        const float q = std::sqrt(D);
        out.r[0] = {-c[1] + q, 0.0f, den};
        out.r[1] = {-c[1] - q, 0.0f, den};
    }
    return D;
}

// Monic-free cubic c0 + c1 t + c2 t^2 + c3 t^3 in binary64: Cardano for one real root, trigonometric
// form for three. Roots are homogeneous: numerator = -(c2 + u_k), denominator = 3 c3.
// Returns the standard discriminant (> 0 <=> three distinct real roots) and the index of a real root.
inline float solve_cubic(const float cf[4], float eps, Roots& out, int& real_root) {
    real_root = 0;
    if (std::fabs(cf[3]) <= eps) {
        const float D = solve_quadratic(cf, eps, out);
        return D;
    }
    const double a = cf[3], b = cf[2], c = cf[1], d = cf[0];
    const double d0 = b * b - 3.0 * a * c;
    const double d1 = 2.0 * b * b * b - 9.0 * a * b * c + 27.0 * a * a * d;
    const double inner_disc = d1 * d1 - 4.0 * d0 * d0 * d0; // = -27 a^2 * discriminant
    const double disc = 18.0 * a * b * c * d - 4.0 * b * b * b * d + b * b * c * c - 4.0 * a * c * c * c - 27.0 * a * a * d * d;
    const float den = (float)(3.0 * a);
    out.n = 3;
    if (inner_disc >= 0.0) {
        const double s = std::sqrt(inner_disc);
        const double C = crh_d_cbrt((d1 + (d1 < 0.0 ? -s : s)) * 0.5);
        if (C == 0.0) {
            for (int k = 0; k < 3; ++k) out.r[k] = {(float)(-b), 0.0f, den};
        } else {
            const double q = d0 / C;
            const double re = -0.5 * (C + q);
            const double im = 0.86602540378443864676 * (C - q);
            out.r[0] = {(float)(-(b + (C + q))), 0.0f, den};
            out.r[1] = {(float)(-(b + re)), (float)(-im), den};
            out.r[2] = {(float)(-(b + re)), (float)(im), den};
        }
    } else {
        const double m = 2.0 * std::sqrt(d0);
        const double theta = crh_d_atan2(std::sqrt(-inner_disc), d1) * (1.0 / 3.0);
        for (int k = 0; k < 3; ++k) {
            double sn, cs;
            crh_d_sincos(theta + 2.09439510239319549231 * (double)k, &sn, &cs);
            out.r[k] = {(float)(-(b + m * cs)), 0.0f, den};
        }
    }
    return (float)disc;
}

// Quartic c0 + ... + c4 t^4 in binary64 by Ferrari: depressed quartic y^4 + p y^2 + q y + r, real root m > 0
// of the resolvent 8 m^3 + 8 p m^2 + (2 p^2 - 8 r) m - q^2, then two quadratics. Roots are returned with
// denominator 1: first the (+,-) pair of y^2 + s y + .., then of y^2 - s y + ...
inline float solve_quartic(const float cf[5], float eps, Roots& out) {
    if (std::fabs(cf[4]) <= eps) {
        int real_root;
        return solve_cubic(cf, eps, out, real_root);
    }
    const double a4 = cf[4];
    const double b = cf[3] / a4, c = cf[2] / a4, d = cf[1] / a4, e = cf[0] / a4;
    const double p = c - 0.375 * b * b;
    const double q = 0.125 * b * b * b - 0.5 * b * c + d;
    const double r = -0.01171875 * b * b * b * b + 0.0625 * b * b * c - 0.25 * b * d + e;
    const double shift = -0.25 * b;
    out.n = 4;
    auto push_quadratic = [&](int at, double s1, double s0) { // y^2 + s1 y + s0
        const double D = s1 * s1 - 4.0 * s0;
        if (D < 0.0) {
            const double im = 0.5 * std::sqrt(-D);
            out.r[at] = {(float)(-0.5 * s1 + shift), (float)im, 1.0f};
            out.r[at + 1] = {(float)(-0.5 * s1 + shift), (float)(-im), 1.0f};
        } else {
            const double sq = std::sqrt(D);
            out.r[at] = {(float)(0.5 * (-s1 + sq) + shift), 0.0f, 1.0f};
            out.r[at + 1] = {(float)(0.5 * (-s1 - sq) + shift), 0.0f, 1.0f};
        }
    };
    if (std::fabs(q) <= 1e-12 * (1.0 + std::fabs(p) + std::fabs(r))) {
        // biquadratic: y^2 = (-p +- sqrt(p^2 - 4 r)) / 2
        const double D = p * p - 4.0 * r;
        if (D < 0.0) {
            // y^2 complex: y = +-sqrt(z), z = (-p +- i sqrt(-D)) / 2
            const double zr = -0.5 * p, zi = 0.5 * std::sqrt(-D);
            const double mag = std::sqrt(std::sqrt(zr * zr + zi * zi));
            double sn, cs;
            crh_d_sincos(0.5 * crh_d_atan2(zi, zr), &sn, &cs);
            out.r[0] = {(float)(mag * cs + shift), (float)(mag * sn), 1.0f};
            out.r[1] = {(float)(-mag * cs + shift), (float)(-mag * sn), 1.0f};
            out.r[2] = {(float)(mag * cs + shift), (float)(-mag * sn), 1.0f};
            out.r[3] = {(float)(-mag * cs + shift), (float)(mag * sn), 1.0f};
        } else {
            const double sq = std::sqrt(D);
            const double z[2] = {0.5 * (-p + sq), 0.5 * (-p - sq)};
            for (int k = 0; k < 2; ++k) {
                if (z[k] >= 0.0) {
                    const double y = std::sqrt(z[k]);
                    out.r[2 * k] = {(float)(y + shift), 0.0f, 1.0f};
                    out.r[2 * k + 1] = {(float)(-y + shift), 0.0f, 1.0f};
                } else {
                    const double y = std::sqrt(-z[k]);
                    out.r[2 * k] = {(float)shift, (float)y, 1.0f};
                    out.r[2 * k + 1] = {(float)shift, (float)(-y), 1.0f};
                }
            }
        }
        return (float)D;
    }
    // resolvent cubic m^3 + p m^2 + (p^2/4 - r) m - q^2/8 = 0, largest real root (positive because q != 0)
    const double rb = p, rc = 0.25 * p * p - r, rd = -0.125 * q * q;
    const double d0 = rb * rb - 3.0 * rc;
    const double d1 = 2.0 * rb * rb * rb - 9.0 * rb * rc + 27.0 * rd;
    const double inner_disc = d1 * d1 - 4.0 * d0 * d0 * d0;
    double m;
    if (inner_disc >= 0.0) {
        const double s = std::sqrt(inner_disc);
        const double C = crh_d_cbrt((d1 + (d1 < 0.0 ? -s : s)) * 0.5);
        m = (C == 0.0) ? -rb * (1.0 / 3.0) : -(rb + C + d0 / C) * (1.0 / 3.0);
    } else {
        const double mm = 2.0 * std::sqrt(d0);
        const double theta = crh_d_atan2(std::sqrt(-inner_disc), d1) * (1.0 / 3.0);
        m = -1e300;
        for (int k = 0; k < 3; ++k) {
            double sn, cs;
            crh_d_sincos(theta + 2.09439510239319549231 * (double)k, &sn, &cs);
            const double cand = -(rb + mm * cs) * (1.0 / 3.0);
            if (cand > m) m = cand;
        }
    }
    if (!(m > 0.0)) m = 0.0;
    const double s = std::sqrt(2.0 * m);
    if (s == 0.0) {
        push_quadratic(0, 0.0, 0.5 * p);
        push_quadratic(2, 0.0, 0.5 * p);
    } else {
        push_quadratic(0, s, 0.5 * p + m - q / (2.0 * s));
        push_quadratic(2, -s, 0.5 * p + m + q / (2.0 * s));
    }
    return (float)inner_disc;
}

} // namespace oracle
