// csrc/fill.hpp — per-lane implicit-curve fill of ONE cubic segment (device side of fill.rs:14-250).
//
// One lane owns one segment. The routine is written against a `Sink` so that the count pass (how many records
// will this lane emit?) and the emit pass (write them at the scanned offsets) run the very same arithmetic.
// No per-segment heap Vecs as in the reference: at most 12 curve vertices and 6 polygon vertices leave the lane,
// straight from registers.
#pragma once
#include "../../include/contrast_hip.h"
#include "ga.hpp"

namespace crh {

// fill.rs:34-49
CRH_D void weight_derivatives(float w[4][4], int column, Root r0, Root r1, Root r2) {
    const float p0 = r0.re * r1.re * r2.re;
    const float p1 = -r0.den * r1.re * r2.re - r0.re * r1.den * r2.re - r0.re * r1.re * r2.den;
    const float p2 = r0.re * r1.den * r2.den + r0.den * r1.re * r2.den + r0.den * r1.den * r2.re;
    const float p3 = -r0.den * r1.den * r2.den;
    w[0][column] = p0;
    w[1][column] = p0 + p1 * 1.0f / 3.0f;
    w[2][column] = p0 + p1 * 2.0f / 3.0f + p2 * 1.0f / 3.0f;
    w[3][column] = p0 + p1 + p2 + p3;
}

// fill.rs:51-68
CRH_D void cubic_weights(float discriminant, const Root r[3], float w[4][4]) {
    if (discriminant == 0.0f) {
        weight_derivatives(w, 0, r[0], r[0], r[2]);
        weight_derivatives(w, 1, r[0], r[0], r[0]);
        weight_derivatives(w, 2, r[0], r[0], r[0]);
    } else if (discriminant < 0.0f) {
        weight_derivatives(w, 0, r[0], r[1], r[2]);
        weight_derivatives(w, 1, r[0], r[0], r[1]);
        weight_derivatives(w, 2, r[1], r[1], r[0]);
    } else {
        weight_derivatives(w, 0, r[0], r[1], r[2]);
        weight_derivatives(w, 1, r[0], r[0], r[0]);
        weight_derivatives(w, 2, r[1], r[1], r[1]);
    }
    weight_derivatives(w, 3, r[2], r[2], r[2]);
}

// (A v B) v C in ppga3d for points (w, x, y, z): Pluecker minors of the line, then 3-term sums. fill.rs:77-79.
struct P4 {
    float a, b, c, d;
};
CRH_D P4 join3(P4 A, P4 B, P4 C) {
    const float m01 = A.a * B.b - A.b * B.a;
    const float m02 = A.a * B.c - A.c * B.a;
    const float m03 = A.a * B.d - A.d * B.a;
    const float m12 = A.b * B.c - A.c * B.b;
    const float m13 = A.b * B.d - A.d * B.b;
    const float m23 = A.c * B.d - A.d * B.c;
    return {m12 * C.d - m13 * C.c + m23 * C.b, -(m02 * C.d - m03 * C.c + m23 * C.a), m01 * C.d - m03 * C.b + m13 * C.a,
            -(m01 * C.c - m02 * C.b + m12 * C.a)};
}

// fill.rs:70-96: the gradient of k^3 - l m n at control point 0, through the four weight planes
CRH_D Pl implicit_gradient_at_start(const Pt cp[4], const float w[4][4]) {
    Pl planes[4];
    for (int i = 0; i < 4; ++i) {
        const P4 p0 = {cp[0].w, cp[0].x, cp[0].y, w[0][i]};
        const P4 p1 = {cp[1].w, cp[1].x, cp[1].y, w[1][i]};
        const P4 p2 = {cp[2].w, cp[2].x, cp[2].y, w[2][i]};
        P4 plane = join3(p0, p1, p2);
        if (plane.b * plane.b + plane.c * plane.c + plane.d * plane.d < kErrorMargin) {
            const P4 p3 = {cp[3].w, cp[3].x, cp[3].y, w[3][i]};
            plane = join3(p0, p1, p3);
        }
        const float s = 1.0f / -plane.d;
        planes[i] = {plane.a * s, plane.b * s, plane.c * s};
    }
    return planes[0] * (3.0f * w[0][0] * w[0][0]) - planes[1] * (w[0][2] * w[0][3]) - planes[2] * (w[0][1] * w[0][3]) - planes[3] * (w[0][1] * w[0][2]);
}

// triangulate_cubic_curve_quadrilateral! + emit_cubic_curve_triangle! (fill.rs:116-204)
template <class Sink>
CRH_D void triangulate_quadrilateral(const Pt cp[4], float w[4][4], Sink& sink, uint32_t& err) {
    for (int j = 0; j < 4; ++j) {
        const float s = 1.0f / cp[j].w;
        for (int i = 0; i < 4; ++i) w[j][i] = w[j][i] * s;
    }
    float area[4];
    area[0] = triple(cp[1], cp[2], cp[3]);
    area[1] = triple(cp[0], cp[2], cp[3]);
    area[2] = triple(cp[0], cp[1], cp[3]);
    area[3] = triple(cp[0], cp[1], cp[2]);
    const float area_sum = fabsf(area[0]) + fabsf(area[1]) + fabsf(area[2]) + fabsf(area[3]);
    const float equilibrium = 0.5f * area_sum;
    int enclosing = -1;
    for (int i = 0; i < 4; ++i)
        if (fabsf(equilibrium - fabsf(area[i])) <= kErrorMargin) enclosing = enclosing < 0 ? i : -1;
    float2 v[4];
    for (int j = 0; j < 4; ++j) v[j] = point_to_vec(cp[j]);
    // (which corner is left out is known at run time only: the corners are picked with selects — indexing the register arrays with a
    // variable sent v[], w[][] and area[] to scratch memory, 176 B per lane of the emitting kernel)
    auto pick = [](int i, float a0, float a1, float a2, float a3) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); };
    auto emit_corner = [&](int i) {
        const float wi[4] = {pick(i, w[0][0], w[1][0], w[2][0], w[3][0]), pick(i, w[0][1], w[1][1], w[2][1], w[3][1]), pick(i, w[0][2], w[1][2], w[2][2], w[3][2]),
                             pick(i, w[0][3], w[1][3], w[2][3], w[3][3])};
        sink.curve(make_float2(pick(i, v[0].x, v[1].x, v[2].x, v[3].x), pick(i, v[0].y, v[1].y, v[2].y, v[3].y)), wi);
    };
    auto emit_triangle = [&](int skip) {
        const float a = pick(skip, area[0], area[1], area[2], area[3]);
        if (fabsf(a) > kErrorMargin) {
            int i0 = skip == 0 ? 1 : 0;
            int i1 = skip <= 1 ? 2 : 1;
            int i2 = skip == 3 ? 2 : 3;
            if (a < 0.0f) {
                const int t = i0;
                i0 = i2;
                i2 = t;
            }
            emit_corner(i0);
            emit_corner(i1);
            emit_corner(i2);
        }
    };
    if (enclosing >= 0) {
        emit_triangle(enclosing);
    } else {
        int opposite = 0;
        for (int j = 1; j < 4; ++j) {
            const float side_of_d = area[0] * (j == 2 ? -1.0f : 1.0f);
            if (area[j] * side_of_d < 0.0f) {
                if (opposite != 0) err = CRH_ERR_DEGENERATE_CUBIC; // assert_eq!, fill.rs:174
                opposite = j;
            }
        }
        if (opposite == 0) err = CRH_ERR_DEGENERATE_CUBIC; // assert_ne!, fill.rs:178
        emit_triangle(0);
        emit_triangle(opposite);
    }
    // interior control points join the fan polygon (fill.rs:191-201)
    const bool in1 = enclosing != 1 && (w[1][0] * w[1][0] * w[1][0] - w[1][1] * w[1][2] * w[1][3]) < 0.0f;
    const bool in2 = enclosing != 2 && (w[2][0] * w[2][0] * w[2][0] - w[2][1] * w[2][2] * w[2][3]) < 0.0f;
    if (in1 && in2 && area[0] * area[1] < 0.0f) {
        sink.solid(v[2]);
        sink.solid(v[1]);
    } else {
        if (in1) sink.solid(v[1]);
        if (in2) sink.solid(v[2]);
    }
}

// emit_cubic_curve! (fill.rs:218-250) for control points cp (weighted for rational segments, fill.rs:337-342)
template <class Sink>
CRH_D void cubic_fill(const Pt cp[4], bool integral, Sink& sink, uint32_t& err) {
    Pt pb[4];
    cubic_power_basis(cp, pb);
    float d[4];
    inflection_coefficients(pb, integral, d);
    Root roots[3];
    const float discriminant = integral ? integral_inflection_points(d, true, roots) : rational_inflection_points(d, true, roots);
    float w[4][4];
    cubic_weights(discriminant, roots, w);
    const Pl gradient = implicit_gradient_at_start(cp, w);
    const Pl tangent = cubic_tangent(pb, 0.0f);
    if (dot(tangent, gradient) > 0.0f) { // normalize_implicit_curve_side, fill.rs:98-114
        for (int j = 0; j < 4; ++j) {
            w[j][0] *= -1.0f;
            w[j][1] *= -1.0f;
        }
    }
    // find_double_point_issue, fill.rs:14-32
    bool split = false;
    float param = -1.0f;
    if (discriminant < 0.0f) {
        int inside = 0;
        for (int k = 0; k < 3; ++k) {
            if (roots[k].den != 0.0f) {
                const float t = roots[k].re / roots[k].den;
                if (0.0f < t && t < 1.0f) {
                    param = t;
                    inside += 1;
                }
            }
        }
        split = inside == 1;
    }
    if (split) { // split_curve_at! on the points and on the weights (fill.rs:206-216,232-241)
        const float u = 1.0f - param;
        auto lerp_pt = [&](Pt a, Pt b) { return a * u + b * param; };
        const Pt p10 = lerp_pt(cp[0], cp[1]), p11 = lerp_pt(cp[1], cp[2]), p12 = lerp_pt(cp[2], cp[3]);
        const Pt p20 = lerp_pt(p10, p11), p21 = lerp_pt(p11, p12);
        const Pt p30 = lerp_pt(p20, p21);
        float wa[4][4], wb[4][4];
        for (int i = 0; i < 4; ++i) {
            const float w10 = w[0][i] * u + w[1][i] * param, w11 = w[1][i] * u + w[2][i] * param, w12 = w[2][i] * u + w[3][i] * param;
            const float w20 = w10 * u + w11 * param, w21 = w11 * u + w12 * param;
            const float w30 = w20 * u + w21 * param;
            wa[0][i] = w[0][i];
            wa[1][i] = w10;
            wa[2][i] = w20;
            wa[3][i] = w30;
            const float flip = i < 2 ? -1.0f : 1.0f;
            wb[0][i] = i < 2 ? w30 * flip : w30;
            wb[1][i] = i < 2 ? w21 * flip : w21;
            wb[2][i] = i < 2 ? w12 * flip : w12;
            wb[3][i] = i < 2 ? w[3][i] * flip : w[3][i];
        }
        const Pt cpa[4] = {cp[0], p10, p20, p30};
        const Pt cpb[4] = {p30, p21, p12, cp[3]};
        triangulate_quadrilateral(cpa, wa, sink, err);
        sink.solid(point_to_vec(cpb[0]));
        triangulate_quadrilateral(cpb, wb, sink, err);
    } else {
        triangulate_quadrilateral(cp, w, sink, err);
    }
    sink.hull(point_to_vec(cp[1]));
    sink.hull(point_to_vec(cp[2]));
    const float2 end = point_to_vec(cp[3]);
    sink.hull(end);
    sink.solid(end);
}

} // namespace crh
